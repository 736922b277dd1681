#!/bin/bash
# build tools/_variants/<name>.so = the shipped objects with rollout_coop.hip recompiled with extra flags (e.g. -DCOOP_TIMING)
# usage: [SRC=policy_mfma.hip] tools/build_variant.sh <name> [extra hipcc flags...]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd "$root/me-trpo_amd/csrc"
make -s
mkdir -p "$root/tools/_variants"
src=${SRC:-rollout_coop.hip}
vf=""; { [ "$src" = rollout_coop.hip ] || [ "$src" = rollout_resident.hip ]; } && vf="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function $vf "$@" -c $src -o /tmp/rc_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "^${src%.hip}.o\$") /tmp/rc_$name.o -ldl -o "$root/tools/_variants/$name.so"
echo built tools/_variants/$name.so
