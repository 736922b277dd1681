#!/bin/bash
# Rollout kernel families at the small per-GPU shares of C2 / C3 (strong scaling, verdict r5 item 2): default pick vs NO_PERSIST (per-step stream-K launches) vs
# NO_STREAMK (tile GEMMs per step).   usage: tools/share_sweep.sh > gpurun_out/share_sweep.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print("%-4s B=%-5s %-28s iter %8.3f ms  rollout %8.3f ms  %-20s frac %.3f" % (sys.argv[1],sys.argv[2],sys.argv[3],d["ms_per_step"],d["rollout"]["ms"],d["rollout"]["kernel"],d["roofline"]["frac"]))'
for cfg in C2 C3; do for B in 312 625 1250 2500; do
  for v in "" "METRPO_NO_PERSIST=1" "METRPO_NO_PERSIST=1 METRPO_STREAMK_LATE=1" "METRPO_NO_STREAMK=1"; do
    env $v python bench.py --config $cfg --B $B --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$show" $cfg $B "${v:-default}"
  done; done; done
