#!/bin/bash
# kernel trace of the 100-50-25 policy's update on the fused kernels (tools/f3_time.py).  usage (GPU box): bash tools/f3_trace.sh <tag> N...
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; tag=${1:-rXX}; shift; out=$R/gpurun_out/round; mkdir -p $out
for N in "$@"; do
  (cd /tmp && TMPDIR=/tmp F3_ONLY=1 rocprofv3 --kernel-trace -d /tmp/prof_f3 -o t -- python $R/tools/f3_time.py $N > $out/${tag}_f3_N$N.txt 2>/dev/null)
  python tools/rocpd_stats.py /tmp/prof_f3/t_results.db > $out/${tag}_f3_N$N.kernel_stats.txt; rm -rf /tmp/prof_f3
done
