#!/bin/bash
# Kernel trace of a short C1 bench run: per-kernel average durations (rocprofv3 --kernel-trace) -> gpurun_out/fvp_trace_<tag>.txt
# usage: tools/fvp_trace.sh <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; shift
out=$R/gpurun_out; mkdir -p $out
(cd /tmp && env "$@" rocprofv3 --kernel-trace -d $out/prof_$tag -o t -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $out/fvp_trace_$tag.json 2> $out/fvp_trace_$tag.err)
python $R/tools/rocpd_stats.py $out/prof_$tag/t_results.db > $out/fvp_trace_$tag.txt
rm -rf $out/prof_$tag
head -12 $out/fvp_trace_$tag.txt
