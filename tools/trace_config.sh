#!/bin/bash
# Kernel trace of a short bench run of one config: per-kernel durations -> gpurun_out/trace_<config>_<tag>.txt
# usage: tools/trace_config.sh <config> <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=$1; tag=$2; shift; shift
out=$R/gpurun_out; mkdir -p $out
(cd /tmp && env "$@" rocprofv3 --kernel-trace -d $out/prof_$tag -o t -- python $R/bench.py --config $cfg --no-cpu-baseline --steps 5 --warmup 2 > $out/trace_${cfg}_$tag.json 2> $out/trace_${cfg}_$tag.err)
python $R/tools/rocpd_stats.py $out/prof_$tag/t_results.db > $out/trace_${cfg}_$tag.txt
python $R/tools/rocpd_sequence.py $out/prof_$tag/t_results.db > $out/seq_${cfg}_$tag.txt 2>&1
rm -rf $out/prof_$tag
head -14 $out/trace_${cfg}_$tag.txt
python -c "import json,sys; r=json.load(open('$out/trace_${cfg}_$tag.json')); print('ms_per_step', r['ms_per_step'], 'rollout', r['rollout']['ms'], 'update', r['roofline']['update']['ms'])"
