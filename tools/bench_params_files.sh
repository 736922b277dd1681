#!/bin/bash
# bench.py over the reference's own params-file shapes (+ C0 / C2s) -> gpurun_out/params/<tag>_bench_<config>.json; kernel traces of Ant and Humanoid.
# usage: tools/bench_params_files.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; out=$R/gpurun_out/params; mkdir -p $out
cd $R
for c in C0 C2s C0p C0hc C0ho C0sn C0an C0hu; do python bench.py --config $c --no-cpu-baseline > $out/${tag}_bench_$c.json 2> $out/$c.err; done
for c in C0an C0hu; do
  (cd /tmp && rocprofv3 --kernel-trace -d $out/prof_$c -o t -- python $R/bench.py --config $c --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1)
  python tools/rocpd_stats.py $out/prof_$c/t_results.db > $out/${tag}_bench_$c.kernel_stats.txt
done
rm -rf $out/prof_*
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/${tag}_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().split('\n')[-1])
        print(f.split('_bench_')[1][:-5], round(d['ms_per_step'], 3), 'rollout', round(d['rollout']['ms'], 3), d['rollout']['kernel'], 'frac', round(d['roofline']['frac'], 3), 'update', round(d['roofline']['update']['ms'], 3))
    except Exception as e:
        print(f, 'FAILED', e)
PY
