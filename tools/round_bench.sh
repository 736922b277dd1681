#!/bin/bash
# End-of-pass measurements: bench lines of every config + kernel traces of the configs whose kernels changed.  usage (GPU box): bash tools/round_bench.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; tag=${1:-rXX}; out=$R/gpurun_out/round; mkdir -p $out
for c in C1 C0 C2s C0p C0hc C0ho C0sn C0an C0hu C2 C3; do python bench.py --config $c --no-cpu-baseline > $out/${tag}_bench_$c.json 2> $out/$c.err; done
python bench.py --config C4 --no-cpu-baseline --steps 4 --warmup 1 > $out/${tag}_bench_C4.json 2> $out/C4.err
for c in C0hu C1; do
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_$c -o t -- python $R/bench.py --config $c --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>&1)
  python tools/rocpd_stats.py /tmp/prof_$c/t_results.db > $out/${tag}_bench_$c.kernel_stats.txt; rm -rf /tmp/prof_$c
done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_C4 -o t -- python $R/bench.py --config C4 --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1)
python tools/rocpd_stats.py /tmp/prof_C4/t_results.db > $out/${tag}_bench_C4.kernel_stats.txt; rm -rf /tmp/prof_C4
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/${tag}_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().split('\n')[-1]); u = d['roofline']['update']
        print(f.split('_bench_')[1][:-5], round(d['ms_per_step'], 3), 'rollout', round(d['rollout']['ms'], 3), d['rollout']['kernel'], 'frac', round(d['roofline']['frac'], 3), 'update', round(u['ms'], 3), u['path'], round(u['frac'], 3))
    except Exception as e:
        print(f, 'FAILED', e)
PY
