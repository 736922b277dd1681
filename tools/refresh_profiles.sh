#!/bin/bash
# Re-measure every BASELINE config and the C1/C3/C4 kernel traces into gpurun_out/refresh (run on the GPU box; copy what is kept to profiles/).
# usage: tools/refresh_profiles.sh <tag>      e.g.  r02_c
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; out=$R/gpurun_out/refresh; mkdir -p $out
cd $R
python bench.py > $out/${tag}_bench_C1.json 2> $out/C1.err
for c in C0 C0p C2 C2s C3; do python bench.py --config $c --no-cpu-baseline > $out/${tag}_bench_$c.json 2> $out/$c.err; done
python bench.py --config C4 --no-cpu-baseline --steps 3 --warmup 1 > $out/${tag}_bench_C4.json 2> $out/C4.err
for c in C1 C3; do
  (cd /tmp && rocprofv3 --kernel-trace -d $out/prof_$c -o t -- python $R/bench.py --config $c --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1)
  python tools/rocpd_stats.py $out/prof_$c/t_results.db > $out/${tag}_bench_$c.kernel_stats.txt
done
(cd /tmp && rocprofv3 --kernel-trace -d $out/prof_C4 -o t -- python $R/bench.py --config C4 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/rocpd_stats.py $out/prof_C4/t_results.db > $out/${tag}_bench_C4.kernel_stats.txt
rm -rf $out/prof_*
tail -c 400 $out/*.json | head -60
