#!/bin/bash
# Everything a round's numbers come from, on the GPU box: the GPU suite, the smoke test, a bench line per config and the kernel traces of C1 / C2 / C3 / C4 / Ant / Humanoid
# params files -> gpurun_out/round/<tag>_*; copy what is kept to profiles/.   usage: tools/run_all_round.sh <tag>      e.g.  r04_e
cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-rXX}; out=gpurun_out/round; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
show='import json,sys;d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]);r=d["roofline"];print(sys.argv[2],round(d["ms_per_step"],4),"median",round(d["ms_per_step_median"],4),"rollout",round(d["rollout"]["ms"],4),d["rollout"]["kernel"],round(r["frac"],4),"update",round(r["update"]["ms"],4),round(r["update"]["frac"],4))'
python bench.py > $out/${tag}_bench_C1.json 2> $out/err_C1.txt; python -c "$show" $out/${tag}_bench_C1.json C1
for c in C0 C0p C0hc C0ho C0sn C0an C0hu C2s; do python bench.py --config $c --no-cpu-baseline > $out/${tag}_bench_$c.json 2> $out/err_$c.txt; python -c "$show" $out/${tag}_bench_$c.json $c; done
for c in C2 C3; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $out/${tag}_bench_$c.json 2> $out/err_$c.txt; python -c "$show" $out/${tag}_bench_$c.json $c; done
python bench.py --config C4 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_C4.json 2> $out/err_C4.txt; python -c "$show" $out/${tag}_bench_C4.json C4
# C1 (the metric's config) under the tracer with the DRIVER's command (--steps 20 --warmup 5) and with the default (40 + 10): kernel sums vs ms_per_step
for v in "20 5 drv" "40 10 def"; do set -- $v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_C1$3 -o t -- python $OLDPWD/bench.py --steps $1 --warmup $2 --no-cpu-baseline > $OLDPWD/$out/${tag}_C1_$3_traced.json 2>/dev/null)
  python tools/rocpd_stats.py /tmp/prof_C1$3/t_results.db > $out/${tag}_C1_$3.kernel_stats.txt; rm -rf /tmp/prof_C1$3
done
for c in C2 C3 C0an C0hu C0p; do
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_$c -o t -- python $OLDPWD/bench.py --config $c --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1)
  python tools/rocpd_stats.py /tmp/prof_$c/t_results.db > $out/${tag}_bench_$c.kernel_stats.txt; rm -rf /tmp/prof_$c
done
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_C4 -o t -- python $OLDPWD/bench.py --config C4 --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1)
python tools/rocpd_stats.py /tmp/prof_C4/t_results.db > $out/${tag}_bench_C4.kernel_stats.txt; rm -rf /tmp/prof_C4
head -8 $out/${tag}_C1_drv.kernel_stats.txt
