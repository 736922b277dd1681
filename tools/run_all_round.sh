cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for c in C1 C0 C0p C0hc C0ho C0sn C0an C0hu C2s; do python bench.py --config $c > gpurun_out/r03_b_bench_$c.json 2>gpurun_out/err_$c.txt; python -c "
import json;d=json.load(open('gpurun_out/r03_b_bench_$c.json'));print('$c',round(d['ms_per_step'],4),round(d['ms_per_step_median'],4),round(d['rollout']['ms'],4),round(d['roofline']['update']['ms'],4),round(d['roofline']['frac'],4),round(d['roofline']['update']['frac'],4),d['rollout']['kernel'])"; done
for c in C2 C3 C4; do python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r03_b_bench_$c.json 2>gpurun_out/err_$c.txt; python -c "
import json;d=json.load(open('gpurun_out/r03_b_bench_$c.json'));print('$c',round(d['ms_per_step'],3),round(d['rollout']['ms'],3),round(d['roofline']['frac'],4))"; done
bash tools/trace_config.sh C1 r03b > /dev/null 2>&1; head -16 gpurun_out/trace_C1_r03b.txt
