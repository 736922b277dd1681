#!/bin/bash
# HBM-side bytes of one kernel family under `python bench.py --config <C> --steps 1 --warmup 1` (rocprofv3 PMC; FETCH_SIZE, WRITE_SIZE and the TCC hit counters in their
# own passes -- MI355X_MICROARCH.md, HBM / rocprofv3).  Prints per-kernel dispatch counts, SUMS and means, so per-step figures can be formed for kernels that run
# many steps per launch.   usage (GPU box): bash tools/pmc_traffic.sh <config> <kernel substring> <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=$1; kern=$2; tag=$3; shift 3
out=$R/gpurun_out/pmct_${cfg}_$tag; rm -rf $out; mkdir -p $out
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  env "$@" timeout 900 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/g$i -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_$i.json 2>/dev/null
  i=$((i+1))
done
cd $R
{ echo "# rocprofv3 PMC over \`$* python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline\` (tools/pmc_traffic.sh), kernels matching '$kern'; FETCH_SIZE / WRITE_SIZE in KiB";
  grep -o '"env steps per rollout [0-9.]*' $out/bench_0.json | head -1;
  for f in $(find $out -name '*counter_collection.csv' | sort); do python - "$f" "$kern" <<'PY'
import sys, pandas as pd
df = pd.read_csv(sys.argv[1]); df = df[df['Kernel_Name'].str.contains(sys.argv[2], regex=False)]
df['k'] = df['Kernel_Name'].str.slice(0, 60)
for (k, c), g in df.groupby(['k', 'Counter_Name']):
    print('%-62s %-14s dispatches %6d  sum %.6g  mean %.6g' % (k, c, g['Dispatch_Id'].nunique(), g['Counter_Value'].sum(), g['Counter_Value'].mean()))
PY
  done; } > gpurun_out/pmc_traffic_${cfg}_$tag.txt 2>&1
rm -rf $out
cat gpurun_out/pmc_traffic_${cfg}_$tag.txt
