#!/bin/bash
# HBM-side bytes of the stream-K ensemble kernel per launch (rocprofv3 PMC, FETCH_SIZE / WRITE_SIZE / TCC hits in their own passes), with the XCD-aware
# ranges and with the plain blockIdx-ordered split.  usage (GPU box): bash tools/pmc_streamk_traffic.sh [C3|C2|C4] -> gpurun_out/pmc_streamk_traffic_<config>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=${1:-C3}
out=$R/gpurun_out/pmcskt_$cfg; rm -rf $out; mkdir -p $out
for v in xcd plain; do
  i=0
  for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    if [ $v = plain ]; then export METRPO_STREAMK_PLACE=flat; else unset METRPO_STREAMK_PLACE; fi
    timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/${v}_g$i -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
    i=$((i+1))
  done
done
unset METRPO_STREAMK_PLACE
cd $R
{ echo "# rocprofv3 PMC passes over \`python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline\` (tools/pmc_streamk_traffic.sh): per-launch MEANS of k_mlp_sk; FETCH_SIZE / WRITE_SIZE in KiB";
  for v in xcd plain; do echo "## ranges: $v"; for f in $(find $out -path "*${v}_g*" -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 6 k_mlp_sk; done; done; } > gpurun_out/pmc_streamk_traffic_$cfg.txt 2>&1
rm -rf $out
cat gpurun_out/pmc_streamk_traffic_$cfg.txt
