#!/bin/bash
# rocprofv3 PMC passes (counters only, one group per pass) over a short bench run of a step-wise config: the stream-K ensemble kernel k_mlp_sk.
# usage (GPU box): bash tools/pmc_streamk.sh [C3|C2|C4]  -> gpurun_out/pmc_streamk_<config>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cfg=${1:-C3}
out=$R/gpurun_out/pmcsk_$cfg; rm -rf $out; mkdir -p $out
i=0
for g in "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $out/g$i -- python $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  i=$((i+1))
done
cd $R
{ echo "# rocprofv3 PMC passes over \`python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline\` (tools/pmc_streamk.sh): per-kernel MEANS over the run's dispatches";
  for f in $(find $out -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 6 k_mlp_sk; done; } > gpurun_out/pmc_streamk_$cfg.txt 2>&1
rm -rf $out
cat gpurun_out/pmc_streamk_$cfg.txt
