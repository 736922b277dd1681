cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmc3/g$i -- python $R/tools/rollout_one.py > /dev/null 2>&1
  i=$((i+1))
done
cd $R
find gpurun_out/pmc3 -name '*counter_collection.csv' | sort | xargs python tools/pmc_summary.py
