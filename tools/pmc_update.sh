# rocprofv3 PMC passes of the policy-update kernels at C1 (one counter group per pass, --kernel-trace --pmc only); run on the GPU box:
#   bash tools/pmc_update.sh > gpurun_out/pmc_update.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcu
i=0
for g in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmcu/g$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  i=$((i+1))
done
cd $R
for f in $(find gpurun_out/pmcu -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 8; done
