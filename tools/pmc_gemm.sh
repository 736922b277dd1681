# rocprofv3 PMC passes of the step-wise GEMM rollout at the C3 share (tools/big_sweep.py C3); run on the GPU box:  bash tools/pmc_gemm.sh [config]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cfg=${1:-C3}
mkdir -p $R/gpurun_out/pmcg
i=0
for g in "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmcg/g$i -- python $R/tools/big_sweep.py $cfg > /dev/null 2>&1
  i=$((i+1))
done
cd $R
for f in $(find gpurun_out/pmcg -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 5; done
