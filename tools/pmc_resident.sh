cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcr
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmcr/g$i -- python $R/tools/resident_one.py > $R/gpurun_out/pmcr/log$i.txt 2>&1
  i=$((i+1))
done
cd $R
for f in $(find gpurun_out/pmcr -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 3; done
for f in $(find gpurun_out/pmcr -name '*kernel_trace.csv' | sort | head -1); do python - "$f" <<'PY'
import sys, pandas as pd
d = pd.read_csv(sys.argv[1]); d['us'] = (d['End_Timestamp'] - d['Start_Timestamp']) / 1e3
print(d[d['Kernel_Name'].str.contains('resident')][['Kernel_Name', 'us']].assign(Kernel_Name=lambda x: x['Kernel_Name'].str.slice(0, 40)))
PY
done
