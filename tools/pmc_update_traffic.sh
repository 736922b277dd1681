cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmct
i=0
for g in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmct/g$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  i=$((i+1))
done
cd $R
for f in $(find gpurun_out/pmct -name '*counter_collection.csv' | sort); do python tools/pmc_summary.py $f 8; done
