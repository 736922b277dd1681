#!/bin/bash
# C1 (the metric's config) under the kernel tracer with the DRIVER's command (--steps 20 --warmup 5) and with the default (40 + 10): per-kernel durations whose sum must stay
# at or below the line's ms_per_step.   usage (GPU box): bash tools/c1_traces.sh <tag>
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; tag=${1:-rXX}; out=$R/gpurun_out/round; mkdir -p $out
for v in "20 5 drv" "40 10 def"; do set -- $v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/prof_C1$3 -o t -- python $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline > $out/${tag}_C1_$3_traced.json 2>/dev/null)
  python tools/rocpd_stats.py /tmp/prof_C1$3/t_results.db > $out/${tag}_C1_$3.kernel_stats.txt
  python tools/rocpd_iter_check.py /tmp/prof_C1$3/t_results.db $2 $1; rm -rf /tmp/prof_C1$3
  python - $out/${tag}_C1_$3_traced.json $out/${tag}_C1_$3.kernel_stats.txt $(( 2 * $1 + $2 + 1 )) <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); n_it = int(sys.argv[3])
tot = other = 0.0
for l in open(sys.argv[2]).read().splitlines()[1:]:
    f = l.split()
    try:
        us = float(f[-5])
    except Exception:
        continue
    if l.startswith(('k_probe', 'void k_probe', 'k_cu_census', 'k_repack')):      # the peak probes / set-up kernels run once, outside the timed region
        other += us
    else:
        tot += us
print('%s: ms_per_step %.4f (median %.4f); kernel time of the iterations %.1f us over %d iterations (timed + warm-up + the instrumented pass + the FVP-timing one) = %.4f ms per iteration; probes / set-up %.1f us'
      % (sys.argv[1].split('/')[-1], d['ms_per_step'], d['ms_per_step_median'], tot, n_it, tot / n_it / 1e3, other))
PY
done
