cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print("%-4s %-34s iter %8.3f ms  rollout %8.3f ms  %-20s frac %.4f" % (sys.argv[1],sys.argv[2],d["ms_per_step"],d["rollout"]["ms"],d["rollout"]["kernel"],d["roofline"]["frac"]))'
for cfg in C2 C3; do
  for v in "" "METRPO_PERSIST_NCLOSE=2" "METRPO_PERSIST_NCLOSE=3" "METRPO_PERSIST_NCLOSE=4" "METRPO_PERSIST_NCLOSE=6" "METRPO_PERSIST_NCLOSE=8" "METRPO_PERSIST_WIDE=1" "METRPO_PERSIST_WIDE=0"; do
    env $v python bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$show" $cfg "${v:-default}"
  done; done
