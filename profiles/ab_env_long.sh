#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of load-time switches with LONG timed regions (200 steps after 100 warm-up steps: the
# step time settles ~1.3 % below the 20-step figure), 3 interleaved rounds.   bash profiles/ab_env_long.sh "A=1" "B=2 C=3" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "[$v] round $round: "
    env $v python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-whole-net --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"
  done
done
