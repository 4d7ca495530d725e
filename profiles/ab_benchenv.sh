#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of launch-time environment switches in bench.py's layer order (2 interleaved rounds).
#   bash profiles/ab_benchenv.sh "" "SLFP_PW_STREAM_MAXK=64" "A=1 B=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for envs in "$@"; do
    env $envs python bench.py --no-cpu-baseline --no-whole-net --no-other-configs > gpurun_out/ab_be.json 2> gpurun_out/ab_be.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_be.json").read().strip().splitlines()[-1])
print("[$envs] round $round:", d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
  done
done
