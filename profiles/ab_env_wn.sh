#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of launch-time environment switches on the per-layer sweep AND the whole net (2 rounds).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for envs in "$@"; do
    env $envs python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/ab_wn.json 2> gpurun_out/ab_wn.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_wn.json").read().strip().splitlines()[-1])
w = d["whole_net"]
print("[$envs] round $round: sweep", d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()}, "| whole net stock", w["stock_bn_relu"], "fused", w["fused_bn_relu"], "graph", w.get("fused_hipgraph"))
PY
  done
done
