#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of compile-time switches on BOTH the per-layer sweep and the whole net (where a layer
# reads what the previous one wrote, so cache policy of the stores matters differently).  2 interleaved rounds.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  for flags in "$@"; do
    SLFP_EXTRA_HIPCC_FLAGS="$flags" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/ab_flags_build.log 2>&1 || exit 1
    python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/ab_wn.json 2> gpurun_out/ab_wn.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_wn.json").read().strip().splitlines()[-1])
w = d["whole_net"]
print("[$flags] round $round: sweep", d["value"], "| whole net stock", w["stock_bn_relu"], "fused", w["fused_bn_relu"], "graph", w.get("fused_hipgraph"), "batch8", w.get("batch8"))
PY
  done
done
python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > /dev/null 2>&1
