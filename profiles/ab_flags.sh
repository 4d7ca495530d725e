#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of compile-time switches, bench.py context (cold layer order).
#   bash profiles/ab_flags.sh "<flags A>" "<flags B>" ... ; each variant is built and benched twice, interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for round in 1 2; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    SLFP_EXTRA_HIPCC_FLAGS="$flags" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/ab_flags_build.log 2>&1 || exit 1
    python bench.py --no-cpu-baseline --no-whole-net --no-other-configs > gpurun_out/ab_flags_${i}_$round.json 2> gpurun_out/ab_flags.err || exit 1
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_flags_${i}_$round.json").read().strip().splitlines()[-1])
print("[$flags] round $round:", d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
  done
done
