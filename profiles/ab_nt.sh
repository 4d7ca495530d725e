#!/bin/bash
# Runs ON THE GPU BOX: A/B of the `nt` cache-policy hint on the once-through activation streams
# (SLFP_NT bit 0: loads, bit 1: stores; compile-time, so the library is rebuilt per variant).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for nt in ${1:-0 1 2 3}; do
  SLFP_EXTRA_HIPCC_FLAGS="-DSLFP_NT=$nt" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/ab_nt_build_$nt.log 2>&1 || exit 1
  python bench.py --no-cpu-baseline --no-whole-net --no-other-configs > gpurun_out/ab_nt_$nt.json 2> gpurun_out/ab_nt_$nt.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_nt_$nt.json").read().strip().splitlines()[-1])
print("SLFP_NT=$nt", d["value"], {k: (v["ms_per_step"], v["GB/s"]) for k, v in d["kernels"].items()})
PY
done
