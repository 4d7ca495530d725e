#!/bin/bash
# Runs ON THE GPU BOX: per-family A/B of the `nt` hint (compile-time macros of slfp_device.hpp; library rebuilt per variant).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rebuild() { SLFP_EXTRA_HIPCC_FLAGS="$1" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/ab_nt2_build.log 2>&1; }
for nt in 0 1 2 3; do
  rebuild "-DSLFP_NT_PW=$nt -DSLFP_NT_CODEC=$nt -DSLFP_NT_DENSE=$nt" || exit 1
  echo "=== SLFP_NT_PW = SLFP_NT_CODEC = SLFP_NT_DENSE = $nt"
  python profiles/variants.py --family pw --rounds 5 2> gpurun_out/ab_nt2_pw_$nt.err | tee gpurun_out/ab_nt2_pw_$nt.log || exit 1
  python - <<PY
import torch, json, sys
sys.path.insert(0, "$R")
import bench
from cnns_slfp_quantization_amd import _lib
L = _lib.load()
print("codec", json.dumps(bench.codec_bench(L, torch.device("cuda", 0))))
PY
  python bench.py --net vgg16_224 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-whole-net --no-other-configs > gpurun_out/ab_nt2_vgg_$nt.json 2> gpurun_out/ab_nt2_vgg_$nt.err || exit 1
  python - <<PY
import json
d = json.loads(open("gpurun_out/ab_nt2_vgg_$nt.json").read().strip().splitlines()[-1])
print("vgg16 b128", d["value"], {k: v["ms_per_step"] for k, v in d["kernels"].items()})
PY
done
