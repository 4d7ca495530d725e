#!/bin/bash
# Runs ON THE GPU BOX: ablation of k_pw_tiled (diagnostic builds, WRONG results by construction): what do the deep pointwise
# layers cost without their HBM loads / W loads / quantizer / MFMAs / stores?   bash profiles/ablate_pw.sh "<flags>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for flags in "$@"; do
  SLFP_EXTRA_HIPCC_FLAGS="$flags" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/abl_build.log 2>&1 || { tail gpurun_out/abl_build.log; exit 1; }
  python bench.py --per-layer --no-cpu-baseline --no-whole-net --no-other-configs > gpurun_out/abl.json 2> gpurun_out/abl.err || { tail gpurun_out/abl.err; exit 1; }
  echo "== [$flags]"; grep "pw_mfma" gpurun_out/abl.err | awk '{print $2 $3, $7, $8}' | tr '\n' ';'; echo
done
python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > /dev/null 2>&1
