#!/bin/bash
# Round 3: per-layer VGG-16 (batch 128) times for forced dense tilings and two / three weight buffers (runs on the GPU box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "X=0" "SLFP_DENSE_NWB=2" "SLFP_DENSE_CFG=244" "SLFP_DENSE_CFG=424" "SLFP_DENSE_CFG=424 SLFP_DENSE_NWB=2" "SLFP_DENSE_CFG=422" "SLFP_DENSE_CFG=812"; do
  echo "== $v"
  env $v python bench.py --net vgg16_224 --batch 128 --steps 3 --warmup 1 --per-layer --no-other-configs --no-cpu-baseline --no-whole-net 2>&1 >/dev/null | grep dense_mfma | sed 's/dense_mfma_f16x1//; s/ k3 s1//; s/ GB\/s//' | awk '{printf "%s%s@%s %.0fus | ", $1, $2, $3, $5*1000} END {print ""}'
done
