#!/bin/bash
# Round 3: per-layer VGG-16 (batch 128) times: unrolled 3x3 kernel vs the general one, forced tilings, two / three weight buffers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "X=0" "SLFP_DENSE_GENERIC=1" "SLFP_DENSE_CFG=244" "SLFP_DENSE_CFG=424" "SLFP_DENSE_CFG=421" "SLFP_DENSE_CFG=811" "SLFP_DENSE_CFG=244 SLFP_DENSE_NWB=2"; do
  echo "== $v"
  env $v python bench.py --net vgg16_224 --batch 128 --steps 3 --warmup 1 --per-layer --no-other-configs --no-cpu-baseline --no-whole-net 2>/tmp/err.log | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  grep -E "dense_mfma|stem" /tmp/err.log | awk '{printf "%s %s | ", $(NF-3), $(NF-1)} END {print ""}'
done
