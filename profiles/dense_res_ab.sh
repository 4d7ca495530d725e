#!/bin/bash
# A/B of the weights-resident persistent 3x3 kernel (k_dense3x3_res, C_in <= 64) on BASELINE configs 3-5: default vs SLFP_DENSE_NORES=1
# (runs on the GPU box; two rounds per variant, alternating)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in "X=0" "SLFP_DENSE_NORES=1"; do
  for cfg in "vgg16_224 128 8" "resnet50_imagenet224 128 8" "squeezenet1_0_imagenet224 256 7"; do
    set -- $cfg
    echo -n "$v $1: "
    env $v python bench.py --net $1 --batch $2 --qbits $3 --steps 10 --warmup 3 --no-other-configs --no-cpu-baseline --no-whole-net 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d.get('codes_path', {}).get('value') if isinstance(d.get('codes_path'), dict) else None, {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
  done
done
done
