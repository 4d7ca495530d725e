#!/bin/bash
# A/B of the dense k x k path on BASELINE configs 3-5 + AlexNet: default vs SLFP_DENSE_GENERIC=1 (runs on the GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "X=0" "SLFP_DENSE_GENERIC=1"; do
  for cfg in "vgg16_224 128 8" "resnet50_imagenet224 128 8" "squeezenet1_0_imagenet224 256 7"; do
    set -- $cfg
    echo -n "$v $1: "
    env $v python bench.py --net $1 --batch $2 --qbits $3 --steps 3 --warmup 1 --no-other-configs --no-cpu-baseline --no-whole-net 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
  done
done
