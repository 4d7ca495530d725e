#!/bin/bash
# Same-box, same-binary A/B of an environment switch of libslfp_hip.so (e.g. SLFP_LONG_ENCODE=1 = the long-form
# quantizer instead of the threshold table):   bash profiles/ab_env.sh SLFP_LONG_ENCODE [bench.py args]
# Alternates unset / set three times (box-to-box spread on the pool is larger than most single optimisations).
VAR="$1"; shift
for i in 1 2 3; do
  for v in off on; do
    if [ "$v" = on ]; then export "$VAR"=1; else unset "$VAR"; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-whole-net --passes 1 "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], {k:(v['ms_per_step'], v['GB/s']) for k,v in d['kernels'].items()})"
  done
done
