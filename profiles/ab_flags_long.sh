#!/bin/bash
# Runs ON THE GPU BOX: A/B of compile-time switches with long timed regions (200 steps after 100 warm-up); each variant is built
# once and benched 3 times; the first variant is repeated at the end (drift check).   bash profiles/ab_flags_long.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() {
  SLFP_EXTRA_HIPCC_FLAGS="$1" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/ab_flags_build.log 2>&1 || { tail gpurun_out/ab_flags_build.log; exit 1; }
  for r in 1 2 3; do
    echo -n "[$1] $r: "
    python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-whole-net --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],4) for k,v in d['kernels'].items()})"
  done
}
for flags in "$@"; do run "$flags"; done
run "$1"
python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > /dev/null 2>&1
