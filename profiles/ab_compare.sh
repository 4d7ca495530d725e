#!/bin/bash
# Same-box A/B of two builds of libslfp_hip.so (box-to-box spread on the gpurun pool is +-2-4 %, larger than
# most single optimisations, so "before/after" numbers from different calls are not comparable).
#
#   (build container)  bash profiles/ab_compare.sh prepare <git-rev>     # worktree _ab_old/ at <rev>, built
#   (gpurun)           bash profiles/ab_compare.sh run [bench.py args]   # alternates old/new 3 times
#   (build container)  bash profiles/ab_compare.sh clean
#
# Both sides run the OLD checkout's bench.py (same allocation order, same script), only the library differs.
set -e
case "$1" in
  prepare)
    git worktree add -f _ab_old "$2" -q
    (cd _ab_old && python -m cnns_slfp_quantization_amd.build > /dev/null)
    cp _ab_old/cnns_slfp_quantization_amd/libslfp_hip.so _ab_old/cnns_slfp_quantization_amd/libslfp_hip_old.so
    python -m cnns_slfp_quantization_amd.build > /dev/null
    cp cnns_slfp_quantization_amd/libslfp_hip.so _ab_old/cnns_slfp_quantization_amd/libslfp_hip_new.so
    ;;
  run)
    shift
    cd _ab_old
    for i in 1 2 3; do
      for v in old new; do
        cp cnns_slfp_quantization_amd/libslfp_hip_$v.so cnns_slfp_quantization_amd/libslfp_hip.so
        python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-whole-net --passes 1 "$@" 2>/dev/null | \
          python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
      done
    done
    ;;
  clean)
    git worktree remove --force _ab_old; git worktree prune
    ;;
esac
