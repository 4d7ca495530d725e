#!/bin/bash
# Runs ON THE GPU BOX: diagnostic build with in-kernel stamps, then the default library again.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SLFP_EXTRA_HIPCC_FLAGS="-DSLFP_PW_STAMPS ${STAMP_FLAGS:-}" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/stamps_build.log 2>&1 || { tail gpurun_out/stamps_build.log; exit 1; }
python profiles/stamps_tiled.py ${1:-512} ${2:-512}; python profiles/stamps_gap.py ${1:-512} ${2:-512}
python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > /dev/null 2>&1
