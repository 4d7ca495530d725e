#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
SLFP_EXTRA_HIPCC_FLAGS="-DSLFP_PW_STAMPS -DSLFP_PW_STAMPS2" python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > gpurun_out/stamps_build.log 2>&1 || { tail gpurun_out/stamps_build.log; exit 1; }
python profiles/stamps_fine.py ${1:-512} ${2:-512}
python -c "from cnns_slfp_quantization_amd import build; build.build(force=True)" > /dev/null 2>&1
