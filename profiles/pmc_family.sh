#!/bin/bash
# Runs ON THE GPU BOX: PMC counters (separate passes) for the layers of ONE kernel family, via profiles/variants.py.
#   bash profiles/pmc_family.sh <tag> <family> [variants.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; FAM=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -- python3 $R/profiles/variants.py --family $FAM --rounds 2 "$@" > $OUT/$name.log 2>&1 || echo "$name pass failed"
done
python3 $R/profiles/pmc_family_summary.py $OUT
