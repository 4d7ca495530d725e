#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the same passes as run_profile.sh for an arbitrary python script of this repo.
#   profiles/run_profile_cmd.sh TAG profiles/codes_layers.py --reps 3
# Raw output: gpurun_out/prof_TAG/ (scratch); condense with `python profiles/summarize.py TAG "title"`.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
SCRIPT=$R/$1; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $SCRIPT $ARGS > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $SCRIPT $ARGS > $OUT/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $SCRIPT $ARGS > $OUT/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 $SCRIPT $ARGS > $OUT/sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $SCRIPT $ARGS > $OUT/sq2.log 2>&1 || echo "sq2 pass failed"
du -sh $OUT
