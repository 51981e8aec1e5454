#!/bin/bash
# Run on the GPU box (via gpurun): PMC counter groups (separate passes, kernel-trace only) for one bench_configs selection.
#   gpurun -- 'bash scripts/pmc_kernel.sh <out-subdir> <configs> [steps]'
# The largest dispatch of every kernel is summarised (scripts/pmc_largest.py): spans, not the per-batch launches beside them.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
CFG=$2
STEPS=${3:-4}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" "FETCH_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- python $ROOT/scripts/bench_configs.py --configs $CFG --steps $STEPS > $OUT/pmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_largest.py $OUT/summary.json $OUT/pmc_* > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*agent_info.csv" -delete
cat $OUT/summary.txt
