#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 evidence for the row-chained dctr_embed_mlp_fwd kernel (round 2, second half),
# copied to gpurun_out/profiles_r02b for commit under profiles/ (r02b_*).
#   gpurun -- 'bash scripts/profile_r02b.sh'
# kernel-trace/stats and every PMC group are SEPARATE passes (counters are never combined with API traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r02b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# PMC passes: 4 calls of 16 batches = 65,536 rows each = ONE chain_kernel<2,8> launch per call (one 256-row pass per CU)
BENCH="python $ROOT/bench.py --steps 64 --launch-batches 16 --warmup 16 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench20 -o bench20 -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench20_under_rocprof.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- $BENCH > $OUT/pmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_bench_summary.json $OUT/pmc_* > $OUT/pmc_bench_summary.txt 2>&1
python $ROOT/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT; ls $OUT
