#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 evidence of round 2, copied to gpurun_out/profiles_r02 for commit under profiles/.
#   gpurun -- 'bash scripts/profile_r02.sh'
# kernel-trace/stats and every PMC group are SEPARATE passes (counters are never combined with API traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 40 --launch-batches 20 --warmup 20 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- $BENCH > $OUT/pmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_bench_summary.json $OUT/pmc_* > $OUT/pmc_bench_summary.txt 2>&1
# the other BASELINE configs: per-kernel stats + MFMA-busy counters of cin_kernel / din_score_kernel / cross kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -o configs -- python $ROOT/scripts/bench_configs.py > $OUT/bench_configs.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/cfgpmc -o p -- python $ROOT/scripts/bench_configs.py --quick > $OUT/cfgpmc.log 2>&1
python $ROOT/scripts/pmc_summary.py $OUT/pmc_configs_summary.json $OUT/cfgpmc > $OUT/pmc_configs_summary.txt 2>&1
python $ROOT/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
# keep only summaries + stats (traces are large)
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
du -sh $OUT; ls $OUT
