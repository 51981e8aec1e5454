#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 evidence of a round, written to gpurun_out/profiles_<tag> for commit under profiles/ (<tag>_*).
#   gpurun -- 'bash scripts/profile_round.sh r05'        (the round-4 files came from this script as scripts/profile_r04.sh)
# kernel-trace/stats and every PMC group are SEPARATE passes (counters are never combined with API traces).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B20="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
B256="python $ROOT/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench20 -o bench20 -- $B20 > $OUT/bench20_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench256 -o bench256 -- $B256 > $OUT/bench256_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -o configs -- python $ROOT/scripts/bench_configs.py --configs c1,c2_e64,c3,c3_span,dcn_v,dcn_v_span,dcn_m,dcn_m_span,c4,c4_span --steps 16 > $OUT/configs_under_rocprof.log 2>&1
PM="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --prewarm-ms 10 --regions 3"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -o p -- $PM > $OUT/pmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_bench_summary.json $OUT/pmc_* > $OUT/pmc_bench_summary.txt 2>&1
# PMC on the DCN one-launch span and the DIN folded-lookup call (FETCH / WRITE / MFMA busy)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/cfgpmc_$tag -o p -- python $ROOT/scripts/bench_configs.py --configs dcn_v,dcn_v_span,c4,c4_span --steps 8 > $OUT/cfgpmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_configs_summary.json $OUT/cfgpmc_* > $OUT/pmc_configs_summary.txt 2>&1
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python $ROOT/bench.py > $OUT/bench_steps256.json 2> $OUT/bench_steps256.err
python $ROOT/bench.py --workload c5 --steps 20 --warmup 5 > $OUT/bench_c5_steps20.json 2> $OUT/bench_c5_steps20.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $ROOT/scripts/mfma_valu_lab.cpp -o /tmp/mfma_valu_lab > /dev/null 2>&1 && /tmp/mfma_valu_lab > $OUT/mfma_valu_lab.log 2>&1
python $ROOT/scripts/bench_configs.py > $OUT/bench_configs.log 2>&1
for m in DeepFM DCN DCNM xDeepFM DIN; do python $ROOT/scripts/bench_train.py --model $m --batches $([ $m = DIN ] && echo 2048 || echo 4096) >> $OUT/train_steps.log 2>&1; done
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT | head -50
