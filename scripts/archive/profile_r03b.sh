#!/bin/bash
# Round 3, second half (cin_kernel fold, regularised training step, side-stream dW): evidence for profiles/r03b_*.
#   gpurun -- 'bash scripts/profile_r03b.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -o configs -- python $ROOT/scripts/bench_configs.py --configs c3,c3_span,dcn_v,dcn_m,c4,afm,pnn --steps 16 > $OUT/configs_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_c3_mfma -o p -- python $ROOT/scripts/bench_configs.py --configs c3_span > $OUT/pmc_c3_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c3_fetch -o p -- python $ROOT/scripts/bench_configs.py --configs c3_span > $OUT/pmc_c3_fetch.log 2>&1
cd $ROOT
python scripts/bench_configs.py --configs c1,c2,c2_span,c2_hash,c2_varlen,c2_wide,c3,c3_span,dcn_v,dcn_m,dcn_mix,nfm,afm,pnn,c4 --steps 64 2>&1 | grep -v amdgpu.ids > $OUT/bench_configs.log
( cd scripts && ./_bin/cin_lab && ./_bin/cin_lab 65536 26 16 128 128 ) > $OUT/cin_lab.log 2>&1
for m in DeepFM DeepFMdrop DeepFMbn DCN DCNM DCNMix xDeepFM DIN; do python scripts/bench_train.py --model $m 2>&1 | grep -v amdgpu.ids; done > $OUT/train_steps.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python bench.py --no-cpu-baseline > $OUT/bench_steps256.json 2> $OUT/bench_steps256.err
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +3M -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT
tail -3 $OUT/train_steps.log; cat $OUT/train_steps_no_side_stream.log; cat $OUT/bench_configs.log
