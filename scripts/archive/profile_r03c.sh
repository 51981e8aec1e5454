#!/bin/bash
# Round 3, third part (training step on the chained DNN backward + touched bytes, DCN fused head): evidence for profiles/r03c_*.
#   gpurun -- 'bash scripts/profile_r03c.sh'          (MODELS / BATCHES / SKIP_PMC=1 narrow it)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r03c
mkdir -p $OUT
cd $ROOT
for m in ${MODELS:-DeepFM xDeepFM DCNM DIN}; do
  bash scripts/kstats.sh train_$m python $ROOT/scripts/bench_train.py --model $m --batches ${BATCHES:-4096} --steps 30 > $OUT/train_${m}_top.txt 2>&1
  cp gpurun_out/kstats_train_$m/*kernel_stats.csv $OUT/train_${m}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/kstats_train_$m/run.log $OUT/train_${m}_run.log 2>/dev/null; rm -rf gpurun_out/kstats_train_$m
  echo "== $m"; cat $OUT/train_${m}_top.txt
done
if [ -z "${SKIP_PMC:-}" ]; then
  cd /tmp && export TMPDIR=/tmp
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_train_$tag -o p -- python $ROOT/scripts/bench_train.py --model DeepFM --batches 4096 --steps 12 > $OUT/pmc_train_$tag.log 2>&1
  done
  cd $ROOT
  python scripts/pmc_summary.py $OUT/pmc_train_summary.json $OUT/pmc_train_FETCH_SIZE $OUT/pmc_train_WRITE_SIZE $OUT/pmc_train_SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE > $OUT/pmc_train_summary.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
fi
for m in DeepFM DeepFMdrop DeepFMbn DCN DCNM DCNMix xDeepFM DIN; do python scripts/bench_train.py --model $m 2>&1 | grep -v amdgpu.ids; done > $OUT/train_steps.log
python scripts/bench_configs.py --configs c1,c2,c2_span,c2_hash,c2_varlen,c2_wide,c3,c3_span,dcn_v,dcn_v_span,dcn_m,dcn_m_span,dcn_mix,nfm,afm,pnn,c4 --steps 64 2>&1 | grep -v amdgpu.ids > $OUT/bench_configs.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python bench.py --no-cpu-baseline > $OUT/bench_steps256.json 2> $OUT/bench_steps256.err
du -sh $OUT; cat $OUT/train_steps.log; cat $OUT/pmc_train_summary.txt 2>/dev/null | head -60
