#!/bin/bash
# Round 3, third part: per-kernel statistics of the training steps on the library's own GEMM (where the step time goes).
#   gpurun -- 'bash scripts/profile_r03c.sh'
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
