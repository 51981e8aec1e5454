# Run on the GPU box (via gpurun): training-step timings and rocprofv3 kernel stats of the HIP step
export TMPDIR=/tmp
OUT=gpurun_out/train
MODELS=${MODELS:-"DeepFM xDeepFM DCN DCNM DCNMix DIN"}
PROF=${PROF-"DeepFM xDeepFM DCN DIN"}      # PROF="" = no rocprofv3 passes
mkdir -p $OUT
for m in $MODELS; do
  python scripts/bench_train.py --steps 50 --model $m --batches ${BATCHES:-$([ $m = DIN ] && echo 2048 || echo 4096)} > $OUT/$m.log 2>&1
done
for m in $PROF; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o t -- python scripts/bench_train.py --steps 30 --model $m --batches $([ $m = DIN ] && echo 2048 || echo 4096) > $OUT/prof_$m.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/*.log | grep "train step"
