# Run on the GPU box (via gpurun): training-step timings and rocprofv3 kernel stats of the HIP step
export TMPDIR=/tmp
OUT=gpurun_out/train
mkdir -p $OUT
for m in DeepFM xDeepFM DCN; do
  python scripts/bench_train.py --steps 50 --batches 4096 --model $m > $OUT/$m.log 2>&1
done
for m in DeepFM xDeepFM; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o t -- python scripts/bench_train.py --steps 30 --batches 4096 --model $m > $OUT/prof_$m.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/DeepFM.log $OUT/xDeepFM.log $OUT/DCN.log | grep "train step"
