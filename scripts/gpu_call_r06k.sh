#!/bin/bash
# round 6: the four failures of the third sweep (cut off before its summary) re-run by seed; training tests + step times after the FM rows
# of the scatter kernel moved to dnn_in
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
export DCTR_FUZZ_SEEDS=2436,2437,2438,3126,3127,3128,3546,3547,3548,4086,4087,4088
export DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=1
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "matches_the_oracle and not din" > $O/pytest_seeds.log 2>&1
tail -3 $O/pytest_seeds.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_seeds.log | cut -c1-400 | head -20
unset DCTR_FUZZ_SEEDS DCTR_FUZZ_DIN_SEEDS DCTR_FUZZ_FIT_SEEDS
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_fit.py tests/test_gpu_din_train.py tests/test_gpu_rank_path.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_train.log 2>&1
tail -3 $O/pytest_train.log | cut -c1-300; grep -n "^FAILED" $O/pytest_train.log | head
export DCTR_FUZZ_SEEDS=1
export DCTR_FUZZ_DIN_SEEDS=1
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -3 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fitfuzz.log | head
for m in DeepFM DCN xDeepFM; do python scripts/bench_train.py --model $m --batches 4096 2>&1 | grep -v amdgpu.ids; done | tee $O/train_steps.log
