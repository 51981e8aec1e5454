#!/bin/bash
# round 6, session 5: fit()'s loss with every step's l2 penalties summed by the optimizer launch (dctr_opt_multi_l2): fit / train tests,
# the two fit-fuzz seeds whose loss missed its bar under the two-ends average (569, 581), step times with the penalty sum in the launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_train.py tests/test_gpu_train_reg.py tests/test_gpu_din_train.py tests/test_gpu_rank_path.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_train.log 2>&1
tail -2 $O/pytest_train.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_train.log | cut -c1-300 | head -20
export DCTR_FUZZ_SEEDS=1 DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in list(range(0,160))+[569,581]+list(range(480,640))))")
timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -1 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fitfuzz.log | cut -c1-300 | head -20
for m in DeepFM xDeepFM; do python scripts/bench_train.py --model $m --batches 4096 2>&1 | grep -v "amdgpu.ids\|parameterization"; done | tee $O/train_steps.log
