#!/bin/bash
# round 6: the scatter's "owner stores" form — training tests, fit fuzz, step times with and without
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_fit.py tests/test_gpu_din_train.py tests/test_gpu_rank_path.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_train.log 2>&1
tail -3 $O/pytest_train.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_train.log | cut -c1-300 | head
export DCTR_FUZZ_SEEDS=1
export DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(0,400)))")
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -3 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fitfuzz.log | cut -c1-300 | head -20
for m in DeepFM DCN DCNM xDeepFM DIN; do python scripts/bench_train.py --model $m --batches $([ $m = DIN ] && echo 2048 || echo 4096) 2>&1 | grep -v "amdgpu.ids\|parameterization"; done | tee $O/train_steps.log
DCTR_NO_OWNER=1 python scripts/bench_train.py --model DeepFM --batches 4096 2>&1 | grep -v amdgpu.ids | sed 's/^/atomics only: /' | tee -a $O/train_steps.log
bash scripts/kstats.sh r06m_train python $GRAFT_REPO_ROOT/scripts/bench_train.py --model DeepFM --batches 4096 2>&1 | tail -14 | tee $O/train_kstats.log
