#!/bin/bash
# round 6, session 5: the ABI-13 routes (CIN in slices, streaming AFM / inner product, long DIN histories, wide CrossNet training forms, the
# penalty sum of the optimizer launch) with torch's caching allocator off (every tensor its own hipMalloc: an out-of-bounds access faults)
# and with every torch.empty poisoned
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
K="past_128 or more_maps or past_the_lds or past_4096 or wider_than_128 or criteo_width or any_key_width or crossnet_bwd or every_steps_l2 or test_cin"
PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_train.py tests/test_gpu_fit.py -q -m gpu -p no:cacheprovider --tb=line -rf -k "$K" > $O/pytest_nocache.log 2>&1
echo "no-caching rc=$?"; tail -2 $O/pytest_nocache.log | cut -c1-250; grep -n "fault\|Abort\|Fatal Python" $O/pytest_nocache.log | head -5
DCTR_POISON_EMPTY=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_train.py tests/test_gpu_fit.py -q -m gpu -p no:cacheprovider --tb=line -rf -k "$K" > $O/pytest_poison.log 2>&1
echo "poison rc=$?"; tail -2 $O/pytest_poison.log | cut -c1-250
