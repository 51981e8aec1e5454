#!/bin/bash
# round 6, last tree: the whole GPU suite with every torch.empty poisoned (NaN / max-int), and the routes added after r06u with torch's
# caching allocator off (every tensor its own hipMalloc: an out-of-bounds access faults)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ad; mkdir -p $O
DCTR_POISON_EMPTY=1 timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=line -rf > $O/pytest_poison.log 2>&1
echo "poison rc=$?"; tail -1 $O/pytest_poison.log | cut -c1-250; grep -n "^FAILED" $O/pytest_poison.log | head
K="past_every_tile or past_the_register_file or cin_bwd or linear or criteo_width or any_key_width or every_steps_l2 or trains_alike"
export DCTR_FUZZ_FIT_SEEDS=14,28,42,49,91,112,119,306,312,329,569,581,5,20,35,50
PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_train.py tests/test_gpu_fit.py -q -m gpu -p no:cacheprovider --tb=line -rf -k "$K" > $O/pytest_nocache.log 2>&1
echo "no-caching rc=$?"; tail -1 $O/pytest_nocache.log | cut -c1-250; grep -n "fault\|Abort\|Fatal Python\|^FAILED" $O/pytest_nocache.log | head -5
