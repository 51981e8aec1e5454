#!/bin/bash
# round 5: the randomised parity test (+ the op tests added with it) on the GPU box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_fuzz
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider --tb=short -rf > gpurun_out/r05_fuzz/pytest.log 2>&1
grep -n "^FAILED\|^E  .*Error\|passed\|failed" gpurun_out/r05_fuzz/pytest.log | tail -40
