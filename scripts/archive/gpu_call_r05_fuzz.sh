#!/bin/bash
# round 5: the randomised parity tests on the GPU box; FUZZ_K: a -k expression without spaces; FUZZ_FILES: further test files;
# DCTR_POISON_EMPTY=1: torch.empty() filled with NaN / max-int (tests/conftest.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_fuzz
timeout 2400 python -m pytest tests/test_gpu_fuzz.py ${FUZZ_FILES} -q -m gpu -p no:cacheprovider --tb=short -rf --maxfail=60 ${FUZZ_K:+-k $FUZZ_K} > gpurun_out/r05_fuzz/pytest${FUZZ_TAG}.log 2>&1
grep -n "^E  .*Error\|passed\|failed\|fault\|Abort" gpurun_out/r05_fuzz/pytest${FUZZ_TAG}.log | cut -c1-300 | tail -70
