#!/bin/bash
# round 5: the randomised parity tests (+ the op tests added with them) on the GPU box; FUZZ_K: a -k expression without spaces
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_fuzz
timeout 2400 python -m pytest tests/test_gpu_fuzz.py ${FUZZ_FILES} -q -m gpu -p no:cacheprovider --tb=short -rf --maxfail=60 ${FUZZ_K:+-k $FUZZ_K} > gpurun_out/r05_fuzz/pytest.log 2>&1
grep -n "^E  .*Error\|passed\|failed" gpurun_out/r05_fuzz/pytest.log | cut -c1-300 | tail -70
