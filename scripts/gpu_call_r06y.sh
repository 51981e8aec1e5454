#!/bin/bash
# round 6, last session: model / DIN fuzz over fresh seeds (4400 .. 5399 / 600 .. 999) on the ABI-13 tree, per-test timeout
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
export DCTR_FUZZ_FIT_SEEDS=1
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(4400,5400)))")
export DCTR_FUZZ_DIN_SEEDS=$(python -c "print(','.join(str(i) for i in range(600,1000)))")
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "matches_the_oracle and random" --timeout 120 --timeout-method=thread > $O/pytest_fuzz.log 2>&1
tail -1 $O/pytest_fuzz.log | cut -c1-300; grep -n "^FAILED\|Timeout" $O/pytest_fuzz.log | cut -c1-300 | head -30
