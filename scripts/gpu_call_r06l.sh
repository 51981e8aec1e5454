#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
export DCTR_FUZZ_SEEDS=2509,3199,3619,4159
export DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=1
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "matches_the_oracle and not din" > $O/pytest_seeds.log 2>&1
tail -3 $O/pytest_seeds.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_seeds.log | cut -c1-400 | head -20
