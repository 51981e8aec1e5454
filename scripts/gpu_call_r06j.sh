#!/bin/bash
# round 6: third one-off sweep of the randomised parity tests (seeds the suite and the first two sweeps do not hold)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(2400,4400)))")
export DCTR_FUZZ_DIN_SEEDS=$(python -c "print(','.join(str(i) for i in range(590,1090)))")
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(640,900)))")
timeout 3300 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_fuzz_sweep3.log 2>&1
tail -3 $O/pytest_fuzz_sweep3.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fuzz_sweep3.log | cut -c1-400 | head -80
