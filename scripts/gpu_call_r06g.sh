#!/bin/bash
# round 6: the randomised parity tests over seeds the suite does not hold (800 model configurations, 200 DIN, 320 fit) — a one-off sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(400,1200)))")
export DCTR_FUZZ_DIN_SEEDS=$(python -c "print(','.join(str(i) for i in range(90,290)))")
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(160,480)))")
timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_fuzz_sweep.log 2>&1
tail -3 $O/pytest_fuzz_sweep.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fuzz_sweep.log | cut -c1-400 | head -60
