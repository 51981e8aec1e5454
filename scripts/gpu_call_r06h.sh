#!/bin/bash
# round 6: the sweep's findings fixed (more than four extra logit vectors; DIN widths on the HIP step) + a second sweep over new seeds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
export DCTR_FUZZ_SEEDS=690,910,930,1050,1170,789
export DCTR_FUZZ_FIT_SEEDS=295,355,365,410
export DCTR_FUZZ_DIN_SEEDS=1
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_ops.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "fuzz or random or adds_any" > $O/pytest_fixes.log 2>&1
tail -4 $O/pytest_fixes.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fixes.log | cut -c1-300 | head
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(1200,2400)))")
export DCTR_FUZZ_DIN_SEEDS=$(python -c "print(','.join(str(i) for i in range(290,590)))")
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(480,640)))")
timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_fuzz_sweep2.log 2>&1
tail -3 $O/pytest_fuzz_sweep2.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fuzz_sweep2.log | cut -c1-400 | head -60
