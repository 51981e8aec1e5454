#!/bin/bash
# round 6: the tree after the reverted scatter experiment — whole GPU suite, then the fit fuzz over seeds 160 .. 479
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head
export DCTR_FUZZ_SEEDS=1
export DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(160,480)))")
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -3 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fitfuzz.log | cut -c1-300 | head -20
