#!/bin/bash
# round 6, session 5: fit fuzz over seeds 160 .. 639 with the probe fixed (BatchNormalization / Dice configurations no longer skipped), and
# model / DIN fuzz over fresh seeds 4400 .. 5199 / 600 .. 899 on the ABI-13 library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06r; mkdir -p $O
export DCTR_FUZZ_SEEDS=1
export DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(160,640)))")
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rfs -k "trains_alike" > $O/pytest_fitfuzz.log 2>&1
tail -1 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fitfuzz.log | cut -c1-300 | head -30
grep "^SKIPPED" $O/pytest_fitfuzz.log | sed 's/fit fuzz [0-9]* //; s/(.*//' | cut -c1-120 | sort | uniq -c | sort -rn | head -12
export DCTR_FUZZ_FIT_SEEDS=1
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(4400,5200)))")
export DCTR_FUZZ_DIN_SEEDS=$(python -c "print(','.join(str(i) for i in range(600,900)))")
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "matches_the_oracle and random" > $O/pytest_fuzz.log 2>&1
tail -1 $O/pytest_fuzz.log | cut -c1-300; grep -n "^FAILED" $O/pytest_fuzz.log | cut -c1-300 | head -30
