#!/bin/bash
# round 6, last session: fit fuzz over fresh seeds 900 .. 1899 on the final tree (every model kind incl. linear-only features, BN / Dice)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ac; mkdir -p $O
export DCTR_FUZZ_SEEDS=1 DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_FIT_SEEDS=$(python -c "print(','.join(str(i) for i in range(900,1900)))")
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rfs -k "trains_alike" --timeout 120 --timeout-method=thread > $O/pytest_fitfuzz.log 2>&1
tail -1 $O/pytest_fitfuzz.log | cut -c1-300; grep -n "^FAILED\|^E  \|Timeout" $O/pytest_fitfuzz.log | cut -c1-300 | head -40
grep "^SKIPPED" $O/pytest_fitfuzz.log | sed 's/fit fuzz [0-9]* //; s/(.*//; s/bs=[0-9]*//' | cut -c1-120 | sort | uniq -c | sort -rn | head -10
