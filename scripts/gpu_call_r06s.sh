#!/bin/bash
# round 6, session 5: the model fuzz over seeds 4400 .. 5199 stopped making progress 693 tests into one process (r06r): again, with a
# per-test timeout that dumps every thread's stack
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
export DCTR_FUZZ_FIT_SEEDS=1 DCTR_FUZZ_DIN_SEEDS=1
export DCTR_FUZZ_SEEDS=$(python -c "print(','.join(str(i) for i in range(4400,5200)))")
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf -k "test_random_configuration_matches_the_oracle" \
   --timeout 100 --timeout-method=thread > $O/pytest_fuzz.log 2>&1
tail -60 $O/pytest_fuzz.log | cut -c1-220
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
