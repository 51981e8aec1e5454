#!/bin/bash
# round 6: DIN attention of any key width (materialised route) — the sweep's DIN seeds, the DIN tests, the attention op tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
export DCTR_FUZZ_SEEDS=2399,789
export DCTR_FUZZ_FIT_SEEDS=295,365,410,569
export DCTR_FUZZ_DIN_SEEDS=295,323,355,365,1,2,3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest_fixes.log 2>&1
tail -4 $O/pytest_fixes.log | cut -c1-300; grep -n "^FAILED\|^E  " $O/pytest_fixes.log | cut -c1-300 | head -20
unset DCTR_FUZZ_SEEDS DCTR_FUZZ_FIT_SEEDS DCTR_FUZZ_DIN_SEEDS
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_din_train.py -q -m gpu -p no:cacheprovider --tb=short -rf -x > $O/pytest_din.log 2>&1
tail -3 $O/pytest_din.log | cut -c1-300
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_lab.cpp -o /tmp/mfma_valu_lab > /dev/null 2>&1 && /tmp/mfma_valu_lab > $O/mfma_valu_lab.log 2>&1; cat $O/mfma_valu_lab.log
