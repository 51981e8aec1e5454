#!/bin/bash
# round 6, second pass: layered DNN route, ask-before-launch, data-parallel fit with batch statistics, keras evaluate semantics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fit.py tests/test_gpu_rank_path.py tests/test_gpu_models.py -q -p no:cacheprovider --tb=short -rf -x > $O/pytest_a.log 2>&1
tail -15 $O/pytest_a.log | cut -c1-600
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -p no:cacheprovider --tb=short -rf > $O/pytest_fuzz.log 2>&1
tail -30 $O/pytest_fuzz.log | cut -c1-400
