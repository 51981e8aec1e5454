#!/bin/bash
# round 5: the fixtures of oracle/make_golden.py gen_widths() through the GPU model tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_fuzz
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider --tb=short -k "d12 or auto or e80" > gpurun_out/r05_fuzz/pytest_widths.log 2>&1
tail -15 gpurun_out/r05_fuzz/pytest_widths.log | cut -c1-250
