#!/bin/bash
# round 5: the whole GPU suite + smoke on the GPU box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_suite
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf -x --maxfail=15 > gpurun_out/r05_suite/pytest.log 2>&1
tail -25 gpurun_out/r05_suite/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_suite/smoke.log 2>&1; tail -2 gpurun_out/r05_suite/smoke.log
