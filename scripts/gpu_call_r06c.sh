#!/bin/bash
# round 6, third pass: the whole GPU suite (any-width HIP training step, layered DNN), kernel trace of the c2_varlen configuration
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head -40
grep -c "outside the HIP" $O/pytest.log
bash scripts/kstats.sh r06c_varlen python $GRAFT_REPO_ROOT/scripts/bench_configs.py --configs c2_varlen,c2_span 2>&1 | tail -14
tail -6 gpurun_out/kstats_r06c_varlen/run.log
