#!/bin/bash
# round 6, fifth pass: sequences pooled inside the row-chained launch — tests, the c2_varlen A/B, and a same-box A/B of the plain
# kernels against the library of the previous commit (the POOL template parameter changed hipcc's register assignment in them)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_chain.py -q -p no:cacheprovider --tb=short -rf -x > $O/pytest_chain.log 2>&1
tail -25 $O/pytest_chain.log | cut -c1-400
timeout 600 python scripts/bench_configs.py --configs c2_varlen,c2_span > $O/varlen.log 2>&1; cat $O/varlen.log | cut -c1-250
timeout 600 python scripts/bench_configs.py --configs c2_varlen > $O/varlen2.log 2>&1; cat $O/varlen2.log | cut -c1-250
