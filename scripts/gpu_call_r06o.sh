#!/bin/bash
# round 6, session 5: ABI 13 (CIN in slices of d, streaming AFM / inner product, long DIN histories, *_supported queries): the GPU suite,
# then the configurations whose launchers were touched (C3 / DCN-matrix / C4), against what profiles/r06_bench_configs.log holds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED\|^ERROR" $O/pytest.log | cut -c1-300 | head
timeout 600 python scripts/bench_configs.py --configs c3,c3_span,c3_dnn_in,dcn_m,dcn_m_span,c4,c4_span > $O/bench_configs.log 2>&1; tail -12 $O/bench_configs.log | cut -c1-250
