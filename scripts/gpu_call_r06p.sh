#!/bin/bash
# round 6, session 5: the HIP training step past its old width gates (matrix / vector CrossNet of any width, DIN over any key width,
# xDeepFM over embedding_dim > 128) — the new tests and the families they touch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_train.py -q -m gpu -p no:cacheprovider --tb=short -rfs \
  -k "criteo_width or any_key_width or wider_than_128 or crossnet_bwd or dcn_hip or trains_alike or din_hip or xdeepfm" > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED\|^ERROR" $O/pytest.log | cut -c1-300 | head -20
grep "^SKIPPED" $O/pytest.log | cut -c1-260 | sort | uniq -c | sort -rn | head -20
