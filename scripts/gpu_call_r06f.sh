#!/bin/bash
# round 6, sixth pass: whole GPU suite after the in-launch pooling work
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head -40
