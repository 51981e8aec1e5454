#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_chain.py -x -q -k "shared_weight_stream or record_form" 2>&1 | tail -25 > $OUT/tests_ring.log; cat $OUT/tests_ring.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; tail -c 300 $OUT/bench20.err
python - <<'P'
import json,os
j=json.loads(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05h/bench20.json").read().strip().splitlines()[-1])
print("value %.1fM"%(j["value"]/1e6))
for k in j["kernels"]: print("  %-100s %8.2f us frac %.3f"%(k["kernel"][:100],k["us_per_launch"],k["frac"]))
print("per_batch", j["one_launch_per_batch"])
P
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/gputests.log; cat $OUT/gputests.log
