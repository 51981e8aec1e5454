#!/bin/bash
# round 6, fourth pass: after the bf16x3 removal — the chain / model / fuzz tests, the driver's bench command with rocprofv3 kernel statistics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head -40
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06d/bench_steps20.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.5f  frac %.3f traffic %s e2e %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"],
      {k: round(v["ratio_to_value"], 3) for k, v in d["predict_e2e"]["legs"].items()}))
PY
