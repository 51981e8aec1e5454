#!/bin/bash
# round 5, closing pass: the GPU suite, smoke, then the bench line with its rocprofv3 kernel statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_final
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > gpurun_out/r05_final/pytest.log 2>&1
tail -4 gpurun_out/r05_final/pytest.log | cut -c1-300; grep -n "^FAILED" gpurun_out/r05_final/pytest.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_final/smoke.log 2>&1; tail -1 gpurun_out/r05_final/smoke.log | cut -c1-200
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final/bench_steps20.json 2> gpurun_out/r05_final/bench_steps20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_final/bench_steps20.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.5f  frac %.3f  long_run ratio %.3f  e2e %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["long_run"]["ratio_to_value_region_rate"],
      {k: round(v["ratio_to_value"], 3) for k, v in d["predict_e2e"]["legs"].items()}))
PY
