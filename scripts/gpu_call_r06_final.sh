#!/bin/bash
# round 6, closing pass: the GPU suite, smoke, rocprofv3 kernel statistics + PMC passes + bench lines + configs + training steps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-250
timeout 3000 bash scripts/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
python - <<'PY'
import json
for f in ("bench_steps20", "bench_steps256", "bench_c5_steps20"):
    try:
        d = json.loads(open("gpurun_out/profiles_r06/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.4g  ms/step %.5f  frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    except Exception as e:
        print(f, "ERR", repr(e))
PY
