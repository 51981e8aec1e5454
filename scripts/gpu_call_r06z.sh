#!/bin/bash
# round 6, closing pass of the last session (ABI 13 tree): the GPU suite, smoke, rocprofv3 kernel statistics + PMC passes + bench lines +
# configs + training steps -> gpurun_out/profiles_r06z (committed under profiles/ as r06z_*)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --tb=short -rf > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-300; grep -n "^FAILED" $O/pytest.log | cut -c1-300 | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-250
timeout 2400 bash scripts/profile_round.sh r06z > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
python - <<'PY'
import json
for f in ("bench_steps20", "bench_steps256", "bench_c5_steps20"):
    try:
        d = json.loads(open("gpurun_out/profiles_r06z/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.4g  ms/step %.5f  frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
    except Exception as e:
        print(f, "ERR", repr(e))
PY
