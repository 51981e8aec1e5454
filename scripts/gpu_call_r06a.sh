#!/bin/bash
# round 6, first pass: bench.py's own launcher (--gpus 2 on one GPU), the driver's command, the c5 workload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rank_path.py -q -p no:cacheprovider --tb=short -rf -k "bench" > $O/pytest_rank.log 2>&1
tail -4 $O/pytest_rank.log | cut -c1-400
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc $?"; tail -3 $O/bench_steps20.err | cut -c1-300
timeout 900 python bench.py --workload c5 --steps 20 --warmup 5 > $O/bench_c5_steps20.json 2> $O/bench_c5_steps20.err; echo "c5 rc $?"; tail -3 $O/bench_c5_steps20.err | cut -c1-300
python bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_gpus8_refused.out 2>&1; echo "gpus8 on 1 GPU rc $? (must be non-zero)"; tail -1 $O/bench_gpus8_refused.out | cut -c1-200
python - <<'PY'
import json
for f in ("bench_steps20", "bench_c5_steps20"):
    try:
        d = json.loads(open("gpurun_out/r06a/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value %.4g  ms/step %.5f  frac %.3f traffic %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"]))
        print("  mixed:", json.dumps(d.get("criteo_vocabularies"))[:300])
        print("  gather_hbm:", json.dumps(d.get("gather_hbm_resident"))[:600])
        print("  cpu:", json.dumps(d.get("cpu_baseline"))[:300])
        for k in d["kernels"]:
            print("   %-120s %8.1f us  frac %.3f" % (k["kernel"][:120], k["us_per_launch"], k["frac"]))
    except Exception as e:
        print(f, "ERR", repr(e))
PY
