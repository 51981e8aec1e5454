#!/bin/bash
# round 5: same-box A/B of two library builds on the K = 20 bench line (deepctr_amd/lib/ab/{old,new}.so), interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_ab
cp deepctr_amd/lib/libdctr_hip.so /tmp/_cur.so
for r in 1 2 3; do
  for name in old new; do
    cp deepctr_amd/lib/ab/$name.so deepctr_amd/lib/libdctr_hip.so
    timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic > gpurun_out/r05_ab/${name}_r$r.json 2> gpurun_out/r05_ab/${name}_r$r.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_ab/${name}_r$r.json").read().strip().splitlines()[-1])
print("%-4s round $r  %7.1f M samples/s  frac %.4f  %8.1f us per launch" % ("$name", d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["us_per_launch"]), flush=True)
PY
  done
done | tee gpurun_out/r05_ab/ab.log
cp /tmp/_cur.so deepctr_amd/lib/libdctr_hip.so
