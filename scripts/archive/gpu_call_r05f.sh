#!/bin/bash
# same-box A/B: record-form tables vs plain tables behind chain_kernel (K = 20 and K = 256), two rounds interleaved
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $ROOT
for r in 1 2; do
  for k in 20 256; do
    python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline > $OUT/rec_k${k}_$r.json 2>/dev/null
    python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline --no-records > $OUT/plain_k${k}_$r.json 2>/dev/null
  done
done
python - <<'P'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r05f/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-22s value %7.1f M  frac %.3f  us/launch %8.1f  regions %s"%(os.path.basename(f), j["value"]/1e6, j["roofline"]["frac"], j["roofline"]["us_per_launch"], " ".join("%.4f"%x for x in j["regions_ms"])))
P
