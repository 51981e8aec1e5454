#!/bin/bash
# Same-box A/B of library builds (boxes of the pool differ by a few percent): every build listed is copied over
# deepctr_amd/lib/libdctr_hip.so in turn and the bench line taken, REPS rounds interleaved.
#   gpurun -- 'bash scripts/ab_libs.sh <out-subdir> "<name>=<path.so> ..." [reps]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
LIBS=$2
REPS=${3:-2}
mkdir -p $OUT
cp $ROOT/deepctr_amd/lib/libdctr_hip.so $OUT/_current.so
for r in $(seq 1 $REPS); do
  for kv in $LIBS; do
    name=${kv%%=*}; path=${kv#*=}
    if [ "$path" = "deepctr_amd/lib/libdctr_hip.so" ]; then cp $OUT/_current.so $ROOT/deepctr_amd/lib/libdctr_hip.so; else cp $ROOT/$path $ROOT/deepctr_amd/lib/libdctr_hip.so; fi
    for K in 20 256; do
      W=$([ $K = 20 ] && echo 5 || echo 32)
      python $ROOT/bench.py --steps $K --warmup $W --no-cpu-baseline --no-secondary > $OUT/${name}_k${K}_r${r}.json 2> $OUT/${name}_k${K}_r${r}.err
      python - <<PY
import json
d=json.loads(open("$OUT/${name}_k${K}_r${r}.json").read().strip().splitlines()[-1])
print("%-8s K=%-3d round $r  %7.1f M samples/s  frac %.4f  %8.1f us per launch  parity %.2e" % ("$name", $K, d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["us_per_launch"], d.get("parity_max_rel") or 0), flush=True)
PY
    done
  done
done | tee $OUT/ab.log
cp $OUT/_current.so $ROOT/deepctr_amd/lib/libdctr_hip.so; rm -f $OUT/_current.so
