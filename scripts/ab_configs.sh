#!/bin/bash
# Same-box A/B of library builds on scripts/bench_configs.py configurations (see scripts/ab_libs.sh).
#   gpurun -- 'bash scripts/ab_configs.sh <out-subdir> "<name>=<path.so> ..." <configs> [reps]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1
LIBS=$2
CFG=$3
REPS=${4:-2}
mkdir -p $OUT
cp $ROOT/deepctr_amd/lib/libdctr_hip.so $OUT/_current.so
for r in $(seq 1 $REPS); do
  for kv in $LIBS; do
    name=${kv%%=*}; path=${kv#*=}
    if [ "$path" = "deepctr_amd/lib/libdctr_hip.so" ]; then cp $OUT/_current.so $ROOT/deepctr_amd/lib/libdctr_hip.so; else cp $ROOT/$path $ROOT/deepctr_amd/lib/libdctr_hip.so; fi
    python $ROOT/scripts/bench_configs.py --configs $CFG 2>&1 | grep -v amdgpu.ids | sed -e "s/^/$name r$r  /"
  done
done | tee $OUT/ab.log
cp $OUT/_current.so $ROOT/deepctr_amd/lib/libdctr_hip.so; rm -f $OUT/_current.so
