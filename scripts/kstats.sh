#!/bin/bash
# rocprofv3 per-kernel statistics of a command, on the GPU box:  scripts/kstats.sh <tag> <command ...>
# prints the top kernels and leaves the CSV under gpurun_out/kstats_<tag>/
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- "$@" > $OUT/run.log 2>&1
f=$(ls $OUT/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %6s  avg %10.2f us  total %8.2f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
