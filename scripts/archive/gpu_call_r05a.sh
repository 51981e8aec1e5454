#!/bin/bash
# round-5 first GPU call: long_run diagnosis
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/scripts/longrun_probe.py > $OUT/longrun_probe_plain.log 2>&1
timeout 300 python $ROOT/scripts/longrun_probe.py --events 0 --phases warm,idle,parity,idle_synced,warm2 > $OUT/longrun_probe_noevents.log 2>&1
timeout 300 python $ROOT/scripts/longrun_probe.py --steps 256 --reps 21 > $OUT/longrun_probe_k256.log 2>&1
timeout 400 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/scripts/longrun_probe.py --events 0 --phases warm,idle,parity,idle_synced > $OUT/longrun_probe_traced.log 2>&1
python $ROOT/scripts/longrun_trace.py $OUT/trace > $OUT/longrun_trace_summary.txt 2>&1
find $OUT -name "*.csv" -size +200k -delete
find $OUT -name "*agent_info.csv" -delete
cat $OUT/longrun_probe_plain.log $OUT/longrun_probe_noevents.log $OUT/longrun_probe_k256.log $OUT/longrun_trace_summary.txt
