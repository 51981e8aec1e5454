#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$ROOT/scripts/_bin/gather_bw_lab 16 100000 262144 > $OUT/gather_bw_lab.log 2>&1
$ROOT/scripts/_bin/gather_bw_lab 16 100000 65536 >> $OUT/gather_bw_lab.log 2>&1
cat $OUT/gather_bw_lab.log
for grp in "TCP_TCC_READ_REQ_sum" "FETCH_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$grp -o p -- $ROOT/scripts/_bin/gather_bw_lab 16 100000 262144 > $OUT/pmc_$grp.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_gather_records.json $OUT/pmc_* > $OUT/pmc_gather_records.txt 2>&1
cat $OUT/pmc_gather_records.txt | cut -c1-200
find $OUT -name "*.csv" -size +200k -delete
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; tail -c 300 $OUT/bench20.err
