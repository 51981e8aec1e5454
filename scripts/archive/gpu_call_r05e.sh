#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -k "record_form or hashed_at_stage or vs_oracle" 2>&1 | tail -25 > $OUT/tests_records.log; cat $OUT/tests_records.log
python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; tail -c 300 $OUT/bench20.err
python bench.py > $OUT/bench256.json 2> $OUT/bench256.err; tail -c 300 $OUT/bench256.err
