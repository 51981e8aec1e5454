#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace/stats of bench.py and of the GPU test-suite, copied to
# gpurun_out/ so that the summaries can be committed under profiles/ (named per round).
#   gpurun -- 'bash scripts/profile_round.sh r01'
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --steps 128 --warmup 16 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
# separate PMC passes (never combined with API traces): read / write bytes at the L2 memory side
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python $ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-graph > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python $ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-graph > $OUT/pmc_write.log 2>&1
python $ROOT/bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
ls -R $OUT | head -40
