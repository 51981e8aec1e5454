#!/bin/bash
# Round 4, first GPU call: the evidence VERDICT r03 item 3 asks for — a PMC pass on gather_fm_kernel ITSELF (large launches, logits-only
# and -> dnn_in, against the pure random-row-read kernels of scripts/gather_bw_lab.cpp) and on chain_kernel in its exploratory bf16x3 mode —
# plus this box's baseline bench lines.  kernel-trace and every PMC group are SEPARATE passes.
#   gpurun -- 'bash scripts/profile_r04a.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_r04a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LAB=$ROOT/scripts/_bin/gather_bw_lab
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for cfg in "16 100000 262144" "32 2000000 131072" "16 100000 4096"; do
  $LAB $cfg >> $OUT/gather_bw_lab.log 2>&1
done
for cfg in "16 100000 262144" "32 2000000 131072"; do
  ctag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gl_${ctag}_stats -o s -- $LAB $cfg > /dev/null 2>&1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/gl_${ctag}_pmc_$tag -o p -- $LAB $cfg > $OUT/gl_${ctag}_pmc_$tag.log 2>&1
  done
  python $ROOT/scripts/pmc_summary.py $OUT/pmc_gather_${ctag}.json $OUT/gl_${ctag}_pmc_* > $OUT/pmc_gather_${ctag}.txt 2>&1
done
# chain_kernel, fp32 and the exploratory bf16x3 instantiation (the bench's exploratory leg), K = 256
PM="python $ROOT/bench.py --no-cpu-baseline --prewarm-ms 10 --regions 2"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/bf3_pmc_$tag -o p -- $PM > $OUT/bf3_pmc_$tag.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT/pmc_chain_bf3.json $OUT/bf3_pmc_* > $OUT/pmc_chain_bf3.txt 2>&1
# this box's baseline
python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python $ROOT/scripts/bench_configs.py --configs c2,c2_span,c3,dcn_v,dcn_m,c4 > $OUT/bench_configs.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +1M -delete
find $OUT -name "*counter_collection.csv" -size +1M -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT; cat $OUT/gather_bw_lab.log; cat $OUT/pmc_gather_16_100000_262144.txt | head -80
