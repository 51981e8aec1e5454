#!/bin/bash
# A/B of the training step's backward DNN: chained form (mlp_bwd_kernels.hip + grouped dW GEMM) against the layer-by-layer form,
# the k-block of the 64 x 64 GEMM tiling and the rows per dW slice.   gpurun -- 'bash scripts/gemm_lab.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
run() { echo "## $*"; env "$@" python scripts/bench_train.py --model ${MODEL:-DeepFM} --batches ${BATCHES:-4096,16384} 2>&1 | grep -v amdgpu.ids; }
run DCTR_MLP_BWD_CHAIN=0 DCTR_GEMM_BK=16
run DCTR_MLP_BWD_CHAIN=0 DCTR_GEMM_BK=32
run DCTR_MLP_BWD_CHAIN=1 DCTR_GEMM_BK=16 DCTR_DW_ROWS=256
run DCTR_MLP_BWD_CHAIN=1 DCTR_GEMM_BK=32 DCTR_DW_ROWS=128
run DCTR_MLP_BWD_CHAIN=1 DCTR_GEMM_BK=32 DCTR_DW_ROWS=256
run DCTR_MLP_BWD_CHAIN=1 DCTR_GEMM_BK=32 DCTR_DW_ROWS=512
