// Plain fp32 GEMMs of the training step (dW = X^T dZ, dH = dZ W^T, the recomputed pre-activations, CrossNet / CrossNetMix
// projections) on v_mfma_f32_16x16x4_f32 — own kernels (gemm_kernels.hip), no BLAS library behind the C ABI.
// Column-major BLAS semantics, so that the call sites read like the algebra they implement:
//     C (m x n, ldc) = op(A) (m x k) * op(B) (k x n) + beta * C,      beta in {0, 1}
// with op = none / transpose of a column-major matrix with leading dimension lda / ldb.  Exact fp32 products, fp32 accumulation
// in the MFMA's order; results of different tile shapes differ in summation order only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dctr_gemm {

enum Op { OP_N = 0, OP_T = 1 };

// 0 on success, a negative DCTR_E_* / positive hipError_t otherwise
int sgemm(hipStream_t stream, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, const float* B, int ldb, float beta,
          float* C, int ldc);
int sgemm_strided_batched(hipStream_t stream, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, int64_t stride_a,
                          const float* B, int ldb, int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c, int batch);

// Several independent products in ONE launch (64 x 64 tiles, beta = 0): the dW = X^T dZ slices of every layer of a DNN.
// ones_last: op(B) holds only n - 1 columns in memory; column n - 1 reads as 1.0 for every k, so that column n - 1 of C is
// sum_k op(A)(:, k) — the bias gradient colsum(dZ) rides on the dW product as one more output row.
struct GroupDesc {
    Op op_a, op_b;
    int m, n, k;
    const float* A;
    int lda;
    int64_t stride_a;
    const float* B;
    int ldb;
    int64_t stride_b;
    float* C;
    int ldc;
    int64_t stride_c;
    int batch;
    int ones_last;
    int accumulate;             // 1: C += the product (beta = 1); with k_slices > 1 and slice_stride_c == 0 the slices add with float atomics
    int k_slices;               // > 1: k is cut into k_slices slices (as dctr_gemm::k_slices counts them) and slice s STORES its partial
    int64_t slice_stride_c;     //      product at C + s * slice_stride_c (the caller sums the slices: deterministic, no atomics)
};
int sgemm_grouped(hipStream_t stream, const GroupDesc* groups, int n_groups);
// number of slices (<= 32, each a multiple of 32 long except the last) a reduction of length k is cut into at ~rows_per_slice
int k_slices(int k, int rows_per_slice);

}  // namespace dctr_gemm
