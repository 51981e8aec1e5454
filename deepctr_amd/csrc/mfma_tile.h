// Row-tile GEMM on the f32-input matrix cores (v_mfma_f32_16x16x4_f32): exact fp32 (bitwise a k-ordered
// fmaf chain), 64 FLOP/clk/SIMD = the fp32 vector peak but with one VGPR per operand and the VALU left
// free for the epilogue.  gfx950 has no xf32/TF32 path; logits must hold 1e-4 vs the reference, so the
// dense parts of the path (DNN, CrossNet-matrix, DIN attention MLP, CIN) stay in exact fp32.
//
// Shape handled by one wave:  C[16 x 16*TPW] += A[16 x K] * B[K x (16*TPW columns)]
//   * A lives in LDS, row-major with row stride `lda` floats, zero-padded to 4*KQ columns (KQ % 4 == 0).
//   * MFMA operand layout (16x16x4): lane l supplies A[i = l&15][k-slot g = l>>4] and B[g][j = l&15].
//     The four k-slots of one MFMA are NOT adjacent k: slot g walks k = g*KQ + t, so a lane reads
//     CONTIGUOUS k from LDS (one ds_read_b128 feeds 4 MFMAs per column tile).
//   * Column mapping: lane j owns columns n_base + TPW*j + c (c < TPW), so one global_load_dwordx{TPW}
//     of the Keras-layout weight row W[k, :] feeds TPW MFMAs.
//   * C/D layout: acc[c][r] = C[row = 4*g + r][col = n_base + TPW*j + c].
#pragma once
#include <hip/hip_runtime.h>

namespace dctr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int pad16(int k) { return (k + 15) & ~15; }

template <int TPW>
__device__ __forceinline__ void load_cols(const float* p, float (&b)[TPW]) {
    if constexpr (TPW == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
    } else if constexpr (TPW == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        b[0] = t.x; b[1] = t.y;
    } else {
        b[0] = *p;
    }
}

// W in Keras layout [K, N] row-major (y = x W).  Columns >= N are clamped for the load (results for
// them are discarded by the caller); rows k >= K are clamped too (A is zero there).
template <int TPW>
__device__ __forceinline__ void tile_gemm_kn(const float* A, int lda, int K, int KQ, const float* __restrict__ W, int N,
                                             int n_base, f32x4 (&acc)[TPW]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const float* arow = A + j * lda + g * KQ;
    int n0 = n_base + TPW * j;
    if (n0 + TPW > N) n0 = N - TPW;  // TPW > 1 only when N % (16*TPW) == 0, so this clamps TPW == 1 tails
    const float* wcol = W + n0;
    const int k_last = K - 1;
#pragma unroll 2
    for (int t0 = 0; t0 < KQ; t0 += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + t0);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float b[4][TPW];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int k = min(g * KQ + t0 + tt, k_last);
            load_cols<TPW>(wcol + (int64_t)k * N, b[tt]);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b[tt][c], acc[c], 0, 0, 0);
        }
    }
}

// W transposed [N, K] row-major (out_n = sum_k W[n,k] x_k — CrossNet 'matrix', interaction.py:418).
template <int TPW>
__device__ __forceinline__ void tile_gemm_nk(const float* A, int lda, int K, int KQ, const float* __restrict__ W, int N,
                                             int n_base, f32x4 (&acc)[TPW]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const float* arow = A + j * lda + g * KQ;
    const int k_last = K - 1;
    const float* wrow[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) wrow[c] = W + (int64_t)min(n_base + TPW * j + c, N - 1) * K;
    for (int t0 = 0; t0 < KQ; t0 += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + t0);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float b[4][TPW];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int k = min(g * KQ + t0 + tt, k_last);
#pragma unroll
            for (int c = 0; c < TPW; ++c) b[tt][c] = wrow[c][k];
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b[tt][c], acc[c], 0, 0, 0);
        }
    }
}

// sigmoid on the hardware transcendental units: v_exp_f32 (via __expf) and the RAW v_rcp_f32 (__builtin_amdgcn_rcpf, 1 ulp), ~2 ulp
// together; the IEEE expf + division pair costs ~40 VALU instructions = 160 cycles per wave64 element, which dominated the Dice
// epilogues.  (Until round 4 this read __frcp_rn, which hipcc expands to the correctly rounded division — v_div_scale, v_rcp, three
// Newton steps, v_div_fmas, v_div_fixup: ten instructions where one was meant.)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// tanh on the same units: |x| < 1/4 by its odd series through x^9 (next term < 1e-8 of the result), else 1 - 2 / (1 + e^{2|x|}) with
// the sign put back — a few fp32 ulp of the result, a dozen instructions where libm's tanhf is ~40 with branches
__device__ __forceinline__ float tanh_fast(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float small = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
    const float big = fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __expf(2.f * ax)), 1.f);
    return ax < 0.25f ? small : copysignf(big, x);
}

// activation codes = DCTR_ACT_* (include/dctr.h)
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return sigmoidf_(v);
        case 3: return tanhf(v);
        default: return v;
    }
}

// Dice, inference form (reference layers/activation.py:59-64):
//   x_p = sigmoid((x - mean) * rsqrt(var + eps));  y = alpha*(1-x_p)*x + x_p*x
__device__ __forceinline__ float dice_pre(float v, float alpha, float inv, float shift) {   // inv, shift per column
    const float xp = sigmoidf_(v * inv + shift);
    return alpha * (1.f - xp) * v + xp * v;
}
__device__ __forceinline__ float dice_act(float v, float alpha, float mean, float var, float eps) {
    const float inv = 1.f / sqrtf(var + eps);
    return dice_pre(v, alpha, inv, -mean * inv);
}

}  // namespace dctr
