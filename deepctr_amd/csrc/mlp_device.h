// Device side of dctr_mlp_fwd / dctr_embed_mlp_fwd: the whole-MLP kernel template (see mlp_kernels.hip for the
// design notes).  Included by one translation unit per row-tile count RT (mlp_kernels_rt{1,2,4}.hip) so that the
// instantiations compile in parallel, and by mlp_kernels.hip for the parameter structs.
#pragma once
#include <type_traits>
#include "dctr_common.h"
#include "embed_device.h"
#include "mfma_tile.h"


namespace dctr_mlp {

constexpr int MAX_LAYERS = 8;
#ifndef DCTR_MLP_WAVES
#define DCTR_MLP_WAVES 8          // waves per 16-row workgroup: 8 = two per SIMD (the second hides the first's waits)
#endif
constexpr int NWAVE = DCTR_MLP_WAVES;
constexpr int NTHR = 64 * NWAVE;

struct MlpParams {
    const float* x;
    int64_t batch;
    int64_t x_stride;
    int32_t in_dim;
    int32_t n_layers;
    int32_t units[MAX_LAYERS];
    const float* W[MAX_LAYERS];
    const float* bias[MAX_LAYERS];
    const float* dice_alpha[MAX_LAYERS];
    const float* dice_mean[MAX_LAYERS];
    const float* dice_var[MAX_LAYERS];
    float dice_eps;
    int32_t activation;
    int32_t has_head;
    int32_t sigmoid_out;
    const float* head_w;
    const float* add[4];
    const float* global_bias;
    float* y;
    int64_t y_stride;
    const float* bn_scale[MAX_LAYERS];   // DNN(use_bn): per-column scale / shift between bias_add and the activation (or NULL)
    const float* bn_shift[MAX_LAYERS];
    float* save[MAX_LAYERS];  // training: layer l's activations [B, units[l]] also go to HBM (NULL = inference)
    unsigned long long* probe;  // measurement aid (NULL normally): {min start, max end} wall-clock stamps of this launch
    int32_t lda;      // LDS row stride (floats) = pad64(max tile width) + 4
    int32_t k_split;  // 0, or the column (multiple of 64) at which the layer-0 input tile is built in two halves
    // backward chain (mlp_bwd_kernels.hip, ACT_BWD): the output of layer l is multiplied by act'(deriv_h[l][b, n]) — the forward's
    // saved activation of that width — before it feeds the next layer (NULL: passed on as it is); deriv_act = the FORWARD's activation
    const float* deriv_h[MAX_LAYERS];
    int32_t deriv_act;
    // CrossNet, vector parameterization, folded into the forward (below): kernels [L, in_dim], bias [L, in_dim], the cross branch's
    // share of the head's Dense(1) kernel [in_dim]; cross_layers = L (0: none, <= CROSS_MAXL)
    const float* cross_w;
    const float* cross_b;
    const float* cross_head;
    const float* cross_const;     // [CROSS_NV] precomputed constants of the recurrence (dctr_crossnet_fold_consts) or NULL
    int32_t cross_layers;
};

// ---- CrossNet (vector parameterization) inside the one-launch forward — reference layers/interaction.py:405-424 (CrossNet.call),
// models/dcn.py:48-66 (Dense(1) over Concatenate([cross_out, deep_out])).
//     x_{l+1} = x_0 (x_l . w_l) + b_l + x_l
// keeps every x_l in span{x_0} + a constant vector: x_l = a_l x_0 + c_l with c_l = b_0 + .. + b_{l-1}, a_0 = 1 and
//     s_l = x_l . w_l = a_l (x_0 . w_l) + c_l . w_l,      a_{l+1} = a_l + s_l,
// so the whole network and the cross branch's share of DCN's final Dense(1), x_L . k_c = a_L (x_0 . k_c) + c_L . k_c, need only the
// L + 1 dot products of the input row with w_0 .. w_{L-1}, k_c — taken from the DNN-input tile while the forward has it on chip — and
// L + 1 row-independent constants.  Same real-number function as the layer-by-layer form (cross_vector_kernel, the oracle); in fp32
// the two differ by rounding only (the dots are fmaf chains over the same 429 products).  Slot l < L holds w_l, slot L holds k_c.
constexpr int CROSS_MAXL = 3;
constexpr int CROSS_NV = CROSS_MAXL + 1;

// the row-independent constants cst[l] = c_l . v_l (v_l = w_l for l < L, k_c for l = L); one wave, result in every lane
__device__ __forceinline__ void cross_constants(const float* w, const float* b, const float* head, int L, int d, int lane,
                                                float (&cst)[CROSS_NV]) {
#pragma unroll
    for (int l = 0; l < CROSS_NV; ++l) cst[l] = 0.f;
    for (int k = lane; k < d; k += 64) {
        float pre = 0.f;                                   // c_l[k]
#pragma unroll
        for (int l = 0; l < CROSS_NV; ++l) {
            if (l <= L) {
                const float v = l < L ? w[(size_t)l * d + k] : head[k];
                cst[l] = fmaf(pre, v, cst[l]);
                if (l < L) pre += b[(size_t)l * d + k];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < CROSS_NV; ++l)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) cst[l] += __shfl_xor(cst[l], m, 64);
}

// the cross branch's logit of one row from its L + 1 dot products
__device__ __forceinline__ float cross_logit(const float (&dots)[CROSS_NV], const float (&cst)[CROSS_NV], int L) {
    float a = 1.f;
#pragma unroll
    for (int l = 0; l < CROSS_MAXL; ++l)
        if (l < L) a += fmaf(a, dots[l], cst[l]);
    float out = 0.f;
#pragma unroll
    for (int l = 0; l < CROSS_NV; ++l)
        if (l == L) out = fmaf(a, dots[l], cst[l]);
    return out;
}

constexpr int ACT_BWD = 64;     // internal epilogue mode of the backward chain (not a DCTR_ACT_* value of the ABI)

// act'(h) from the activation's OUTPUT h (relu, sigmoid, tanh, linear — the forms dctr_mlp_bwd's row kernels use)
__device__ __forceinline__ float act_deriv_from_output(int act, float h) {
    switch (act) {
        case DCTR_ACT_RELU: return h > 0.f ? 1.f : 0.f;
        case DCTR_ACT_SIGMOID: return h * (1.f - h);
        case DCTR_ACT_TANH: return 1.f - h * h;
        default: return 1.f;
    }
}

// gather arguments of the fused path; lpr == 0 selects the plain x-staging path
struct GatherFused : dctr_gather_fm_args_t {
    int32_t fm_logit_used;   // add the FM logit of the gather epilogue to the head
    int32_t lin_logit_used;  // add the linear logit
};
struct FusedGather {
    GatherFused g;
    int32_t lpr;
};

__device__ __forceinline__ int pad64(int k) { return (k + 63) & ~63; }

template <int ACT>
__device__ __forceinline__ float act_t(float v, float al, float mu, float var, float eps) {
    if constexpr (ACT == DCTR_ACT_DICE) return dctr::dice_act(v, al, mu, var, eps);
    else if constexpr (ACT == DCTR_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == DCTR_ACT_SIGMOID) return dctr::sigmoidf_(v);
    else if constexpr (ACT == DCTR_ACT_TANH) return tanhf(v);
    else return v;
}

// B fragments of 8 consecutive k-steps of one wave-tile.  MFMA slot g = lane>>4 takes row k = 4*t + g, so the four
// slots (x the waves' column slices) read 4 ADJACENT weight rows = one contiguous 4 KB; the A tile in LDS is stored
// column-permuted to match (logical column k at (k&3)*KQ + (k>>2)), so a lane still reads its 8 k-steps with two
// ds_read_b128.
// The loads are raw BUFFER loads: the per-lane offset (slot row g, column slice) is constant for the whole tile,
// the row advance is a scalar offset, and rows >= K fall outside num_records and return 0 from the hardware bounds
// check — i.e. ZERO VALU per load.  (With flat 64-bit addressing hipcc spent ~9 VALU incl. two quarter-rate
// v_mad_u64_u32 per load, ~700 issue cycles per 1024-cycle stage that a lone wave per SIMD cannot overlap.)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int TPW>
__device__ __forceinline__ void buf_load_cols(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, float (&b)[TPW]) {
    if constexpr (TPW == 4) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        b[0] = __uint_as_float(t.x); b[1] = __uint_as_float(t.y); b[2] = __uint_as_float(t.z); b[3] = __uint_as_float(t.w);
    } else if constexpr (TPW == 2) {
        const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        b[0] = __uint_as_float(t.x); b[1] = __uint_as_float(t.y);
    } else {
        b[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
}

// stage s covers k-steps t = SD*s .. SD*s+SD-1: row of slot g = 4*t + g -> byte offset (4*t)*N*4 (scalar) + voff (lane)
template <int TPW, int SD>
__device__ __forceinline__ void load_bs(__amdgpu_buffer_rsrc_t rsrc, int voff, int row4_bytes, int s, float (&b)[SD][TPW]) {
#pragma unroll
    for (int tt = 0; tt < SD; ++tt) buf_load_cols<TPW>(rsrc, voff, (SD * s + tt) * row4_bytes, b[tt]);
}

__device__ __forceinline__ int lds_pos(int k, int KQ) { return (k & 3) * KQ + (k >> 2); }

// A fragments of the same SD k-steps: SD/4 ds_read_b128 of the column-permuted LDS tile
template <int SD>
__device__ __forceinline__ void load_as(const float* arow, float (&a)[SD]) {
#pragma unroll
    for (int h = 0; h < SD / 4; ++h) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * h);
        a[4 * h] = v.x; a[4 * h + 1] = v.y; a[4 * h + 2] = v.z; a[4 * h + 3] = v.w;
    }
}

template <int TPW, int RT, int SD>
__device__ __forceinline__ void mfmas(const float (&av)[RT][SD], const float (&b)[SD][TPW], dctr::f32x4 (&acc)[RT][TPW]) {
#pragma unroll
    for (int tt = 0; tt < SD; ++tt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c)
                acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][tt], b[tt][c], acc[rt][c], 0, 0, 0);
}

// C[16*RT x 16*TPW] = A[16*RT x K] * W[K x N] for one wave-tile (every B fragment feeds RT row tiles: the weight
// stream per row drops by RT).  K is walked in stages of 8 k-steps per MFMA slot
// (8*TPW MFMAs); THREE register stages rotate so that the operands of stages s+1 and s+2 are in flight while
// stage s issues its MFMAs (>= 2 x 256*TPW cycles of cover for the L2 latency of the weight stream).  The
// sched_barriers pin the order "issue loads, then MFMAs" — without them hipcc sinks each load next to its first
// use and the wave alternates load-wait / MFMA (measured: 62 cycles per 32-cycle MFMA).
#define DCTR_SB __builtin_amdgcn_sched_barrier(0)
template <int TPW, int RT, int SD>
__device__ __forceinline__ void tile_gemm_pipe(const float* A, int lda, int KQ, int k_rows, const float* __restrict__ W,
                                               int N, int n_base, dctr::f32x4 (&acc)[RT][TPW], int lane = threadIdx.x & 63) {
    // A: LDS tile of 4*KQ (zero-padded) columns in the permuted layout; W: k_rows x N weight rows for those columns
    const int g = lane >> 4, j = lane & 15;
    const float* arow = A + j * lda + g * KQ;
    int n0 = n_base + TPW * j;
    if (n0 + TPW > N) n0 = N - TPW;                       // TPW > 1 only when N % (16*TPW) == 0
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, k_rows * N * 4, 0x00020000);
    const int voff = (g * N + n0) * 4;                    // lane-constant byte offset: slot row g, column slice
    const int row4_bytes = 4 * N * 4;                     // four weight rows
    const int n_it = KQ / SD;                             // >= 2
    const int s_last = n_it - 1;
    float b0[SD][TPW], b1[SD][TPW], b2[SD][TPW];
    float a0[RT][SD], a1[RT][SD], a2[RT][SD];
#define DCTR_STAGE_LOAD(S, AB, BB)                                                      \
    do {                                                                                \
        const int s_ = min((S), s_last);                                                \
        load_bs<TPW, SD>(rsrc, voff, row4_bytes, s_, BB);                               \
        _Pragma("unroll") for (int rt_ = 0; rt_ < RT; ++rt_)                            \
            load_as<SD>(arow + rt_ * 16 * lda + s_ * SD, AB[rt_]);                      \
    } while (0)
    DCTR_STAGE_LOAD(0, a0, b0);
    DCTR_STAGE_LOAD(1, a1, b1);
    for (int it = 0; it < n_it; it += 3) {
        DCTR_STAGE_LOAD(it + 2, a2, b2);
        DCTR_SB;
        mfmas<TPW, RT, SD>(a0, b0, acc);
        DCTR_SB;
        DCTR_STAGE_LOAD(it + 3, a0, b0);
        DCTR_SB;
        if (it + 1 < n_it) mfmas<TPW, RT, SD>(a1, b1, acc);
        DCTR_SB;
        DCTR_STAGE_LOAD(it + 4, a1, b1);
        DCTR_SB;
        if (it + 2 < n_it) mfmas<TPW, RT, SD>(a2, b2, acc);
        DCTR_SB;
    }
#undef DCTR_STAGE_LOAD
}

// k-steps per pipeline stage: 8 with one row tile (two 8-wave workgroups per CU, each wave needs deep cover), 4 with
// two or more row tiles (every B fragment already feeds >= 2x the MFMAs, and the A/B stage registers must stay
// <= 128 VGPRs in total so that two workgroups still share a CU)
template <int RT> struct StageDepth { static constexpr int value = RT == 1 ? 8 : 4; };

// bias + activation of one wave-tile, written to the next layer's LDS tile (permuted layout, K = N)
template <int TPW, int ACT, int RT>
__device__ __forceinline__ void tile_epilogue(const MlpParams& p, int l, float* out, int N, int n_base,
                                              const dctr::f32x4 (&acc)[RT][TPW]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQn = pad64(N) / 4;
    float* const sv = p.save[l];
    const int64_t row0 = (int64_t)blockIdx.x * (16 * RT);
    if constexpr (ACT == ACT_BWD) {
        // dZ_prev = (dZ W^T) .* act'(h_prev): no bias; the product goes to the next layer's tile and (save) to HBM for dW
        const float* const dh = p.deriv_h[l];
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int n = n_base + TPW * j + c;
            if (n < N) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t b = row0 + rt * 16 + 4 * g + r;
                        float v = acc[rt][c][r];
                        if (dh != nullptr) v *= b < p.batch ? act_deriv_from_output(p.deriv_act, dh[b * N + n]) : 0.f;
                        out[(rt * 16 + 4 * g + r) * p.lda + lds_pos(n, KQn)] = v;
                        if (sv != nullptr && b < p.batch) sv[b * N + n] = v;
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int n = n_base + TPW * j + c;
        if (n < N) {
            const float bv = p.bias[l] != nullptr ? p.bias[l][n] : 0.f;
            const bool bn = p.bn_scale[l] != nullptr;
            const float bsc = bn ? p.bn_scale[l][n] : 1.f, bsh = bn ? p.bn_shift[l][n] : 0.f;
            float al = 0.f, mu = 0.f, var = 1.f;
            if constexpr (ACT == DCTR_ACT_DICE) {
                al = p.dice_alpha[l][n];
                mu = p.dice_mean[l][n];
                var = p.dice_var[l][n];
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float z = acc[rt][c][r] + bv;
                    if (bn) z = fmaf(z, bsc, bsh);              // keras: x * inv + (beta - mean * inv)
                    const float v = act_t<ACT>(z, al, mu, var, p.dice_eps);
                    out[(rt * 16 + 4 * g + r) * p.lda + lds_pos(n, KQn)] = v;
                    if (sv != nullptr) {
                        const int64_t b = row0 + rt * 16 + 4 * g + r;
                        if (b < p.batch) sv[b * N + n] = v;
                    }
                }
        }
    }
}

// zero the K padding the NEXT layer reads: columns [N, pad64(N))
template <int RT>
__device__ __forceinline__ void zero_k_padding(const MlpParams& p, float* out, int N) {
    const int npad = pad64(N) - N;
    const int KQn = pad64(N) / 4;
    if (npad > 0) {
        for (int i = threadIdx.x; i < 16 * RT * 64; i += NTHR) {
            const int r = i >> 6, c = i & 63;
            if (c < npad) out[r * p.lda + lds_pos(N + c, KQn)] = 0.f;
        }
    }
}

template <int TPW, int ACT, int RT>
__device__ __forceinline__ void layer_tiles(const MlpParams& p, int l, const float* in, float* out, int K, int N) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    for (int wt = wave; wt < n_tiles; wt += NWAVE) {
        const int n_base = wt * 16 * TPW;
        f32x4 acc[RT][TPW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm_pipe<TPW, RT, StageDepth<RT>::value>(in, p.lda, pad64(K) / 4, K, p.W[l], N, n_base, acc);
        tile_epilogue<TPW, ACT, RT>(p, l, out, N, n_base, acc);
    }
    zero_k_padding<RT>(p, out, N);
}

template <int ACT, int RT>
__device__ __forceinline__ void layer_dispatch(const MlpParams& p, int l, const float* in, float* out, int K, int N) {
    // widest column slice per wave that still gives every wave of the workgroup a tile.  TPW = 4 (96 B-operand
    // registers in the 3-stage pipeline) would push the kernel past 128 VGPRs, i.e. below 4 waves per SIMD = two
    // co-resident workgroups per CU, so 32 columns per wave is the widest slice.
    if (N % 32 == 0 && N >= 32 * NWAVE) layer_tiles<2, ACT, RT>(p, l, in, out, K, N);
    else layer_tiles<1, ACT, RT>(p, l, in, out, K, N);
}

// ---------------------------------------------------------------------------------------------------
// Input producers.  The layer-0 input tile is built in LDS either whole or in two K-halves ("chunks": columns
// [c0, c0 + 4*KQ) of the zero-padded row, stored in the permuted layout with stride KQ); with two halves the
// workgroup needs half the tile and two workgroups of 32 rows share a CU, so one's gather (latency-bound) runs
// under the other's MFMAs.
// ---------------------------------------------------------------------------------------------------
struct Chunk {
    int c0;        // first DNN-input column
    int KQ;        // (padded width) / 4
    int f_lo, f_hi;  // gather fields of this chunk (fused path)
    bool first, last;
};

// The dot products of this chunk's columns of the input tile with the cross vectors, added to xp [CROSS_NV][ROWS] (stored by the
// first chunk).  Wave w takes rows w * RPW ..; lanes walk the tile positions (position i holds local column 4 (i % KQ) + i / KQ).
template <int RT>
__device__ __forceinline__ void cross_partial(const MlpParams& p, const float* tile, const Chunk& ck, float* xp, const float* xv) {
    constexpr int ROWS = 16 * RT, RPW = ROWS / NWAVE;
    static_assert(ROWS % NWAVE == 0, "rows per wave");
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int kpad = pad64(p.in_dim);                       // xv: [CROSS_NV][kpad], zero beyond the layers / the input width
    float acc[RPW][CROSS_NV];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int l = 0; l < CROSS_NV; ++l) acc[r][l] = 0.f;
    for (int i = lane; i < 4 * ck.KQ; i += 64) {
        const int k = ck.c0 + 4 * (i % ck.KQ) + i / ck.KQ;
        float v[CROSS_NV];
#pragma unroll
        for (int l = 0; l < CROSS_NV; ++l) v[l] = xv[l * kpad + k];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float x = tile[(wave * RPW + r) * p.lda + i];
#pragma unroll
            for (int l = 0; l < CROSS_NV; ++l) acc[r][l] = fmaf(x, v[l], acc[r][l]);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int l = 0; l < CROSS_NV; ++l) {
            float a = acc[r][l];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            if (lane == 0) {
                float* q = xp + l * ROWS + wave * RPW + r;
                *q = ck.first ? a : *q + a;
            }
        }
}

// Plain path: x rows from HBM (rows beyond the batch and the K padding are zero).  Division-free mapping: wave w
// takes rows w, w+NWAVE, ...; lanes walk the float4 groups of a row.  Loads are unconditional (clamped address,
// masked afterwards) and all issued before the first LDS store.
template <int RT>
__device__ __forceinline__ void stage_x_chunk(const MlpParams& p, float* tile, int64_t b0, const Chunk& ck) {
    constexpr int ROWS = 16 * RT;
    const int KQc = ck.KQ;
    const int in4 = (p.in_dim + 3) / 4;                        // float4 groups of a real row
    const int c40 = ck.c0 / 4;
    const bool vec = (p.x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15u) == 0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int RPW = ROWS / NWAVE > 0 ? ROWS / NWAVE : 1;   // rows per wave
    for (int q0 = 0; q0 < KQc; q0 += 128) {
        float4 v[RPW][2];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = (wave + rr * NWAVE) & (ROWS - 1);
            const int64_t b = min(b0 + r, p.batch - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c4 = min(c40 + q0 + h * 64 + lane, in4 - 1);
                const float* src = p.x + b * p.x_stride + 4 * c4;
                if (vec) {
                    v[rr][h] = *reinterpret_cast<const float4*>(src);
                } else {
                    const int c = 4 * c4, last = p.in_dim - 1;
                    v[rr][h] = make_float4(src[0], src[min(c + 1, last) - c], src[min(c + 2, last) - c],
                                           src[min(c + 3, last) - c]);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave + rr * NWAVE;
            const bool rowok = r < ROWS && b0 + r < p.batch;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q4 = q0 + h * 64 + lane;                 // float4 group inside the chunk
                const int c = 4 * (c40 + q4);                      // its first real column
                if (r < ROWS && q4 < KQc) {
                    float4 t = v[rr][h];
                    if (!rowok || c >= p.in_dim) t.x = 0.f;        // never let stride padding of x into the tile
                    if (!rowok || c + 1 >= p.in_dim) t.y = 0.f;
                    if (!rowok || c + 2 >= p.in_dim) t.z = 0.f;
                    if (!rowok || c + 3 >= p.in_dim) t.w = 0.f;
                    float* dst = tile + r * p.lda + q4;            // local columns 4*q4+s -> position s*KQc + q4
                    dst[0] = t.x;
                    dst[KQc] = t.y;
                    dst[2 * KQc] = t.z;
                    dst[3 * KQc] = t.w;
                }
            }
        }
    }
    __syncthreads();
}

// Fused path (dctr_embed_mlp_fwd): the tile of the workgroup's samples is GATHERED straight into LDS — embedding
// rows, dense passthrough — and the linear + FM logits of the gather epilogue stay in LDS for the head.  Versus
// dctr_embed_gather_fm + dctr_mlp_fwd this removes the [B, 432] fp32 tile's trip through HBM (1.7 KB written and
// read back per sample), the [B] logit vectors and one kernel launch.  The waves of the workgroup split the FIELDS
// (as the stand-alone gather does at small batch); lane (s, q) owns chunk q of sample s; samples are covered in
// ROWS*LPR/64 passes.  Per-lane partial sums (sum_f e [4], sum_f e^2, linear) are parked in `red` between the two
// K-halves and combined across waves after the last one.
template <int LPR, bool HASH, int RT>
__device__ __forceinline__ void fused_gather_chunk(const MlpParams& p, const GatherFused& g, float* tile, float* red,
                                                   float* extra, int64_t b0, const Chunk& ck) {
    constexpr int VEC = 4;
    constexpr int ROWS = 16 * RT;                             // samples of this workgroup
    constexpr int SPW = 64 / LPR;                             // samples per wave pass
    constexpr int PASSES = SPW >= ROWS ? 1 : ROWS / SPW;
    constexpr int RW = VEC + 2;                                // parked per lane: sum[4], sum of squares, linear
    static_assert(PASSES * 64 <= NTHR, "the combine step gives one wave per pass");
    const int KQc = ck.KQ, c0 = ck.c0;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int s = lane / LPR, q = lane % LPR;

    // columns past the real input (K padding) and rows past the batch are zero; nothing to do for a chunk of real
    // columns in a full workgroup, and only the padding columns otherwise
    {
        const bool full_rows = b0 + ROWS <= g.batch;
        const int real4 = full_rows ? min(max((p.in_dim - c0) / 4, 0), KQc) : 0;    // leading float4 groups that need no fill
        const int nfill = KQc - real4;
        for (int i = threadIdx.x; i < ROWS * nfill; i += NTHR) {
            const int r = i / nfill, c4 = real4 + (i - r * nfill);
            const bool rowok = b0 + r < g.batch;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!rowok || c0 + 4 * c4 + k >= p.in_dim) tile[r * p.lda + k * KQc + c4] = 0.f;
        }
    }

    // dense features (last chunk): lanes 0..ROWS-1 of the last wave take one sample each.  The first 16 values are
    // requested NOW and consumed after the gather, whose two memory round trips hide theirs.
    const bool dense_wave = ck.last && g.n_dense > 0 && wave == NWAVE - 1;
    float dx[16];
    if (dense_wave) {
        const int r = min(lane, ROWS - 1);
        const float* src = g.dense + min(b0 + r, g.batch - 1) * g.dense_stride;
#pragma unroll
        for (int m = 0; m < 16; ++m) dx[m] = src[min(m, g.n_dense - 1)];
    }

    // The wave's work items are (field, pass) pairs: field f_lo + wave + NWAVE*fi, samples pass*SPW + s.  They are
    // processed eight at a time in branch-free phases — 8 id loads, then 8 row + 8 linear loads, then use — so a chunk
    // costs two memory round trips whatever PASSES is (one pass at a time cost 2*PASSES of them).
    static_assert(8 % PASSES == 0, "PASSES divides the batch");
    float sum[PASSES][VEC], sqs[PASSES], lin[PASSES];
    int oor = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const float* rp = red + ((wave * PASSES + ps) * RW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < VEC; ++c) sum[ps][c] = ck.first ? 0.f : rp[c * 64];
        sqs[ps] = ck.first ? 0.f : rp[VEC * 64];
        lin[ps] = ck.first ? 0.f : rp[(VEC + 1) * 64];
    }
    cfield_ptr Fd = (cfield_ptr)g.fields;
    const int f_last = ck.f_hi - 1;
    auto run_batches = [&](auto UC) {
    constexpr int U = decltype(UC)::value, FPB = U / PASSES;   // items, fields per batch
    for (int fb = ck.f_lo + wave; fb < ck.f_hi; fb += NWAVE * FPB) {
        RawId raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(fb + NWAVE * (u / PASSES), f_last);
            const int r = (u % PASSES) * SPW + s;
            const int64_t b = b0 + r;
            const int64_t bb = (r < ROWS && b < g.batch) ? b : 0;
            raw[u] = load_id(g.ids, (int64_t)f * g.ids_stride_f + bb * g.ids_stride_b, g.ids_is_i64);
        }
        FieldRegs fr[FPB];
#pragma unroll
        for (int k = 0; k < FPB; ++k) fr[k] = load_field(Fd, min(fb + NWAVE * k, f_last));
        int64_t row[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const FieldRegs& f = fr[u / PASSES];
            const int r = (u % PASSES) * SPW + s;
            const int64_t b = b0 + r;
            const bool live = r < ROWS && b < g.batch && fb + NWAVE * (u / PASSES) <= f_last;
            int64_t rw = f.identity ? b : id_value(raw[u], g.ids_is_i64);
            if constexpr (HASH) {
                if (f.hash_mode != 0) rw = resolve_row(rw, f.hash_mode, g.ids_is_i64, f.vocab);
            }
            ok[u] = live && (uint64_t)rw < (uint64_t)f.vocab;
            if (live && !ok[u]) oor = 1;
            row[u] = ok[u] ? rw : 0;
        }
        float v[U][VEC], lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const FieldRegs& f = fr[u / PASSES];
            const int qq = (q * VEC < f.dim) ? q * VEC : 0;
            load_vec<VEC>(f.table + row[u] * f.pitch + qq, v[u]);
            const float* lp = f.lin_table != nullptr ? f.lin_table + row[u] * f.lin_pitch : reinterpret_cast<const float*>(g.fields);
            lv[u] = *lp;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const FieldRegs& f = fr[u / PASSES];
            const int ps = u % PASSES;
            const bool act = ok[u] && q * VEC < f.dim;
            if (ok[u] && q == 0 && f.lin_table != nullptr) lin[ps] += lv[u];
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[u][c] = act ? v[u][c] : 0.f;
            if (f.in_fm) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    sum[ps][c] += v[u][c];
                    sqs[ps] = fmaf(v[u][c], v[u][c], sqs[ps]);
                }
            }
            const int r = ps * SPW + s;
            const bool live = r < ROWS && b0 + r < g.batch && fb + NWAVE * (u / PASSES) <= f_last;
            if (f.out_offset >= 0 && live && q * VEC < f.dim) {
                float* dst = tile + r * p.lda + ((f.out_offset + q * VEC - c0) >> 2);   // local col+k -> k*KQc + col/4
                dst[0] = v[u][0];
                dst[KQc] = v[u][1];
                dst[2 * KQc] = v[u][2];
                dst[3 * KQc] = v[u][3];
            }
        }
    }
    };
    // four items per batch (C2: 2 fields x 2 passes per K-half, or 4 fields x 1 pass): eight would push the kernel past
    // 128 VGPRs, i.e. spill around the MFMA loops
    run_batches(std::integral_constant<int, (PASSES <= 4 ? 4 : 8)>{});
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        float* rp = red + ((wave * PASSES + ps) * RW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < VEC; ++c) rp[c * 64] = sum[ps][c];
        rp[VEC * 64] = sqs[ps];
        rp[(VEC + 1) * 64] = lin[ps];
    }
    if (g.status != nullptr && __any(oor) && lane == 0) atomicOr(g.status, (int)DCTR_STATUS_INDEX_OOR);
    if (!ck.last) {
        __syncthreads();
        return;
    }

    // dense features: passthrough + dense . Linear.kernel
    if (dense_wave && lane < ROWS) {
        const int r = lane;
        const bool valid = b0 + r < g.batch;
        const float* src = g.dense + (valid ? b0 + r : 0) * g.dense_stride;
        float dlin = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < g.n_dense && valid) {
                if (g.dense_out_offset >= 0 && m < g.dense_copy_cols)
                    tile[r * p.lda + lds_pos(g.dense_out_offset + m - c0, KQc)] = dx[m];
                dlin = fmaf(dx[m], g.dense_lin_w != nullptr ? g.dense_lin_w[m] : 0.f, dlin);
            }
        }
        for (int k0 = 16; k0 < g.n_dense; k0 += 8) {
            float x[8], w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int k = min(k0 + m, g.n_dense - 1);
                x[m] = src[k];
                w[m] = g.dense_lin_w != nullptr ? g.dense_lin_w[k] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int k = k0 + m;
                if (k < g.n_dense && valid) {
                    if (g.dense_out_offset >= 0 && k < g.dense_copy_cols)
                        tile[r * p.lda + lds_pos(g.dense_out_offset + k - c0, KQc)] = x[m];
                    dlin = fmaf(x[m], w[m], dlin);
                }
            }
        }
        extra[ROWS + r] = dlin;
    }
    __syncthreads();

    // combine the waves' partial sums: FM = 0.5 * (sum_d (sum_f e)^2 - sum_d sum_f e^2), linear = sum of 1-wide rows
    if (threadIdx.x < PASSES * 64) {
        const int pass = threadIdx.x >> 6;
        const int r = pass * SPW + s;
        float S[VEC], Q = 0.f, lin = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) S[c] = 0.f;
        for (int w = 0; w < NWAVE; ++w) {
            const float* rp = red + ((w * PASSES + pass) * RW) * 64 + lane;
#pragma unroll
            for (int c = 0; c < VEC; ++c) S[c] += rp[c * 64];
            Q += rp[VEC * 64];
            lin += rp[(VEC + 1) * 64];
        }
        float fm = -Q;
#pragma unroll
        for (int c = 0; c < VEC; ++c) fm = fmaf(S[c], S[c], fm);
        fm = 0.5f * reduce_lpr<LPR>(fm);
        lin = reduce_lpr<LPR>(lin);
        if (q == 0 && r < ROWS) {
            if (g.n_dense > 0) lin += extra[ROWS + r];
            const int64_t b = b0 + r;
            if (b < g.batch) {
                if (g.fm_logit != nullptr) g.fm_logit[b] = fm;
                if (g.lin_logit != nullptr) g.lin_logit[b] = lin;
            }
            extra[r] = (g.fm_logit_used ? fm : 0.f) + (g.lin_logit_used ? lin : 0.f);
        }
    }
    __syncthreads();     // `red` aliases the tile the first layer writes
}

template <int RT>
__device__ __forceinline__ void produce_chunk(const MlpParams& p, const FusedGather& fg, float* tile, float* red,
                                              float* extra, int64_t b0, const Chunk& ck) {
    if constexpr (RT <= 2) {
        if (fg.lpr != 0) {
            const GatherFused& g = fg.g;
#define DCTR_FUSED(L)                                                                   \
    do {                                                                                \
        if (g.any_hash) fused_gather_chunk<L, true, RT>(p, g, tile, red, extra, b0, ck);  \
        else fused_gather_chunk<L, false, RT>(p, g, tile, red, extra, b0, ck);            \
    } while (0)
            // host: 4 lanes per row up to dim 16, 8 to 32, 16 to 64 (the last only with 16-row workgroups: eight passes
            // of partial sums do not fit the 128-VGPR budget of the 32-row kernel)
            if (fg.lpr == 4) DCTR_FUSED(4);
            else if (fg.lpr == 8) DCTR_FUSED(8);
            else if constexpr (RT == 1) DCTR_FUSED(16);
#undef DCTR_FUSED
            return;
        }
    }
    stage_x_chunk<RT>(p, tile, b0, ck);
}

template <int TPW, int RT>
__device__ __forceinline__ void zero_acc(dctr::f32x4 (&acc)[RT][TPW]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < TPW; ++c) acc[rt][c] = dctr::f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int RT>
__global__ __launch_bounds__(NTHR, RT <= 2 ? 4 : 2) void mlp_kernel(MlpParams p, FusedGather fg) {
    constexpr int ROWS = 16 * RT;
    constexpr int SD = StageDepth<RT>::value;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + ROWS * p.lda;
    float* extra = smem + 2 * ROWS * p.lda;                        // [2*ROWS]: per-row fused logits, dense partials
    float* xp = extra + 2 * ROWS;                                  // [CROSS_NV][ROWS] cross dot products (cross_layers > 0 only)
    float* xcs = xp + CROSS_NV * ROWS;                             // [8] constants of the recurrence
    float* xv = xcs + 8;                                           // [CROSS_NV][pad64(in_dim)] the cross vectors, zero-padded
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (p.probe != nullptr && threadIdx.x == 0) atomicMin(p.probe, (unsigned long long)wall_clock64());

    if (p.cross_layers > 0) {
        // the cross vectors go to LDS while the gather's requests are in flight (published by produce_chunk's barriers); the
        // constants come precomputed (cross_const) or are computed here by the last wave
        const int kpad = pad64(p.in_dim), L = p.cross_layers, d = p.in_dim;
        for (int i = threadIdx.x; i < CROSS_NV * kpad; i += NTHR) {
            const int v = i / kpad, k = i - v * kpad;
            xv[i] = (v <= L && k < d) ? (v < L ? p.cross_w[(size_t)v * d + k] : p.cross_head[k]) : 0.f;
        }
        if (p.cross_const != nullptr) {
            if (threadIdx.x < CROSS_NV) xcs[threadIdx.x] = p.cross_const[threadIdx.x];
        } else if (wave == NWAVE - 1) {
            float cst[CROSS_NV];
            cross_constants(p.cross_w, p.cross_b, p.cross_head, L, d, threadIdx.x & 63, cst);
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int v = 0; v < CROSS_NV; ++v) xcs[v] = cst[v];
            }
        }
    }
    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    int l = 0;
    const int n_fields = fg.lpr != 0 ? fg.g.n_fields : 0;
    if (p.k_split == 0) {
        const Chunk ck{0, pad64(p.in_dim) / 4, 0, n_fields, true, true};
        produce_chunk<RT>(p, fg, buf0, buf1, extra, b0, ck);       // partial sums live in the (still unused) 2nd tile
        if (p.cross_layers > 0) cross_partial<RT>(p, buf0, ck, xp, xv);
    } else {
        // layer 0 in two K-halves: every wave owns ONE wave-tile of the layer output (host guarantees it) and keeps
        // its accumulators in registers while the second half of the input tile replaces the first in LDS
        using dctr::f32x4;
        const int N = p.units[0];
        const bool wide = N % 32 == 0 && N >= 32 * NWAVE;
        f32x4 acc2[RT][2];
        f32x4 acc1[RT][1];
        zero_acc<2, RT>(acc2);
        zero_acc<1, RT>(acc1);
        const int n_tiles = (N + (wide ? 32 : 16) - 1) / (wide ? 32 : 16);
        const int n_base = wave * (wide ? 32 : 16);
        const int kpad = pad64(p.in_dim);
        const int cw0 = p.k_split, cw1 = kpad - p.k_split;
        const int f_mid = fg.lpr != 0 ? fg.g.split_field : 0;
        const float* W0 = p.W[0];
        {
            const Chunk ck{0, cw0 / 4, 0, f_mid, true, false};
            produce_chunk<RT>(p, fg, buf0, buf1, extra, b0, ck);
            if (p.cross_layers > 0) cross_partial<RT>(p, buf0, ck, xp, xv);
            if (wave < n_tiles) {
                if (wide) tile_gemm_pipe<2, RT, SD>(buf0, p.lda, cw0 / 4, cw0, W0, N, n_base, acc2);
                else tile_gemm_pipe<1, RT, SD>(buf0, p.lda, cw0 / 4, cw0, W0, N, n_base, acc1);
            }
            __syncthreads();                                      // buf0 is rebuilt by the second half
        }
        {
            const Chunk ck{cw0, cw1 / 4, f_mid, n_fields, false, true};
            produce_chunk<RT>(p, fg, buf0, buf1, extra, b0, ck);
            if (p.cross_layers > 0) cross_partial<RT>(p, buf0, ck, xp, xv);
            if (wave < n_tiles) {
                const int k_rows = p.in_dim - cw0;
                const float* W = W0 + (size_t)cw0 * N;
                if (wide) tile_gemm_pipe<2, RT, SD>(buf0, p.lda, cw1 / 4, k_rows, W, N, n_base, acc2);
                else tile_gemm_pipe<1, RT, SD>(buf0, p.lda, cw1 / 4, k_rows, W, N, n_base, acc1);
            }
        }
        if (wave < n_tiles) {
#define DCTR_EPI(ACT)                                                              \
    do {                                                                           \
        if (wide) tile_epilogue<2, ACT, RT>(p, 0, buf1, N, n_base, acc2);          \
        else tile_epilogue<1, ACT, RT>(p, 0, buf1, N, n_base, acc1);               \
    } while (0)
            switch (p.activation) {
                case DCTR_ACT_RELU: DCTR_EPI(DCTR_ACT_RELU); break;
                case DCTR_ACT_SIGMOID: DCTR_EPI(DCTR_ACT_SIGMOID); break;
                case DCTR_ACT_TANH: DCTR_EPI(DCTR_ACT_TANH); break;
                case DCTR_ACT_DICE: DCTR_EPI(DCTR_ACT_DICE); break;
                default: DCTR_EPI(DCTR_ACT_LINEAR); break;
            }
#undef DCTR_EPI
        }
        zero_k_padding<RT>(p, buf1, N);
        __syncthreads();
        in = buf1;
        out = buf0;
        K = N;
        l = 1;
    }

    for (; l < p.n_layers; ++l) {
        const int N = p.units[l];
        switch (p.activation) {
            case DCTR_ACT_RELU: layer_dispatch<DCTR_ACT_RELU, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_SIGMOID: layer_dispatch<DCTR_ACT_SIGMOID, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_TANH: layer_dispatch<DCTR_ACT_TANH, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_DICE: layer_dispatch<DCTR_ACT_DICE, RT>(p, l, in, out, K, N); break;
            default: layer_dispatch<DCTR_ACT_LINEAR, RT>(p, l, in, out, K, N); break;
        }
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        K = N;
    }

    if (p.has_head) {
        // logit[row] = h[row,:] . head_w (+ extra logits + global bias), sigmoid for task == binary
        const int part = threadIdx.x & 15;
        const int KQh = pad64(K) / 4;
        float xcst[CROSS_NV];
        if (p.cross_layers > 0) {
#pragma unroll
            for (int v = 0; v < CROSS_NV; ++v) xcst[v] = xcs[v];
        }
        for (int row = threadIdx.x >> 4; row < ROWS; row += NTHR / 16) {
            float acc = 0.f;
            for (int n = part; n < K; n += 16) acc = fmaf(in[row * p.lda + lds_pos(n, KQh)], p.head_w[n], acc);
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            const int64_t b = b0 + row;
            if (part == 0 && b < p.batch) {
                float v = acc;
                if (fg.lpr != 0) v += extra[row];
                if (p.cross_layers > 0) {
                    float dots[CROSS_NV];
#pragma unroll
                    for (int l = 0; l < CROSS_NV; ++l) dots[l] = xp[l * ROWS + row];
                    v += cross_logit(dots, xcst, p.cross_layers);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p.add[i] != nullptr) v += p.add[i][b];
                if (p.global_bias != nullptr) v += p.global_bias[0];
                if (p.sigmoid_out) v = dctr::sigmoidf_(v);
                p.y[b] = v;
            }
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * K; i += NTHR) {
            const int r = i / K, c = i % K;
            const int64_t b = b0 + r;
            if (b < p.batch) p.y[b * p.y_stride + c] = in[r * p.lda + lds_pos(c, pad64(K) / 4)];
        }
    }
    if (p.probe != nullptr && threadIdx.x == 0) atomicMax(p.probe + 1, (unsigned long long)wall_clock64());
}


// ---------------------------------------------------------------------------------------------------
// The 16-row kernel with LDS-DMA weight streams (mlp_kernels_ring.hip) — launches of at most 16 rows per CU, where every CU works on
// 16 rows and the whole DNN's weights (603 KB at C2) must reach every CU.
// mlp_kernel<1> streams each wave's column slice of W from L2 into a three-stage REGISTER pipeline: the bytes a wave keeps in flight
// are bounded by its registers (8 KB), and 8 waves x 8 KB per L2 round trip is ~20 B/clk/CU — the 12.5 us that bound a 4096-row launch
// (§4).  The part itself delivers the same 603 KB to all 256 CUs in 3.7 us (profiles/r05_wstream_lab.log: 40+ B/clk/CU).
// Here every wave pulls ITS OWN column slice (16 TPW columns of 32 weight rows = one chunk of 2 / 4 KiB) by LDS-DMA
// (global_load_lds_dwordx4: no registers; per-lane source addresses, so eight (four) lanes fetch the 128 (64) contiguous bytes of a row's
// slice and one instruction lays 8 (16) rows down as a dense [row][column] image) into a private ring of three chunks in LDS, two chunks
// ahead of its MFMAs.  Nothing is shared between waves: NO barrier inside a layer, a wave waits on its own vmcnt only, and the two waves
// of a SIMD fill each other's LDS / DMA waits.  (A first form with ONE ring shared by the eight waves and a barrier per 32-KiB chunk lost
// ~1,350 cycles per chunk to the lockstep — DMA issue, barrier skew, exposed LDS latency — against 1,024 of MFMA: 30.8 us, slower than
// the kernel it was to replace; profiles/r05_ring_lab.log.)
// B fragments: ds_read_b64 / _b32 from the dense image (rows 4 tt + g of a lane group g lie 128 / 64 B apart: the two groups of a read
// cover 256 / 128 contiguous bytes — conflict-free); A fragments from the activation tile as mlp_kernel reads them.  Same MFMAs on the same
// operands in the same k order as mlp_kernel<RT>: bit-identical results.  Gather front end, epilogues (every activation,
// BatchNormalization, Dice, saved activations), folded CrossNet and head are the tile kernel's own code.
// Shapes: every layer width a multiple of 16 (a wave's slice is whole 16-B pieces), the input tile whole (no layer-0 K split).
constexpr int RING_KS = 8;              // k-steps (of 4 weight rows) per chunk
constexpr int RING_WAVE_F = 2 * 1024;   // floats of LDS per wave: two chunks of 32 rows x 32 columns (a third chunk sits in registers)

// the DMA side of a wave's weight stream for one wave-tile (columns n0 .. n0 + 16 TPW of W [k_rows, N]): per-lane offsets inside a chunk are
// constants (the chunk's base moves as a SCALAR) — fp32 MFMAs run on the vector lanes, every VALU instruction of either wave of a SIMD is
// matrix time lost (~4 cycles each; 40-60 of them per chunk held the first form of this loop at 65 % of the MFMA rate)
template <int TPW>
struct RingStream {
    static constexpr int LPR = 4 * TPW;                 // lanes per row slice (16 B each)
    static constexpr int RPI = 64 / LPR;                // weight rows per DMA instruction
    static constexpr int NI = 32 / RPI;                 // DMA instructions per chunk
    static constexpr int CF = 32 * 16 * TPW;            // floats per chunk
    const float* W;
    float* wring;
    uint32_t row_b, col_b, rl;
    int nch, k_rows;
    uint32_t vq[NI];                                    // per-lane offsets inside a chunk whose 32 rows all exist

    __device__ __forceinline__ RingStream(const float* W_, int N, int k_rows_, int n_base, float* wring_, int lane)
        : W(W_), wring(wring_), k_rows(k_rows_) {
        int n0 = n_base;
        if (n0 + 16 * TPW > N) n0 = N - 16 * TPW;       // (host: N % 16 == 0; TPW = 2 only when N % 32 == 0)
        nch = pad64(k_rows) / (4 * RING_KS);
        rl = (uint32_t)lane / LPR;
        const uint32_t pc = (uint32_t)lane % LPR;
        row_b = (uint32_t)N * 4u;
        col_b = (uint32_t)(n0 + 4 * (int)pc) * 4u;
#pragma unroll
        for (int q = 0; q < NI; ++q) vq[q] = ((uint32_t)(RPI * q) + rl) * row_b + col_b;
    }
    __device__ __forceinline__ void issue(int i, int slot) const {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        if (32 * i + 32 <= k_rows) {                                                           // (scalar branch: no per-lane select)
            const char* Wc = reinterpret_cast<const char*>(W) + (size_t)(32 * i) * row_b;      // (scalar)
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const uint32_t lds_addr = (uint32_t)(size_t)(lds_ptr_t)(wring + slot * CF + q * 256);
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vq[q]), "s"(Wc), "s"(lds_addr) : "memory");
            }
        } else {                     // the zero-padded K tail (the layer's last one or two chunks): rows past K read row K - 1 — x is 0 there
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const uint32_t voff = (uint32_t)min(32 * i + RPI * q + (int)rl, k_rows - 1) * row_b + col_b;
                const uint32_t lds_addr = (uint32_t)(size_t)(lds_ptr_t)(wring + slot * CF + q * 256);
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(W), "s"(lds_addr) : "memory");
            }
        }
    }
    // chunks 0 and 1 -> slots 0 and 1 (the wave's earlier reads of its ring must be complete: the caller waits lgkmcnt(0))
    __device__ __forceinline__ void prologue() const {
        issue(0, 0);
        if (nch > 1) issue(1, 1);
    }
};

// C[16 x 16 TPW] = A[16 x K] * W[K x N] for one wave-tile: A fragments from the activation tile (as tile_gemm_pipe reads them), B
// fragments from the wave's own LDS-DMA ring.  `prefetched`: the caller has issued st.prologue() earlier (under the gather / the previous
// layer's epilogue) and nothing of this loop's accounting was in flight before it.
template <int TPW>
__device__ __forceinline__ void tile_gemm_dma(const float* A, int lda, int KQ, const RingStream<TPW>& st, bool prefetched,
                                              dctr::f32x4 (&acc)[1][TPW], int lane) {
    constexpr int NI = RingStream<TPW>::NI, CF = RingStream<TPW>::CF;
    const int g = lane >> 4, j = lane & 15;
    const int nch = st.nch;
    const float* const bbase0 = st.wring + g * (16 * TPW) + TPW * j;       // two precomputed bases + immediate offsets: no address VALU
    const float* const bbase1 = bbase0 + CF;
    const float* ap = A + j * lda + g * KQ;
    auto frags = [&](int slot, float (&a)[1][RING_KS], float (&b)[RING_KS][TPW]) {
        load_as<RING_KS>(ap, a[0]);
        ap += RING_KS;
        const float* brow = slot ? bbase1 : bbase0;
#pragma unroll
        for (int tt = 0; tt < RING_KS; ++tt) {
            if constexpr (TPW == 2) {
                const float2 v = *reinterpret_cast<const float2*>(brow + 4 * tt * (16 * TPW));
                b[tt][0] = v.x;
                b[tt][1] = v.y;
            } else {
                b[tt][0] = brow[4 * tt * (16 * TPW)];
            }
        }
    };
    // (splitting even / odd k-steps over two accumulator sets — longer distances between dependent MFMAs — measured no gain and would
    //  give up the tile kernel's summation order: one set, the tile kernel's bits)
    auto mfma2 = [&](const float (&a)[1][RING_KS], const float (&b)[RING_KS][TPW]) { mfmas<TPW, 1, RING_KS>(a, b, acc); };
    // Software pipeline, per wave.  Invariant at the top of step i: chunk i sits in REGISTERS (ac, bc); LDS slot (i + 1) % 2 holds
    // chunk i + 1 (landed or landing), slot i % 2 chunk i + 2.  Step i: chunk i + 1 -> the other register set (its LDS latency runs
    // under this step's MFMAs), the MFMAs of chunk i, then the DMA of chunk i + 3 into the slot chunk i + 1 has just left.
    float a0[1][RING_KS], b0[RING_KS][TPW], a1[1][RING_KS], b1[RING_KS][TPW];
    if (!prefetched) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a clean counter: only this loop's DMAs are counted from here on
        st.prologue();
        if (nch > 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunks 0 and 1 (issued long ago) and whatever the epilogue loaded since
    }
    frags(0, a0, b0);
    if (nch > 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st.issue(2, 0);
    }
    auto step = [&](int i, int slot_n, float (&ac)[1][RING_KS], float (&bc)[RING_KS][TPW], float (&an)[1][RING_KS], float (&bn)[RING_KS][TPW]) {
        if (i + 1 < nch) {
            if (i + 2 < nch) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NI) : "memory");      // chunk i + 1 landed (i + 2 may be on its way)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            frags(slot_n, an, bn);
        }
        mfma2(ac, bc);
        if (i + 3 < nch) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (chunk i + 1 is in registers: its slot is free)
            st.issue(i + 3, slot_n);
        }
    };
    for (int i = 0; i < nch; i += 2) {
        step(i, 1, a0, b0, a1, b1);
        if (i + 1 < nch) step(i + 1, 0, a1, b1, a0, b0);
    }
}

// the wave-tile rule of a layer (the tile kernel's: layer_dispatch): 32-column tiles when they give every wave one, else 16
__device__ __forceinline__ bool ring_wide(int N) { return N % 32 == 0 && N >= 32 * NWAVE; }

// chunks 0 and 1 of this wave's FIRST tile of layer l, issued ahead of time (before the gather; under the previous layer's epilogue)
__device__ __forceinline__ void ring_prefetch(const MlpParams& p, int l, float* wring, int wave, int lane) {
    const int N = p.units[l], K = l == 0 ? p.in_dim : p.units[l - 1];
    if (ring_wide(N)) {
        if (wave * 32 < N) RingStream<2>(p.W[l], N, K, wave * 32, wring, lane).prologue();
    } else {
        if (wave * 16 < N) RingStream<1>(p.W[l], N, K, wave * 16, wring, lane).prologue();
    }
}

template <int TPW, int ACT>
__device__ __forceinline__ void layer_tiles_dma(const MlpParams& p, int l, const float* in, float* out, int K, int N, float* wring, bool prefetched) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    for (int wt = wave; wt < n_tiles; wt += NWAVE) {
        const int n_base = wt * 16 * TPW;
        f32x4 acc[1][TPW];
        zero_acc<TPW, 1>(acc);
        const RingStream<TPW> st(p.W[l], N, K, n_base, wring, lane);
        tile_gemm_dma<TPW>(in, p.lda, pad64(K) / 4, st, prefetched && wt == wave, acc, lane);
        if (wt + NWAVE >= n_tiles && l + 1 < p.n_layers) {            // this wave's last tile of the layer: the next layer's first chunks
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // travel under the epilogue and the barrier (the ring is read out)
            ring_prefetch(p, l + 1, wring, wave, lane);
        }
        tile_epilogue<TPW, ACT, 1>(p, l, out, N, n_base, acc);
    }
    if (wave >= n_tiles && l + 1 < p.n_layers) ring_prefetch(p, l + 1, wring, wave, lane);     // (a wave without a tile in this layer)
    zero_k_padding<1>(p, out, N);
}

template <int ACT>
__device__ __forceinline__ void layer_dispatch_dma(const MlpParams& p, int l, const float* in, float* out, int K, int N, float* wring, bool prefetched) {
    if (ring_wide(N)) layer_tiles_dma<2, ACT>(p, l, in, out, K, N, wring, prefetched);
    else layer_tiles_dma<1, ACT>(p, l, in, out, K, N, wring, prefetched);
}

template <int UNUSED = 0>      // (a template so that the header may be included by several translation units; instantiated in mlp_kernels_ring.hip)
__global__ __launch_bounds__(NTHR, 1) void mlp_ring_kernel(MlpParams p, FusedGather fg, int ring_off) {
    constexpr int RT = 1, ROWS = 16;
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + ROWS * p.lda;
    float* extra = smem + 2 * ROWS * p.lda;
    float* xp = extra + 2 * ROWS;
    float* xcs = xp + CROSS_NV * ROWS;
    float* xv = xcs + 8;
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float* wring = smem + ring_off + wave * RING_WAVE_F;
    if (p.probe != nullptr && threadIdx.x == 0) atomicMin(p.probe, (unsigned long long)wall_clock64());

    if (p.cross_layers > 0) {
        const int kpad = pad64(p.in_dim), L = p.cross_layers, d = p.in_dim;
        for (int i = threadIdx.x; i < CROSS_NV * kpad; i += NTHR) {
            const int v = i / kpad, k = i - v * kpad;
            xv[i] = (v <= L && k < d) ? (v < L ? p.cross_w[(size_t)v * d + k] : p.cross_head[k]) : 0.f;
        }
        if (p.cross_const != nullptr) {
            if (threadIdx.x < CROSS_NV) xcs[threadIdx.x] = p.cross_const[threadIdx.x];
        } else if (wave == NWAVE - 1) {
            float cst[CROSS_NV];
            cross_constants(p.cross_w, p.cross_b, p.cross_head, L, d, lane, cst);
            if (lane == 0) {
#pragma unroll
                for (int v = 0; v < CROSS_NV; ++v) xcs[v] = cst[v];
            }
        }
    }
    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    // (layer 0's first chunks are NOT requested ahead of the gather: measured, they lengthen it by more than they save)
    {
        const int n_fields = fg.lpr != 0 ? fg.g.n_fields : 0;
        const Chunk ck{0, pad64(p.in_dim) / 4, 0, n_fields, true, true};
        produce_chunk<RT>(p, fg, buf0, buf1, extra, b0, ck);
        if (p.cross_layers > 0) cross_partial<RT>(p, buf0, ck, xp, xv);
    }
    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.units[l];
        switch (p.activation) {
            case DCTR_ACT_RELU: layer_dispatch_dma<DCTR_ACT_RELU>(p, l, in, out, K, N, wring, l > 0); break;
            case DCTR_ACT_SIGMOID: layer_dispatch_dma<DCTR_ACT_SIGMOID>(p, l, in, out, K, N, wring, l > 0); break;
            case DCTR_ACT_TANH: layer_dispatch_dma<DCTR_ACT_TANH>(p, l, in, out, K, N, wring, l > 0); break;
            case DCTR_ACT_DICE: layer_dispatch_dma<DCTR_ACT_DICE>(p, l, in, out, K, N, wring, l > 0); break;
            default: layer_dispatch_dma<DCTR_ACT_LINEAR>(p, l, in, out, K, N, wring, l > 0); break;
        }
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        K = N;
    }

    if (p.has_head) {
        const int part = threadIdx.x & 15;
        const int KQh = pad64(K) / 4;
        float xcst[CROSS_NV];
        if (p.cross_layers > 0) {
#pragma unroll
            for (int v = 0; v < CROSS_NV; ++v) xcst[v] = xcs[v];
        }
        for (int row = threadIdx.x >> 4; row < ROWS; row += NTHR / 16) {
            float acc = 0.f;
            for (int n = part; n < K; n += 16) acc = fmaf(in[row * p.lda + lds_pos(n, KQh)], p.head_w[n], acc);
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            const int64_t b = b0 + row;
            if (part == 0 && b < p.batch) {
                float v = acc;
                if (fg.lpr != 0) v += extra[row];
                if (p.cross_layers > 0) {
                    float dots[CROSS_NV];
#pragma unroll
                    for (int l = 0; l < CROSS_NV; ++l) dots[l] = xp[l * ROWS + row];
                    v += cross_logit(dots, xcst, p.cross_layers);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p.add[i] != nullptr) v += p.add[i][b];
                if (p.global_bias != nullptr) v += p.global_bias[0];
                if (p.sigmoid_out) v = dctr::sigmoidf_(v);
                p.y[b] = v;
            }
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * K; i += NTHR) {
            const int r = i / K, c = i % K;
            const int64_t b = b0 + r;
            if (b < p.batch) p.y[b * p.y_stride + c] = in[r * p.lda + lds_pos(c, pad64(K) / 4)];
        }
    }
    if (p.probe != nullptr && threadIdx.x == 0) atomicMax(p.probe + 1, (unsigned long long)wall_clock64());
}

// launchers, one per translation unit mlp_kernels_rt{1,2,4}.hip
int launch_rt1(const MlpParams& p, const FusedGather& fg, unsigned blocks, size_t lds, hipStream_t stream);
int launch_rt2(const MlpParams& p, const FusedGather& fg, unsigned blocks, size_t lds, hipStream_t stream);
int launch_rt4(const MlpParams& p, const FusedGather& fg, unsigned blocks, size_t lds, hipStream_t stream);
int launch_rt1_ring(const MlpParams& p, const FusedGather& fg, unsigned blocks, size_t lds, int ring_off, hipStream_t stream);   // mlp_kernels_ring.hip
int launch_wide(const MlpParams& p, int kc, unsigned blocks, size_t lds, hipStream_t stream);                                         // mlp_kernels_wide.hip

// backward chain of dctr_mlp_bwd (mlp_bwd_kernels.hip): dz_in [B, units_fwd[L-1]] -> dZ of every earlier layer (-> dz_out[l], dense
// [B, units_fwd[l]]) and, with dx != NULL, the gradient of the DNN input; Wt[l] = W_l^T ([units_fwd[l], K_l] row-major, 16-B aligned)
int launch_bwd_chain(hipStream_t stream, int64_t batch, int in_dim, int n_layers, const int32_t* units_fwd, const float* const* Wt,
                     const float* const* acts, int activation, const float* dz_in, float* const* dz_out, float* dx, int64_t dx_stride);
bool bwd_chain_fits(int in_dim, int n_layers, const int32_t* units_fwd);

}  // namespace dctr_mlp
