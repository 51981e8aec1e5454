// adjacent to the hot path — DNN.call (+ Dense(1,use_bias=False) head + add_func + PredictionLayer.call)
// reference deepctr/layers/core.py:189-208, :250-259, layers/utils.py:328-333.
//
// One kernel runs the WHOLE multilayer perceptron for a 16-row tile of the batch: activations never leave LDS
// between layers, and the head (Dense(1) + linear/FM logits + global bias + sigmoid) is the epilogue.  Replaces,
// per layer, the reference's tensordot + bias_add + activation (+ Dice) kernels, then Dense, Add, bias_add, sigmoid.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32, see mfma_tile.h).  At B = 4096 the grid is 256 workgroups = one
// per CU, four waves = one per SIMD, so there is NO second wave to hide latency behind: every wave owns a
// 16*TPW-column slice of the layer output, walks K, and keeps its B operand in a two-stage REGISTER pipeline —
// the global_load_dwordx{TPW} of the next 8 k-steps (32 MFMAs = 1024 issue cycles) are in flight while the
// current 8 are consumed, which covers the L2 latency of the weight stream (603 KB per workgroup, L2-resident).
// Measured alternatives (scripts/mlp_lab.cpp): loads issued and waited per 4 MFMAs 58 us; weights staged through
// LDS in double-buffered 32-KB chunks 44 us (ds_write + barrier per chunk cost more than the MFMAs they feed).
// fp32 MFMA floor for 429-256-128-64: 301.7 kFLOP/sample -> 7.9 us per 4096 rows.
#include "dctr_common.h"
#include "embed_device.h"
#include "mfma_tile.h"

#ifdef DCTR_LAB_TIMING
__device__ unsigned long long dctr_lab_ts[64];
#define LAB_TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) dctr_lab_ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LAB_TS(i) do {} while (0)
#endif

namespace {

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
    int32_t lda;  // LDS row stride (floats) = pad64(max width) + 4
};

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

// stage s covers k-steps t = 8s .. 8s+7: row of slot g = 4*t + g  ->  byte offset (4*t)*N*4 (scalar) + voff (lane)
template <int TPW>
__device__ __forceinline__ void load_b8(__amdgpu_buffer_rsrc_t rsrc, int voff, int row4_bytes, int s, float (&b)[8][TPW]) {
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) buf_load_cols<TPW>(rsrc, voff, (8 * s + tt) * row4_bytes, b[tt]);
}

__device__ __forceinline__ int lds_pos(int k, int KQ) { return (k & 3) * KQ + (k >> 2); }

// A fragments of the same 8 k-steps: two ds_read_b128 of the column-permuted LDS tile
__device__ __forceinline__ void load_a8(const float* arow, float (&a)[8]) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow);
    const float4 a1 = *reinterpret_cast<const float4*>(arow + 4);
    a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
    a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
}

template <int TPW, int RT>
__device__ __forceinline__ void mfma8(const float (&av)[RT][8], const float (&b)[8][TPW], dctr::f32x4 (&acc)[RT][TPW]) {
#pragma unroll
    for (int tt = 0; tt < 8; ++tt)
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
#ifdef DCTR_LAB_NO_SB
#define DCTR_SB do {} while (0)
#else
#define DCTR_SB __builtin_amdgcn_sched_barrier(0)
#endif
template <int TPW, int RT>
__device__ __forceinline__ void tile_gemm_pipe(const float* A, int lda, int K, const float* __restrict__ W, int N,
                                               int n_base, dctr::f32x4 (&acc)[RT][TPW]) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQ = pad64(K) / 4;
    const float* arow = A + j * lda + g * KQ;
    int n0 = n_base + TPW * j;
    if (n0 + TPW > N) n0 = N - TPW;                       // TPW > 1 only when N % (16*TPW) == 0
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, K * N * 4, 0x00020000);
    const int voff = (g * N + n0) * 4;                    // lane-constant byte offset: slot row g, column slice
    const int row4_bytes = 4 * N * 4;                     // four weight rows
    const int n_it = KQ / 8;                              // >= 2
    const int s_last = n_it - 1;
    float b0[8][TPW], b1[8][TPW], b2[8][TPW];
    float a0[RT][8], a1[RT][8], a2[RT][8];
#define DCTR_STAGE_LOAD(S, AB, BB)                                                      \
    do {                                                                                \
        const int s_ = min((S), s_last);                                                \
        load_b8<TPW>(rsrc, voff, row4_bytes, s_, BB);                                   \
        _Pragma("unroll") for (int rt_ = 0; rt_ < RT; ++rt_)                            \
            load_a8(arow + rt_ * 16 * lda + s_ * 8, AB[rt_]);                           \
    } while (0)
    DCTR_STAGE_LOAD(0, a0, b0);
    DCTR_STAGE_LOAD(1, a1, b1);
    for (int it = 0; it < n_it; it += 3) {
        DCTR_STAGE_LOAD(it + 2, a2, b2);
        DCTR_SB;
        mfma8<TPW, RT>(a0, b0, acc);
        DCTR_SB;
        DCTR_STAGE_LOAD(it + 3, a0, b0);
        DCTR_SB;
        if (it + 1 < n_it) mfma8<TPW, RT>(a1, b1, acc);
        DCTR_SB;
        DCTR_STAGE_LOAD(it + 4, a1, b1);
        DCTR_SB;
        if (it + 2 < n_it) mfma8<TPW, RT>(a2, b2, acc);
        DCTR_SB;
    }
#undef DCTR_STAGE_LOAD
}

template <int TPW, int ACT, int RT>
__device__ __forceinline__ void layer_tiles(const MlpParams& p, int l, const float* in, float* out, int K, int N) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    const int KQn = pad64(N) / 4;                       // the next layer reads this tile with K = N
    for (int wt = wave; wt < n_tiles; wt += NWAVE) {
        const int n_base = wt * 16 * TPW;
        f32x4 acc[RT][TPW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef DCTR_LAB_COPIES
        tile_gemm_pipe<TPW, RT>(in, p.lda, K, p.W[l] + (size_t)((blockIdx.x / 8) % DCTR_LAB_COPIES) * 110080, N, n_base, acc);
#else
        tile_gemm_pipe<TPW, RT>(in, p.lda, K, p.W[l], N, n_base, acc);
#endif
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int n = n_base + TPW * j + c;
            if (n < N) {
                const float bv = p.bias[l] != nullptr ? p.bias[l][n] : 0.f;
                float al = 0.f, mu = 0.f, var = 1.f;
                if constexpr (ACT == DCTR_ACT_DICE) {
                    al = p.dice_alpha[l][n];
                    mu = p.dice_mean[l][n];
                    var = p.dice_var[l][n];
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(rt * 16 + 4 * g + r) * p.lda + lds_pos(n, KQn)] =
                            act_t<ACT>(acc[rt][c][r] + bv, al, mu, var, p.dice_eps);
            }
        }
    }
    // zero the K padding the NEXT layer reads: columns [N, pad64(N))
    const int npad = pad64(N) - N;
    if (npad > 0) {
        for (int i = threadIdx.x; i < 16 * RT * 64; i += NTHR) {
            const int r = i >> 6, c = i & 63;
            if (c < npad) out[r * p.lda + lds_pos(N + c, KQn)] = 0.f;
        }
    }
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
// Fused input producer (dctr_embed_mlp_fwd): the DNN-input tile of the workgroup's 16 samples is GATHERED
// straight into LDS — embedding rows, dense passthrough — and the linear + FM logits of the gather epilogue stay
// in LDS for the head.  Versus dctr_embed_gather_fm + dctr_mlp_fwd this removes the [B, 432] fp32 tile's trip
// through HBM (1.7 KB written and read back per sample), the [B] logit vectors and one kernel launch.
// The waves of the workgroup split the FIELDS (as the stand-alone gather does at small batch); lane (s, q) owns
// chunk q of sample s; samples are covered in 16*LPR/64 passes.
// ---------------------------------------------------------------------------------------------------
template <int LPR, bool HASH, int RT>
__device__ __forceinline__ void fused_gather_stage(const MlpParams& p, const GatherFused& g, float* tile, float* red,
                                                   float* extra, int64_t b0) {
    constexpr int VEC = 4;
    constexpr int ROWS = 16 * RT;                             // samples of this workgroup
    constexpr int SPW = 64 / LPR;                             // samples per wave pass
    constexpr int PASSES = SPW >= ROWS ? 1 : ROWS / SPW;
    static_assert(PASSES * 64 <= NTHR, "the combine step gives one wave per pass");
    constexpr int RW = 2 * VEC + 1;                            // partials per lane: sum[4], sq[4], lin
    const int KQ0 = pad64(p.in_dim) / 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int s = lane / LPR, q = lane % LPR;

    // columns past the real input (K padding) and rows past the batch are zero
    for (int i = threadIdx.x; i < ROWS * KQ0; i += NTHR) {
        const int r = i / KQ0, c4 = i - r * KQ0;
        const bool rowok = b0 + r < g.batch;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!rowok || 4 * c4 + k >= p.in_dim) tile[r * p.lda + k * KQ0 + c4] = 0.f;
    }

#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int r = pass * SPW + s;
        const int64_t b = b0 + r;
        const bool valid = r < ROWS && b < g.batch;
        float sum[VEC], sq[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) sum[c] = sq[c] = 0.f;
        GatherAcc acc{0.f, 0};
        float* const trow = tile + (r & (ROWS - 1)) * p.lda;
        auto store = [trow, KQ0](int col, const float (&v)[VEC]) {     // col % 4 == 0: columns col+k -> k*KQ0 + col/4
            float* dst = trow + (col >> 2);
            dst[0] = v[0];
            dst[KQ0] = v[1];
            dst[2 * KQ0] = v[2];
            dst[3 * KQ0] = v[3];
        };
        gather_fields<VEC, LPR, HASH>(g, wave, NWAVE, b, valid, q, sum, sq, acc, store);
        float* rp = red + ((wave * PASSES + pass) * RW) * 64 + lane;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            rp[c * 64] = sum[c];
            rp[(VEC + c) * 64] = sq[c];
        }
        rp[2 * VEC * 64] = acc.lin;
        if (g.status != nullptr && __any(acc.oor) && lane == 0) atomicOr(g.status, (int)DCTR_STATUS_INDEX_OOR);
    }

    // dense features: lanes 0..ROWS-1 of the last wave take one sample each (passthrough + dense . Linear.kernel)
    float dlin = 0.f;
    if (g.n_dense > 0 && wave == NWAVE - 1 && lane < ROWS) {
        const int r = lane;
        const bool valid = b0 + r < g.batch;
        const float* src = g.dense + (valid ? b0 + r : 0) * g.dense_stride;
        for (int k0 = 0; k0 < g.n_dense; k0 += 8) {
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
                        tile[r * p.lda + lds_pos(g.dense_out_offset + k, KQ0)] = x[m];
                    dlin = fmaf(x[m], w[m], dlin);
                }
            }
        }
        extra[ROWS + r] = dlin;
    }
    __syncthreads();

    // combine the waves' partial sums: FM = 0.5 * sum_d((sum_f e)^2 - sum_f e^2), linear = sum of the 1-wide rows
    if (threadIdx.x < PASSES * 64) {
        const int pass = threadIdx.x >> 6;
        const int r = pass * SPW + s;
        float S[VEC], Q[VEC], lin = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) S[c] = Q[c] = 0.f;
        for (int w = 0; w < NWAVE; ++w) {
            const float* rp = red + ((w * PASSES + pass) * RW) * 64 + lane;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                S[c] += rp[c * 64];
                Q[c] += rp[(VEC + c) * 64];
            }
            lin += rp[2 * VEC * 64];
        }
        float fm = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) fm += S[c] * S[c] - Q[c];
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
}

template <int RT>
__global__ __launch_bounds__(NTHR) void mlp_kernel(MlpParams p, FusedGather fg) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + ROWS * p.lda;
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    LAB_TS(0);

    float* extra = smem + 2 * ROWS * p.lda;                        // [2*ROWS]: per-row fused logits, dense partials
    if constexpr (RT <= 2) if (fg.lpr != 0) {
        float* red = buf1;                                         // partial sums live in the (still unused) 2nd tile
        const GatherFused& g = fg.g;
#define DCTR_FUSED(L)                                                              \
    do {                                                                           \
        if (g.any_hash) fused_gather_stage<L, true, RT>(p, g, buf0, red, extra, b0);  \
        else fused_gather_stage<L, false, RT>(p, g, buf0, red, extra, b0);            \
    } while (0)
        switch (fg.lpr) {
            case 1: DCTR_FUSED(1); break;
            case 2: DCTR_FUSED(2); break;
            case 4: DCTR_FUSED(4); break;
            case 8: DCTR_FUSED(8); break;
            default: DCTR_FUSED(16); break;
        }
#undef DCTR_FUSED
    }
    if (fg.lpr == 0) {
    // stage the input tile (rows beyond the batch and the K padding are zero) into the column-permuted layout.
    // Division-free mapping: wave w takes rows w, w+NWAVE, ...; lanes walk the float4 groups of a row.  Loads are
    // unconditional (clamped address, masked afterwards) and all issued before the first LDS store.
    {
        const int KQ0 = pad64(p.in_dim) / 4;                       // float4 groups per row incl. zero padding
        const int in4 = (p.in_dim + 3) / 4;
        const bool vec = (p.x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15u) == 0);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        constexpr int RPW = ROWS / NWAVE > 0 ? ROWS / NWAVE : 1;   // rows per wave
        for (int c0 = 0; c0 < KQ0; c0 += 128) {
            float4 v[RPW][2];
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = (wave + rr * NWAVE) & (ROWS - 1);
                const int64_t b = min(b0 + r, p.batch - 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c4 = min(c0 + h * 64 + lane, in4 - 1);
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
                    const int c4 = c0 + h * 64 + lane;
                    if (r < ROWS && c4 < KQ0) {
                        float4 t = v[rr][h];
                        if (!rowok || 4 * c4 >= p.in_dim) t.x = 0.f;   // never let stride padding of x into the tile
                        if (!rowok || 4 * c4 + 1 >= p.in_dim) t.y = 0.f;
                        if (!rowok || 4 * c4 + 2 >= p.in_dim) t.z = 0.f;
                        if (!rowok || 4 * c4 + 3 >= p.in_dim) t.w = 0.f;
                        float* dst = buf0 + r * p.lda + c4;             // columns 4*c4+s -> position s*KQ0 + c4
                        dst[0] = t.x;
                        dst[KQ0] = t.y;
                        dst[2 * KQ0] = t.z;
                        dst[3 * KQ0] = t.w;
                    }
                }
            }
        }
    }
    }
    __syncthreads();
    LAB_TS(1);

    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.units[l];
        switch (p.activation) {
            case DCTR_ACT_RELU: layer_dispatch<DCTR_ACT_RELU, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_SIGMOID: layer_dispatch<DCTR_ACT_SIGMOID, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_TANH: layer_dispatch<DCTR_ACT_TANH, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_DICE: layer_dispatch<DCTR_ACT_DICE, RT>(p, l, in, out, K, N); break;
            default: layer_dispatch<DCTR_ACT_LINEAR, RT>(p, l, in, out, K, N); break;
        }
        __syncthreads();
        LAB_TS(2 + l);
        float* t = in;
        in = out;
        out = t;
        K = N;
    }

    if (p.has_head) {
        // logit[row] = h[row,:] . head_w (+ extra logits + global bias), sigmoid for task == binary
        const int part = threadIdx.x & 15;
        const int KQh = pad64(K) / 4;
        for (int row = threadIdx.x >> 4; row < ROWS; row += NTHR / 16) {
            float acc = 0.f;
            for (int n = part; n < K; n += 16) acc = fmaf(in[row * p.lda + lds_pos(n, KQh)], p.head_w[n], acc);
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            const int64_t b = b0 + row;
            if (part == 0 && b < p.batch) {
                float v = acc;
                if (fg.lpr != 0) v += extra[row];
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
    LAB_TS(10);
}

int mlp_lda(const dctr_mlp_args_t* a) {
    int w = a->in_dim;
    for (int l = 0; l < a->n_layers; ++l) w = a->units[l] > w ? a->units[l] : w;
    return ((w + 63) & ~63) + 4;
}

}  // namespace

extern "C" size_t dctr_mlp_workspace_bytes(const dctr_mlp_args_t*) { return 0; }  // activations live in LDS

static int mlp_launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* ga, int fm_used, int lin_used, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "mlp_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->in_dim >= 1 && a->n_layers >= 0 && a->n_layers <= MAX_LAYERS, DCTR_E_DIM,
                 "mlp_fwd: bad sizes (batch=%lld in_dim=%d layers=%d, max %d layers)", (long long)a->batch, a->in_dim,
                 a->n_layers, MAX_LAYERS);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE((ga != nullptr || a->x) && a->y, DCTR_E_NULL, "mlp_fwd: null x / y");
    DCTR_REQUIRE(a->n_layers == 0 || (a->units && a->kernels && a->biases), DCTR_E_NULL, "mlp_fwd: null layer arrays");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_DICE, DCTR_E_ENUM, "mlp_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(!a->has_head || a->head_w, DCTR_E_NULL, "mlp_fwd: has_head without head_w");
    if (a->activation == DCTR_ACT_DICE && a->n_layers > 0)
        DCTR_REQUIRE(a->dice_alpha && a->dice_mean && a->dice_var, DCTR_E_NULL, "mlp_fwd: dice without parameters");
    MlpParams p{};
    p.x = a->x;
    p.batch = a->batch;
    p.x_stride = a->x_stride;
    p.in_dim = a->in_dim;
    p.n_layers = a->n_layers;
    for (int l = 0; l < a->n_layers; ++l) {
        DCTR_REQUIRE(a->units[l] >= 1, DCTR_E_DIM, "mlp_fwd: units[%d]=%d", l, a->units[l]);
        DCTR_REQUIRE(a->kernels[l] != nullptr, DCTR_E_NULL, "mlp_fwd: kernels[%d] null", l);
        DCTR_REQUIRE(dctr_aligned16(a->kernels[l]), DCTR_E_ALIGN, "mlp_fwd: kernels[%d] not 16-B aligned", l);
        p.units[l] = a->units[l];
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
        if (a->activation == DCTR_ACT_DICE) {
            DCTR_REQUIRE(a->dice_alpha[l] && a->dice_mean[l] && a->dice_var[l], DCTR_E_NULL, "mlp_fwd: dice[%d] null", l);
            p.dice_alpha[l] = a->dice_alpha[l];
            p.dice_mean[l] = a->dice_mean[l];
            p.dice_var[l] = a->dice_var[l];
        }
    }
    p.dice_eps = a->dice_eps;
    p.activation = a->activation;
    p.has_head = a->has_head;
    p.sigmoid_out = a->sigmoid_out;
    p.head_w = a->head_w;
    for (int i = 0; i < 4; ++i) p.add[i] = a->add[i];
    p.global_bias = a->global_bias;
    p.y = a->y;
    p.y_stride = a->y_stride;
    p.lda = mlp_lda(a);
    FusedGather fg{};
    // rows per workgroup.  auto: 16 while that still gives every CU a workgroup (latency), else 32
    int rt = a->tile_rows / 16;
    DCTR_REQUIRE(a->tile_rows == 0 || ((rt == 1 || rt == 2 || rt == 4) && a->tile_rows % 16 == 0), DCTR_E_DIM,
                 "mlp_fwd: tile_rows %d (0, 16, 32 or 64)", a->tile_rows);
    if (rt == 0) rt = a->batch > 16 * 2 * 256 ? 2 : 1;
    while (rt > 1 && ((size_t)2 * 16 * rt * p.lda + 2 * 16 * rt) * sizeof(float) > 160 * 1024) rt >>= 1;
    if (ga != nullptr && rt > 2) rt = 2;
    const int rows = 16 * rt;
    if (ga != nullptr) {
        DCTR_REQUIRE(ga->batch == a->batch, DCTR_E_DIM, "embed_mlp_fwd: gather batch %lld != mlp batch %lld",
                     (long long)ga->batch, (long long)a->batch);
        DCTR_REQUIRE(ga->n_fields >= 1 && ga->fields && ga->ids, DCTR_E_NULL, "embed_mlp_fwd: needs >= 1 gather field");
        DCTR_REQUIRE(ga->all_dim4 && ga->max_dim >= 1 && ga->max_dim <= 64, DCTR_E_UNSUPPORTED,
                     "embed_mlp_fwd: fused path needs every embedding_dim %% 4 == 0 and <= 64 (got max_dim %d, all_dim4 %d)",
                     ga->max_dim, ga->all_dim4);
        DCTR_REQUIRE(ga->n_dense == 0 || (ga->dense && ga->dense_stride >= ga->n_dense), DCTR_E_NULL,
                     "embed_mlp_fwd: bad dense matrix");
        DCTR_REQUIRE(ga->dense_copy_cols >= 0 && ga->dense_copy_cols <= ga->n_dense, DCTR_E_DIM,
                     "embed_mlp_fwd: dense_copy_cols outside [0, n_dense]");
        static_cast<dctr_gather_fm_args_t&>(fg.g) = *ga;
        fg.g.fm_logit_used = fm_used;
        fg.g.lin_logit_used = lin_used;
        int lpr = 1;
        while (lpr * 4 < ga->max_dim) lpr <<= 1;
        fg.lpr = lpr;
        const int spw = 64 / lpr;
        const int passes = spw >= rows ? 1 : rows / spw;
        const size_t red_floats = (size_t)NWAVE * passes * 9 * 64;   // aliases the second activation tile
        DCTR_REQUIRE(red_floats <= (size_t)rows * p.lda, DCTR_E_UNSUPPORTED,
                     "embed_mlp_fwd: layer widths too small to hold the gather partial sums (%zu floats)", red_floats);
    }
    const size_t lds = ((size_t)2 * rows * p.lda + 2 * rows) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "mlp_fwd: layer width needs %zu B of LDS (> 160 KiB)", lds);
    const void* fn = rt == 1 ? (const void*)mlp_kernel<1> : rt == 2 ? (const void*)mlp_kernel<2> : (const void*)mlp_kernel<4>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, (int64_t)rows);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: batch too large");
    if (rt == 1) DCTR_LAUNCH(mlp_kernel<1>, dim3((unsigned)blocks), dim3(NTHR), lds, (hipStream_t)stream, p, fg);
    else if (rt == 2) DCTR_LAUNCH(mlp_kernel<2>, dim3((unsigned)blocks), dim3(NTHR), lds, (hipStream_t)stream, p, fg);
    else DCTR_LAUNCH(mlp_kernel<4>, dim3((unsigned)blocks), dim3(NTHR), lds, (hipStream_t)stream, p, fg);
    return dctr_launch_status("dctr_mlp_fwd");
}

extern "C" int dctr_mlp_fwd(const dctr_mlp_args_t* a, void* stream) { return mlp_launch(a, nullptr, 0, 0, stream); }

extern "C" int dctr_embed_mlp_fwd(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int32_t add_fm_logit,
                                  int32_t add_lin_logit, void* stream) {
    DCTR_REQUIRE(g != nullptr && m != nullptr, DCTR_E_NULL, "embed_mlp_fwd: null args");
    DCTR_REQUIRE(m->has_head || (!add_fm_logit && !add_lin_logit), DCTR_E_DIM,
                 "embed_mlp_fwd: the gather logits can only be added by the fused head");
    return mlp_launch(m, g, add_fm_logit ? 1 : 0, add_lin_logit ? 1 : 0, stream);
}
