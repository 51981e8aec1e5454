// a13 — AttentionSequencePoolingLayer.call + LocalActivationUnit.call (DIN)
// reference deepctr/layers/sequence.py:261-298, layers/core.py:94-108, Dice layers/activation.py:59-64.
//
// One workgroup per sample.  Positions are processed in chunks of 64 (four 16-row MFMA tiles):
//   att_input[t] = [q, k_t, q - k_t, q * k_t]   (core.py:99-102) is formed straight into LDS,
//   the attention MLP (default 80-40) runs on v_mfma_f32_16x16x4_f32 with activations kept in LDS,
//   score_t = h . kernel + bias (core.py:106), masked with 0 — or -2^32+1 followed by a softmax when
//   weight_normalization — (sequence.py:280-288), and out = sum_t score_t * k_t (sequence.py:291).
// The reference materialises [B,T,4E], [B,T,80], [B,T,40] in HBM; here only keys are read (once from
// HBM, once more from L2 for the weighted sum) and [B,E] is written.
#include <math.h>

#include <stdlib.h>

#include "dctr_common.h"
#include "mfma_tile.h"

namespace dctr_din_chain {      // din_chain_kernels.hip: the row-chained form of the score kernel (two-layer attention MLPs)
int try_launch(const float* query, const float* keys, int64_t batch, int T, int E, int n_layers, const int32_t* units,
               const float* const* kernels, const float* const* biases, int activation, const float* const* dice_alpha,
               const float* const* dice_mean, const float* const* dice_var, float dice_eps, const float* out_kernel,
               const float* out_bias, float* raw, hipStream_t stream, const dctr_din_gather_t* gd, const uint8_t* key_mask,
               void* compact_ws, void* image_ws);
size_t compact_bytes(int64_t rows);
size_t image_bytes();
}


namespace {

constexpr int DIN_MAX_LAYERS = 6;
constexpr int RT = 4;        // row tiles per chunk
constexpr int CHUNK = 64;    // positions per chunk

struct DinParams {
    const float* query;
    const float* keys;
    const uint8_t* key_mask;
    int64_t batch;
    int32_t T, E, n_layers, activation;
    int32_t units[DIN_MAX_LAYERS];
    const float* W[DIN_MAX_LAYERS];
    const float* bias[DIN_MAX_LAYERS];
    const float* dice_alpha[DIN_MAX_LAYERS];
    const float* dice_mean[DIN_MAX_LAYERS];
    const float* dice_var[DIN_MAX_LAYERS];
    float dice_eps;
    int32_t weight_normalization;
    const float* out_kernel;
    const float* out_bias;
    float* out;
    int64_t out_stride;
    float* scores;
    int32_t lda;   // LDS row stride of the activation buffers
};

// C[64 x 16*TPW] slice of one layer for all four row tiles; B fragment loaded once per k-step.
template <int TPW>
__device__ __forceinline__ void din_layer(const DinParams& p, int l, const float* in, float* out, int K, int N) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQ = dctr::pad16(K) / 4;
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    const int act = p.activation;
    const int k_last = K - 1;
    for (int wt = wave; wt < n_tiles; wt += 4) {
        const int n_base = wt * 16 * TPW;
        int n0 = n_base + TPW * j;
        if (n0 + TPW > N) n0 = N - TPW;
        const float* wcol = p.W[l] + n0;
        f32x4 acc[RT][TPW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < KQ; t0 += 4) {
            float av[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 a4 = *reinterpret_cast<const float4*>(in + (rt * 16 + j) * p.lda + g * KQ + t0);
                av[rt][0] = a4.x; av[rt][1] = a4.y; av[rt][2] = a4.z; av[rt][3] = a4.w;
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int k = min(g * KQ + t0 + tt, k_last);
                float b[TPW];
                dctr::load_cols<TPW>(wcol + (int64_t)k * N, b);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int c = 0; c < TPW; ++c)
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][tt], b[c], acc[rt][c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int n = n_base + TPW * j + c;
            if (n < N) {
                const float bv = p.bias[l] != nullptr ? p.bias[l][n] : 0.f;
                float al = 0.f, mu = 0.f, var = 1.f;
                if (act == DCTR_ACT_DICE) {
                    al = p.dice_alpha[l][n];
                    mu = p.dice_mean[l][n];
                    var = p.dice_var[l][n];
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[rt][c][r] + bv;
                        v = act == DCTR_ACT_DICE ? dctr::dice_act(v, al, mu, var, p.dice_eps) : dctr::apply_act(v, act);
                        out[(rt * 16 + 4 * g + r) * p.lda + n] = v;
                    }
            }
        }
    }
    const int NP = dctr::pad16(N);
    for (int i = threadIdx.x; i < CHUNK * (NP - N); i += 256) {
        const int r = i / (NP - N), c = N + i % (NP - N);
        out[r * p.lda + c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void din_attn_kernel(DinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int E = p.E, T = p.T;
    float* buf0 = smem;                      // [64][lda]
    float* buf1 = buf0 + CHUNK * p.lda;      // [64][lda]
    float* qs = buf1 + CHUNK * p.lda;        // [E]
    float* sc = qs + ((E + 3) & ~3);         // [T] scores
    float* red = sc + T;                     // [2] softmax max / denominator (kept in the dynamic region: a
                                             // static __shared__ would shift its 16-B alignment)
    const int64_t b = blockIdx.x;
    const float* kb = p.keys + b * (int64_t)T * E;
    for (int i = threadIdx.x; i < E; i += 256) qs[i] = p.query[b * E + i];
    __syncthreads();

    const int K0 = 4 * E;
    const int KP0 = dctr::pad16(K0);
    for (int c0 = 0; c0 < T; c0 += CHUNK) {
        // keys tile for positions c0 .. c0+63 -> buf1 (all global loads of a pass issued before the LDS stores),
        // then att_input = [q, k, q-k, q*k] is formed LDS -> LDS (rows past T are zero)
        {
            constexpr int U = 8;
            const int total = CHUNK * E;
            for (int base = 0; base < total; base += 256 * U) {
                float v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = base + u * 256 + threadIdx.x;
                    const int r = i / E, e = i % E;
                    v[u] = (i < total && c0 + r < T) ? kb[(int64_t)(c0 + r) * E + e] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = base + u * 256 + threadIdx.x;
                    if (i < total) buf1[(i / E) * p.lda + (i % E)] = v[u];
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CHUNK * KP0; i += 256) {
            const int r = i / KP0, c = i % KP0;
            const int t = c0 + r;
            float v = 0.f;
            if (t < T && c < K0) {
                const int part = c / E, e = c % E;
                const float q = qs[e], kv = buf1[r * p.lda + e];
                v = part == 0 ? q : (part == 1 ? kv : (part == 2 ? q - kv : q * kv));
            }
            buf0[r * p.lda + c] = v;
        }
        __syncthreads();
        float* in = buf0;
        float* out = buf1;
        int K = K0;
        for (int l = 0; l < p.n_layers; ++l) {
            const int N = p.units[l];
            if (N % 32 == 0) din_layer<2>(p, l, in, out, K, N);
            else din_layer<1>(p, l, in, out, K, N);
            __syncthreads();
            float* t_ = in;
            in = out;
            out = t_;
            K = N;
        }
        // attention_score = att_out . kernel + bias (core.py:106), then the mask (sequence.py:280-285)
        {
            const int r = threadIdx.x >> 2, part = threadIdx.x & 3;   // 64 rows x 4 partial sums
            float acc = 0.f;
            for (int n = part; n < K; n += 4) acc = fmaf(in[r * p.lda + n], p.out_kernel[n], acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            const int t = c0 + r;
            if (part == 0 && t < T) {
                float v = acc + p.out_bias[0];
                const bool m = p.key_mask[b * (int64_t)T + t] != 0;
                v = m ? v : (p.weight_normalization ? -4294967296.f : 0.f);
                sc[t] = v;
            }
        }
        __syncthreads();
    }

    if (p.weight_normalization) {   // softmax over T (sequence.py:287-288)
        if (threadIdx.x < 64) {
            float mx = -INFINITY;
            for (int t = threadIdx.x; t < T; t += 64) mx = fmaxf(mx, sc[t]);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
            float den = 0.f;
            for (int t = threadIdx.x; t < T; t += 64) den += expf(sc[t] - mx);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
            if (threadIdx.x == 0) {
                red[0] = mx;
                red[1] = den;
            }
        }
        __syncthreads();
        const float mx = red[0], den = red[1];
        for (int t = threadIdx.x; t < T; t += 256) sc[t] = expf(sc[t] - mx) / den;
        __syncthreads();
    }
    if (p.scores != nullptr)
        for (int t = threadIdx.x; t < T; t += 256) p.scores[b * (int64_t)T + t] = sc[t];
    // outputs = scores[1,T] @ keys[T,E]  (sequence.py:291): serial over t per output element (deterministic)
    for (int e = threadIdx.x; e < E; e += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(sc[t], kb[(int64_t)t * E + e], acc);
        p.out[b * p.out_stride + e] = acc;
    }
}


// ---------------------------------------------------------------------------------------------------
// Fast path (needs a [B*T] float workspace): the attention MLP as a ROW problem over all B*T (sample, position)
// pairs, with the (small) MLP weights resident in LDS.
//   din_score_kernel: persistent workgroups (one per CU).  The weights of every layer are copied once into LDS
//     (zero-padded, row stride = 16 mod 32 floats so the MFMA B-operand reads of the four k-slots hit distinct
//     banks).  Each WAVE then loops over 16-row tiles on its own — no workgroup barrier in the loop: it stages the
//     16 key rows and their query rows into a wave-private LDS tile (coalesced float4 loads, column-permuted so a
//     lane reads four k-steps with one ds_read_b128), forms [q, k, q-k, q*k] in registers as the A operand, runs
//     all layers with activations in the same private tile, and reduces the last layer against `kernel`
//     (core.py:106) to one raw score per row.
//   din_pool_kernel: one wave per sample: mask (sequence.py:280-285), optional softmax, out = scores @ keys.
// Versus one-workgroup-per-sample this removes the 14 idle rows of every sample's last tile (T = 50 -> 64), the
// per-sample weight stream (95 KB from L2 per 50 rows) and every workgroup barrier from the MFMA loop.
// ---------------------------------------------------------------------------------------------------
constexpr int FAST_MAX_TILES = 6;     // layer widths up to 96 (two register sets of 4*NT B fragments + NT accumulators)

struct DinFastParams {
    const float* query;
    const float* keys;
    int64_t rows;                      // B*T
    int32_t T, E, n_layers, activation;
    int32_t units[DIN_MAX_LAYERS];
    const float* W[DIN_MAX_LAYERS];
    const float* bias[DIN_MAX_LAYERS];
    const float* dice_alpha[DIN_MAX_LAYERS];
    const float* dice_mean[DIN_MAX_LAYERS];
    const float* dice_var[DIN_MAX_LAYERS];
    float dice_eps;
    const float* out_kernel;
    const float* out_bias;
    float* raw;                        // [B*T] un-masked scores
    int32_t w_off[DIN_MAX_LAYERS];     // LDS float offset of layer l's weights
    int32_t np[DIN_MAX_LAYERS];        // LDS row stride of layer l (floats)
    int32_t kp[DIN_MAX_LAYERS];        // K rows of layer l incl. zero padding (multiple of 16)
    int32_t pb_off[DIN_MAX_LAYERS];    // LDS float offset of layer l's column parameters: bias, alpha, inv, shift [np] each
    int32_t ok_off;                    // LDS float offset of the output kernel (padded to np[last]), then out_bias,
                                       // then the workgroup's tile ticket counter (int)
    int32_t wave_off;                  // LDS float offset of the per-wave regions
    int32_t wave_floats;               // floats per wave region
    int32_t ldq;                       // row stride of the q / k tiles
    int32_t lda;                       // row stride of the activation tile
    int32_t n_waves;
};

// epilogue of one layer for a wave's 16-row tile: bias + activation (column parameters from LDS), then either the
// next layer's A tile (permuted: column n at (n&3)*KQn + (n>>2)) or, for the last layer, the dot with `kernel`
// -> raw score per row
template <int NT, bool LAST>
__device__ __forceinline__ void din_epilogue(const DinFastParams& p, int l, const dctr::f32x4 (&acc)[NT], float* tile,
                                             int64_t R0, const float* smem) {
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int N = p.units[l], NP = p.np[l];
    const float* pb = smem + p.pb_off[l];
    const bool dice = p.activation == DCTR_ACT_DICE;
    float v[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * nt + j;                                   // < NP; padded columns carry zeros
        const float bv = pb[n], al = pb[NP + n], inv = pb[2 * NP + n], sh = pb[3 * NP + n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = acc[nt][r] + bv;
            v[nt][r] = dice ? dctr::dice_pre(x, al, inv, sh) : dctr::apply_act(x, p.activation);
            if (n >= N) v[nt][r] = 0.f;                              // K padding of the next layer
        }
    }
    if constexpr (LAST) {
        const float* okp = smem + p.ok_off;
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float ok = okp[16 * nt + j];
#pragma unroll
            for (int r = 0; r < 4; ++r) part[r] = fmaf(v[nt][r], ok, part[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) part[r] += __shfl_xor(part[r], m, 64);
        if (j == 0) {
            const float ob = okp[NP];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t R = R0 + 4 * g + r;
                if (R < p.rows) p.raw[R] = part[r] + ob;
            }
        }
    } else {
        const int KQn = p.kp[l + 1] / 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 16 * nt + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(4 * g + r) * p.lda + (n & 3) * KQn + (n >> 2)] = v[nt][r];
        }
    }
}

// B fragments (NT column tiles) of the four k-steps of group `gi`: rows 16*gi + 4u + g of the layer's LDS weights
template <int NT>
__device__ __forceinline__ void din_load_b(const float* bcol, int NP, int gi, float (&b)[4][NT]) {
    const float* bs = bcol + (size_t)(16 * gi) * NP;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[u][nt] = bs[(4 * u) * NP + 16 * nt];
}

// the same for k-steps u = 2H, 2H+1 only
template <int NT, int H>
__device__ __forceinline__ void din_load_b2(const float* bcol, int NP, int gi, float (&b)[4][NT]) {
    const float* bs = bcol + (size_t)(16 * gi) * NP;
#pragma unroll
    for (int u = 2 * H; u < 2 * H + 2; ++u)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[u][nt] = bs[(4 * u) * NP + 16 * nt];
}

template <int NT, int H>
__device__ __forceinline__ void din_mfma2(const float (&a)[4], const float (&b)[4][NT], dctr::f32x4 (&acc)[NT]) {
#pragma unroll
    for (int u = 2 * H; u < 2 * H + 2; ++u)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][nt], acc[nt], 0, 0, 0);
}

// One pipeline step: the MFMAs of the current group (operands ac, bc) with the loads of the next group (-> bn)
// issued in two halves between them, so that never more than ~8 LDS instructions are outstanding when a wait for
// the older group's operands is needed (lgkmcnt saturates at 15: a whole group of 13-15 loads in flight forced the
// wait to cover part of the NEW group, i.e. exposed the LDS latency once per group).
#define DIN_STEP(NT, ac, bc, bn, bcol, NP, gnext, LOAD_A_NEXT)   \
    do {                                                        \
        LOAD_A_NEXT;                                            \
        din_load_b2<NT, 0>(bcol, NP, gnext, bn);                \
        DIN_SB;                                                 \
        din_mfma2<NT, 0>(ac, bc, acc);                          \
        DIN_SB;                                                 \
        din_load_b2<NT, 1>(bcol, NP, gnext, bn);                \
        DIN_SB;                                                 \
        din_mfma2<NT, 1>(ac, bc, acc);                          \
        DIN_SB;                                                 \
    } while (0)

#define DIN_SB __builtin_amdgcn_sched_barrier(0)

// Both layer kinds walk K in groups of four k-steps (one ds_read_b128 of A, 4*NT ds_read_b32 of B) with the
// operands of group gi+1 loaded into a second register set before the 4*NT MFMAs of group gi issue: a wave runs
// alone on its SIMD here (the LDS is full of weights), so nothing else would hide the LDS latency.  The
// sched_barriers keep hipcc from re-serialising load -> wait -> MFMA (measured: 6x the MFMA time).

// layer 0.  att_input = [q, k, q-k, q*k] (core.py:99-102) times W = [Wq; Wk; Wd; Wp] is evaluated as
//     q (Wq + Wd) + k (Wk - Wd) + (q*k) Wp
// i.e. K = 3E instead of 4E: a quarter fewer MFMAs and a quarter less LDS for the weights (which is what lets a
// second wave per SIMD fit); the two weight sums are formed once per workgroup while the weights are copied to LDS.
// Same value up to fp32 rounding (the reference's own GEMM does not fix a summation order either).
// One call = one of the three parts: the A operand is formed in registers from the wave's q / k tiles.  The raw q / k fragments of group gi+1 are loaded before the MFMAs of group
// gi and turned into the operand after them (any earlier and the VALU would wait on the B loads just issued).
template <int NT, int PART>
__device__ __forceinline__ void din_layer0_part(const float* qrow, const float* krow, const float* bcol, int NP, int GP,
                                                dctr::f32x4 (&acc)[NT]) {
    auto load_raw = [&](int gi, float4& q4, float4& k4) {
        const int i = 4 * min(gi, GP - 1);
        if constexpr (PART != 1) q4 = *reinterpret_cast<const float4*>(qrow + i);
        if constexpr (PART != 0) k4 = *reinterpret_cast<const float4*>(krow + i);
    };
    auto make_a = [&](const float4& q4, const float4& k4, float (&a)[4]) {
        if constexpr (PART == 0) { a[0] = q4.x; a[1] = q4.y; a[2] = q4.z; a[3] = q4.w; }
        else if constexpr (PART == 1) { a[0] = k4.x; a[1] = k4.y; a[2] = k4.z; a[3] = k4.w; }
        else { a[0] = q4.x * k4.x; a[1] = q4.y * k4.y; a[2] = q4.z * k4.z; a[3] = q4.w * k4.w; }
    };
    const float* bpart = bcol + (size_t)(16 * PART * GP) * NP;
    float4 rq = make_float4(0.f, 0.f, 0.f, 0.f), rk = rq;
    float a0[4], a1[4], b0[4][NT], b1[4][NT];
    load_raw(0, rq, rk);
    din_load_b<NT>(bpart, NP, 0, b0);
    make_a(rq, rk, a0);
    for (int gi = 0; gi < GP; gi += 2) {
        DIN_STEP(NT, a0, b0, b1, bpart, NP, min(gi + 1, GP - 1), load_raw(gi + 1, rq, rk));
        make_a(rq, rk, a1);
        if (gi + 1 < GP) {
            DIN_STEP(NT, a1, b1, b0, bpart, NP, min(gi + 2, GP - 1), load_raw(gi + 2, rq, rk));
            make_a(rq, rk, a0);
        }
    }
}

template <int NT, bool LAST>
__device__ __forceinline__ void din_layer0(const DinFastParams& p, const float* smem, float* qt, float* kt, float* tile,
                                           int64_t R0) {
    using dctr::f32x4;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int EQ = p.E / 4, NP = p.np[0];
    const int GP = EQ / 4;                 // groups of four k-steps per part
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* qrow = qt + j * p.ldq + g * EQ;
    const float* krow = kt + j * p.ldq + g * EQ;
    const float* bcol = smem + p.w_off[0] + g * NP + j;
    din_layer0_part<NT, 0>(qrow, krow, bcol, NP, GP, acc);
    din_layer0_part<NT, 1>(qrow, krow, bcol, NP, GP, acc);
    din_layer0_part<NT, 2>(qrow, krow, bcol, NP, GP, acc);
    din_epilogue<NT, LAST>(p, 0, acc, tile, R0, smem);
}

// layer l >= 1: A from the wave's activation tile
template <int NT, bool LAST>
__device__ __forceinline__ void din_layer(const DinFastParams& p, int l, const float* smem, float* tile, int64_t R0) {
    const float* wl = smem + p.w_off[l];
    using dctr::f32x4;
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQ = p.kp[l] / 4, NP = p.np[l];
    const int G = KQ / 4;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = tile + j * p.lda + g * KQ;
    const float* bcol = wl + g * NP + j;
    auto load_a = [&](int gi, float (&a)[4]) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 4 * min(gi, G - 1));
        a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
    };
    float a0[4], a1[4], b0[4][NT], b1[4][NT];
    load_a(0, a0);
    din_load_b<NT>(bcol, NP, 0, b0);
    for (int gi = 0; gi < G; gi += 2) {
        DIN_STEP(NT, a0, b0, b1, bcol, NP, min(gi + 1, G - 1), load_a(gi + 1, a1));
        if (gi + 1 < G) DIN_STEP(NT, a1, b1, b0, bcol, NP, min(gi + 2, G - 1), load_a(gi + 2, a0));
    }
    din_epilogue<NT, LAST>(p, l, acc, tile, R0, smem);
}

#define DIN_NT_SWITCH(NTV, CALL)            \
    switch (NTV) {                          \
        case 1: { constexpr int NT = 1; CALL; } break; \
        case 2: { constexpr int NT = 2; CALL; } break; \
        case 3: { constexpr int NT = 3; CALL; } break; \
        case 4: { constexpr int NT = 4; CALL; } break; \
        case 5: { constexpr int NT = 5; CALL; } break; \
        default: { constexpr int NT = 6; CALL; } break; \
    }

__global__ __launch_bounds__(512) void din_score_kernel(DinFastParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nthr = blockDim.x;
    // weights -> LDS, zero-padded to [kp][np]; batches of loads in flight per thread, then the stores.
    // LDS row k of layer 0 = W[r1] + sg * W[r2]:  k < E: Wq + Wd;  k < 2E: Wk - Wd;  k < 3E: Wp   (see din_layer0_part)
    for (int l = 0; l < p.n_layers; ++l) {
        const int K = l == 0 ? 3 * p.E : p.units[l - 1], N = p.units[l], NP = p.np[l];
        const int E = p.E;
        float* wl = smem + p.w_off[l];
        const float* src = p.W[l];
        auto rows_of = [&](int k, int& r1, int& r2, float& sg) {
            r1 = k; r2 = k; sg = 0.f;
            if (l == 0) {
                if (k < E) { r2 = 2 * E + k; sg = 1.f; }
                else if (k < 2 * E) { r2 = E + k; sg = -1.f; }
                else { r1 = E + k; r2 = r1; }
            }
        };
        if (N % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
            const int NP4 = NP / 4, total4 = p.kp[l] * NP4;
            for (int base = 0; base < total4; base += 4 * nthr) {
                float4 v[4], w[4];
                float sgn[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * nthr + threadIdx.x;
                    const int k = idx / NP4, n = 4 * (idx - k * NP4);
                    int r1, r2;
                    rows_of(min(k, K - 1), r1, r2, sgn[u]);
                    v[u] = *reinterpret_cast<const float4*>(src + (size_t)r1 * N + min(n, N - 4));
                    w[u] = *reinterpret_cast<const float4*>(src + (size_t)r2 * N + min(n, N - 4));
                    if (k >= K || n >= N) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); sgn[u] = 0.f; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * nthr + threadIdx.x;
                    if (idx < total4)
                        reinterpret_cast<float4*>(wl)[idx] = make_float4(v[u].x + sgn[u] * w[u].x, v[u].y + sgn[u] * w[u].y,
                                                                         v[u].z + sgn[u] * w[u].z, v[u].w + sgn[u] * w[u].w);
                }
            }
        } else {
            const int total = p.kp[l] * NP;
            for (int base = 0; base < total; base += 8 * nthr) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * nthr + threadIdx.x;
                    const int k = idx / NP, n = idx - k * NP;
                    int r1, r2;
                    float sg;
                    rows_of(min(k, K - 1), r1, r2, sg);
                    v[u] = src[(size_t)r1 * N + min(n, N - 1)] + sg * src[(size_t)r2 * N + min(n, N - 1)];
                    if (k >= K || n >= N) v[u] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * nthr + threadIdx.x;
                    if (idx < total) wl[idx] = v[u];
                }
            }
        }
        // column parameters: bias, Dice alpha, inv = rsqrt(var + eps), shift = -mean * inv (zeros past N)
        float* pb = smem + p.pb_off[l];
        for (int n = threadIdx.x; n < NP; n += nthr) {
            const bool in = n < N;
            float al = 0.f, inv = 0.f, sh = 0.f;
            if (in && p.activation == DCTR_ACT_DICE) {
                al = p.dice_alpha[l][n];
                inv = 1.f / sqrtf(p.dice_var[l][n] + p.dice_eps);
                sh = -p.dice_mean[l][n] * inv;
            }
            pb[n] = (in && p.bias[l] != nullptr) ? p.bias[l][n] : 0.f;
            pb[NP + n] = al;
            pb[2 * NP + n] = inv;
            pb[3 * NP + n] = sh;
            if (l == p.n_layers - 1) smem[p.ok_off + n] = in ? p.out_kernel[n] : 0.f;
        }
        if (l == p.n_layers - 1 && threadIdx.x == 0) {
            smem[p.ok_off + NP] = p.out_bias[0];
            reinterpret_cast<int*>(smem)[p.ok_off + NP + 1] = p.n_waves;       // first tickets = the waves' own indices
        }
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float* region = smem + p.wave_off + wave * p.wave_floats;
    float* qt = region;                       // [16][ldq]
    float* kt = region + 16 * p.ldq;          // [16][ldq]
    float* tile = region;                     // [16][lda] activations alias the q / k tiles (in-order LDS per wave)
    const int EQ = p.E / 4;
    const int64_t n_tiles = (p.rows + 15) / 16;
    // chunk c of a tile = float4 ii of row jj; a lane owns chunks lane, lane+64, ... (<= MAXC of them: E <= 256)
    constexpr int MAXC = 4;
    const int n_chunks = 16 * EQ;
    int jj[MAXC], ii[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const int c = min(u * 64 + lane, n_chunks - 1);
        jj[u] = c / EQ;
        ii[u] = c - jj[u] * EQ;
    }
    float4 kv[MAXC], qv[MAXC];
    auto fetch_rows = [&](int64_t tt) {            // rows of tile tt -> registers (clamped; stale data is never stored)
        const int64_t R0f = min(tt, n_tiles - 1) * 16;
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const int64_t R = min(R0f + jj[u], p.rows - 1);
            const int64_t b = (int64_t)((uint32_t)R / (uint32_t)p.T);      // host: rows < 2^31
            kv[u] = *reinterpret_cast<const float4*>(p.keys + R * p.E + 4 * ii[u]);
            qv[u] = *reinterpret_cast<const float4*>(p.query + b * p.E + 4 * ii[u]);
        }
    };
    // tiles [t_lo, t_hi) belong to this workgroup; its waves draw them from an LDS ticket counter (waves that share
    // a SIMD run slower than one that has its SIMD to itself, so a static split would leave SIMDs idle)
    const int64_t per_wg = (n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t t_lo = per_wg * blockIdx.x, t_hi = min(t_lo + per_wg, n_tiles);
    int* ticket = reinterpret_cast<int*>(smem) + p.ok_off + p.np[p.n_layers - 1] + 1;
    int64_t t = t_lo + wave;
    if (t < t_hi) fetch_rows(t);
    while (t < t_hi) {
        const int64_t R0 = t * 16;
        // registers -> the wave's q / k tiles, permuted (column 4i+s -> s*EQ + i); then prefetch the next tile's rows,
        // which arrive while this tile's MFMAs run
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            if (u * 64 + lane < n_chunks) {
                float* kd = kt + jj[u] * p.ldq + ii[u];
                float* qd = qt + jj[u] * p.ldq + ii[u];
                kd[0] = kv[u].x; kd[EQ] = kv[u].y; kd[2 * EQ] = kv[u].z; kd[3 * EQ] = kv[u].w;
                qd[0] = qv[u].x; qd[EQ] = qv[u].y; qd[2 * EQ] = qv[u].z; qd[3 * EQ] = qv[u].w;
            }
        }
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(ticket, 1);
        const int64_t t_next = t_lo + __builtin_amdgcn_readfirstlane(nxt);
        fetch_rows(t_next);
        const int nt0 = (p.units[0] + 15) / 16;
        if (p.n_layers == 1) {
            DIN_NT_SWITCH(nt0, (din_layer0<NT, true>(p, smem, qt, kt, tile, R0)));
        } else {
            DIN_NT_SWITCH(nt0, (din_layer0<NT, false>(p, smem, qt, kt, tile, R0)));
            for (int l = 1; l < p.n_layers; ++l) {
                const int ntl = (p.units[l] + 15) / 16;
                if (l == p.n_layers - 1) {
                    DIN_NT_SWITCH(ntl, (din_layer<NT, true>(p, l, smem, tile, R0)));
                } else {
                    DIN_NT_SWITCH(ntl, (din_layer<NT, false>(p, l, smem, tile, R0)));
                }
            }
        }
        t = t_next;
    }
}

// one wave per sample: mask, softmax (weight_normalization), out = scores @ keys, serial over t per output element
__global__ __launch_bounds__(256) void din_pool_kernel(const float* raw, const float* keys, const uint8_t* key_mask,
                                                       int64_t batch, int T, int E, int weight_normalization, float* out,
                                                       int64_t out_stride, float* scores) {
    extern __shared__ float sc_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    float* sc = sc_all + wave * T;
    const float* kb = keys + b * (int64_t)T * E;
    for (int t = lane; t < T; t += 64) {
        const bool m = key_mask[b * (int64_t)T + t] != 0;
        sc[t] = m ? raw[b * (int64_t)T + t] : (weight_normalization ? -4294967296.f : 0.f);
    }
    if (weight_normalization) {
        float mx = -INFINITY;
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sc[t]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        float den = 0.f;
        for (int t = lane; t < T; t += 64) den += expf(sc[t] - mx);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
        for (int t = lane; t < T; t += 64) sc[t] = expf(sc[t] - mx) / den;
    }
    if (scores != nullptr)
        for (int t = lane; t < T; t += 64) scores[b * (int64_t)T + t] = sc[t];
    for (int e = lane; e < E; e += 64) {
        float acc = 0.f;
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            float kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kv[u] = kb[(int64_t)(t + u) * E + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sc[t + u], kv[u], acc);
        }
        for (; t < T; ++t) acc = fmaf(sc[t], kb[(int64_t)t * E + e], acc);
        out[b * out_stride + e] = acc;
    }
}

// the same with the keys read from the embedding tables (dctr_din_attn_gather_fwd): the sample's ids sit in LDS, position t counts
// iff every mask_zero feature's id != 0 (keras Concat.compute_mask over the history features, reference layers/utils.py:198-228)
struct DinPoolGather {
    int32_t nf, ids_i64;
    const void* hist_ids[2];
    int64_t hist_stride;
    const float* hist_table[2];
    int64_t hist_vocab[2];
    int32_t mask_zero[2];
};
__global__ __launch_bounds__(256) void din_pool_gather_kernel(const float* raw, DinPoolGather gd, int64_t batch, int T, int E,
                                                              int weight_normalization, float* out, int64_t out_stride, float* scores) {
    extern __shared__ float sc_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    float* sc = sc_all + wave * (3 * T);
    int* row = reinterpret_cast<int*>(sc + T);              // [nf][T] table rows (vocabularies < 2^31 on this path: host)
    const int EH = E / gd.nf;
    for (int t = lane; t < T; t += 64) {
        bool m = true;
        for (int h = 0; h < gd.nf; ++h) {
            const int64_t idx = b * gd.hist_stride + t;
            const int64_t id = gd.ids_i64 ? reinterpret_cast<const int64_t*>(gd.hist_ids[h])[idx]
                                          : (int64_t)reinterpret_cast<const int32_t*>(gd.hist_ids[h])[idx];
            if (gd.mask_zero[h]) m = m && id != 0;
            row[h * T + t] = (uint64_t)id < (uint64_t)gd.hist_vocab[h] ? (int)id : 0;
        }
        sc[t] = m ? raw[b * (int64_t)T + t] : (weight_normalization ? -4294967296.f : 0.f);
    }
    if (weight_normalization) {
        float mx = -INFINITY;
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, sc[t]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        float den = 0.f;
        for (int t = lane; t < T; t += 64) den += expf(sc[t] - mx);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
        for (int t = lane; t < T; t += 64) sc[t] = expf(sc[t] - mx) / den;
    }
    if (scores != nullptr)
        for (int t = lane; t < T; t += 64) scores[b * (int64_t)T + t] = sc[t];
    for (int e = lane; e < E; e += 64) {
        const int h = e / EH, eo = e - h * EH;
        const float* tb = gd.hist_table[h] + eo;
        const int* rw = row + h * T;
        float acc = 0.f;
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            float kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kv[u] = tb[(int64_t)rw[t + u] * EH];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sc[t + u], kv[u], acc);
        }
        for (; t < T; ++t) acc = fmaf(sc[t], tb[(int64_t)rw[t] * EH], acc);
        out[b * out_stride + e] = acc;
    }
}

// LDS plan of the fast path; returns false when the shape does not qualify
bool din_fast_plan(const dctr_din_attn_args_t* a, DinFastParams& p, size_t& lds_bytes) {
    if (a->n_layers < 1 || a->dim % 16 != 0 || a->dim > 64) return false;   // <= 4 float4 chunks per lane and tile
    int off = 0;
    int width = 0;
    for (int l = 0; l < a->n_layers; ++l) {
        const int K = l == 0 ? 3 * a->dim : a->units[l - 1], N = a->units[l];     // layer 0 folded to K = 3E
        if (N > 16 * FAST_MAX_TILES) return false;
        const int n16 = (N + 15) & ~15;
        p.np[l] = (n16 % 32 == 16) ? n16 : n16 + 16;          // = 16 mod 32: the four k-slots read distinct banks
        p.kp[l] = (K + 15) & ~15;
        p.w_off[l] = off;
        off += p.kp[l] * p.np[l];
        if (l > 0) width = p.kp[l] > width ? p.kp[l] : width;
    }
    for (int l = 0; l < a->n_layers; ++l) {
        p.pb_off[l] = off;
        off += 4 * p.np[l];
    }
    p.ok_off = off;
    off += p.np[a->n_layers - 1] + 16;                        // + out_bias, ticket counter (keeps 16-B alignment)
    p.wave_off = off;
    p.ldq = a->dim + 4;
    p.lda = width + 4;
    const int need = 2 * 16 * p.ldq > 16 * p.lda ? 2 * 16 * p.ldq : 16 * p.lda;
    p.wave_floats = (need + 3) & ~3;
    for (int nw = 8; nw >= 4; --nw) {
        const size_t bytes = ((size_t)off + (size_t)nw * p.wave_floats) * sizeof(float);
        if (bytes <= 160 * 1024) {
            p.n_waves = nw;
            lds_bytes = bytes;
            return true;
        }
    }
    return false;
}

}  // namespace

// [B * T] raw scores, then the list of the rows that count (din_chain_kernels.hip: masked positions are skipped)
static size_t din_raw_bytes(int64_t rows) { return (((size_t)rows * sizeof(float)) + 15) & ~(size_t)15; }

extern "C" size_t dctr_din_attn_workspace_bytes(const dctr_din_attn_args_t* a) {
    if (a == nullptr || a->batch <= 0 || a->maxlen <= 0) return 0;
    const int64_t rows = a->batch * (int64_t)a->maxlen;
    return din_raw_bytes(rows) + dctr_din_chain::compact_bytes(rows) + dctr_din_chain::image_bytes();
}

// the compaction area of a workspace that is large enough for it, else NULL (a [B * T]-float workspace of earlier ABIs: all rows scored)
static void* din_compact_area(const dctr_din_attn_args_t* a, int64_t rows) {
    const size_t cb = dctr_din_chain::compact_bytes(rows);
    if (cb == 0 || a->workspace == nullptr || a->workspace_bytes < din_raw_bytes(rows) + cb) return nullptr;
    return static_cast<char*>(a->workspace) + din_raw_bytes(rows);
}

// the LDS-image area behind it (workspaces of dctr_din_attn_workspace_bytes), else NULL
static void* din_image_area(const dctr_din_attn_args_t* a, int64_t rows) {
    const size_t off = din_raw_bytes(rows) + dctr_din_chain::compact_bytes(rows);
    if (dctr_din_chain::compact_bytes(rows) == 0 || a->workspace == nullptr || a->workspace_bytes < off + dctr_din_chain::image_bytes()) return nullptr;
    return static_cast<char*>(a->workspace) + off;
}

extern "C" int dctr_din_attn_pool_fwd(const dctr_din_attn_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "din_attn_pool_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->maxlen >= 1 && a->dim >= 1 && a->n_layers >= 0 && a->n_layers <= DIN_MAX_LAYERS,
                 DCTR_E_DIM, "din_attn_pool_fwd: bad sizes (B=%lld T=%d E=%d layers=%d)", (long long)a->batch, a->maxlen,
                 a->dim, a->n_layers);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->query && a->keys && a->key_mask && a->out && a->out_kernel && a->out_bias, DCTR_E_NULL,
                 "din_attn_pool_fwd: null pointer");
    DCTR_REQUIRE(a->n_layers == 0 || (a->units && a->kernels && a->biases), DCTR_E_NULL,
                 "din_attn_pool_fwd: null layer arrays");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_DICE, DCTR_E_ENUM,
                 "din_attn_pool_fwd: activation %d", a->activation);
    DCTR_REQUIRE(a->out_stride >= a->dim, DCTR_E_DIM, "din_attn_pool_fwd: out_stride < dim");
    if (a->activation == DCTR_ACT_DICE && a->n_layers > 0)
        DCTR_REQUIRE(a->dice_alpha && a->dice_mean && a->dice_var, DCTR_E_NULL, "din_attn_pool_fwd: dice without parameters");
    // fast path: weights resident in LDS, rows = B*T, two launches; needs the [B*T] float workspace
    {
        DinFastParams f{};
        size_t lds = 0;
        const int64_t rows = a->batch * (int64_t)a->maxlen;
        // (histories past 4,096 positions — din_pool_kernel keeps 4 samples' scores in 64 KiB of LDS — take the one-kernel form below,
        //  whose score row of one sample fits up to ~30,000 positions: nothing is launched here for a shape that would then be refused)
        if (a->workspace != nullptr && a->workspace_bytes >= (size_t)rows * sizeof(float) && rows < 0x7fffffffLL &&
            (size_t)4 * a->maxlen * sizeof(float) <= 64 * 1024 && dctr_aligned16(a->query) && dctr_aligned16(a->keys) && din_fast_plan(a, f, lds)) {
            f.query = a->query;
            f.keys = a->keys;
            f.rows = rows;
            f.T = a->maxlen;
            f.E = a->dim;
            f.n_layers = a->n_layers;
            f.activation = a->activation;
            for (int l = 0; l < a->n_layers; ++l) {
                DCTR_REQUIRE(a->units[l] >= 1, DCTR_E_DIM, "din_attn_pool_fwd: units[%d]=%d", l, a->units[l]);
                DCTR_REQUIRE(a->kernels[l] != nullptr, DCTR_E_NULL, "din_attn_pool_fwd: kernels[%d] null", l);
                f.units[l] = a->units[l];
                f.W[l] = a->kernels[l];
                f.bias[l] = a->biases[l];
                if (a->activation == DCTR_ACT_DICE) {
                    DCTR_REQUIRE(a->dice_alpha[l] && a->dice_mean[l] && a->dice_var[l], DCTR_E_NULL,
                                 "din_attn_pool_fwd: dice[%d] null", l);
                    f.dice_alpha[l] = a->dice_alpha[l];
                    f.dice_mean[l] = a->dice_mean[l];
                    f.dice_var[l] = a->dice_var[l];
                }
            }
            f.dice_eps = a->dice_eps;
            f.out_kernel = a->out_kernel;
            f.out_bias = a->out_bias;
            f.raw = static_cast<float*>(a->workspace);
            // the row-chained score kernel for the shapes it is instantiated for (DCTR_DIN_NO_CHAIN=1: lab switch back)
            static const bool no_chain = dctr_lab_env("DCTR_DIN_NO_CHAIN") != nullptr;
            const size_t pool_lds0 = (size_t)4 * a->maxlen * sizeof(float);
            if (!no_chain && pool_lds0 <= 64 * 1024 &&
                dctr_din_chain::try_launch(a->query, a->keys, a->batch, a->maxlen, a->dim, a->n_layers, a->units, a->kernels, a->biases,
                                           a->activation, a->dice_alpha, a->dice_mean, a->dice_var, a->dice_eps, a->out_kernel,
                                           a->out_bias, f.raw, (hipStream_t)stream, nullptr, a->key_mask, din_compact_area(a, rows),
                                           din_image_area(a, rows))) {
                hipLaunchKernelGGL(din_pool_kernel, dim3((unsigned)dctr_ceil_div(a->batch, (int64_t)4)), dim3(256), pool_lds0,
                                   (hipStream_t)stream, f.raw, a->keys, a->key_mask, a->batch, a->maxlen, a->dim,
                                   a->weight_normalization, a->out, a->out_stride, a->scores);
                return dctr_launch_status("dctr_din_attn_pool_fwd");
            }
            if (lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)din_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                DCTR_REQUIRE(e == hipSuccess, (int)e, "din_attn_pool_fwd: cannot raise dynamic LDS to %zu B: %s", lds,
                             hipGetErrorString(e));
            }
            const int64_t n_tiles = (rows + 15) / 16;
            int64_t grid = dctr_ceil_div(n_tiles, (int64_t)f.n_waves);     // persistent: one workgroup per CU (the
            if (grid > 256) grid = 256;                                    // kernel's ~250 VGPRs allow 2 waves per SIMD)
            DCTR_LAUNCH(din_score_kernel, dim3((unsigned)grid), dim3(64 * f.n_waves), lds, (hipStream_t)stream, f);
            const size_t pool_lds = (size_t)4 * a->maxlen * sizeof(float);
            DCTR_REQUIRE(pool_lds <= 64 * 1024, DCTR_E_UNSUPPORTED, "din_attn_pool_fwd: maxlen %d too long", a->maxlen);
            hipLaunchKernelGGL(din_pool_kernel, dim3((unsigned)dctr_ceil_div(a->batch, (int64_t)4)), dim3(256), pool_lds,
                               (hipStream_t)stream, f.raw, a->keys, a->key_mask, a->batch, a->maxlen, a->dim,
                               a->weight_normalization, a->out, a->out_stride, a->scores);
            return dctr_launch_status("dctr_din_attn_pool_fwd");
        }
    }
    DinParams p{};
    p.query = a->query;
    p.keys = a->keys;
    p.key_mask = a->key_mask;
    p.batch = a->batch;
    p.T = a->maxlen;
    p.E = a->dim;
    p.n_layers = a->n_layers;
    p.activation = a->activation;
    int w = 4 * a->dim;
    for (int l = 0; l < a->n_layers; ++l) {
        DCTR_REQUIRE(a->units[l] >= 1, DCTR_E_DIM, "din_attn_pool_fwd: units[%d]=%d", l, a->units[l]);
        DCTR_REQUIRE(a->kernels[l] != nullptr, DCTR_E_NULL, "din_attn_pool_fwd: kernels[%d] null", l);
        DCTR_REQUIRE((((uintptr_t)a->kernels[l]) & 7u) == 0, DCTR_E_ALIGN, "din_attn_pool_fwd: kernels[%d] not 8-B aligned", l);
        p.units[l] = a->units[l];
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
        if (a->activation == DCTR_ACT_DICE) {
            DCTR_REQUIRE(a->dice_alpha[l] && a->dice_mean[l] && a->dice_var[l], DCTR_E_NULL, "din_attn_pool_fwd: dice[%d] null", l);
            p.dice_alpha[l] = a->dice_alpha[l];
            p.dice_mean[l] = a->dice_mean[l];
            p.dice_var[l] = a->dice_var[l];
        }
        w = a->units[l] > w ? a->units[l] : w;
    }
    p.dice_eps = a->dice_eps;
    p.weight_normalization = a->weight_normalization;
    p.out_kernel = a->out_kernel;
    p.out_bias = a->out_bias;
    p.out = a->out;
    p.out_stride = a->out_stride;
    p.scores = a->scores;
    p.lda = ((w + 15) & ~15) + 4;
    const size_t lds = ((size_t)2 * CHUNK * p.lda + ((a->dim + 3) & ~3) + a->maxlen + 2) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024 - 64, DCTR_E_UNSUPPORTED, "din_attn_pool_fwd: needs %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024 - 64) {
        hipError_t e = hipFuncSetAttribute((const void*)din_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "din_attn_pool_fwd: cannot raise dynamic LDS to %zu B: %s", lds,
                     hipGetErrorString(e));
    }
    DCTR_REQUIRE(a->batch <= 0x7fffffffLL, DCTR_E_DIM, "din_attn_pool_fwd: batch too large");
    DCTR_LAUNCH(din_attn_kernel, dim3((unsigned)a->batch), dim3(256), lds, (hipStream_t)stream, p);
    return dctr_launch_status("dctr_din_attn_pool_fwd");
}

// a13 with the lookups folded in (reference models/sequence/din.py:62-76 + layers/sequence.py:261-298): the behaviour sequences'
// key rows and the query rows are read from the embedding tables inside the kernels, [B, T, E] keys never exist in HBM.
extern "C" int dctr_din_attn_gather_fwd(const dctr_din_attn_args_t* a, const dctr_din_gather_t* g, void* stream) {
    DCTR_REQUIRE(a != nullptr && g != nullptr, DCTR_E_NULL, "din_attn_gather_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->maxlen >= 1 && a->dim >= 1 && a->n_layers >= 0 && a->n_layers <= DIN_MAX_LAYERS, DCTR_E_DIM,
                 "din_attn_gather_fwd: bad sizes (B=%lld T=%d E=%d layers=%d)", (long long)a->batch, a->maxlen, a->dim, a->n_layers);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->out && a->out_kernel && a->out_bias && a->units && a->kernels && a->biases, DCTR_E_NULL, "din_attn_gather_fwd: null pointer");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_DICE, DCTR_E_ENUM, "din_attn_gather_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(a->out_stride >= a->dim, DCTR_E_DIM, "din_attn_gather_fwd: out_stride < dim");
    DCTR_REQUIRE(g->n_feats >= 1 && g->n_feats <= 2, DCTR_E_UNSUPPORTED, "din_attn_gather_fwd: %d history features (1 or 2)", g->n_feats);
    DCTR_REQUIRE(a->dim % (16 * g->n_feats) == 0, DCTR_E_UNSUPPORTED,
                 "din_attn_gather_fwd: every history feature must be dim / n_feats wide, a multiple of 16 (dim %d, %d features)", a->dim, g->n_feats);
    const int64_t rows = a->batch * (int64_t)a->maxlen;
    DCTR_REQUIRE(rows < 0x7fffffffLL, DCTR_E_DIM, "din_attn_gather_fwd: batch x maxlen too large");
    for (int h = 0; h < g->n_feats; ++h) {
        DCTR_REQUIRE(g->hist_ids[h] && g->query_ids[h] && g->hist_table[h] && g->query_table[h], DCTR_E_NULL, "din_attn_gather_fwd: feature %d null", h);
        DCTR_REQUIRE(g->hist_vocab[h] >= 1 && g->hist_vocab[h] < 0x7fffffffLL && g->query_vocab[h] >= 1 && g->query_vocab[h] < 0x7fffffffLL, DCTR_E_DIM,
                     "din_attn_gather_fwd: vocabulary of feature %d", h);
        DCTR_REQUIRE(dctr_aligned16(g->hist_table[h]) && dctr_aligned16(g->query_table[h]), DCTR_E_ALIGN, "din_attn_gather_fwd: tables 16-B aligned");
    }
    DCTR_REQUIRE(g->hist_stride >= a->maxlen && g->query_stride >= 1, DCTR_E_DIM, "din_attn_gather_fwd: id strides");
    DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= (size_t)rows * sizeof(float), DCTR_E_NULL,
                 "din_attn_gather_fwd: needs the [B * T] float workspace (dctr_din_attn_workspace_bytes)");
    if (a->activation == DCTR_ACT_DICE)
        DCTR_REQUIRE(a->dice_alpha && a->dice_mean && a->dice_var, DCTR_E_NULL, "din_attn_gather_fwd: dice without parameters");
    const size_t pool_lds = (size_t)4 * 3 * a->maxlen * sizeof(float);
    DCTR_REQUIRE(pool_lds <= 64 * 1024, DCTR_E_UNSUPPORTED, "din_attn_gather_fwd: maxlen %d too long", a->maxlen);
    float* raw = static_cast<float*>(a->workspace);
    const int ok = dctr_din_chain::try_launch(nullptr, nullptr, a->batch, a->maxlen, a->dim, a->n_layers, a->units, a->kernels, a->biases,
                                              a->activation, a->dice_alpha, a->dice_mean, a->dice_var, a->dice_eps, a->out_kernel,
                                              a->out_bias, raw, (hipStream_t)stream, g, nullptr, din_compact_area(a, rows),
                                              din_image_area(a, rows));
    DCTR_REQUIRE(ok, DCTR_E_UNSUPPORTED,
                 "din_attn_gather_fwd: the row-chained score kernel takes two-layer attention MLPs (units[0] <= 112, units[1] <= 64) and "
                 "dim 16 / 32 / 64 — use the lookups + dctr_din_attn_pool_fwd");
    DinPoolGather pg{};
    pg.nf = g->n_feats;
    pg.ids_i64 = g->ids_is_i64;
    pg.hist_stride = g->hist_stride;
    for (int h = 0; h < g->n_feats; ++h) {
        pg.hist_ids[h] = g->hist_ids[h];
        pg.hist_table[h] = g->hist_table[h];
        pg.hist_vocab[h] = g->hist_vocab[h];
        pg.mask_zero[h] = g->mask_zero[h];
    }
    hipLaunchKernelGGL(din_pool_gather_kernel, dim3((unsigned)dctr_ceil_div(a->batch, (int64_t)4)), dim3(256), pool_lds, (hipStream_t)stream,
                       raw, pg, a->batch, a->maxlen, a->dim, a->weight_normalization, a->out, a->out_stride, a->scores);
    return dctr_launch_status("dctr_din_attn_gather_fwd");
}
