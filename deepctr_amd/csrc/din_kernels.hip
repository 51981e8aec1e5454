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

#include "dctr_common.h"
#include "mfma_tile.h"

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

}  // namespace

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
