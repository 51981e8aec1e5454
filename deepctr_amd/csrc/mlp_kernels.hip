// adjacent to the hot path — DNN.call (+ Dense(1,use_bias=False) head + add_func + PredictionLayer.call)
// reference deepctr/layers/core.py:189-208, :250-259, layers/utils.py:328-333.
//
// One kernel runs the WHOLE multilayer perceptron for a 16-row tile of the batch: activations never
// leave LDS between layers, weights (603 KB for 429-256-128-64) stream from L2, and the head
// (Dense(1) + linear/FM logits + global bias + sigmoid) is the epilogue.  Replaces per layer in the
// reference: tensordot + bias_add + activation (+ BN/Dice) kernels, then Dense, Add, bias_add, sigmoid.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32) — see mfma_tile.h.  At B = 4096 that is 256
// workgroups = one per CU, four waves each (one per SIMD); each wave owns a 16*TPW-column slice of the
// layer output and walks K.  fp32 MFMA floor for 429-256-128-64: 301.7 kFLOP/sample -> 7.9 us / 4096.
#include "dctr_common.h"
#include "mfma_tile.h"

namespace {

constexpr int MAX_LAYERS = 8;

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
    int32_t lda;  // LDS row stride (floats), >= pad16(max width) + 4
};

// ---------------------------------------------------------------------------------------------------
// One layer:  out[16 x N] = act(in[16 x K] @ W[K x N] + bias)
//
// W streams through LDS in K-chunks of KC rows (a chunk = KC*N contiguous floats of the Keras-layout kernel,
// <= 32 KB), double-buffered: while the four waves issue the MFMAs of chunk c, every thread already has the
// global_load_dwordx4s of chunk c+1 in flight (fully coalesced: 4 KB per wave-instruction), and parks them in
// the other LDS buffer before the single barrier of the iteration.  A wave's B fragment then is ONE
// ds_read_b128/b64/b32 per k-step and its A fragment two ds_read_b128 per chunk — the global-memory latency
// that used to sit between every four MFMAs (one wave per SIMD, nothing to switch to) is off the critical path.
// k-slot mapping inside a chunk: MFMA slot g = lane>>4 owns k = k0 + g*(KC/4) + t.
// ---------------------------------------------------------------------------------------------------
constexpr int WBUF_FLOATS = 8192;   // one weight buffer: 32 KB
constexpr int MAXV = WBUF_FLOATS / (256 * 4);   // float4 per thread per chunk (8)
constexpr int ACC_SLOTS = 8;        // f32x4 accumulators per wave (MAXT wave-tiles x TPW columns)

__device__ __forceinline__ int chunk_rows(int K, int N) {
    int kc = (WBUF_FLOATS / N) & ~15;
    const int kp = dctr::pad16(K);
    return kc < kp ? kc : kp;
}

template <int ACT>
__device__ __forceinline__ float act_t(float v, float al, float mu, float var, float eps) {
    if constexpr (ACT == DCTR_ACT_DICE) return dctr::dice_act(v, al, mu, var, eps);
    else if constexpr (ACT == DCTR_ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == DCTR_ACT_SIGMOID) return dctr::sigmoidf_(v);
    else if constexpr (ACT == DCTR_ACT_TANH) return tanhf(v);
    else return v;
}

// All MAXV loads are unconditional (clamped to a safe address) and carry no control flow, so they are issued
// back-to-back and stay in flight during the MFMAs; masking happens when they are parked in LDS.
__device__ __forceinline__ void chunk_fetch(const float* __restrict__ W, int K, int N, int k0, float4 (&r)[MAXV]) {
    const int64_t kn = (int64_t)K * N;
    const int64_t g_safe = kn >= 4 ? ((kn - 4) & ~(int64_t)3) : 0;
    const int64_t g0 = (int64_t)k0 * N;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        int64_t gi = g0 + (v * 256 + threadIdx.x) * 4;
        gi = gi < g_safe ? gi : g_safe;
        r[v] = *reinterpret_cast<const float4*>(W + gi);
    }
}

__device__ __forceinline__ void chunk_park(float* wbuf, const float* __restrict__ W, int K, int N, int k0, int KC,
                                           const float4 (&r)[MAXV]) {
    const int total = KC * N;
    const int valid = max(0, min(KC, K - k0)) * N;
    const int64_t kn = (int64_t)K * N;
    const int64_t g0 = (int64_t)k0 * N;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int i = (v * 256 + threadIdx.x) * 4;
        if (i < total) {
            float4 t = r[v];
            if (i + 3 >= valid) {                               // chunk tail: rows past K are zero
                const bool whole = g0 + i + 3 < kn;             // was the float4 read in place (not clamped)?
                float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (i + c >= valid) e[c] = 0.f;
                    else if (!whole) e[c] = W[g0 + i + c];       // <= 3 floats of the very last row of W
                }
                t = make_float4(e[0], e[1], e[2], e[3]);
            }
            *reinterpret_cast<float4*>(wbuf + i) = t;
        }
    }
}

// MFMAs of one chunk for the NT wave-tiles this wave owns (compile-time NT: no per-step branches)
template <int TPW, int NT>
__device__ __forceinline__ void chunk_compute(const float* arow, const float* brow, int KQ, int N, const int* ncol,
                                              dctr::f32x4 (&acc)[ACC_SLOTS / TPW][TPW]) {
    for (int t0 = 0; t0 < KQ; t0 += 4) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + t0);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        float b[4][NT][TPW];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < NT; ++i) dctr::load_cols<TPW>(brow + (t0 + tt) * N + ncol[i], b[tt][i]);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int cc = 0; cc < TPW; ++cc)
                    acc[i][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tt], b[tt][i][cc], acc[i][cc], 0, 0, 0);
    }
}

template <int TPW, int ACT>
__device__ __forceinline__ void layer_chunked(const MlpParams& p, int l, const float* in, float* out, float* wbuf0,
                                              float* wbuf1, int K, int N) {
    using dctr::f32x4;
    constexpr int MAXT = ACC_SLOTS / TPW;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KC = chunk_rows(K, N);
    const int KQ = KC / 4;                                   // k per MFMA slot per chunk, multiple of 4
    const int n_chunks = (K + KC - 1) / KC;
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    const int nt_w = wave < n_tiles ? (n_tiles - wave + 3) / 4 : 0;     // wave-tiles owned by this wave
    const float* W = p.W[l];

    f32x4 acc[MAXT][TPW];
#pragma unroll
    for (int i = 0; i < MAXT; ++i)
#pragma unroll
        for (int c = 0; c < TPW; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ncol[MAXT];                                          // first column of this lane in wave-tile i (clamped)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        int n0 = (wave + 4 * i) * 16 * TPW + TPW * j;
        if (n0 + TPW > N) n0 = N - TPW;
        ncol[i] = n0;
    }

    float4 stage[MAXV];
    chunk_fetch(W, K, N, 0, stage);
    chunk_park(wbuf0, W, K, N, 0, KC, stage);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const float* wb = (c & 1) ? wbuf1 : wbuf0;
        float* wnext = (c & 1) ? wbuf0 : wbuf1;
        const bool more = c + 1 < n_chunks;
#ifndef DCTR_LAB_NO_FETCH
        if (more) chunk_fetch(W, K, N, (c + 1) * KC, stage);               // in flight during the MFMAs below
#endif
        const float* arow = in + j * p.lda + c * KC + g * KQ;
        const float* brow = wb + (g * KQ) * N;
#ifndef DCTR_LAB_NO_MFMA
        switch (nt_w) {
            case 1: chunk_compute<TPW, 1>(arow, brow, KQ, N, ncol, acc); break;
            case 2: if constexpr (MAXT >= 2) chunk_compute<TPW, 2>(arow, brow, KQ, N, ncol, acc); break;
            case 3: if constexpr (MAXT >= 4) chunk_compute<TPW, 3>(arow, brow, KQ, N, ncol, acc); break;
            case 4: if constexpr (MAXT >= 4) chunk_compute<TPW, 4>(arow, brow, KQ, N, ncol, acc); break;
            case 5: if constexpr (MAXT >= 8) chunk_compute<TPW, 5>(arow, brow, KQ, N, ncol, acc); break;
            case 6: if constexpr (MAXT >= 8) chunk_compute<TPW, 6>(arow, brow, KQ, N, ncol, acc); break;
            case 7: if constexpr (MAXT >= 8) chunk_compute<TPW, 7>(arow, brow, KQ, N, ncol, acc); break;
            case 8: if constexpr (MAXT >= 8) chunk_compute<TPW, 8>(arow, brow, KQ, N, ncol, acc); break;
            default: break;
        }
#endif
#ifndef DCTR_LAB_NO_PARK
        if (more) chunk_park(wnext, W, K, N, (c + 1) * KC, KC, stage);
#endif
        __syncthreads();
    }

    // epilogue: bias + activation -> LDS (C layout: row = 4g + r, col = tile base + TPW*j + c)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        if ((wave + 4 * i) < n_tiles) {
#pragma unroll
            for (int cc = 0; cc < TPW; ++cc) {
                const int n = (wave + 4 * i) * 16 * TPW + TPW * j + cc;
                if (n < N) {
                    const float bv = p.bias[l] != nullptr ? p.bias[l][n] : 0.f;
                    float al = 0.f, mu = 0.f, var = 1.f;
                    if constexpr (ACT == DCTR_ACT_DICE) {
                        al = p.dice_alpha[l][n];
                        mu = p.dice_mean[l][n];
                        var = p.dice_var[l][n];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(4 * g + r) * p.lda + n] = act_t<ACT>(acc[i][cc][r] + bv, al, mu, var, p.dice_eps);
                }
            }
        }
    }
    // zero the K padding the NEXT layer will read: columns [N, lda)
    for (int i = threadIdx.x; i < 16 * (p.lda - N); i += 256) {
        const int r = i / (p.lda - N), c = N + i % (p.lda - N);
        out[r * p.lda + c] = 0.f;
    }
}

template <int ACT>
__device__ __forceinline__ void layer_dispatch(const MlpParams& p, int l, const float* in, float* out, float* wbuf0,
                                               float* wbuf1, int K, int N) {
    if (N % 64 == 0) layer_chunked<4, ACT>(p, l, in, out, wbuf0, wbuf1, K, N);
    else if (N % 32 == 0) layer_chunked<2, ACT>(p, l, in, out, wbuf0, wbuf1, K, N);
    else layer_chunked<1, ACT>(p, l, in, out, wbuf0, wbuf1, K, N);
}

__global__ __launch_bounds__(256) void mlp_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + 16 * p.lda;
    float* wbuf0 = smem + 32 * p.lda;
    float* wbuf1 = wbuf0 + WBUF_FLOATS;
    const int64_t b0 = (int64_t)blockIdx.x * 16;

    // stage the input tile (rows beyond the batch and the K padding are zero).  All global loads of a pass are
    // issued before the first LDS store: a plain load->store loop serialises ~27 dependent memory round trips.
    {
        const int n4 = p.lda / 4;                                  // float4 per LDS row (lda % 4 == 0)
        const int in4 = (p.in_dim + 3) / 4;
        const bool vec = (p.x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15u) == 0);
        constexpr int U = 8;
        for (int base = 0; base < 16 * n4; base += 256 * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256 + threadIdx.x;
                const int r = idx / n4, c4 = idx % n4;
                const int64_t b = b0 + r;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < 16 * n4 && b < p.batch && c4 < in4) {
                    const float* src = p.x + b * p.x_stride + 4 * c4;
                    if (vec) {
                        v[u] = *reinterpret_cast<const float4*>(src);
                    } else {
                        v[u].x = src[0];
                        if (4 * c4 + 1 < p.in_dim) v[u].y = src[1];
                        if (4 * c4 + 2 < p.in_dim) v[u].z = src[2];
                        if (4 * c4 + 3 < p.in_dim) v[u].w = src[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256 + threadIdx.x;
                if (idx < 16 * n4) {
                    const int c4 = idx % n4;
                    float4 t = v[u];
                    if (4 * c4 + 1 >= p.in_dim) t.y = 0.f;          // never let stride padding of x into the tile
                    if (4 * c4 + 2 >= p.in_dim) t.z = 0.f;
                    if (4 * c4 + 3 >= p.in_dim) t.w = 0.f;
                    if (4 * c4 >= p.in_dim) t.x = 0.f;
                    *reinterpret_cast<float4*>(buf0 + 4 * idx) = t;
                }
            }
        }
    }
    __syncthreads();

    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.units[l];
        switch (p.activation) {
            case DCTR_ACT_RELU: layer_dispatch<DCTR_ACT_RELU>(p, l, in, out, wbuf0, wbuf1, K, N); break;
            case DCTR_ACT_SIGMOID: layer_dispatch<DCTR_ACT_SIGMOID>(p, l, in, out, wbuf0, wbuf1, K, N); break;
            case DCTR_ACT_TANH: layer_dispatch<DCTR_ACT_TANH>(p, l, in, out, wbuf0, wbuf1, K, N); break;
            case DCTR_ACT_DICE: layer_dispatch<DCTR_ACT_DICE>(p, l, in, out, wbuf0, wbuf1, K, N); break;
            default: layer_dispatch<DCTR_ACT_LINEAR>(p, l, in, out, wbuf0, wbuf1, K, N); break;
        }
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        K = N;
    }

    if (p.has_head) {
        // logit[row] = h[row,:] . head_w (+ extra logits + global bias), sigmoid for task == binary
        const int row = threadIdx.x >> 4, part = threadIdx.x & 15;
        float acc = 0.f;
        for (int n = part; n < K; n += 16) acc = fmaf(in[row * p.lda + n], p.head_w[n], acc);
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        const int64_t b = b0 + row;
        if (part == 0 && b < p.batch) {
            float v = acc;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (p.add[i] != nullptr) v += p.add[i][b];
            if (p.global_bias != nullptr) v += p.global_bias[0];
            if (p.sigmoid_out) v = dctr::sigmoidf_(v);
            p.y[b] = v;
        }
    } else {
        for (int i = threadIdx.x; i < 16 * K; i += 256) {
            const int r = i / K, c = i % K;
            const int64_t b = b0 + r;
            if (b < p.batch) p.y[b * p.y_stride + c] = in[r * p.lda + c];
        }
    }
}

// LDS row stride: every layer reads its input padded to n_chunks*KC columns (zero beyond K)
int mlp_lda(const dctr_mlp_args_t* a) {
    int need = (a->in_dim + 15) & ~15;
    int K = a->in_dim;
    for (int l = 0; l < a->n_layers; ++l) {
        const int N = a->units[l];
        int kc = (WBUF_FLOATS / N) & ~15;
        const int kp = (K + 15) & ~15;
        kc = kc < kp ? kc : kp;
        const int padded = ((K + kc - 1) / kc) * kc;
        need = padded > need ? padded : need;
        need = N > need ? N : need;
        K = N;
    }
    return ((need + 15) & ~15) + 4;
}

}  // namespace

extern "C" size_t dctr_mlp_workspace_bytes(const dctr_mlp_args_t*) { return 0; }  // activations live in LDS

extern "C" int dctr_mlp_fwd(const dctr_mlp_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "mlp_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->in_dim >= 1 && a->n_layers >= 0 && a->n_layers <= MAX_LAYERS, DCTR_E_DIM,
                 "mlp_fwd: bad sizes (batch=%lld in_dim=%d layers=%d, max %d layers)", (long long)a->batch, a->in_dim,
                 a->n_layers, MAX_LAYERS);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->y, DCTR_E_NULL, "mlp_fwd: null x / y");
    DCTR_REQUIRE(a->n_layers == 0 || (a->units && a->kernels && a->biases), DCTR_E_NULL, "mlp_fwd: null layer arrays");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_DICE, DCTR_E_ENUM, "mlp_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(!a->has_head || a->head_w, DCTR_E_NULL, "mlp_fwd: has_head without head_w");
    if (a->activation == DCTR_ACT_DICE)
        DCTR_REQUIRE(a->dice_alpha && a->dice_mean && a->dice_var, DCTR_E_NULL, "mlp_fwd: dice without parameters");
    MlpParams p{};
    p.x = a->x;
    p.batch = a->batch;
    p.x_stride = a->x_stride;
    p.in_dim = a->in_dim;
    p.n_layers = a->n_layers;
    for (int l = 0; l < a->n_layers; ++l) {
        DCTR_REQUIRE(a->units[l] >= 1 && a->units[l] <= 512, DCTR_E_UNSUPPORTED,
                     "mlp_fwd: units[%d]=%d outside [1, 512] (one 32-KB weight chunk must hold >= 16 rows)", l, a->units[l]);
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
    const size_t lds = ((size_t)2 * 16 * p.lda + 2 * WBUF_FLOATS) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "mlp_fwd: layer width needs %zu B of LDS (> 160 KiB)", lds);
    {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, 16);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: batch too large");
    DCTR_LAUNCH(mlp_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    return dctr_launch_status("dctr_mlp_fwd");
}
