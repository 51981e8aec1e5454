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

template <int TPW>
__device__ __forceinline__ void layer_tiles(const MlpParams& p, int l, const float* in, float* out, int K, int N) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQ = dctr::pad16(K) / 4;
    const int n_tiles = (N + 16 * TPW - 1) / (16 * TPW);
    const int act = p.activation;
    for (int wt = wave; wt < n_tiles; wt += 4) {
        const int n_base = wt * 16 * TPW;
        f32x4 acc[TPW];
#pragma unroll
        for (int c = 0; c < TPW; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        dctr::tile_gemm_kn<TPW>(in, p.lda, K, KQ, p.W[l], N, n_base, acc);
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
                for (int r = 0; r < 4; ++r) {
                    float v = acc[c][r] + bv;
                    v = act == DCTR_ACT_DICE ? dctr::dice_act(v, al, mu, var, p.dice_eps) : dctr::apply_act(v, act);
                    out[(4 * g + r) * p.lda + n] = v;
                }
            }
        }
    }
    // zero the K padding of the next layer: columns [N, pad16(N))
    const int NP = dctr::pad16(N);
    for (int i = threadIdx.x; i < 16 * (NP - N); i += 256) {
        const int r = i / (NP - N), c = N + i % (NP - N);
        out[r * p.lda + c] = 0.f;
    }
}

__global__ __launch_bounds__(256) void mlp_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + 16 * p.lda;
    const int64_t b0 = (int64_t)blockIdx.x * 16;

    // stage the input tile (rows beyond the batch and the K padding are zero)
    {
        const int KP = dctr::pad16(p.in_dim);
        for (int i = threadIdx.x; i < 16 * KP; i += 256) {
            const int r = i / KP, c = i % KP;
            const int64_t b = b0 + r;
            buf0[r * p.lda + c] = (b < p.batch && c < p.in_dim) ? p.x[b * p.x_stride + c] : 0.f;
        }
    }
    __syncthreads();

    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.units[l];
        if (N % 64 == 0) layer_tiles<4>(p, l, in, out, K, N);
        else if (N % 32 == 0) layer_tiles<2>(p, l, in, out, K, N);
        else layer_tiles<1>(p, l, in, out, K, N);
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

int mlp_lda(const dctr_mlp_args_t* a) {
    int w = a->in_dim;
    for (int l = 0; l < a->n_layers; ++l) w = a->units[l] > w ? a->units[l] : w;
    return ((w + 15) & ~15) + 4;
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
    const size_t lds = (size_t)2 * 16 * p.lda * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "mlp_fwd: layer width needs %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, 16);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: batch too large");
    DCTR_LAUNCH(mlp_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    return dctr_launch_status("dctr_mlp_fwd");
}
