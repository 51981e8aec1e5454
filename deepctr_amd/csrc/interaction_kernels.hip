// a8, a9, a11, a12 — stand-alone feature-interaction layers over an already concatenated [B,F,E] tile:
//   FM.call               deepctr/layers/interaction.py:588-604
//   CrossNet.call         deepctr/layers/interaction.py:405-424   (vector: VALU + wave reduction;
//                                                                  matrix: f32 MFMA row-tile GEMM)
//   AFMLayer.call         deepctr/layers/interaction.py:116-146
//   InnerProductLayer.call deepctr/layers/interaction.py:655-678
// All four are bandwidth-trivial once the [F,E] tile of a sample sits in LDS / registers; the
// reference materialises [B, F(F-1)/2, E] pair tensors in HBM for AFM / InnerProduct.
#include <math.h>

#include "dctr_common.h"
#include "mfma_tile.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// FM: lane (s, q) owns embedding dims d = q, q+LPR, ... of sample s and walks the fields.
// ---------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void fm_kernel(const float* __restrict__ x, int64_t batch, int64_t x_stride, int F,
                                                 int E, float* __restrict__ y) {
    constexpr int SPB = 256 / LPR;
    const int s = threadIdx.x / LPR, q = threadIdx.x % LPR;
    const int64_t b = (int64_t)blockIdx.x * SPB + s;
    float acc = 0.f;
    if (b < batch) {
        const float* xb = x + b * x_stride;
        for (int d = q; d < E; d += LPR) {
            float sum = 0.f, sq = 0.f;
            for (int f = 0; f < F; ++f) {
                const float v = xb[f * E + d];
                sum += v;
                sq = fmaf(v, v, sq);
            }
            acc += sum * sum - sq;
        }
    }
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (q == 0 && b < batch) y[b] = 0.5f * acc;
}

// ---------------------------------------------------------------------------------------------------
// CrossNet 'vector': one wave per sample, x0 / x_l kept in registers (NR values per lane).
// ---------------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(256) void cross_vector_kernel(const float* __restrict__ x, int64_t batch, int d,
                                                           int64_t x_stride, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int layers,
                                                           float* __restrict__ y, int64_t y_stride) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    float x0[NR], xl[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = lane + 64 * r;
        x0[r] = i < d ? x[b * x_stride + i] : 0.f;
        xl[r] = x0[r];
    }
    for (int l = 0; l < layers; ++l) {
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = lane + 64 * r;
            dot = fmaf(xl[r], i < d ? w[(int64_t)l * d + i] : 0.f, dot);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = lane + 64 * r;
            const float bv = i < d ? bias[(int64_t)l * d + i] : 0.f;
            xl[r] = x0[r] * dot + bv + xl[r];   // interaction.py:415-416: dot_ + bias + x_l
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int i = lane + 64 * r;
        if (i < d) y[b * y_stride + i] = xl[r];
    }
}

// ---------------------------------------------------------------------------------------------------
// CrossNet 'matrix': 16-sample tile per workgroup, x0 and x_l in LDS, W_l x_l on f32 MFMA.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_matrix_kernel(const float* __restrict__ x, int64_t batch, int d,
                                                           int64_t x_stride, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int layers,
                                                           float* __restrict__ y, int64_t y_stride, int lda) {
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* x0 = smem;                 // [16][lda]
    float* xa = smem + 16 * lda;      // x_l
    float* xb = smem + 32 * lda;      // x_{l+1}
    const int64_t b0 = (int64_t)blockIdx.x * 16;
    const int KP = dctr::pad16(d);
    for (int i = threadIdx.x; i < 16 * KP; i += 256) {
        const int r = i / KP, c = i % KP;
        const float v = (b0 + r < batch && c < d) ? x[(b0 + r) * x_stride + c] : 0.f;
        x0[r * lda + c] = v;
        xa[r * lda + c] = v;
        xb[r * lda + c] = 0.f;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int KQ = KP / 4;
    const int n_tiles = (d + 15) / 16;
    for (int l = 0; l < layers; ++l) {
        const float* W = w + (int64_t)l * d * d;
        for (int wt = wave; wt < n_tiles; wt += 4) {
            f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
            dctr::tile_gemm_nk<1>(xa, lda, d, KQ, W, d, wt * 16, acc);
            const int n = wt * 16 + j;
            if (n < d) {
                const float bv = bias[(int64_t)l * d + n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * g + r;
                    // interaction.py:419-420: x_l = x_0 * (W x_l + b) + x_l
                    xb[row * lda + n] = x0[row * lda + n] * (acc[0][r] + bv) + xa[row * lda + n];
                }
            }
        }
        __syncthreads();
        float* t = xa;
        xa = xb;
        xb = t;
    }
    for (int i = threadIdx.x; i < 16 * d; i += 256) {
        const int r = i / d, c = i % d;
        if (b0 + r < batch) y[(b0 + r) * y_stride + c] = xa[r * lda + c];
    }
}

// ---------------------------------------------------------------------------------------------------
// AFM: one wave per sample; the [F,E] tile, W, b, h, p in LDS; lanes walk the F(F-1)/2 pairs.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_ij(int p, int F, int& i, int& j) {
    // row-major enumeration of i<j (itertools.combinations order, interaction.py:126-128)
    int ii = 0, rem = p;
    while (rem >= F - 1 - ii) {
        rem -= F - 1 - ii;
        ++ii;
    }
    i = ii;
    j = ii + 1 + rem;
}

__global__ __launch_bounds__(256) void afm_kernel(const float* __restrict__ x, int64_t batch, int F, int E,
                                                  const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                  const float* __restrict__ proj_h, const float* __restrict__ proj_p,
                                                  int A, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    float* wsh = smem;                       // [E*A] attention_W, then b[A], h[A], p[E]
    float* bsh = wsh + E * A;
    float* hsh = bsh + A;
    float* psh = hsh + A;
    float* per_wave = psh + E;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = per_wave + wave * (F * E + P);   // [F*E] sample tile
    float* logit = xs + F * E;                   // [P]
    for (int i = threadIdx.x; i < E * A; i += 256) wsh[i] = att_w[i];
    for (int i = threadIdx.x; i < A; i += 256) {
        bsh[i] = att_b[i];
        hsh[i] = proj_h[i];
    }
    for (int i = threadIdx.x; i < E; i += 256) psh[i] = proj_p[i];
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool valid = b < batch;
    if (valid)
        for (int i = lane; i < F * E; i += 64) xs[i] = x[b * (int64_t)F * E + i];
    __syncthreads();
    if (!valid) return;

    // pass 1: attention logits per pair (interaction.py:132-139)
    float mx = -INFINITY;
    for (int p = lane; p < P; p += 64) {
        int i, j;
        pair_ij(p, F, i, j);
        float lg = 0.f;
        for (int a = 0; a < A; ++a) {
            float t = bsh[a];
            for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e] * xs[j * E + e], wsh[e * A + a], t);
            lg = fmaf(fmaxf(t, 0.f), hsh[a], lg);
        }
        logit[p] = lg;
        mx = fmaxf(mx, lg);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    // pass 2: softmax over pairs, weighted sum of the pair products, projection (interaction.py:138-145)
    float den = 0.f;
    for (int p = lane; p < P; p += 64) {
        const float e_ = expf(logit[p] - mx);
        logit[p] = e_;
        den += e_;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
    float out = 0.f;
    for (int p = lane; p < P; p += 64) {
        int i, j;
        pair_ij(p, F, i, j);
        const float sc = logit[p] / den;
        float t = 0.f;
        for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e] * xs[j * E + e], psh[e], t);
        out = fmaf(sc, t, out);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) out += __shfl_xor(out, m, 64);
    if (lane == 0) y[b] = out;
}

// ---------------------------------------------------------------------------------------------------
// InnerProduct: one wave per sample.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inner_product_kernel(const float* __restrict__ x, int64_t batch, int F, int E,
                                                            int reduce_sum, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = smem + wave * F * E;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool valid = b < batch;
    if (valid)
        for (int i = lane; i < F * E; i += 64) xs[i] = x[b * (int64_t)F * E + i];
    __syncthreads();
    if (!valid) return;
    if (reduce_sum) {
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_ij(p, F, i, j);
            float t = 0.f;
            for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e], xs[j * E + e], t);
            y[b * (int64_t)P + p] = t;
        }
    } else {
        const int64_t total = (int64_t)P * E;
        for (int64_t o = lane; o < total; o += 64) {
            const int p = (int)(o / E), e = (int)(o % E);
            int i, j;
            pair_ij(p, F, i, j);
            y[b * total + o] = xs[i * E + e] * xs[j * E + e];
        }
    }
}

int pow2_at_least(int v, int cap) {
    int l = 1;
    while (l < v && l < cap) l <<= 1;
    return l;
}

}  // namespace

extern "C" int dctr_fm_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, float* y,
                           void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 1 && dim >= 1, DCTR_E_DIM, "fm_fwd: bad sizes B=%lld F=%d E=%d", (long long)batch,
                 fields, dim);
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y, DCTR_E_NULL, "fm_fwd: null pointer");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim, DCTR_E_DIM, "fm_fwd: x_stride < fields*dim");
    const int lpr = pow2_at_least(dim, 64);
    const int64_t blocks = dctr_ceil_div(batch, 256 / lpr);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "fm_fwd: batch too large");
    hipStream_t st = (hipStream_t)stream;
#define CALL_FM(L) \
    DCTR_LAUNCH((fm_kernel<L>), dim3((unsigned)blocks), dim3(256), 0, st, x, batch, x_stride, fields, dim, y)
    switch (lpr) {
        case 1: CALL_FM(1); break;
        case 2: CALL_FM(2); break;
        case 4: CALL_FM(4); break;
        case 8: CALL_FM(8); break;
        case 16: CALL_FM(16); break;
        case 32: CALL_FM(32); break;
        default: CALL_FM(64); break;
    }
#undef CALL_FM
    return dctr_launch_status("dctr_fm_fwd");
}

extern "C" int dctr_crossnet_fwd(const float* x, int64_t batch, int32_t dim, int64_t x_stride, const float* kernels,
                                 const float* bias, int32_t layers, int32_t mode, float* y, int64_t y_stride,
                                 void* stream) {
    DCTR_REQUIRE(batch >= 0 && dim >= 1 && layers >= 0, DCTR_E_DIM, "crossnet_fwd: bad sizes");
    DCTR_REQUIRE(mode == DCTR_CROSS_VECTOR || mode == DCTR_CROSS_MATRIX, DCTR_E_ENUM, "crossnet_fwd: mode %d", mode);
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y && (layers == 0 || (kernels && bias)), DCTR_E_NULL, "crossnet_fwd: null pointer");
    DCTR_REQUIRE(x_stride >= dim && y_stride >= dim, DCTR_E_DIM, "crossnet_fwd: stride < dim");
    hipStream_t st = (hipStream_t)stream;
    if (mode == DCTR_CROSS_VECTOR || layers == 0) {
        DCTR_REQUIRE(dim <= 64 * 32, DCTR_E_UNSUPPORTED, "crossnet_fwd(vector): dim %d > 2048", dim);
        const int64_t blocks = dctr_ceil_div(batch, 4);
        DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "crossnet_fwd: batch too large");
        const int nr = pow2_at_least((dim + 63) / 64, 32);
#define CALL_CV(N)                                                                                                  \
    DCTR_LAUNCH((cross_vector_kernel<N>), dim3((unsigned)blocks), dim3(256), 0, st, x, batch, dim, x_stride, \
                       kernels, bias, layers, y, y_stride)
        switch (nr) {
            case 1: CALL_CV(1); break;
            case 2: CALL_CV(2); break;
            case 4: CALL_CV(4); break;
            case 8: CALL_CV(8); break;
            case 16: CALL_CV(16); break;
            default: CALL_CV(32); break;
        }
#undef CALL_CV
    } else {
        const int lda = ((dim + 15) & ~15) + 4;
        const size_t lds = (size_t)3 * 16 * lda * sizeof(float);
        DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "crossnet_fwd(matrix): dim %d needs %zu B of LDS", dim, lds);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)cross_matrix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds);
            DCTR_REQUIRE(e == hipSuccess, (int)e, "crossnet_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        }
        const int64_t blocks = dctr_ceil_div(batch, 16);
        DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "crossnet_fwd: batch too large");
        DCTR_LAUNCH(cross_matrix_kernel, dim3((unsigned)blocks), dim3(256), lds, st, x, batch, dim, x_stride, kernels,
                           bias, layers, y, y_stride, lda);
    }
    return dctr_launch_status("dctr_crossnet_fwd");
}

extern "C" int dctr_afm_fwd(const float* x, int64_t batch, int32_t fields, int32_t dim, const float* att_w,
                            const float* att_b, const float* proj_h, const float* proj_p, int32_t att_factor, float* y,
                            void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 2 && dim >= 1 && att_factor >= 1, DCTR_E_DIM, "afm_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && att_w && att_b && proj_h && proj_p && y, DCTR_E_NULL, "afm_fwd: null pointer");
    const int P = fields * (fields - 1) / 2;
    const size_t lds = ((size_t)dim * att_factor + 2 * att_factor + dim + 4 * ((size_t)fields * dim + P)) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "afm_fwd: needs %zu B of LDS", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)afm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "afm_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(batch, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "afm_fwd: batch too large");
    DCTR_LAUNCH(afm_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, batch, fields, dim,
                       att_w, att_b, proj_h, proj_p, att_factor, y);
    return dctr_launch_status("dctr_afm_fwd");
}

extern "C" int dctr_inner_product_fwd(const float* x, int64_t batch, int32_t fields, int32_t dim, int32_t reduce_sum,
                                      float* y, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 2 && dim >= 1, DCTR_E_DIM, "inner_product_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y, DCTR_E_NULL, "inner_product_fwd: null pointer");
    const size_t lds = (size_t)4 * fields * dim * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "inner_product_fwd: needs %zu B of LDS", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)inner_product_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "inner_product_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(batch, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "inner_product_fwd: batch too large");
    DCTR_LAUNCH(inner_product_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, batch, fields,
                       dim, reduce_sum, y);
    return dctr_launch_status("dctr_inner_product_fwd");
}
