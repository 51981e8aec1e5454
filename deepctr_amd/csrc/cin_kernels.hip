// a10 — CIN.call (reference deepctr/layers/interaction.py:277-325) on the f32 matrix cores.
//
// Reference per layer k: z[b,d,i*F_k+j] = x0[b,i,d] * x_k[b,j,d]  (tf.matmul of split tensors + reshape,
// :288-295 — MATERIALISED as [D,B,F0*F_k]: 436 MB at C3 layer 1), then conv1d(k=1) = z @ W_k [F0*F_k, H_k]
// (:299-300), bias, activation, transpose to [B,H,D], split_half (first half -> next hidden, second half
// -> output, :308-317), and finally sum over D of the concatenated direct maps (:322-323).
//
// Here the whole network runs in ONE kernel and z is never materialised: a workgroup owns SB samples,
// i.e. M = SB*D GEMM rows (b,d); the A operand of v_mfma_f32_16x16x4_f32 is formed in registers as
// x0[row,i] * x_k[row,j] from two LDS-resident tiles, the four k-slots of one MFMA being four consecutive
// j of the same i; B streams W_k rows from L2; the layer output y[b,h,d] is written back to LDS where it
// is both the next layer's x_{k+1} and the source of the sum over D.  Exact fp32 (fmaf-chain numerics).
// Cost model (C3, F0=26, D=16, H=128,128): 9.58 MFLOP/sample -> 155 TF f32-MFMA => >= 253 us / 4096.
#include "dctr_common.h"
#include "mfma_tile.h"

namespace {

constexpr int CIN_MAX_LAYERS = 8;
constexpr int RT_MAX = 4;  // row tiles (16 rows each) per workgroup

struct CinParams {
    const float* x;
    int64_t batch;
    int64_t x_stride;
    int32_t F0, D, n_layers, split_half, activation;
    int32_t SB;        // samples per workgroup
    int32_t RT;        // row tiles = ceil(SB*D/16)
    int32_t Hmax;      // LDS per-sample stride of the y buffers, in maps
    int32_t out_dim;   // featuremap_num
    int32_t H[CIN_MAX_LAYERS];
    const float* W[CIN_MAX_LAYERS];
    const float* bias[CIN_MAX_LAYERS];
    float* out;
};

template <int TPW>
__device__ __forceinline__ void cin_layer(const CinParams& p, int k, const float* x0s, const float* xk, int xk_stride,
                                          int Fk, float* ycur) {
    using dctr::f32x4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, jl = lane & 15;
    const int D = p.D, F0 = p.F0, H = p.H[k];
    const int M = p.SB * D;
    // per row tile: LDS offsets of this lane's row (s,d) in the x0 tile and in the x_k tile
    int off0[RT_MAX], offk[RT_MAX];
    bool rowok[RT_MAX];
#pragma unroll
    for (int rt = 0; rt < RT_MAX; ++rt) {
        const int m = rt * 16 + jl;
        rowok[rt] = rt < p.RT && m < M;
        const int mm = rowok[rt] ? m : 0;
        const int s = mm / D, d = mm % D;
        off0[rt] = s * F0 * D + d;
        offk[rt] = s * xk_stride + d;
    }
    const int n_tiles = (H + 16 * TPW - 1) / (16 * TPW);
    const int JT = (Fk + 3) / 4;
    const float* Wk = p.W[k];
    for (int wt = wave; wt < n_tiles; wt += 4) {
        const int n_base = wt * 16 * TPW;
        int n0 = n_base + TPW * jl;
        if (n0 + TPW > H) n0 = H - TPW;
        f32x4 acc[RT_MAX][TPW];
#pragma unroll
        for (int rt = 0; rt < RT_MAX; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < F0; ++i) {
            float xi[RT_MAX];
#pragma unroll
            for (int rt = 0; rt < RT_MAX; ++rt) xi[rt] = rowok[rt] ? x0s[off0[rt] + i * D] : 0.f;
            const float* wrow = Wk + (int64_t)i * Fk * H + n0;
#pragma unroll 2
            for (int jt = 0; jt < JT; ++jt) {
                const int j = 4 * jt + g;
                const bool jok = j < Fk;
                const int jj = jok ? j : Fk - 1;
                float b[TPW];
                dctr::load_cols<TPW>(wrow + (int64_t)jj * H, b);
                float a[RT_MAX];
#pragma unroll
                for (int rt = 0; rt < RT_MAX; ++rt) a[rt] = jok ? xi[rt] * xk[offk[rt] + jj * D] : 0.f;
#pragma unroll
                for (int rt = 0; rt < RT_MAX; ++rt)
#pragma unroll
                    for (int c = 0; c < TPW; ++c)
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b[c], acc[rt][c], 0, 0, 0);
            }
        }
        // epilogue: bias + activation, store y[s][n][d] (C layout: row = 4g + r, col = n_base + TPW*jl + c)
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int n = n_base + TPW * jl + c;
            if (n < H) {
                const float bv = p.bias[k][n];
#pragma unroll
                for (int rt = 0; rt < RT_MAX; ++rt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = rt * 16 + 4 * g + r;
                        if (rt < p.RT && m < M) {
                            const int s = m / D, d = m % D;
                            ycur[(s * p.Hmax + n) * D + d] = dctr::apply_act(acc[rt][c][r] + bv, p.activation);
                        }
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void cin_kernel(CinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = p.D, F0 = p.F0, SB = p.SB;
    float* x0s = smem;                              // [SB][F0][D]
    float* y0 = x0s + ((SB * F0 * D + 3) & ~3);     // [SB][Hmax][D]
    float* y1 = y0 + SB * p.Hmax * D;
    const int64_t b0 = (int64_t)blockIdx.x * SB;
    for (int i = threadIdx.x; i < SB * F0 * D; i += 256) {
        const int s = i / (F0 * D);
        x0s[i] = (b0 + s < p.batch) ? p.x[(b0 + s) * p.x_stride + (i - s * F0 * D)] : 0.f;
    }
    __syncthreads();

    const float* xk = x0s;
    int xk_stride = F0 * D, Fk = F0;
    float* ycur = y0;
    float* ynext = y1;
    int out_off = 0;
    for (int k = 0; k < p.n_layers; ++k) {
        const int H = p.H[k];
        if (H % 32 == 0) cin_layer<2>(p, k, x0s, xk, xk_stride, Fk, ycur);
        else cin_layer<1>(p, k, x0s, xk, xk_stride, Fk, ycur);
        __syncthreads();
        // split (interaction.py:308-317): maps [0, Hn) feed the next layer, maps [d0, H) go to the output
        const bool last = k == p.n_layers - 1;
        int Hn, d0;
        if (p.split_half) {
            Hn = last ? 0 : H / 2;
            d0 = last ? 0 : H / 2;
        } else {
            Hn = H;
            d0 = 0;
        }
        const int nd = H - d0;
        // result = reduce_sum(concat(direct), -1): deterministic serial sum over d
        for (int t = threadIdx.x; t < SB * nd; t += 256) {
            const int s = t / nd, n = d0 + t % nd;
            if (b0 + s < p.batch) {
                const float* yp = ycur + (s * p.Hmax + n) * D;
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += yp[d];
                p.out[(b0 + s) * p.out_dim + out_off + (n - d0)] = acc;
            }
        }
        out_off += nd;
        xk = ycur;
        xk_stride = p.Hmax * D;
        Fk = Hn;
        float* t = ycur;
        ycur = ynext;
        ynext = t;
        // no barrier needed here: the next layer writes the OTHER y buffer and only reads this one
    }
}

int cin_out_dim(const dctr_cin_args_t* a) {
    int o = 0;
    for (int k = 0; k < a->n_layers; ++k) {
        const int H = a->layer_size[k];
        const bool last = k == a->n_layers - 1;
        o += a->split_half ? (last ? H : H - H / 2) : H;
    }
    return o;
}

}  // namespace

extern "C" size_t dctr_cin_workspace_bytes(const dctr_cin_args_t*) { return 0; }  // intermediates live in LDS

extern "C" int dctr_cin_fwd(const dctr_cin_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "cin_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->fields >= 1 && a->dim >= 1 && a->n_layers >= 1 && a->n_layers <= CIN_MAX_LAYERS,
                 DCTR_E_DIM, "cin_fwd: bad sizes (B=%lld F=%d D=%d layers=%d)", (long long)a->batch, a->fields, a->dim,
                 a->n_layers);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->out && a->layer_size && a->filters && a->bias, DCTR_E_NULL, "cin_fwd: null pointer");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_TANH, DCTR_E_ENUM, "cin_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(a->x_stride >= (int64_t)a->fields * a->dim, DCTR_E_DIM, "cin_fwd: x_stride < fields*dim");
    DCTR_REQUIRE(a->dim <= 64, DCTR_E_UNSUPPORTED, "cin_fwd: embedding_dim %d > 64 not supported", a->dim);
    CinParams p{};
    p.x = a->x;
    p.batch = a->batch;
    p.x_stride = a->x_stride;
    p.F0 = a->fields;
    p.D = a->dim;
    p.n_layers = a->n_layers;
    p.split_half = a->split_half ? 1 : 0;
    p.activation = a->activation;
    p.SB = 64 / a->dim;
    if (p.SB < 1) p.SB = 1;
    p.RT = (p.SB * a->dim + 15) / 16;
    int hmax = 1;
    for (int k = 0; k < a->n_layers; ++k) {
        const int H = a->layer_size[k];
        DCTR_REQUIRE(H >= 1, DCTR_E_DIM, "cin_fwd: layer_size[%d]=%d", k, H);
        if (a->split_half && k != a->n_layers - 1)
            DCTR_REQUIRE(H % 2 == 0, DCTR_E_DIM,
                         "cin_fwd: layer_size must be even except for the last layer when split_half=True");
        DCTR_REQUIRE(a->filters[k] && a->bias[k], DCTR_E_NULL, "cin_fwd: filters/bias[%d] null", k);
        DCTR_REQUIRE((((uintptr_t)a->filters[k]) & 7u) == 0, DCTR_E_ALIGN, "cin_fwd: filters[%d] not 8-B aligned", k);
        p.H[k] = H;
        p.W[k] = a->filters[k];
        p.bias[k] = a->bias[k];
        hmax = H > hmax ? H : hmax;
    }
    p.Hmax = hmax;
    p.out_dim = cin_out_dim(a);
    p.out = a->out;
    const size_t lds = ((size_t)((p.SB * p.F0 * p.D + 3) & ~3) + (size_t)2 * p.SB * p.Hmax * p.D) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "cin_fwd: needs %zu B of LDS (> 160 KiB)", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)cin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "cin_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, p.SB);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "cin_fwd: batch too large");
    DCTR_LAUNCH(cin_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    return dctr_launch_status("dctr_cin_fwd");
}
