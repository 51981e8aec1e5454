// a8, a9, a11, a12 — stand-alone feature-interaction layers over an already concatenated [B,F,E] tile:
//   FM.call               deepctr/layers/interaction.py:588-604
//   CrossNet.call         deepctr/layers/interaction.py:405-424   (vector: VALU + wave reduction;
//                                                                  matrix: f32 MFMA row-tile GEMM)
//   AFMLayer.call         deepctr/layers/interaction.py:116-146
//   InnerProductLayer.call deepctr/layers/interaction.py:655-678
// All four are bandwidth-trivial once the [F,E] tile of a sample sits in LDS / registers; the
// reference materialises [B, F(F-1)/2, E] pair tensors in HBM for AFM / InnerProduct.
#include <math.h>

#include <stdlib.h>
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
                                                           float* __restrict__ y, int64_t y_stride,
                                                           const float* __restrict__ head_w, float* __restrict__ logit) {
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
    if (y != nullptr) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = lane + 64 * r;
            if (i < d) y[b * y_stride + i] = xl[r];
        }
    }
    if (head_w != nullptr) {            // this branch's share of the model's Dense(1): logit[b] = x_L[b, :] . head_w
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int i = lane + 64 * r;
            dot = fmaf(xl[r], i < d ? head_w[i] : 0.f, dot);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
        if (lane == 0) logit[b] = dot;
    }
}

// The same layer for rows past the register file (d > 8192; the reference has no limit, interaction.py:405-424).  The vector recurrence has a
// closed form: with s_l = x_l . w_l,  x_{l+1} = x_0 s_l + b_l + x_l  gives  x_l = (1 + S_l) x_0 + B_l,  S_l = s_0 + .. + s_{l-1},
// B_l = b_0 + .. + b_{l-1}, hence  s_l = (1 + S_l) (x_0 . w_l) + B_l . w_l: a row needs its L (+ 1: the head) dot products with x_0 — ONE walk
// over the row — and the weight-only constants c_l = B_l . v_l (v_l = w_l, v_L = head_w), which every workgroup forms for itself (no scratch:
// O(L^2 d) multiply-adds per four samples).  x_L is then written in a second walk.  Same arithmetic as the folded CrossNet of the one-launch
// forward (mlp_device.h: cross_logit); up to CROSS_STREAM_MAXL layers.
constexpr int CROSS_STREAM_MAXL = 8;
__global__ __launch_bounds__(256) void cross_vector_stream_kernel(const float* __restrict__ x, int64_t batch, int d, int64_t x_stride,
                                                                  const float* __restrict__ w, const float* __restrict__ bias, int layers,
                                                                  float* __restrict__ y, int64_t y_stride, const float* __restrict__ head_w,
                                                                  float* __restrict__ logit) {
    __shared__ float cs[CROSS_STREAM_MAXL + 1];
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nv = layers + (head_w != nullptr ? 1 : 0);
    for (int l = 0; l < nv; ++l) {                        // c_l = (b_0 + .. + b_{l-1}) . v_l
        const float* v = l < layers ? w + (int64_t)l * d : head_w;
        float acc = 0.f;
        for (int i = threadIdx.x; i < d; i += 256) {
            float bs = 0.f;
            for (int j = 0; j < l; ++j) bs += bias[(int64_t)j * d + i];
            acc = fmaf(bs, v[i], acc);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) cs[l] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    const float* x0 = x + b * x_stride;
    float p[CROSS_STREAM_MAXL + 1];
#pragma unroll
    for (int l = 0; l <= CROSS_STREAM_MAXL; ++l) p[l] = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float xv = x0[i];
#pragma unroll
        for (int l = 0; l <= CROSS_STREAM_MAXL; ++l)
            if (l < nv) p[l] = fmaf(xv, (l < layers ? w + (int64_t)l * d : head_w)[i], p[l]);
    }
#pragma unroll
    for (int l = 0; l <= CROSS_STREAM_MAXL; ++l)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) p[l] += __shfl_xor(p[l], m, 64);
    float S = 0.f;
#pragma unroll
    for (int l = 0; l < CROSS_STREAM_MAXL; ++l)
        if (l < layers) S += fmaf(1.f + S, p[l], cs[l]);
    if (y != nullptr)
        for (int i = lane; i < d; i += 64) {
            float bs = 0.f;
            for (int j = 0; j < layers; ++j) bs += bias[(int64_t)j * d + i];
            y[b * y_stride + i] = fmaf(1.f + S, x0[i], bs);
        }
    if (head_w != nullptr && lane == 0) {
        float ph = 0.f;
#pragma unroll
        for (int l = 0; l <= CROSS_STREAM_MAXL; ++l)
            if (l == layers) ph = p[l];
        logit[b] = fmaf(1.f + S, ph, cs[layers]);
    }
}

// ---------------------------------------------------------------------------------------------------
// CrossNet 'matrix' (interaction.py:416-420): x_{l+1} = x_0 * (W_l x_l + b_l) + x_l.
// 16-sample tile per workgroup of 8 waves; x_0, x_l, x_{l+1} in LDS; W_l x_l on v_mfma_f32_16x16x4_f32.
// W_l is [out n, in k] row-major, i.e. the MFMA B operand walks k ALONG a row: a lane loads 16 B = 4 consecutive k
// of row n0 + j (raw buffer load, lane-constant offset + scalar k advance), and the MFMA k-slot g of step u takes
// k = 16t + 4g + u, so that the matching A fragment is one ds_read_b128 of the (un-permuted) x_l tile.  Three
// register stages of 32 k each keep two stages of loads in flight under the MFMAs (the 736 KB of W_l at d = 429
// stream from L2 once per workgroup and layer; that stream, ~16 B/clk per CU, bounds the kernel, not the MFMAs).
// When d % 4 != 0 (or W is not 16-B aligned) the rows are first re-packed to an aligned stride in the workspace.
// ---------------------------------------------------------------------------------------------------
typedef unsigned int cross_u32x4 __attribute__((ext_vector_type(4)));
constexpr int CROSS_WAVES = 8;

__device__ __forceinline__ void cross_load_stage(__amdgpu_buffer_rsrc_t rsrc, int voff, const float* arow, int s,
                                                 float4 (&b)[2], float4 (&a)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const cross_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (2 * s + h) * 64, 0);
        b[h] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
        a[h] = *reinterpret_cast<const float4*>(arow + (2 * s + h) * 16);
    }
}

// W rows re-packed to a 16-B aligned stride (zero tail): dwordx4 loads need 16-B aligned addresses, and the raw-buffer
// bounds check drops a WHOLE dwordx4 that straddles the end of the array
__global__ __launch_bounds__(256) void cross_repack_kernel(const float* __restrict__ w, int d, int dp, int64_t rows,
                                                           float* __restrict__ out) {
    const int64_t total = rows * dp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dp;
        const int c = (int)(i - r * dp);
        out[i] = c < d ? w[r * d + c] : 0.f;
    }
}

// RT row tiles of 16 samples per workgroup.  RT = 1 (x_0, x_l, x_{l+1} in LDS): every CU has a workgroup from 4096 rows on, but
// each weight fragment feeds ONE MFMA tile.  RT = 2 (launches of >= 32 rows per CU, i.e. predict()'s 16,384-row spans): the weight
// stream per row halves — a B fragment feeds both row tiles — and x_0 is re-read from the input (L2) in the epilogue instead of
// holding a third LDS tile, so that the two 32-row tiles still fit the 160 KiB.
template <int RT>
__device__ __forceinline__ void cross_load_stage_rt(__amdgpu_buffer_rsrc_t rsrc, int voff, const float* arow, int lda, int s,
                                                    float4 (&b)[2], float4 (&a)[RT][2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const cross_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (2 * s + h) * 64, 0);
        b[h] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt][h] = *reinterpret_cast<const float4*>(arow + rt * 16 * lda + (2 * s + h) * 16);
    }
}

template <int RT>
__device__ __forceinline__ void cross_mfma_stage_rt(const float4 (&a)[RT][2], const float4 (&b)[2], dctr::f32x4 (&acc)[RT]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][h].x, b[h].x, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][h].y, b[h].y, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][h].z, b[h].z, acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][h].w, b[h].w, acc[rt], 0, 0, 0);
    }
}

template <int RT>
__global__ __launch_bounds__(64 * CROSS_WAVES) void cross_matrix_kernel(const float* __restrict__ x, int64_t batch, int d,
                                                                        int64_t x_stride, const float* __restrict__ w,
                                                                        int wstride, const float* __restrict__ bias,
                                                                        int layers, float* __restrict__ y, int64_t y_stride,
                                                                        int lda, const float* __restrict__ head_w,
                                                                        float* __restrict__ logit, float* __restrict__ save_u,
                                                                        float* __restrict__ save_x) {
    using dctr::f32x4;
    constexpr int NTHR = 64 * CROSS_WAVES;
    constexpr int ROWS = 16 * RT;
    constexpr bool X0_LDS = RT == 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xa = smem;                                   // x_l       [ROWS][lda], columns >= d zero up to the next multiple of 32
    float* xb = smem + ROWS * lda;                      // x_{l+1}
    float* x0 = smem + 2 * ROWS * lda;                  // x_0 (RT == 1 only)
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const int KP = (d + 31) & ~31;
    for (int base = 0; base < ROWS * KP; base += NTHR * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NTHR + threadIdx.x;
            const int r = min(i / KP, ROWS - 1), c = i % KP;
            const int64_t b = min(b0 + r, batch - 1);
            v[u] = x[b * x_stride + min(c, d - 1)];
            if (b0 + r >= batch || c >= d) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NTHR + threadIdx.x;
            if (i < ROWS * KP) {
                const int r = i / KP, c = i % KP;
                if constexpr (X0_LDS) x0[r * lda + c] = v[u];
                xa[r * lda + c] = v[u];
                xb[r * lda + c] = 0.f;
            }
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int n_tiles = (d + 15) / 16;
    const int n_stage = KP / 32;
    for (int l = 0; l < layers; ++l) {
        const float* W = w + (int64_t)l * d * wstride;     // rows 16-B aligned: wstride % 4 == 0
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, d * wstride * 4, 0x00020000);
        const float* arow = xa + j * lda + 4 * g;
        for (int wt = wave; wt < n_tiles; wt += CROSS_WAVES) {
            const int n = wt * 16 + j;
            const int voff = (min(n, d - 1) * wstride + 4 * g) * 4;   // row n, first k of this lane's slot
            f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 bA[2], bB[2], bC[2], aA[RT][2], aB[RT][2], aC[RT][2];
            const int s_last = n_stage - 1;
            // x_0 of this wave-tile (RT > 1): requested now, used after the k loop
            float x0v[RT][4];
            if constexpr (!X0_LDS) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t bb = min(b0 + rt * 16 + 4 * g + r, batch - 1);
                        x0v[rt][r] = x[bb * x_stride + min(n, d - 1)];
                    }
            }
            cross_load_stage_rt<RT>(rsrc, voff, arow, lda, 0, bA, aA);
            cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(1, s_last), bB, aB);
            for (int s = 0; s < n_stage; s += 3) {
                cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(s + 2, s_last), bC, aC);
                __builtin_amdgcn_sched_barrier(0);
                cross_mfma_stage_rt<RT>(aA, bA, acc);
                __builtin_amdgcn_sched_barrier(0);
                cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(s + 3, s_last), bA, aA);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < n_stage) cross_mfma_stage_rt<RT>(aB, bB, acc);
                __builtin_amdgcn_sched_barrier(0);
                cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(s + 4, s_last), bB, aB);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 2 < n_stage) cross_mfma_stage_rt<RT>(aC, bC, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (n < d) {
                const float bv = bias[(int64_t)l * d + n];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + 4 * g + r;
                        float x0e;
                        if constexpr (X0_LDS) x0e = x0[row * lda + n];
                        else x0e = b0 + row < batch ? x0v[rt][r] : 0.f;
                        // interaction.py:419-420: x_l = x_0 * (W x_l + b) + x_l
                        const float xn = x0e * (acc[rt][r] + bv) + xa[row * lda + n];
                        xb[row * lda + n] = xn;
                        if (save_u != nullptr && b0 + row < batch) {     // training: what dctr_crossnet_bwd would recompute
                            const int64_t o = ((int64_t)l * batch + b0 + row) * d + n;
                            save_u[o] = acc[rt][r];
                            if (save_x != nullptr && l + 1 < layers) save_x[o] = xn;
                        }
                    }
            }
        }
        __syncthreads();
        float* t = xa;
        xa = xb;
        xb = t;
    }
    if (y != nullptr) {
        for (int i = threadIdx.x; i < ROWS * d; i += NTHR) {
            const int r = i / d, c = i % d;
            if (b0 + r < batch) y[(b0 + r) * y_stride + c] = xa[r * lda + c];
        }
    }
    if (head_w != nullptr) {            // this branch's share of the model's Dense(1): TPR consecutive lanes per row
        constexpr int TPR = NTHR / ROWS;
        const int r = threadIdx.x / TPR, part = threadIdx.x % TPR;
        float dot = 0.f;
        for (int n = part; n < d; n += TPR) dot = fmaf(xa[r * lda + n], head_w[n], dot);
#pragma unroll
        for (int m = TPR / 2; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
        if (part == 0 && b0 + r < batch) logit[b0 + r] = dot;
    }
}

// Four row tiles (64 rows) per workgroup, x_{l+1} IN PLACE (round 4b; launches of >= 64 rows per CU).  A weight fragment feeds four
// MFMAs: the stream from L2 per row halves again (RT = 2 asks ~32 B/clk of a CU at the full matrix rate where ~20 arrive).  Two 64-row
// tiles do not fit the 160 KiB, so a layer's results wait in REGISTERS — a wave owns column tiles wave, wave + 8, ... (<= MAXT of
// them: d <= 128 MAXT) — until every wave has finished reading x_l (one barrier), and are then written over it.  x_0 of a wave's column
// tiles stays in registers for the whole network.  Same arithmetic per element as the other forms: the same fmaf chain over k, then
// x_0 (u + b) + x_l.
// dctr_crossnet_gather_head_fwd: fields == NULL reads the rows from x; else the workgroup's [64, d] tile is gathered from the embedding
// tables (field f of row b = E floats at column f E: inputs.py:101-117 + layers/utils.py:336-346 inside the kernel) and the dense columns
struct CrossGather {
    const dctr_field_t* fields;
    const void* ids;
    int64_t ids_stride_f;
    int32_t ids_i64, n_fields, E;
    const float* dense;
    int64_t dense_stride;
    int32_t n_dense, dense_off;
    int32_t* status;
};

template <int MAXT>
__global__ __launch_bounds__(64 * CROSS_WAVES) void cross_matrix_inplace_kernel(CrossGather cg, const float* __restrict__ x, int64_t batch, int d,
                                                                                int64_t x_stride, const float* __restrict__ w,
                                                                                int wstride, const float* __restrict__ bias,
                                                                                int layers, float* __restrict__ y, int64_t y_stride,
                                                                                int lda, const float* __restrict__ head_w,
                                                                                float* __restrict__ logit, float* __restrict__ save_u,
                                                                                float* __restrict__ save_x) {
    using dctr::f32x4;
    constexpr int RT = 4;
    constexpr int NTHR = 64 * CROSS_WAVES;
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xa = smem;                                   // x_l [ROWS][lda], columns >= d zero up to the next multiple of 32
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const int KP = (d + 31) & ~31;
    // the workgroup's rows of the input as a buffer: [rows_here, x_stride] floats from its first row (< 2^31 B: host)
    const int rows_here = (int)(batch - b0 < ROWS ? batch - b0 : ROWS);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + b0 * x_stride), 0,
                                                                           (int)(((int64_t)(rows_here - 1) * x_stride + d) * 4), 0x00020000);
    if (cg.fields != nullptr) {
        // item (row r, field f, 16-B piece q): eight threads per row (r = thread / 8), thread t % 8 takes the row's items t % 8, + 8, ...
        // — f and q by shift and mask (E / 4 a power of two: host), no integer division per item (with i / (F Q) and % Q per item the
        // prologue was ~1,700 vector instructions per wave beside the network's 3,000 MFMAs); four ids, then their rows, in flight
        // together; ids outside the vocabulary read row 0 and raise the status flag; rows past the batch: zeros
        const int Q = cg.E >> 2, qs = __builtin_ctz(Q), fq = cg.n_fields * Q;
        const int r = threadIdx.x >> 3, t8 = threadIdx.x & 7;
        const int64_t brow = min(b0 + r, batch - 1);
        const bool live = b0 + r < batch;
        float* xrow = xa + r * lda;
        for (int base = t8; base < fq; base += 32) {
            int64_t id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = min(base + 8 * u, fq - 1);
                const int64_t eo = (int64_t)(it >> qs) * cg.ids_stride_f + brow;
                id[u] = cg.ids_i64 ? reinterpret_cast<const int64_t*>(cg.ids)[eo] : (int64_t)reinterpret_cast<const int32_t*>(cg.ids)[eo];
            }
            float4 v[4];
            bool bad = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = min(base + 8 * u, fq - 1);
                const int f = it >> qs, q = it & (Q - 1);
                const bool ok = (uint64_t)id[u] < (uint64_t)cg.fields[f].vocab;
                bad = bad || (!ok && base + 8 * u < fq && live);
                v[u] = *reinterpret_cast<const float4*>(cg.fields[f].table + (ok ? id[u] : 0) * cg.E + 4 * q);
                if (!live) v[u] = float4{0.f, 0.f, 0.f, 0.f};
            }
            if (bad && cg.status != nullptr) atomicOr(cg.status, (int)DCTR_STATUS_INDEX_OOR);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + 8 * u < fq) *reinterpret_cast<float4*>(xrow + 4 * (base + 8 * u)) = v[u];     // column f E + 4 q = 4 (f Q + q)
        }
        const int tail0 = cg.n_fields * cg.E;               // dense columns, then zeros up to KP
        for (int i = threadIdx.x; i < ROWS * (KP - tail0); i += NTHR) {
            const int r = i / (KP - tail0), c = tail0 + i % (KP - tail0);
            float v = 0.f;
            if (c >= cg.dense_off && c < cg.dense_off + cg.n_dense && b0 + r < batch) v = cg.dense[(b0 + r) * cg.dense_stride + (c - cg.dense_off)];
            xa[r * lda + c] = v;
        }
    } else {
        for (int base = 0; base < ROWS * KP; base += NTHR * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * NTHR + threadIdx.x;
                const int r = min(i / KP, ROWS - 1), c = i % KP;
                v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrsrc, (r * (int)x_stride + min(c, d - 1)) * 4, 0, 0));
                if (c >= d) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * NTHR + threadIdx.x;
                if (i < ROWS * KP) xa[(i / KP) * lda + i % KP] = v[u];
            }
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const int n_tiles = (d + 15) / 16;
    const int n_stage = KP / 32;
    // x_0 of this wave's column tiles: registers for the whole network (the tile in LDS is overwritten layer by layer)
    f32x4 x0r[MAXT][RT];
#pragma unroll
    for (int ti = 0; ti < MAXT; ++ti) {
        const int nc = min((wave + CROSS_WAVES * ti) * 16 + j, d - 1);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x0r[ti][rt][r] = xa[(rt * 16 + 4 * g + r) * lda + nc];
    }
    for (int l = 0; l < layers; ++l) {
        const float* W = w + (int64_t)l * d * wstride;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, d * wstride * 4, 0x00020000);
        const float* arow = xa + j * lda + 4 * g;
        f32x4 res[MAXT][RT];
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int wt = wave + CROSS_WAVES * ti;
            if (wt < n_tiles) {                              // (wave-uniform)
                const int n = wt * 16 + j;
                const int voff = (min(n, d - 1) * wstride + 4 * g) * 4;
                f32x4 acc[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // two register stages (a stage is 32 MFMAs per wave = 2k cycles for the two waves of a SIMD: one stage of cover is
                // an L2 round trip; three stages of four row tiles' operands would be 120 registers beside the 64 of `res`)
                float4 bA[2], bB[2], aA[RT][2], aB[RT][2];
                const int s_last = n_stage - 1;
                cross_load_stage_rt<RT>(rsrc, voff, arow, lda, 0, bA, aA);
                for (int s = 0; s < n_stage; s += 2) {
                    cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(s + 1, s_last), bB, aB);
                    __builtin_amdgcn_sched_barrier(0);
                    cross_mfma_stage_rt<RT>(aA, bA, acc);
                    __builtin_amdgcn_sched_barrier(0);
                    cross_load_stage_rt<RT>(rsrc, voff, arow, lda, min(s + 2, s_last), bA, aA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < n_stage) cross_mfma_stage_rt<RT>(aB, bB, acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const float bv = bias[(int64_t)l * d + min(n, d - 1)];
                const int nc = min(n, d - 1);
                // (lane parts of the epilogue's addresses from an OPAQUE copy of g, rebuilt per tile and layer: as loop invariants of
                //  the layer loop hipcc keeps 32 address registers per column tile alive — 128 of the 256 at MAXT = 4)
                int go = g;
                asm volatile("" : "+v"(go));
                const float* xal = xa + 4 * go * lda + nc;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)       // interaction.py:419-420: x_l = x_0 * (W x_l + b) + x_l
                        res[ti][rt][r] = x0r[ti][rt][r] * (acc[rt][r] + bv) + xal[(rt * 16 + r) * lda];
                __builtin_amdgcn_sched_barrier(0);           // (a tile at a time: the next tile's operand stages must not overlap this epilogue)
            }
        }
        __syncthreads();                                     // every wave has read x_l for the last time
#pragma unroll
        for (int ti = 0; ti < MAXT; ++ti) {
            const int wt = wave + CROSS_WAVES * ti;
            const int n = wt * 16 + j;
            if (wt < n_tiles && n < d) {
                int go = g;
                asm volatile("" : "+v"(go));
                float* xal = xa + 4 * go * lda + n;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xal[(rt * 16 + r) * lda] = res[ti][rt][r];
            }
        }
        __syncthreads();
    }
    if (y != nullptr) {
        for (int i = threadIdx.x; i < ROWS * d; i += NTHR) {
            const int r = i / d, c = i % d;
            if (b0 + r < batch) y[(b0 + r) * y_stride + c] = xa[r * lda + c];
        }
    }
    if (head_w != nullptr) {            // this branch's share of the model's Dense(1): TPR consecutive lanes per row
        constexpr int TPR = NTHR / ROWS;
        const int r = threadIdx.x / TPR, part = threadIdx.x % TPR;
        float dot = 0.f;
        for (int n = part; n < d; n += TPR) dot = fmaf(xa[r * lda + n], head_w[n], dot);
#pragma unroll
        for (int m = TPR / 2; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
        if (part == 0 && b0 + r < batch) logit[b0 + r] = dot;
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

__global__ __launch_bounds__(256) void afm_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F, int E,
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
        for (int i = lane; i < F * E; i += 64) xs[i] = x[b * x_stride + i];
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

// The layer for shapes whose sample tile + pair logits do not fit the LDS (fields x embedding_dim x 16 B + 16 B per pair > 160 KiB:
// the reference has no such limit, interaction.py:116-146): nothing is staged.  One wave per sample; a lane walks pairs p = lane,
// lane + 64, ... ONCE — a pair's attention logit and its projected product p . (e_i * e_j) come out of the same walk over e — and keeps
// a running (max, sum of exp, weighted sum): softmax over the pairs without storing a logit (the streaming form of :138-145: the weights
// exp(l_p - max) / sum are the same numbers, summed in another order).  x, attention_W and the projections are read through the L1.
__global__ __launch_bounds__(256) void afm_stream_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F, int E,
                                                         const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                         const float* __restrict__ proj_h, const float* __restrict__ proj_p,
                                                         int A, float* __restrict__ y) {
    const int P = F * (F - 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    const float* xs = x + b * x_stride;
    float mx = -INFINITY, den = 0.f, num = 0.f;
    for (int p = lane; p < P; p += 64) {
        int i, j;
        pair_ij(p, F, i, j);
        const float* xi = xs + (int64_t)i * E;
        const float* xj = xs + (int64_t)j * E;
        float lg = 0.f;
        for (int a = 0; a < A; ++a) {
            float t = att_b[a];
            for (int e = 0; e < E; ++e) t = fmaf(xi[e] * xj[e], att_w[(int64_t)e * A + a], t);
            lg = fmaf(fmaxf(t, 0.f), proj_h[a], lg);
        }
        float t = 0.f;
        for (int e = 0; e < E; ++e) t = fmaf(xi[e] * xj[e], proj_p[e], t);
        if (lg > mx) {                          // rescale what was summed under the old maximum
            const float r = expf(mx - lg);      // (exp(-inf) = 0 the first time)
            den *= r;
            num *= r;
            mx = lg;
        }
        const float w = expf(lg - mx);
        den += w;
        num = fmaf(w, t, num);
    }
    // lanes -> one (max, den, num)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float omx = __shfl_xor(mx, m, 64), oden = __shfl_xor(den, m, 64), onum = __shfl_xor(num, m, 64);
        const float nmx = fmaxf(mx, omx);
        const float ra = mx == -INFINITY ? 0.f : expf(mx - nmx), rb = omx == -INFINITY ? 0.f : expf(omx - nmx);
        den = den * ra + oden * rb;
        num = num * ra + onum * rb;
        mx = nmx;
    }
    if (lane == 0) y[b] = num / den;
}

// The same layer on the matrix pipe (embedding_dim % 16 == 0, attention_factor <= 15).  Per sample the attention net is a
// [P pairs, E] x [E, A] product: tiles of 16 pairs are the MFMA M dimension, the pair products e_i * e_j are formed in registers
// as the A operand (two ds_read_b128 + four multiplies per four k-steps: k-slot g of k-step t is dimension 4g + t, a K permutation
// the B operand follows), the B operand [E, 16] = [attention_W | projection_p | zeros] stays in registers for the whole sample:
// column n < A of a C tile is a pair's pre-activation, column A its projected product p . (e_i * e_j).  ReLU, the dot with
// projection_h over the 16 lanes of a pair's row (DPP adds), then softmax over the pairs and the weighted sum of column A —
// (P / 16) x (E / 4) MFMAs per sample where the VALU form spent ~150 fused multiply-adds and 2 E LDS reads PER PAIR
// (149 -> 14 us per 4096 samples of 26 fields x 16, profiles/r03_*).  One wave per sample, four samples per workgroup.
template <int EB>
__global__ __launch_bounds__(256) void afm_mfma_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                       const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                       const float* __restrict__ proj_h, const float* __restrict__ proj_p,
                                                       int A, float* __restrict__ y) {
    constexpr int E = 16 * EB;
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int PT = (P + 15) / 16;                // pair tiles
    int* ptab = reinterpret_cast<int*>(smem);    // [16 PT] (i << 8) | j per pair (the last pair repeated past P)
    float* per_wave = smem + 16 * PT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    float* xs = per_wave + wave * (F * E + 2 * 16 * PT);   // [F * E] sample tile
    float* logit = xs + F * E;                              // [16 PT]
    float* zs = logit + 16 * PT;                            // [16 PT] projected pair products
    for (int pidx = threadIdx.x; pidx < 16 * PT; pidx += 256) {
        int pi, pj;
        pair_ij(min(pidx, P - 1), F, pi, pj);
        ptab[pidx] = (pi << 8) | pj;
    }
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool valid = b < batch;
    if (valid) {
        const float4* src = reinterpret_cast<const float4*>(x + b * x_stride);
        for (int i = lane; i < F * E / 4; i += 64) reinterpret_cast<float4*>(xs)[i] = src[i];
    }
    // weight operand: lane (g, j) holds column n = j of rows e = 16 c + 4g + t (k-step (c, t), k-slot g)
    float bw[EB][4];
#pragma unroll
    for (int c = 0; c < EB; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = 16 * c + 4 * g + t;
            bw[c][t] = j < A ? att_w[e * A + j] : (j == A ? proj_p[e] : 0.f);
        }
    // the product is taken TRANSPOSED — C'[column][pair] = Wext^T (A operand) x products^T (B operand): lane (g, j), register r then
    // holds column 4g + r of pair j, and a pair's dot with projection_h is four local terms + two cross-lane adds (over g)
    float b4[4], h4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = 4 * g + r;
        b4[r] = n < A ? att_b[n] : 0.f;
        h4[r] = n < A ? proj_h[n] : 0.f;
    }
    __syncthreads();
    if (!valid) return;
    // pass 1: per pair tile, pre-activations of the attention net and the projected product
    for (int q = 0; q < PT; ++q) {
        const int pr = ptab[16 * q + j];
        const float* ei = xs + (pr >> 8) * E + 4 * g;
        const float* ej = xs + (pr & 255) * E + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < EB; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(ei + 16 * c), d = *reinterpret_cast<const float4*>(ej + 16 * c);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[c][0], a.x * d.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[c][1], a.y * d.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[c][2], a.z * d.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[c][3], a.w * d.w, acc, 0, 0, 0);
        }
        // lane (g, j), register r: column 4g + r of pair 16 q + j
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v = fmaf(fmaxf(acc[r] + b4[r], 0.f), h4[r], v);        // columns >= A: h = 0
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (g == 0) logit[16 * q + j] = v;
        if (g == (A >> 2)) zs[16 * q + j] = (A & 3) == 0 ? acc[0] : (A & 3) == 1 ? acc[1] : (A & 3) == 2 ? acc[2] : acc[3];
    }
    // pass 2: softmax over the pairs, weighted sum of the projected products (interaction.py:138-145)
    float mx = -INFINITY;
    for (int p = lane; p < P; p += 64) mx = fmaxf(mx, logit[p]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float den = 0.f, num = 0.f;
    for (int p = lane; p < P; p += 64) {
        const float e_ = expf(logit[p] - mx);
        den += e_;
        num = fmaf(e_, zs[p], num);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        den += __shfl_xor(den, m, 64);
        num += __shfl_xor(num, m, 64);
    }
    if (lane == 0) y[b] = num / den;
}

// ---------------------------------------------------------------------------------------------------
// InnerProduct: one wave per sample.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inner_product_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                            int E, int reduce_sum, float* __restrict__ y, int64_t y_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = smem + wave * F * E;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool valid = b < batch;
    if (valid)
        for (int i = lane; i < F * E; i += 64) xs[i] = x[b * x_stride + i];
    __syncthreads();
    if (!valid) return;
    if (reduce_sum) {
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_ij(p, F, i, j);
            float t = 0.f;
            for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e], xs[j * E + e], t);
            y[b * y_stride + p] = t;
        }
    } else {
        const int64_t total = (int64_t)P * E;
        for (int64_t o = lane; o < total; o += 64) {
            const int p = (int)(o / E), e = (int)(o % E);
            int i, j;
            pair_ij(p, F, i, j);
            y[b * y_stride + o] = xs[i * E + e] * xs[j * E + e];
        }
    }
}

// ... and for sample tiles past the LDS (4 x fields x embedding_dim floats > 160 KiB): the rows straight from global memory (L1 / L2)
__global__ __launch_bounds__(256) void inner_product_stream_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                                   int E, int reduce_sum, float* __restrict__ y, int64_t y_stride) {
    const int P = F * (F - 1) / 2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    const float* xs = x + b * x_stride;
    if (reduce_sum) {
        for (int p = lane; p < P; p += 64) {
            int i, j;
            pair_ij(p, F, i, j);
            float t = 0.f;
            for (int e = 0; e < E; ++e) t = fmaf(xs[(int64_t)i * E + e], xs[(int64_t)j * E + e], t);
            y[b * y_stride + p] = t;
        }
    } else {
        const int64_t total = (int64_t)P * E;
        for (int64_t o = lane; o < total; o += 64) {
            const int p = (int)(o / E), e = (int)(o % E);
            int i, j;
            pair_ij(p, F, i, j);
            y[b * y_stride + o] = xs[(int64_t)i * E + e] * xs[(int64_t)j * E + e];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// BiInteractionPooling (interaction.py:190-203, NFM): y[b, e] = 0.5 * ((sum_f x[b,f,e])^2 - sum_f x[b,f,e]^2)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bi_interaction_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                             int E, float* __restrict__ y, int64_t y_stride) {
    const int64_t total = batch * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / E;
        const int e = (int)(o - b * E);
        const float* xb = x + b * x_stride + e;
        float s = 0.f, q = 0.f;
        for (int f = 0; f < F; ++f) {
            const float v = xb[(int64_t)f * E];
            s += v;
            q = fmaf(v, v, q);
        }
        y[b * y_stride + e] = 0.5f * (s * s - q);
    }
}

// ---------------------------------------------------------------------------------------------------
// CrossNetMix (interaction.py:511-549, DCN-Mix): per layer, per expert e
//     g_e = x_l . G_e ;  v = tanh(V_e^T x_l) ;  v = tanh(C_e v) ;  out_e = x_0 * (U_e v + bias)
//     x_{l+1} = sum_e softmax(g)_e out_e + x_l  =  x_0 * (sum_e U_e (p_e v_e) + bias) + x_l        (sum_e p_e = 1)
// One workgroup owns R rows for every layer: x_0 / x_l stay in LDS ([d][R], so the R rows of a column are one broadcast
// read).  Stage 1: one thread per low-rank column (a d-long dot per row; V [experts,d,r] is read with lanes along r,
// coalesced; with <= 128 columns several threads share one and split the j range), then the gating dots.  Stage 2: the
// r x r mixing and the softmax weights.  Stage 3: two output columns per thread over U^T [experts*r, d] — U transposed once
// per call into the workspace so that lanes along d read consecutive floats.  fp32 FMA: for fp32 the VALU and the MFMA peak of gfx950 are the same 157 TFLOP/s, so the
// bound is keeping WGB weight loads in flight per thread (L2 latency), not the matrix cores.
// ---------------------------------------------------------------------------------------------------
constexpr int WGB = 8;   // weight loads issued back to back before their FMAs

__global__ __launch_bounds__(256) void cross_mix_transpose_kernel(const float* __restrict__ U, int64_t mats, int d, int r,
                                                                  float* __restrict__ Ut) {
    // U [mats, d, r] -> Ut [mats, r, d]
    const int64_t total = mats * d * r;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t m = o / ((int64_t)d * r);
        const int64_t rem = o - m * d * r;
        const int k = (int)(rem / d), jj = (int)(rem - (int64_t)k * d);
        Ut[o] = U[(m * d + jj) * r + k];
    }
}

// acc[n][q] += sum_j a[j*astride + q] * w[n][j*wstride], j in [j0, j1): `a` in LDS (broadcast reads), NC weight columns in
// global memory; the next WGB weights per column are in flight while the current ones are used
template <int RA, int NC>
__device__ __forceinline__ void dot_rows(const float* __restrict__ a, int astride, const float* const (&w)[NC], int64_t wstride,
                                         int j0, int j1, float (&acc)[NC][RA]) {
    float cur[NC][WGB], nxt[NC][WGB];
    const int nfull = (j1 - j0) / WGB;
    int jj = j0;
    if (nfull > 0) {
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int u = 0; u < WGB; ++u) cur[n][u] = w[n][(int64_t)(jj + u) * wstride];
    }
    for (int b = 0; b < nfull; ++b) {
        if (b + 1 < nfull) {
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int u = 0; u < WGB; ++u) nxt[n][u] = w[n][(int64_t)(jj + WGB + u) * wstride];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < WGB; ++u) {
            float av[RA];
#pragma unroll
            for (int q = 0; q < RA; ++q) av[q] = a[(jj + u) * astride + q];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int q = 0; q < RA; ++q) acc[n][q] = fmaf(av[q], cur[n][u], acc[n][q]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NC; ++n)
#pragma unroll
            for (int u = 0; u < WGB; ++u) cur[n][u] = nxt[n][u];
        jj += WGB;
    }
    for (; jj < j1; ++jj) {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const float wv = w[n][(int64_t)jj * wstride];
#pragma unroll
            for (int q = 0; q < RA; ++q) acc[n][q] = fmaf(a[jj * astride + q], wv, acc[n][q]);
        }
    }
}

template <int R>
__global__ __launch_bounds__(256) void cross_mix_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int d,
                                                        const float* __restrict__ Ut, const float* __restrict__ V,
                                                        const float* __restrict__ C, const float* __restrict__ G,
                                                        const float* __restrict__ bias, int layers, int ne, int r,
                                                        float* __restrict__ y, int64_t y_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = ne * r;
    float* x0 = smem;                     // [d][R]
    float* xl = x0 + (size_t)d * R;       // [d][R]
    float* h1 = xl + (size_t)d * R;       // [H][R]   tanh(V^T x_l)
    float* h2 = h1 + (size_t)H * R;       // [H][R]   p_e * tanh(C v)
    float* gate = h2 + (size_t)H * R;     // [8][ne][R] partial gating dots
    float* pw = gate + (size_t)8 * ne * R;    // [ne][R]  softmax weight of every expert
    float* h1p = pw + (size_t)ne * R;         // [P1][H][R] partial projections (P1 * H <= 256)
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * R;
    // few low-rank columns: P1 threads share a column (each a slice of the j range) so that every wave has work
    const int P1 = H <= 128 ? (256 / H < 8 ? 256 / H : 8) : 1;
    // gating dots: (expert, row) pairs x GP slices of the j range
    const int combos = ne * R;
    const int GP = 256 >= combos ? (256 / combos < 8 ? 256 / combos : 8) : 1;
    for (int j = tid; j < d; j += 256) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t b = b0 + q < batch ? b0 + q : batch - 1;
            const float v = x[b * x_stride + j];
            x0[j * R + q] = v;
            xl[j * R + q] = v;
        }
    }
    __syncthreads();
    for (int l = 0; l < layers; ++l) {
        const float* Vl = V + (size_t)l * ne * d * r;
        const float* Utl = Ut + (size_t)l * H * d;
        const float* Cl = C + (size_t)l * ne * r * r;
        // stage 1: low-rank projections ...
        if (P1 > 1) {
            if (tid < P1 * H) {
                const int part = tid / H, c = tid - part * H;
                const int e = c / r, k = c - e * r;
                const int per = (d + P1 - 1) / P1;
                const int j0 = part * per < d ? part * per : d, j1 = j0 + per < d ? j0 + per : d;
                float acc[1][R];
#pragma unroll
                for (int q = 0; q < R; ++q) acc[0][q] = 0.f;
                const float* const w[1] = {Vl + (size_t)e * d * r + k};
                dot_rows<R, 1>(xl, R, w, r, j0, j1, acc);
#pragma unroll
                for (int q = 0; q < R; ++q) h1p[(part * H + c) * R + q] = acc[0][q];
            }
        } else {
            for (int c = tid; c < H; c += 256) {
                const int e = c / r, k = c - e * r;
                float acc[1][R];
#pragma unroll
                for (int q = 0; q < R; ++q) acc[0][q] = 0.f;
                const float* const w[1] = {Vl + (size_t)e * d * r + k};
                dot_rows<R, 1>(xl, R, w, r, 0, d, acc);
#pragma unroll
                for (int q = 0; q < R; ++q) h1[c * R + q] = tanhf(acc[0][q]);
            }
        }
        // ... and gating scores: item = (slice, expert, row)
        for (int it = tid; it < GP * combos; it += 256) {
            const int part = it / combos, eq = it - part * combos;
            const int e = eq / R, q = eq - e * R;
            const int per = (d + GP - 1) / GP;
            const int j0 = part * per < d ? part * per : d, j1 = j0 + per < d ? j0 + per : d;
            float acc[1][1] = {{0.f}};
            const float* const w[1] = {G + (size_t)e * d};
            dot_rows<1, 1>(xl + q, R, w, 1, j0, j1, acc);
            gate[(part * ne + e) * R + q] = acc[0][0];
        }
        __syncthreads();
        if (P1 > 1) {
            for (int o = tid; o < H * R; o += 256) {
                float sum = 0.f;
                for (int pp = 0; pp < P1; ++pp) sum += h1p[pp * H * R + o];
                h1[o] = tanhf(sum);
            }
        }
        // softmax over the experts' gating scores, once per (expert, row)
        for (int it = tid; it < combos; it += 256) {
            const int e = it / R, q = it - e * R;
            float mine = 0.f, mx = -INFINITY, den = 0.f;
            for (int t = 0; t < ne; ++t) {
                float gsum = 0.f;
                for (int pp = 0; pp < GP; ++pp) gsum += gate[(pp * ne + t) * R + q];
                mx = fmaxf(mx, gsum);
            }
            for (int t = 0; t < ne; ++t) {
                float gsum = 0.f;
                for (int pp = 0; pp < GP; ++pp) gsum += gate[(pp * ne + t) * R + q];
                const float ex = expf(gsum - mx);
                den += ex;
                if (t == e) mine = ex;
            }
            pw[e * R + q] = mine / den;
        }
        __syncthreads();
        // stage 2: r x r mixing in the low-rank space, scaled by the expert's softmax weight
        for (int c = tid; c < H; c += 256) {
            const int e = c / r, i = c - e * r;
            float acc[1][R];
#pragma unroll
            for (int q = 0; q < R; ++q) acc[0][q] = 0.f;
            const float* const w[1] = {Cl + ((size_t)e * r + i) * r};
            dot_rows<R, 1>(h1 + (size_t)e * r * R, R, w, 1, 0, r, acc);
#pragma unroll
            for (int q = 0; q < R; ++q) h2[c * R + q] = tanhf(acc[0][q]) * pw[e * R + q];
        }
        __syncthreads();
        // stage 3: back to d columns (two per thread, sharing the broadcast reads), Hadamard with x_0, residual
        for (int j = tid; j < d; j += 512) {
            const int j2 = j + 256 < d ? j + 256 : j;
            float acc[2][R];
#pragma unroll
            for (int q = 0; q < R; ++q) acc[0][q] = acc[1][q] = 0.f;
            const float* const w[2] = {Utl + j, Utl + j2};
            dot_rows<R, 2>(h2, R, w, d, 0, H, acc);
            const float bj = bias[(size_t)l * d + j];
#pragma unroll
            for (int q = 0; q < R; ++q) xl[j * R + q] = fmaf(x0[j * R + q], acc[0][q] + bj, xl[j * R + q]);
            if (j2 != j) {
                const float b2 = bias[(size_t)l * d + j2];
#pragma unroll
                for (int q = 0; q < R; ++q) xl[j2 * R + q] = fmaf(x0[j2 * R + q], acc[1][q] + b2, xl[j2 * R + q]);
            }
        }
        __syncthreads();
    }
    for (int j = tid; j < d; j += 256) {
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (b0 + q < batch) y[(b0 + q) * y_stride + j] = xl[j * R + q];
    }
}

size_t cross_mix_lds_bytes(int d, int ne, int r, int R) {
    return ((size_t)2 * d + 2 * (size_t)ne * r + 9 * (size_t)ne + 256) * R * sizeof(float);
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

extern "C" size_t dctr_crossnet_workspace_bytes(int32_t dim, int32_t layers, int32_t mode, const float* kernels) {
    if (mode != DCTR_CROSS_MATRIX || layers <= 0 || dim < 1) return 0;
    if (dim % 4 == 0 && dctr_aligned16(kernels)) return 0;
    return (size_t)layers * dim * ((dim + 3) & ~3) * sizeof(float);
}

// dry: every shape check and kernel-shape decision of the call, no launch, device pointers not looked at (dctr_crossnet_fwd_supported)
static int crossnet_launch(const dctr_crossnet_args_t* a, void* stream, const dctr_gather_fm_args_t* gg = nullptr, bool dry = false) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "crossnet_fwd: null args");
    const float* x = a->x;
    const int64_t batch = a->batch, x_stride = a->x_stride, y_stride = a->y_stride;
    const int32_t dim = a->dim, layers = a->layers, mode = a->mode;
    const float* kernels = a->kernels;
    const float* bias = a->bias;
    float* y = a->y;
    void* workspace = a->workspace;
    const size_t workspace_bytes = a->workspace_bytes;
    DCTR_REQUIRE(batch >= 0 && dim >= 1 && layers >= 0, DCTR_E_DIM, "crossnet_fwd: bad sizes");
    DCTR_REQUIRE(mode == DCTR_CROSS_VECTOR || mode == DCTR_CROSS_MATRIX, DCTR_E_ENUM, "crossnet_fwd: mode %d", mode);
    if (batch == 0 && !dry) return DCTR_OK;
    if (!dry) {
        DCTR_REQUIRE((x || gg) && (y || a->head_w) && (layers == 0 || (kernels && bias)), DCTR_E_NULL, "crossnet_fwd: null pointer");
        DCTR_REQUIRE((a->head_w == nullptr) == (a->logit == nullptr), DCTR_E_NULL, "crossnet_fwd: head_w and logit go together");
        DCTR_REQUIRE((gg != nullptr || x_stride >= dim) && (y == nullptr || y_stride >= dim), DCTR_E_DIM, "crossnet_fwd: stride < dim");
    }
    DCTR_REQUIRE(a->save_u == nullptr || (mode == DCTR_CROSS_MATRIX && layers >= 1 && (layers == 1 || a->save_x != nullptr)), DCTR_E_UNSUPPORTED,
                 "crossnet_fwd: save_u / save_x exist for the matrix form (save_x with more than one layer)");
    hipStream_t st = (hipStream_t)stream;
    if (mode == DCTR_CROSS_VECTOR || layers == 0) {
        DCTR_REQUIRE(dim <= 64 * 128 || layers <= CROSS_STREAM_MAXL, DCTR_E_UNSUPPORTED,
                     "crossnet_fwd(vector): dim %d > 8192 (rows past the register file: the closed form) with more than %d layers", dim, CROSS_STREAM_MAXL);
        const int64_t blocks = dctr_ceil_div(batch, 4);
        DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "crossnet_fwd: batch too large");
        if (dry) return DCTR_OK;
        if (dim > 64 * 128) {
            DCTR_LAUNCH(cross_vector_stream_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, batch, dim, x_stride, kernels, bias, layers, y, y_stride,
                        a->head_w, a->logit);
            return dctr_launch_status("dctr_crossnet_fwd");
        }
        const int nr = pow2_at_least((dim + 63) / 64, 128);      // values of x_0 / x_l per lane (one wave per sample; 128: 256 VGPRs)
#define CALL_CV(N)                                                                                                  \
    DCTR_LAUNCH((cross_vector_kernel<N>), dim3((unsigned)blocks), dim3(256), 0, st, x, batch, dim, x_stride, \
                       kernels, bias, layers, y, y_stride, a->head_w, a->logit)
        switch (nr) {
            case 1: CALL_CV(1); break;
            case 2: CALL_CV(2); break;
            case 4: CALL_CV(4); break;
            case 8: CALL_CV(8); break;
            case 16: CALL_CV(16); break;
            case 32: CALL_CV(32); break;
            case 64: CALL_CV(64); break;
            default: CALL_CV(128); break;
        }
#undef CALL_CV
    } else {
        const int lda = ((dim + 31) & ~31) + 4;
        // 32 rows per workgroup (two row tiles per weight fragment, x_0 from L2) once that still gives every CU a workgroup
        const size_t lds2 = (size_t)2 * 32 * lda * sizeof(float);
        static const bool rt1_forced = [] { const char* e = dctr_lab_env("DCTR_CROSS_RT"); return e != nullptr && atoi(e) == 1; }();   // A/B switch
        const bool rt2 = !rt1_forced && batch >= (int64_t)32 * dctr_n_cus() && lds2 <= 160 * 1024;
        // 64 rows per workgroup, the layer's output written in place (cross_matrix_inplace_kernel): from 64 rows per CU on, dim <= 512
        const size_t lds4 = (size_t)64 * lda * sizeof(float);
        static const bool rt4_off = [] { const char* e = dctr_lab_env("DCTR_CROSS_RT"); return e != nullptr && atoi(e) == 2; }();     // A/B switch
        const bool rt4 = rt2 && !rt4_off && batch >= (int64_t)64 * dctr_n_cus() && dim <= 128 * 4 && lds4 <= 160 * 1024 && a->save_u == nullptr &&
                         (gg != nullptr || (int64_t)64 * x_stride * 4 < 0x7fffffffLL);   // (inference; a workgroup's rows addressable in 32 bits)
        CrossGather cg{};
        if (gg != nullptr) {
            // the gather of the same batch inside the kernel: plain lookups of one width, the reference's DNN-input layout
            DCTR_REQUIRE(rt4, DCTR_E_UNSUPPORTED, "crossnet_gather_head_fwd: the 64-row kernel takes launches of >= %lld rows, dim <= 512 "
                         "(got %lld rows, dim %d): use dctr_embed_gather_fm + dctr_crossnet_head_fwd", (long long)(64 * dctr_n_cus()), (long long)batch, dim);
            cg.fields = gg->fields;
            cg.ids = gg->ids;
            cg.ids_stride_f = gg->ids_stride_f;
            cg.ids_i64 = gg->ids_is_i64;
            cg.n_fields = gg->n_fields;
            cg.E = gg->uniform_dim;
            cg.dense = gg->dense;
            cg.dense_stride = gg->dense_stride;
            cg.n_dense = gg->dense_copy_cols;
            cg.dense_off = gg->dense_out_offset;
            cg.status = gg->status;
        }
        const size_t lds = rt4 ? lds4 : rt2 ? lds2 : (size_t)3 * 16 * lda * sizeof(float);
        DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "crossnet_fwd(matrix): dim %d needs %zu B of LDS", dim, lds);
        DCTR_REQUIRE((int64_t)dim * (dim + 3) * 4 < 0x7fffffffLL, DCTR_E_UNSUPPORTED, "crossnet_fwd(matrix): dim %d too large", dim);
        if (dry) return DCTR_OK;
        const int maxt = dim <= 128 ? 1 : dim <= 256 ? 2 : 4;          // column tiles a wave of the in-place kernel holds in registers
        if (lds > 64 * 1024) {
            const void* fn = rt4 ? (maxt == 1 ? (const void*)cross_matrix_inplace_kernel<1> : maxt == 2 ? (const void*)cross_matrix_inplace_kernel<2>
                                                                                                        : (const void*)cross_matrix_inplace_kernel<4>)
                                 : rt2 ? (const void*)cross_matrix_kernel<2> : (const void*)cross_matrix_kernel<1>;
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DCTR_REQUIRE(e == hipSuccess, (int)e, "crossnet_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        }
        const int64_t blocks = dctr_ceil_div(batch, rt4 ? 64 : rt2 ? 32 : 16);
        DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "crossnet_fwd: batch too large");
        const size_t need = dctr_crossnet_workspace_bytes(dim, layers, mode, kernels);
        const float* wk = kernels;
        int wstride = dim;
        if (need > 0) {
            DCTR_REQUIRE(workspace != nullptr && workspace_bytes >= need && dctr_aligned16(workspace), DCTR_E_NULL,
                         "crossnet_fwd(matrix): needs a 16-B aligned workspace of %zu B (dctr_crossnet_workspace_bytes)", need);
            wstride = (dim + 3) & ~3;
            if (!a->workspace_ready) {                          // (the caller keeps the re-packed rows while the kernels do not change)
                const int64_t rows = (int64_t)layers * dim;
                const int64_t rb = dctr_ceil_div(rows * wstride, (int64_t)256);
                hipLaunchKernelGGL(cross_repack_kernel, dim3((unsigned)(rb > 2048 ? 2048 : rb)), dim3(256), 0, st, kernels, dim,
                                   wstride, rows, static_cast<float*>(workspace));
            }
            wk = static_cast<const float*>(workspace);
        }
#define CALL_IP(N)                                                                                                         \
    DCTR_LAUNCH(cross_matrix_inplace_kernel<N>, dim3((unsigned)blocks), dim3(64 * CROSS_WAVES), lds, st, cg, x, batch, dim, x_stride, wk, wstride, \
                bias, layers, y, y_stride, lda, a->head_w, a->logit, a->save_u, a->save_x)
        if (rt4) {
            if (maxt == 1) CALL_IP(1);
            else if (maxt == 2) CALL_IP(2);
            else CALL_IP(4);
        } else if (rt2)
            DCTR_LAUNCH(cross_matrix_kernel<2>, dim3((unsigned)blocks), dim3(64 * CROSS_WAVES), lds, st, x, batch, dim, x_stride, wk, wstride,
                        bias, layers, y, y_stride, lda, a->head_w, a->logit, a->save_u, a->save_x);
        else
            DCTR_LAUNCH(cross_matrix_kernel<1>, dim3((unsigned)blocks), dim3(64 * CROSS_WAVES), lds, st, x, batch, dim, x_stride, wk, wstride,
                        bias, layers, y, y_stride, lda, a->head_w, a->logit, a->save_u, a->save_x);
#undef CALL_IP
    }
    return dctr_launch_status("dctr_crossnet_fwd");
}

extern "C" int dctr_crossnet_fwd(const float* x, int64_t batch, int32_t dim, int64_t x_stride, const float* kernels,
                                 const float* bias, int32_t layers, int32_t mode, float* y, int64_t y_stride,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    dctr_crossnet_args_t a = {};
    a.x = x; a.batch = batch; a.dim = dim; a.x_stride = x_stride; a.kernels = kernels; a.bias = bias; a.layers = layers; a.mode = mode;
    a.y = y; a.y_stride = y_stride; a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    return crossnet_launch(&a, stream);
}

extern "C" int dctr_crossnet_head_fwd(const dctr_crossnet_args_t* a, void* stream) { return crossnet_launch(a, stream); }

// ABI 8 — CrossNet (matrix parameterization) + the branch's share of DCN's Dense(1) over the embeddings of a gather (reference
// models/dcn.py:48-66, layers/interaction.py:405-424): the workgroup's [64, dim] tile of the DNN input is read from the tables and the
// dense matrix inside the kernel (a->x is ignored), logit[b] = x_L[b] . head_w leaves (y optional).  Spans only: >= 64 rows per CU.
static int crossnet_gather_checks(const dctr_crossnet_args_t* a, const dctr_gather_fm_args_t* g, bool dry) {
    DCTR_REQUIRE(a != nullptr && g != nullptr, DCTR_E_NULL, "crossnet_gather_head_fwd: null args");
    DCTR_REQUIRE(a->mode == DCTR_CROSS_MATRIX && a->layers >= 1, DCTR_E_UNSUPPORTED, "crossnet_gather_head_fwd: matrix parameterization, >= 1 layer");
    DCTR_REQUIRE((dry || (g->fields != nullptr && g->ids != nullptr)) && g->batch == a->batch, DCTR_E_NULL, "crossnet_gather_head_fwd: gather of another batch / null");
    DCTR_REQUIRE(g->uniform_dim >= 4 && (g->uniform_dim & (g->uniform_dim - 1)) == 0 && g->all_dim4 && !g->any_hash && !g->any_identity &&
                     !g->any_pitch && g->ids_stride_b == 1,
                 DCTR_E_UNSUPPORTED, "crossnet_gather_head_fwd: plain (unhashed, not pre-pooled) lookups of one width (a power of two >= 4), contiguous id rows");
    const int n_dense = g->dense_copy_cols > 0 ? g->dense_copy_cols : 0;
    DCTR_REQUIRE(a->dim == g->n_fields * g->uniform_dim + n_dense && (n_dense == 0 || ((dry || g->dense != nullptr) && g->dense_out_offset == g->n_fields * g->uniform_dim)),
                 DCTR_E_DIM, "crossnet_gather_head_fwd: dim %d is not the gather's DNN-input width (%d fields x %d + %d dense)", a->dim, g->n_fields,
                 g->uniform_dim, n_dense);
    DCTR_REQUIRE(a->save_u == nullptr, DCTR_E_UNSUPPORTED, "crossnet_gather_head_fwd: inference only");
    return DCTR_OK;
}

extern "C" int dctr_crossnet_gather_head_fwd(const dctr_crossnet_args_t* a, const dctr_gather_fm_args_t* g, void* stream) {
    const int rc = crossnet_gather_checks(a, g, false);
    if (rc != DCTR_OK) return rc;
    return crossnet_launch(a, stream, g);
}

// ABI 13 — would dctr_crossnet_head_fwd / dctr_crossnet_fwd (gather == NULL) or dctr_crossnet_gather_head_fwd (gather != NULL) take these
// arguments?  1 = yes, 0 = no (dctr_last_error() says why); nothing is launched, device pointers are not looked at — save_u counts as
// "training forward" when non-NULL, args->kernels only for its 16-B alignment (the re-pack workspace)
extern "C" int dctr_crossnet_fwd_supported(const dctr_crossnet_args_t* a, const dctr_gather_fm_args_t* g) {
    if (a == nullptr) return 0;
    if (g != nullptr && crossnet_gather_checks(a, g, true) != DCTR_OK) return 0;
    return crossnet_launch(a, nullptr, g, true) == DCTR_OK ? 1 : 0;
}

namespace {
__global__ __launch_bounds__(256) void cross_matrix_step_kernel(const float* __restrict__ x0, int64_t x0_stride, const float* xl, int64_t xl_stride,
                                                                const float* __restrict__ u, const float* __restrict__ bias, int64_t batch, int d,
                                                                float* xn, int64_t xn_stride) {
    const int64_t total = batch * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / d;
        const int c = (int)(i - b * d);
        xn[b * xn_stride + c] = x0[b * x0_stride + c] * (u[i] + bias[c]) + xl[b * xl_stride + c];
    }
}
}  // namespace

extern "C" int dctr_crossnet_matrix_step(const float* x0, int64_t x0_stride, const float* xl, int64_t xl_stride, const float* u, const float* bias,
                                         int64_t batch, int32_t dim, float* x_next, int64_t x_next_stride, void* stream) {
    DCTR_REQUIRE(batch >= 0 && dim >= 1, DCTR_E_DIM, "crossnet_matrix_step: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x0 && xl && u && bias && x_next, DCTR_E_NULL, "crossnet_matrix_step: null pointer");
    DCTR_REQUIRE(x0_stride >= dim && xl_stride >= dim && x_next_stride >= dim, DCTR_E_DIM, "crossnet_matrix_step: stride < dim");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 8192) blocks = 8192;
    DCTR_LAUNCH(cross_matrix_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x0, x0_stride, xl, xl_stride, u, bias, batch,
                (int)dim, x_next, x_next_stride);
    return dctr_launch_status("dctr_crossnet_matrix_step");
}

extern "C" int dctr_afm_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim,
                            const float* att_w, const float* att_b, const float* proj_h, const float* proj_p,
                            int32_t att_factor, float* y, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 2 && dim >= 1 && att_factor >= 1, DCTR_E_DIM, "afm_fwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim, DCTR_E_DIM, "afm_fwd: x_stride < fields*dim");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && att_w && att_b && proj_h && proj_p && y, DCTR_E_NULL, "afm_fwd: null pointer");
    const int P = fields * (fields - 1) / 2;
    // the matrix-pipe form: embedding_dim 16 / 32 / 64, attention_factor <= 15, 16-B aligned rows
    if ((dim == 16 || dim == 32 || dim == 64) && att_factor <= 15 && fields <= 255 && x_stride % 4 == 0 && dctr_aligned16(x)) {
        const int PT = (P + 15) / 16;
        const size_t lds_m = ((size_t)16 * PT + 4 * ((size_t)fields * dim + 2 * 16 * PT)) * sizeof(float);
        if (lds_m <= 160 * 1024) {
            const int64_t blocks_m = dctr_ceil_div(batch, 4);
            DCTR_REQUIRE(blocks_m <= 0x7fffffffLL, DCTR_E_DIM, "afm_fwd: batch too large");
#define AFM_M(EBV)                                                                                                              \
            do {                                                                                                                \
                if (lds_m > 64 * 1024) {                                                                                        \
                    hipError_t e = hipFuncSetAttribute((const void*)afm_mfma_kernel<EBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m); \
                    DCTR_REQUIRE(e == hipSuccess, (int)e, "afm_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));       \
                }                                                                                                               \
                DCTR_LAUNCH(afm_mfma_kernel<EBV>, dim3((unsigned)blocks_m), dim3(256), lds_m, (hipStream_t)stream, x, x_stride, batch, fields, \
                            att_w, att_b, proj_h, proj_p, att_factor, y);                                                       \
            } while (0)
            if (dim == 16) AFM_M(1);
            else if (dim == 32) AFM_M(2);
            else AFM_M(4);
#undef AFM_M
            return dctr_launch_status("dctr_afm_fwd");
        }
    }
    const size_t lds = ((size_t)dim * att_factor + 2 * att_factor + dim + 4 * ((size_t)fields * dim + P)) * sizeof(float);
    if (lds > 160 * 1024) {                     // no shape is refused: the streaming form stages nothing
        const int64_t blocks_s = dctr_ceil_div(batch, 4);
        DCTR_REQUIRE(blocks_s <= 0x7fffffffLL, DCTR_E_DIM, "afm_fwd: batch too large");
        DCTR_LAUNCH(afm_stream_kernel, dim3((unsigned)blocks_s), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch, fields, dim, att_w, att_b,
                    proj_h, proj_p, att_factor, y);
        return dctr_launch_status("dctr_afm_fwd");
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)afm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "afm_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(batch, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "afm_fwd: batch too large");
    DCTR_LAUNCH(afm_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, x_stride, batch, fields, dim,
                       att_w, att_b, proj_h, proj_p, att_factor, y);
    return dctr_launch_status("dctr_afm_fwd");
}

extern "C" int dctr_inner_product_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim,
                                      int32_t reduce_sum, float* y, int64_t y_stride, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 2 && dim >= 1, DCTR_E_DIM, "inner_product_fwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim &&
                     y_stride >= (int64_t)(fields * (fields - 1) / 2) * (reduce_sum ? 1 : dim),
                 DCTR_E_DIM, "inner_product_fwd: stride smaller than a row");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y, DCTR_E_NULL, "inner_product_fwd: null pointer");
    const size_t lds = (size_t)4 * fields * dim * sizeof(float);
    if (lds > 160 * 1024) {                     // no shape is refused: rows from global memory
        const int64_t blocks_s = dctr_ceil_div(batch, 4);
        DCTR_REQUIRE(blocks_s <= 0x7fffffffLL, DCTR_E_DIM, "inner_product_fwd: batch too large");
        DCTR_LAUNCH(inner_product_stream_kernel, dim3((unsigned)blocks_s), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch, fields, dim,
                    reduce_sum, y, y_stride);
        return dctr_launch_status("dctr_inner_product_fwd");
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)inner_product_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "inner_product_fwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(batch, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "inner_product_fwd: batch too large");
    DCTR_LAUNCH(inner_product_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, x_stride, batch,
                fields, dim, reduce_sum, y, y_stride);
    return dctr_launch_status("dctr_inner_product_fwd");
}

extern "C" int dctr_bi_interaction_fwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, float* y,
                                       int64_t y_stride, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 1 && dim >= 1, DCTR_E_DIM, "bi_interaction_fwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim && y_stride >= dim, DCTR_E_DIM, "bi_interaction_fwd: stride smaller than a row");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y, DCTR_E_NULL, "bi_interaction_fwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 8192) blocks = 8192;
    DCTR_LAUNCH(bi_interaction_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch, fields, dim, y,
                y_stride);
    return dctr_launch_status("dctr_bi_interaction_fwd");
}

extern "C" size_t dctr_crossnet_mix_workspace_bytes(int32_t dim, int32_t layers, int32_t experts, int32_t low_rank) {
    if (dim < 1 || layers < 1 || experts < 1 || low_rank < 1) return 0;
    return (size_t)layers * experts * low_rank * dim * sizeof(float);      // U transposed to [layers, experts*low_rank, dim]
}

extern "C" int dctr_crossnet_mix_fwd(const float* x, int64_t batch, int32_t dim, int64_t x_stride, const float* U, const float* V,
                                     const float* C, const float* gating, const float* bias, int32_t layers, int32_t experts,
                                     int32_t low_rank, float* y, int64_t y_stride, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    DCTR_REQUIRE(batch >= 0 && dim >= 1 && layers >= 0 && experts >= 1 && low_rank >= 1, DCTR_E_DIM, "crossnet_mix_fwd: bad sizes");
    DCTR_REQUIRE(x_stride >= dim && y_stride >= dim, DCTR_E_DIM, "crossnet_mix_fwd: stride smaller than a row");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && y, DCTR_E_NULL, "crossnet_mix_fwd: null x / y");
    DCTR_REQUIRE(layers == 0 || (U && V && C && gating && bias), DCTR_E_NULL, "crossnet_mix_fwd: null weights");
    const size_t need = dctr_crossnet_mix_workspace_bytes(dim, layers, experts, low_rank);
    DCTR_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need), DCTR_E_NULL,
                 "crossnet_mix_fwd: needs a workspace of %zu B (dctr_crossnet_mix_workspace_bytes)", need);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_cap = 64 * 1024;
    // more rows per workgroup = fewer passes over the weights; fewer when the batch would not fill the chip or LDS is short
    int R = 8;
    while (R > 1 && (cross_mix_lds_bytes(dim, experts, low_rank, R) > lds_cap || dctr_ceil_div(batch, (int64_t)R) < 512)) R >>= 1;
    const size_t lds = cross_mix_lds_bytes(dim, experts, low_rank, R);
    DCTR_REQUIRE(lds <= lds_cap, DCTR_E_UNSUPPORTED, "crossnet_mix_fwd: dim %d x experts %d x low_rank %d does not fit LDS", dim,
                 experts, low_rank);
    float* Ut = static_cast<float*>(workspace);
    if (need > 0) {
        const int64_t tb = dctr_ceil_div((int64_t)(need / sizeof(float)), (int64_t)256);
        hipLaunchKernelGGL(cross_mix_transpose_kernel, dim3((unsigned)(tb > 4096 ? 4096 : tb)), dim3(256), 0, st, U,
                           (int64_t)layers * experts, (int)dim, (int)low_rank, Ut);
    }
    const dim3 grid((unsigned)dctr_ceil_div(batch, (int64_t)R));
#define DCTR_MIX(RR)                                                                                                             \
    DCTR_LAUNCH(cross_mix_kernel<RR>, grid, dim3(256), lds, st, x, x_stride, batch, dim, Ut, V, C, gating, bias, layers, experts, \
                low_rank, y, y_stride)
    if (R == 8) {
        DCTR_MIX(8);
    } else if (R == 4) {
        DCTR_MIX(4);
    } else if (R == 2) {
        DCTR_MIX(2);
    } else {
        DCTR_MIX(1);
    }
#undef DCTR_MIX
    return dctr_launch_status("dctr_crossnet_mix_fwd");
}
