// dctr_gemm: fp32 GEMM on v_mfma_f32_16x16x4_f32 for the training step's plain contractions (see dctr_gemm.h).
//
// Row-major view of the column-major call: C^T (n x m, row stride ldc) = op(B)^T * op(A)^T, i.e. with
//     D[i][j] (i < M = n, j < N = m) = sum_k X[i][k] * Y[k][j],   X = op(B)^T,  Y = op(A)^T
// both operands addressed through element strides (one of each pair is 1), so every op combination is one kernel.
//
// Workgroup = 4 waves, tile (32 TW) x (32 TW) of D with TW in {4, 2}: a wave owns a (16 TW)^2 quarter = TW x TW MFMA tiles.
// A k-block of 16 goes global -> registers -> LDS (double buffered, the next block's loads in flight during this block's MFMAs) in a
// k-major layout whose M / N positions are permuted (position = TW (x % 16) + (x / 16) % TW + 16 TW (x / (16 TW))): the TW tiles
// a lane needs for one k-step are then contiguous — ONE ds_read_b128 (TW = 4) / ds_read_b64 (TW = 2) per operand and k-step
// feeds TW^2 MFMAs (2 LDS reads per 16 MFMAs at TW = 4).  Out-of-range rows / columns / k are zero-filled on the way in and
// masked on the way out: any m, n, k.  TW = 2 is chosen when the 128 x 128 tiling would leave CUs without a workgroup.
#include <stdlib.h>
#include "dctr_common.h"
#include "dctr_gemm.h"

namespace dctr_gemm {

using f32x4 = float __attribute__((ext_vector_type(4)));

struct Params {
    const float* X;
    const float* Y;
    float* D;
    int64_t sxi, sxk, syk, syj, ldd;
    int64_t bx, by, bd;            // batch strides
    int M, N, K;
    int accumulate;                // beta == 1
    int ksplit, kchunk;            // > 1: blockIdx.z = batch * ksplit + s, slice s multiplies k in [s * kchunk, (s + 1) * kchunk) and
                                   // ADDS its product to D with float atomics (D zeroed beforehand when beta == 0)
    int ones_row;                  // >= 0: X row `ones_row` (= M - 1) is not read from memory, it is 1.0 for every k: D's last row
                                   // is then the column sum of Y over k (the bias gradient beside dW = X^T dZ); -1: none
    int64_t sd;                    // ksplit > 1 and sd != 0: slice s STORES its product at D + s * sd (partial products, summed by the caller)
};

// several independent problems in ONE launch (the dW (+ bias) products of every layer of a DNN): blockIdx.x walks the
// concatenated (tile column, tile row, batch slice) lists of the groups
constexpr int MAX_GROUPS = 8;
struct Grouped {
    Params p[MAX_GROUPS];
    int first[MAX_GROUPS + 1];     // first linear workgroup of group i; first[n] = grid size
    int tiles_x[MAX_GROUPS], tiles_y[MAX_GROUPS];
    int n;
};

template <int TW>
__device__ __forceinline__ int perm(int x) { return TW * (x & 15) + ((x >> 4) % TW) + 16 * TW * (x / (16 * TW)); }

// BK = k-block: 16, or 32 with 64 x 64 tiles (TW = 2; an A/B switch: twice the bytes in flight per workgroup did not pay)
template <int TW, int BK>
__device__ __forceinline__ void gemm_tile(const Params& p, int tile_x, int tile_y, int tile_z) {
    constexpr int BT = 32 * TW;                    // tile edge
    constexpr int LD = BT + 4;                     // LDS row stride (floats): 16-B aligned rows
    constexpr int NV = BT * BK / 4 / 256;          // float4 loads per thread and operand: 2 (TW = 4 | BK = 32) / 1 (TW = 2, BK = 16)
    constexpr int KC = BK / 4;                     // 4-k chunks of a row in a k-block
    static_assert(2 * 2 * BK * LD * 4 <= 64 * 1024, "static LDS");
    __shared__ __attribute__((aligned(16))) float Xs[2][BK][LD];
    __shared__ __attribute__((aligned(16))) float Ys[2][BK][LD];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t i0 = (int64_t)tile_y * BT, j0 = (int64_t)tile_x * BT;
    const int bz = tile_z / p.ksplit, ks = tile_z % p.ksplit;
    const int kbeg = ks * p.kchunk;
    const float* X = p.X + (int64_t)bz * p.bx + (int64_t)kbeg * p.sxk;
    const float* Y = p.Y + (int64_t)bz * p.by + (int64_t)kbeg * p.syk;
    float* D = p.D + (int64_t)bz * p.bd + (int64_t)ks * p.sd;
    const int KL = min(p.kchunk, p.K - kbeg);      // this slice's reduction length (the whole K without a split)
    const bool x_kfast = p.sxk == 1;               // X rows contiguous along k
    const bool y_jfast = p.syj == 1;               // Y rows contiguous along j
    const bool x_vec = x_kfast ? ((p.sxi & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0)
                               : ((p.sxk & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0);
    const bool y_vec = y_jfast ? ((p.syk & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0)
                               : ((p.syj & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0);

    // one float4 of a tile: along k (4 consecutive k of one row) or along the row dimension (4 consecutive rows at one k)
    float4 xv[NV], yv[NV];
    // (ones: the row index that reads as 1.0 — X only; rows from it on are not in memory)
    auto load_tile = [&](const float* base, int64_t s_row, int64_t s_k, bool kfast, bool vec, int64_t r0, int n_rows_all, int ones, int k0,
                         float4 (&v)[NV]) {
        const int n_rows = ones >= 0 ? ones : n_rows_all;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = u * 256 + tid;
            int r, k;
            if (kfast) { r = c / KC; k = (c % KC) * 4; }       // KC chunks of 4 k per row
            else { k = c / (BT / 4); r = (c % (BT / 4)) * 4; } // BT / 4 chunks of 4 rows per k
            const int64_t rr = r0 + r;
            const int kk = k0 + k;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kfast) {
                if (rr < n_rows) {
                    const float* src = base + rr * s_row + kk;
                    if (vec && kk + 3 < KL) t = *reinterpret_cast<const float4*>(src);
                    else {
                        if (kk < KL) t.x = src[0];
                        if (kk + 1 < KL) t.y = src[1];
                        if (kk + 2 < KL) t.z = src[2];
                        if (kk + 3 < KL) t.w = src[3];
                    }
                }
            } else {
                if (kk < KL) {
                    const float* src = base + kk * s_k + rr;
                    if (vec && rr + 3 < n_rows) t = *reinterpret_cast<const float4*>(src);
                    else {
                        if (rr < n_rows) t.x = src[0];
                        if (rr + 1 < n_rows) t.y = src[1];
                        if (rr + 2 < n_rows) t.z = src[2];
                        if (rr + 3 < n_rows) t.w = src[3];
                    }
                }
            }
            if (ones >= 0) {
                if (kfast) {
                    if (rr == ones) t = make_float4(kk < KL ? 1.f : 0.f, kk + 1 < KL ? 1.f : 0.f, kk + 2 < KL ? 1.f : 0.f, kk + 3 < KL ? 1.f : 0.f);
                } else if (kk < KL) {
                    if (rr == ones) t.x = 1.f;
                    if (rr + 1 == ones) t.y = 1.f;
                    if (rr + 2 == ones) t.z = 1.f;
                    if (rr + 3 == ones) t.w = 1.f;
                }
            }
            v[u] = t;
        }
    };
    auto store_tile = [&](float (&S)[BK][LD], bool kfast, const float4 (&v)[NV]) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = u * 256 + tid;
            if (kfast) {
                const int r = c / KC, k = (c % KC) * 4, pr = perm<TW>(r);
                S[k][pr] = v[u].x; S[k + 1][pr] = v[u].y; S[k + 2][pr] = v[u].z; S[k + 3][pr] = v[u].w;
            } else {
                const int k = c / (BT / 4), r = (c % (BT / 4)) * 4;
                S[k][perm<TW>(r)] = v[u].x; S[k][perm<TW>(r + 1)] = v[u].y; S[k][perm<TW>(r + 2)] = v[u].z; S[k][perm<TW>(r + 3)] = v[u].w;
            }
        }
    };

    f32x4 acc[TW][TW];
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Pipeline: k-block kb is multiplied out of LDS while block kb + 1 sits in registers (xv / yv) on its way to the other LDS
    // buffer and block kb + 2 is in flight from memory (xw / yw): with few workgroups per CU (small outputs with a long reduction)
    // nothing else hides a ~2 us global round trip, and one block of MFMAs is ~0.2 us
    float4 xw[NV], yw[NV];
    const int nkb = (KL + BK - 1) / BK;
    load_tile(X, p.sxi, p.sxk, x_kfast, x_vec, i0, p.M, p.ones_row, 0, xv);
    load_tile(Y, p.syj, p.syk, !y_jfast, y_vec, j0, p.N, -1, 0, yv);          // (Y's "rows" are its columns j: row stride syj, k stride syk)
    load_tile(X, p.sxi, p.sxk, x_kfast, x_vec, i0, p.M, p.ones_row, BK, xw);          // (past K: zeros)
    load_tile(Y, p.syj, p.syk, !y_jfast, y_vec, j0, p.N, -1, BK, yw);
    store_tile(Xs[0], x_kfast, xv);
    store_tile(Ys[0], !y_jfast, yv);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            xv[u] = xw[u];
            yv[u] = yw[u];
        }
        if (kb + 2 < nkb) {
            load_tile(X, p.sxi, p.sxk, x_kfast, x_vec, i0, p.M, p.ones_row, (kb + 2) * BK, xw);
            load_tile(Y, p.syj, p.syk, !y_jfast, y_vec, j0, p.N, -1, (kb + 2) * BK, yw);
        }
        // lane (g, j): A[i = j][k = g] of M-tile a = Xs[k][16 TW wm + TW j + a];  B[k = g][n = j] of N-tile b likewise
#pragma unroll
        for (int t = 0; t < BK / 4; ++t) {
            float av[TW], bv[TW];
            const float* xa = &Xs[cur][4 * t + g][16 * TW * wm + TW * j];
            const float* yb = &Ys[cur][4 * t + g][16 * TW * wn + TW * j];
            if constexpr (TW == 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(xa), b4 = *reinterpret_cast<const float4*>(yb);
                av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
                bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
            } else {
                const float2 a2 = *reinterpret_cast<const float2*>(xa), b2 = *reinterpret_cast<const float2*>(yb);
                av[0] = a2.x; av[1] = a2.y;
                bv[0] = b2.x; bv[1] = b2.y;
            }
#pragma unroll
            for (int a = 0; a < TW; ++a)
#pragma unroll
                for (int b = 0; b < TW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (kb + 1 < nkb) {
            store_tile(Xs[cur ^ 1], x_kfast, xv);
            store_tile(Ys[cur ^ 1], !y_jfast, yv);
        }
        __syncthreads();
    }
    // C tile (a, b): lane (g, j), register r = D[i0 + 16 TW wm + 16 a + 4g + r][j0 + 16 TW wn + 16 b + j]
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + 16 * TW * wm + 16 * a + 4 * g + r;
            if (i >= p.M) continue;
#pragma unroll
            for (int b = 0; b < TW; ++b) {
                const int64_t jj = j0 + 16 * TW * wn + 16 * b + j;
                if (jj < p.N) {
                    float* d = D + i * p.ldd + jj;
                    if (p.ksplit > 1 && p.sd == 0) unsafeAtomicAdd(d, acc[a][b][r]);
                    else *d = p.accumulate ? *d + acc[a][b][r] : acc[a][b][r];
                }
            }
        }
}

template <int TW, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(Params p) {
    gemm_tile<TW, BK>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

template <int TW, int BK>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(Grouped gp) {
    const int b = (int)blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUPS; ++i)
        if (i < gp.n && b >= gp.first[i]) gi = i;
    int t = b - gp.first[gi];
    const int tx = t % gp.tiles_x[gi];
    t /= gp.tiles_x[gi];
    const int ty = t % gp.tiles_y[gi], tz = t / gp.tiles_y[gi];
    gemm_tile<TW, BK>(gp.p[gi], tx, ty, tz);
}

// k-block of the 64 x 64 tiling: 16 (default) / 32 (DCTR_GEMM_BK=32: the A/B switch of scripts/gemm_lab.sh — measured equal
// at B = 4096 and 3 % slower at B = 16,384 on the DeepFM step, profiles/r03c_gemm_lab.log)
static int small_bk() {
    static const int v = [] {
        const char* e = dctr_lab_env("DCTR_GEMM_BK");
        return (e != nullptr && atoi(e) == 32) ? 32 : 16;
    }();
    return v;
}

static void fill_params(Params& p, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, int64_t stride_a, const float* B, int ldb,
                        int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c) {
    // D = C^T (row-major [n][m], row stride ldc);  X[i][kk] = op(B)(kk, i);  Y[kk][j] = op(A)(j, kk)
    p.X = B;
    p.Y = A;
    p.D = C;
    p.M = n;
    p.N = m;
    p.K = k;
    if (op_b == OP_N) { p.sxi = ldb; p.sxk = 1; } else { p.sxi = 1; p.sxk = ldb; }
    if (op_a == OP_N) { p.syk = lda; p.syj = 1; } else { p.syk = 1; p.syj = lda; }
    p.ldd = ldc;
    p.bx = stride_b;
    p.by = stride_a;
    p.bd = stride_c;
    p.accumulate = beta == 1.f ? 1 : 0;
    p.ksplit = 1;
    p.kchunk = k;
    p.ones_row = -1;
}

static int launch(hipStream_t stream, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, int64_t stride_a, const float* B,
                  int ldb, int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c, int batch) {
    DCTR_REQUIRE(m >= 0 && n >= 0 && k >= 0 && batch >= 0, DCTR_E_DIM, "sgemm: m=%d n=%d k=%d batch=%d", m, n, k, batch);
    DCTR_REQUIRE(beta == 0.f || beta == 1.f, DCTR_E_UNSUPPORTED, "sgemm: beta %g (0 or 1)", (double)beta);
    if (m == 0 || n == 0 || batch == 0) return DCTR_OK;
    DCTR_REQUIRE(A != nullptr && B != nullptr && C != nullptr, DCTR_E_NULL, "sgemm: null pointer");
    Params p{};
    fill_params(p, op_a, op_b, m, n, k, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c);
    const int64_t t128 = dctr_ceil_div(p.M, 128) * dctr_ceil_div(p.N, 128) * batch;
    const bool small = t128 < 2 * (int64_t)dctr_n_cus();       // the 128 x 128 tiling would leave CUs (or their second workgroup slot) idle: 64 x 64 tiles
    const int bt = small ? 64 : 128;
    // a small output with a long reduction (dW = X^T dZ over the batch, the CrossNetMix projections' gradients): split K over
    // workgroups until the chip is filled (>= 256 k per slice), partial products added with float atomics
    const int64_t tiles = dctr_ceil_div(p.N, bt) * dctr_ceil_div(p.M, bt) * batch;
    int ksplit = 1;
    if (tiles * 2 <= (int64_t)dctr_n_cus() && k >= 1024) {
        int64_t want = 2 * (int64_t)dctr_n_cus() / tiles;
        if (want > k / 256) want = k / 256;
        if (want > 32) want = 32;
        ksplit = want > 1 ? (int)want : 1;
    }
    p.ksplit = ksplit;
    p.kchunk = ksplit > 1 ? (int)(dctr_ceil_div(dctr_ceil_div(k, ksplit), 32) * 32) : k;
    if (ksplit > 1) {
        p.ksplit = (int)dctr_ceil_div(k, p.kchunk);            // (no empty slices)
        if (!p.accumulate) {                                   // atomics add to zero
            for (int b = 0; b < batch; ++b) {
                hipError_t e = hipMemset2DAsync(C + (int64_t)b * stride_c, (size_t)ldc * sizeof(float), 0, (size_t)m * sizeof(float), (size_t)n, stream);
                DCTR_REQUIRE(e == hipSuccess, (int)e, "sgemm: clearing C failed: %s", hipGetErrorString(e));
            }
        }
    }
    const dim3 grid((unsigned)dctr_ceil_div(p.N, bt), (unsigned)dctr_ceil_div(p.M, bt), (unsigned)(batch * p.ksplit));
    DCTR_REQUIRE(grid.y <= 65535 && grid.z <= 65535, DCTR_E_DIM, "sgemm: grid too large (n=%d, batch=%d)", n, batch);
    if (small && small_bk() == 32) hipLaunchKernelGGL((gemm_kernel<2, 32>), grid, dim3(256), 0, stream, p);
    else if (small) hipLaunchKernelGGL((gemm_kernel<2, 16>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_kernel<4, 16>), grid, dim3(256), 0, stream, p);
    return dctr_launch_status("dctr_gemm::sgemm");
}

int k_slices(int k, int rows_per_slice) {
    if (k <= 0) return 1;
    int want = (int)dctr_ceil_div(k, rows_per_slice > 0 ? rows_per_slice : k);
    if (want > 32) want = 32;
    if (want < 1) want = 1;
    const int chunk = (int)(dctr_ceil_div(dctr_ceil_div(k, want), 32) * 32);
    return (int)dctr_ceil_div(k, chunk);
}

int sgemm_grouped(hipStream_t stream, const GroupDesc* g, int n_groups) {
    DCTR_REQUIRE(g != nullptr && n_groups >= 1 && n_groups <= MAX_GROUPS, DCTR_E_DIM, "sgemm_grouped: %d groups (1..%d)", n_groups, MAX_GROUPS);
    Grouped gp{};
    int64_t total = 0;
    int ng = 0;
    for (int i = 0; i < n_groups; ++i) {
        const GroupDesc& d = g[i];
        DCTR_REQUIRE(d.m >= 0 && d.n >= 0 && d.k >= 0 && d.batch >= 0, DCTR_E_DIM, "sgemm_grouped[%d]: m=%d n=%d k=%d batch=%d", i, d.m, d.n, d.k, d.batch);
        if (d.m == 0 || d.n == 0 || d.batch == 0) continue;
        DCTR_REQUIRE(d.A != nullptr && d.B != nullptr && d.C != nullptr, DCTR_E_NULL, "sgemm_grouped[%d]: null pointer", i);
        Params& p = gp.p[ng];
        fill_params(p, d.op_a, d.op_b, d.m, d.n, d.k, d.A, d.lda, d.stride_a, d.B, d.ldb, d.stride_b, 0.f, d.C, d.ldc, d.stride_c);
        if (d.ones_last) p.ones_row = d.n - 1;
        // k_slices with slice_stride_c == 0: the slices ADD their products to the one C with float atomics — beta = 1 semantics
        // (accumulate must be set: C holds what is added to)
        DCTR_REQUIRE(!(d.k_slices > 1 && (d.slice_stride_c == 0) != (d.accumulate != 0)), DCTR_E_UNSUPPORTED,
                     "sgemm_grouped[%d]: k_slices store to their own C (slice_stride_c, no accumulate) or add atomically to one C (accumulate)", i);
        p.accumulate = d.accumulate ? 1 : 0;
        int slices = 1;
        if (d.k_slices > 1 && d.k > 0) {                       // k cut into slices of a multiple of 32, each stored to its own C (or added)
            p.kchunk = (int)(dctr_ceil_div(dctr_ceil_div(d.k, d.k_slices), 32) * 32);
            slices = (int)dctr_ceil_div(d.k, p.kchunk);
            DCTR_REQUIRE(slices == d.k_slices, DCTR_E_DIM, "sgemm_grouped[%d]: k=%d does not cut into %d slices of a multiple of 32 (use dctr_gemm::k_slices)", i, d.k, d.k_slices);
            p.ksplit = slices;
            p.sd = d.slice_stride_c;
        }
        gp.tiles_x[ng] = (int)dctr_ceil_div(p.N, 64);
        gp.tiles_y[ng] = (int)dctr_ceil_div(p.M, 64);
        gp.first[ng] = (int)total;
        total += (int64_t)gp.tiles_x[ng] * gp.tiles_y[ng] * d.batch * slices;
        DCTR_REQUIRE(total < 0x7fffffffLL, DCTR_E_DIM, "sgemm_grouped: grid too large");
        ++ng;
    }
    if (ng == 0) return DCTR_OK;
    gp.first[ng] = (int)total;
    gp.n = ng;
    if (small_bk() == 32) hipLaunchKernelGGL((gemm_grouped_kernel<2, 32>), dim3((unsigned)total), dim3(256), 0, stream, gp);
    else hipLaunchKernelGGL((gemm_grouped_kernel<2, 16>), dim3((unsigned)total), dim3(256), 0, stream, gp);
    return dctr_launch_status("dctr_gemm::sgemm_grouped");
}

int sgemm(hipStream_t stream, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, const float* B, int ldb, float beta,
          float* C, int ldc) {
    return launch(stream, op_a, op_b, m, n, k, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1);
}

int sgemm_strided_batched(hipStream_t stream, Op op_a, Op op_b, int m, int n, int k, const float* A, int lda, int64_t stride_a,
                          const float* B, int ldb, int64_t stride_b, float beta, float* C, int ldc, int64_t stride_c, int batch) {
    return launch(stream, op_a, op_b, m, n, k, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c, batch);
}

}  // namespace dctr_gemm

// C ABI (include/dctr.h): the contraction as one entry point, column-major BLAS semantics
extern "C" int dctr_sgemm(int32_t trans_a, int32_t trans_b, int32_t m, int32_t n, int32_t k, const float* A, int32_t lda, int64_t stride_a,
                          const float* B, int32_t ldb, int64_t stride_b, float beta, float* C, int32_t ldc, int64_t stride_c,
                          int32_t batch, void* stream) {
    DCTR_REQUIRE(lda >= 1 && ldb >= 1 && ldc >= m, DCTR_E_DIM, "dctr_sgemm: leading dimensions lda=%d ldb=%d ldc=%d (m=%d)", lda, ldb, ldc, m);
    return dctr_gemm::sgemm_strided_batched((hipStream_t)stream, trans_a ? dctr_gemm::OP_T : dctr_gemm::OP_N,
                                            trans_b ? dctr_gemm::OP_T : dctr_gemm::OP_N, m, n, k, A, lda, stride_a, B, ldb, stride_b, beta, C,
                                            ldc, stride_c, batch);
}
