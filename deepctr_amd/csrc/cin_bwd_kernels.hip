// Backward of one CIN layer (reference deepctr/layers/interaction.py:288-300) with respect to its two inputs, z-free:
//   T[r, i*Fk + j] = sum_h dpre[r, h] W[i*Fk + j, h]           (the gradient of the never-materialised outer product z)
//   dxk[r, j] = sum_i x0[r, i] T[r, i*Fk + j]                   dx0[r, i] += sum_j xk[r, j] T[r, i*Fk + j]
// rows r = (sample, embedding dimension) are independent.  The first version ran T as a rocBLAS GEMM into a [B*D, F0*Fk] buffer
// (436 MB at C3's second layer) and contracted it in a second kernel: 434 + 347 us per step.  Here a workgroup owns 64 rows:
// the dpre and x0 tiles sit in LDS, a wave walks 16-row blocks of W (the MFMA A operand: M = i*Fk + j, the four k-slots of a
// lane being four consecutive h of one 16-B load), T^T tiles [16 (i,j) x 16 rows] come out of v_mfma_f32_16x16x4_f32 and are
// contracted at once, mostly in registers.  W streams from L2 one tile ahead of the MFMAs.
#include "dctr_common.h"
#include "mfma_tile.h"

namespace dctr_cinbwd {

constexpr int RB = 64;        // rows per workgroup: four 16-row N tiles
constexpr int DZ_X0N = 8;     // x0 staging registers per thread: RB * F0 <= 256 * DZ_X0N  (F0 <= 32)
constexpr int MAXKS = 8;      // H = 16 KS <= 128: the W rows of one tile are held (and prefetched) as KS float4 per lane

struct DzParams {
    const float* dpre;        // [rows, H]
    const float* W;           // [F0*Fk, H]
    const float* x0t;         // [rows, F0]
    const float* xk;          // [rows, ldk], first Fk columns
    int64_t ldk;
    float* dx0t;              // [rows, F0]  accumulated
    float* dxk;               // [rows, Fk]  written
    int64_t rows;
    int32_t F0, Fk, H;
    int32_t sd, s0, sk;       // LDS row strides (floats) of the dpre / x0 / xk tiles
};

// KS = H / 16 is a template parameter and the prefetch of the next tile is unconditional (clamped): with branches between the
// loads hipcc cannot count them and waits for vmcnt(0) — i.e. for the tile just requested — before the first MFMA.
//
// Work split.  The (i, j) space is walked as F0 x TPI tiles of 16 consecutive j of one i (Fk padded to a multiple of 16: the
// padding columns read a clamped W row and multiply x_k = 0).  A wave owns j blocks (all i: TPI >= 4) or a share of the i of one
// j block (TPI < 4), so that over its tiles
//   * its x_k values (4 rows-tiles x 4 j per lane) stay in registers,
//   * dxk[r, j] accumulates in registers and reaches LDS once per j block,
//   * dx0[r, i] of a tile is summed over the lane's four j, then over the four k-slot groups with two DPP shuffles: one LDS
//     atomic per row and tile.
// (First version: every T element went to LDS with two atomics — 1.25 ms for C3's second layer against 0.46 ms for the GEMM +
// contraction pair it replaces.)
template <int KS>
__global__ __launch_bounds__(256, 2) void cin_dz_fused_kernel(DzParams p) {
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dps = smem;                     // [RB][sd]
    float* x0s = dps + RB * p.sd;          // [RB][s0]
    float* ax0 = x0s + RB * p.s0;          // [RB][s0]  dx0 of this layer
    float* axk = ax0 + RB * p.s0;          // [RB][sk]  dxk
    const int F0 = p.F0, Fk = p.Fk, H = p.H;
    const int64_t r0 = (int64_t)blockIdx.x * RB;
    {
        // all loads of the tile in flight before the LDS stores (RB * H / 4 = KS * 256 float4: KS per thread, compile-time steps)
        constexpr int H4 = KS * 4, DROWS = 256 / H4;
        const int dm = threadIdx.x / H4, dc = threadIdx.x - dm * H4;
        float4 v[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + dm + u * DROWS < p.rows) v[u] = *reinterpret_cast<const float4*>(p.dpre + (r0 + dm + u * DROWS) * H + 4 * dc);
        }
        float xv[DZ_X0N];
#pragma unroll
        for (int u = 0; u < DZ_X0N; ++u) {                      // the x0 tile is RB * F0 contiguous floats
            const int idx = threadIdx.x + u * 256;
            xv[u] = (idx < RB * F0 && r0 * F0 + idx < p.rows * F0) ? p.x0t[r0 * F0 + idx] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KS; ++u) *reinterpret_cast<float4*>(dps + (dm + u * DROWS) * p.sd + 4 * dc) = v[u];
#pragma unroll
        for (int u = 0; u < DZ_X0N; ++u) {
            const int idx = threadIdx.x + u * 256;
            if (idx < RB * F0) {
                const int m = idx / F0, i = idx - m * F0;
                x0s[m * p.s0 + i] = xv[u];
                ax0[m * p.s0 + i] = 0.f;
            }
        }
        for (int idx = threadIdx.x; idx < RB * p.sk; idx += 256) axk[idx] = 0.f;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, jl = lane & 15;
    const int TPI = (Fk + 15) >> 4;
    int jb0, jbs, i0, is;
    if (TPI >= 4) { jb0 = wave; jbs = 4; i0 = 0; is = 1; }
    else { is = 4 / TPI; jb0 = wave % TPI; jbs = TPI; i0 = wave / TPI; }
    const bool active = i0 < is;             // (TPI == 3: the fourth wave has no share)
    for (int jb = jb0; jb < TPI && active; jb += jbs) {
        float xkv[4][4], dk[4][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 16 + 4 * g + r;
                const int64_t row = r0 + nt * 16 + jl;
                xkv[nt][r] = (j < Fk && row < p.rows) ? p.xk[row * p.ldk + j] : 0.f;
                dk[nt][r] = 0.f;
            }
        float4 wcur[KS], wnext[KS];
        const int jrow = min(jb * 16 + jl, Fk - 1);                        // this lane's A-operand row within the tile
        auto load_w = [&](int i, float4 (&w)[KS]) {
            const float* src = p.W + ((int64_t)i * Fk + jrow) * H + 4 * g;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[ks] = *reinterpret_cast<const float4*>(src + ks * 16);
        };
        load_w(min(i0, F0 - 1), wcur);
        for (int i = i0; i < F0; i += is) {
            load_w(min(i + is, F0 - 1), wnext);
            __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks these loads below the MFMAs to save registers)
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float4 b4[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b4[nt] = *reinterpret_cast<const float4*>(dps + (nt * 16 + jl) * p.sd + ks * 16 + 4 * g);
                const float a[4] = {wcur[ks].x, wcur[ks].y, wcur[ks].z, wcur[ks].w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float b = c == 0 ? b4[nt].x : c == 1 ? b4[nt].y : c == 2 ? b4[nt].z : b4[nt].w;
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b, acc[nt], 0, 0, 0);
                    }
            }
            // acc[nt][r] = T[row nt*16 + jl][(i, j = jb*16 + 4g + r)]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int m = nt * 16 + jl;
                const float x0v = x0s[m * p.s0 + i];
                float sx = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float T = acc[nt][r];
                    dk[nt][r] = fmaf(x0v, T, dk[nt][r]);
                    sx = fmaf(xkv[nt][r], T, sx);
                }
                sx += __shfl_xor(sx, 16, 64);
                sx += __shfl_xor(sx, 32, 64);
                if (g == 0) atomicAdd(&ax0[m * p.s0 + i], sx);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wcur[ks] = wnext[ks];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 16 + 4 * g + r;
                if (j < Fk) atomicAdd(&axk[(nt * 16 + jl) * p.sk + j], dk[nt][r]);
            }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < RB * F0; idx += 256) {
        const int m = idx / F0, i = idx - m * F0;
        if (r0 + m < p.rows) p.dx0t[(r0 + m) * F0 + i] += ax0[m * p.s0 + i];
    }
    if (p.dxk != nullptr) {
        for (int idx = threadIdx.x; idx < RB * Fk; idx += 256) {
            const int m = idx / Fk, j = idx - m * Fk;
            if (r0 + m < p.rows) p.dxk[(r0 + m) * Fk + j] = axk[m * p.sk + j];
        }
    }
}

static size_t dw_lds_bytes(int F0, int Fk, int H, int& sd, int& s0, int& sk);
static size_t lds_bytes(int F0, int Fk, int H, int& sd, int& s0, int& sk) {
    sd = H + 4;
    s0 = F0 | 1;
    sk = Fk | 1;
    return (size_t)RB * (sd + 2 * s0 + sk) * sizeof(float);
}


int launch_dz_fused(const float* dpre, const float* W, const float* x0t, const float* xk, int64_t ldk, int F0, int Fk, int H,
                    int64_t rows, float* dx0t, float* dxk, hipStream_t st) {
    DzParams p{};
    p.dpre = dpre; p.W = W; p.x0t = x0t; p.xk = xk; p.ldk = ldk; p.dx0t = dx0t; p.dxk = dxk; p.rows = rows;
    p.F0 = F0; p.Fk = Fk; p.H = H;
    const size_t lds = lds_bytes(F0, Fk, H, p.sd, p.s0, p.sk);
    const int64_t blocks = dctr_ceil_div(rows, (int64_t)RB);
#define DZ_CASE(KSV)                                                                                                              \
    case KSV: {                                                                                                                   \
        if (lds > 64 * 1024) {                                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)cin_dz_fused_kernel<KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)lds);                                                                         \
            if (e != hipSuccess) return (int)e;                                                                                   \
        }                                                                                                                         \
        hipLaunchKernelGGL(cin_dz_fused_kernel<KSV>, dim3((unsigned)blocks), dim3(256), lds, st, p);                              \
        return 0;                                                                                                                 \
    }
    switch (H / 16) {
        DZ_CASE(1) DZ_CASE(2) DZ_CASE(3) DZ_CASE(4) DZ_CASE(5) DZ_CASE(6) DZ_CASE(7) DZ_CASE(8)
        default: return -1;
    }
#undef DZ_CASE
}

// ---------------------------------------------------------------------------------------------------
// Filter gradient of one CIN layer, z-free:   dW[i*Fk + j, h] += sum_r x0[r, i] xk[r, j] dpre[r, h]
// The first version materialised z [B*D, F0*Fk] (cin_outer_kernel, 436 MB for C3's second layer) and ran z^T dpre as a strided
// batch of rocBLAS GEMMs + a sum over the partial products: 0.65 ms per step.  Here the rows are the K dimension of the MFMA:
// a workgroup owns 4*TW tiles of 16 consecutive j of one i (the M dimension; A = x0[r, i] * xk[r, j] formed in registers from
// two LDS tiles, the four k-slots of a lane being rows 4g + c of a 16-row block) x all H columns (N; B = dpre from LDS) and a
// slice of the rows, walked in 64-row tiles staged through LDS; its partial dW leaves with one atomic per element.
// ---------------------------------------------------------------------------------------------------
constexpr int DW_TR = 64;     // rows per staged tile
constexpr int DW_X0N = 8;     // staging registers per thread: DW_TR * F0 <= 256 * DW_X0N  (F0 <= 32)
constexpr int DW_XKN = 16;    //                               DW_TR * Fk <= 256 * DW_XKN with 16-B loads (Fk <= 64), half of it without

struct DwParams {
    const float* dpre;        // [rows, H]
    const float* x0t;         // [rows, F0]
    const float* xk;          // [rows, ldk], first Fk columns
    int64_t ldk;
    float* parts;             // [n_splits][F0*Fk, H]: this row slice's partial dW, written (the caller sums the slices into dW)
    int64_t rows;
    int64_t rows_per_split;   // multiple of DW_TR
    int32_t F0, Fk, H, n_chunks;
    int32_t sd, s0, sk;       // LDS row strides (floats) of the dpre / x0 / xk tiles
};

// XK4: the x_k tile is requested with 16-B loads (Fk % 4 == 0, ldk % 4 == 0, 16-B aligned: C3's second layer, x_k = the previous
// layer's saved activations); else element-wise (Fk <= 32: the first layer, x_k = x_0 with its odd row length)
template <int TW, int NH, bool XK4>
__global__ __launch_bounds__(256, 2) void cin_dw_fused_kernel(DwParams p) {
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dps = smem;                       // [DW_TR][sd]
    float* x0s = dps + DW_TR * p.sd;         // [DW_TR][s0]
    float* xks = x0s + DW_TR * p.s0;         // [DW_TR][sk]   columns j < TPI*16, zero from Fk on
    const int F0 = p.F0, Fk = p.Fk, H = p.H;
    const int TPI = (Fk + 15) >> 4, Fkp = TPI * 16, NT = TPI * F0;
    const int chunk = blockIdx.x % p.n_chunks;
    const int64_t rlo = (int64_t)(blockIdx.x / p.n_chunks) * p.rows_per_split;
    const int64_t rhi = min(rlo + p.rows_per_split, p.rows);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, jl = lane & 15;
    // this wave's tiles: tau = jb * F0 + i  (jb-major: neighbours share their x_k columns)
    int ti[TW], tj[TW];
    bool tok[TW];
#pragma unroll
    for (int u = 0; u < TW; ++u) {
        const int tau = (chunk * 4 + wave) * TW + u;
        tok[u] = tau < NT;
        const int tc = tok[u] ? tau : 0;
        tj[u] = tc / F0;
        ti[u] = tc - tj[u] * F0;
    }
    f32x4 acc[TW][NH];
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
        for (int nt = 0; nt < NH; ++nt) acc[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging registers: the next 64-row tile is requested from HBM / L2 before the MFMAs of the current one and written to LDS
    // after them (a loop of load -> LDS store per element serialised ~30 memory latencies per tile: 2x the MFMA time)
    float4 sd4[NH];
    float sx0[DW_X0N];
    // dpre tile: DW_TR * H / 4 = NH * 256 float4, element u of a thread = row dm + u * DROWS, float4 column dc (compile-time steps)
    constexpr int H4 = NH * 4, DROWS = 256 / H4;
    const int dm = threadIdx.x / H4, dc = threadIdx.x - dm * H4;
    // x0 / xk tiles as flat [DW_TR][F0] / [DW_TR][Fk] index spaces, (row, column) of element u stepped incrementally
    const int x0_m0 = threadIdx.x / F0, x0_i0 = threadIdx.x - x0_m0 * F0, x0_dm = 256 / F0, x0_di = 256 - x0_dm * F0;
    // xk: XK4: flat over [DW_TR][Fk/4] float4; else flat over [DW_TR][Fk] floats
    constexpr int XKN = XK4 ? DW_XKN / 4 : DW_XKN / 2;
    const int kw = XK4 ? Fk >> 2 : Fk;
    const int xk_m0 = threadIdx.x / kw, xk_j0 = threadIdx.x - xk_m0 * kw, xk_dm = 256 / kw, xk_dj = 256 - xk_dm * kw;
    float4 sxk4[XK4 ? XKN : 1];
    float sxk1[XK4 ? 1 : XKN];
    auto request = [&](int64_t r0) {
        const float* dsrc = p.dpre + (r0 + dm) * H + 4 * dc;
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            sd4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + dm + u * DROWS < rhi) sd4[u] = *reinterpret_cast<const float4*>(dsrc + (int64_t)u * DROWS * H);
        }
        int m = x0_m0, i = x0_i0;
#pragma unroll
        for (int u = 0; u < DW_X0N; ++u) {
            sx0[u] = (m < DW_TR && r0 + m < rhi) ? p.x0t[(r0 + m) * F0 + i] : 0.f;
            m += x0_dm; i += x0_di;
            if (i >= F0) { i -= F0; ++m; }
        }
        m = xk_m0;
        int j = xk_j0;
#pragma unroll
        for (int u = 0; u < XKN; ++u) {
            const bool ok = m < DW_TR && r0 + m < rhi;
            if constexpr (XK4) sxk4[u] = ok ? *reinterpret_cast<const float4*>(p.xk + (r0 + m) * p.ldk + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            else sxk1[u] = ok ? p.xk[(r0 + m) * p.ldk + j] : 0.f;
            m += xk_dm; j += xk_dj;
            if (j >= kw) { j -= kw; ++m; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NH; ++u) *reinterpret_cast<float4*>(dps + (dm + u * DROWS) * p.sd + 4 * dc) = sd4[u];
        int m = x0_m0, i = x0_i0;
#pragma unroll
        for (int u = 0; u < DW_X0N; ++u) {
            if (m < DW_TR) x0s[m * p.s0 + i] = sx0[u];
            m += x0_dm; i += x0_di;
            if (i >= F0) { i -= F0; ++m; }
        }
        m = xk_m0;
        int j = xk_j0;
#pragma unroll
        for (int u = 0; u < XKN; ++u) {
            if (m < DW_TR) {
                if constexpr (XK4) *reinterpret_cast<float4*>(xks + m * p.sk + 4 * j) = sxk4[u];      // sk % 4 == 0
                else xks[m * p.sk + j] = sxk1[u];
            }
            m += xk_dm; j += xk_dj;
            if (j >= kw) { j -= kw; ++m; }
        }
    };
    // the padding columns [Fk, Fkp) of the xk tile stay zero
    for (int idx = threadIdx.x; idx < DW_TR * (Fkp - Fk); idx += 256) {
        const int m = idx / (Fkp - Fk), j = Fk + idx - m * (Fkp - Fk);
        xks[m * p.sk + j] = 0.f;
    }
    request(rlo);
    for (int64_t r0 = rlo; r0 < rhi; r0 += DW_TR) {
        __syncthreads();                    // the previous tile's reads are done
        commit();
        __syncthreads();
        request(r0 + DW_TR);                // (past rhi: zeros, never committed)
#pragma unroll 1
        for (int rb = 0; rb < DW_TR / 16; ++rb) {
            const int mrow = rb * 16 + 4 * g;               // this lane's four k-slot rows: mrow + c
            float bf[NH][4];
#pragma unroll
            for (int nt = 0; nt < NH; ++nt)
#pragma unroll
                for (int c = 0; c < 4; ++c) bf[nt][c] = dps[(mrow + c) * p.sd + nt * 16 + jl];
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                float a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = x0s[(mrow + c) * p.s0 + ti[u]] * xks[(mrow + c) * p.sk + tj[u] * 16 + jl];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int nt = 0; nt < NH; ++nt)
                        acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bf[nt][c], acc[u][nt], 0, 0, 0);
            }
        }
    }
    // acc[u][nt][r] = partial dW[(i, j = jb*16 + 4g + r)][h = nt*16 + jl] of this row slice: plain stores into the slice's own
    // buffer (atomics into dW itself: 1,027 workgroups x 16 K elements = 17 M of them, 0.76 ms — more than the MFMAs)
    float* part = p.parts + (int64_t)(blockIdx.x / p.n_chunks) * ((int64_t)F0 * Fk * H);
#pragma unroll
    for (int u = 0; u < TW; ++u) {
        if (!tok[u]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = tj[u] * 16 + 4 * g + r;
            if (j >= Fk) continue;
            float* dst = part + ((int64_t)ti[u] * Fk + j) * H + jl;
#pragma unroll
            for (int nt = 0; nt < NH; ++nt) dst[nt * 16] = acc[u][nt][r];
        }
    }
}

static size_t dw_lds_bytes(int F0, int Fk, int H, int& sd, int& s0, int& sk) {
    sd = H + 4;
    s0 = F0 | 1;
    sk = ((Fk + 15) / 16) * 16 + 4;
    return (size_t)DW_TR * (sd + s0 + sk) * sizeof(float);
}

// shapes both fused kernels take (then dctr_cin_bwd needs neither z nor dz for the layer)
bool fused_shape_ok(int F0, int Fk, int H) {
    int sd, s0, sk;
    return H % 16 == 0 && H >= 16 && H <= 16 * MAXKS && F0 >= 1 && F0 * DW_TR <= 256 * DW_X0N && Fk >= 1 &&
           (Fk * DW_TR <= 128 * DW_XKN || (Fk % 4 == 0 && Fk * DW_TR <= 256 * DW_XKN)) && lds_bytes(F0, Fk, H, sd, s0, sk) <= 160 * 1024 &&
           dw_lds_bytes(F0, Fk, H, sd, s0, sk) <= 80 * 1024;
}

static int n_cus() { return dctr_n_cus(); }

constexpr int DW_TW = 2;      // (i, j-block) tiles per wave

// row slices of the dW kernel: ~1,024 workgroups (two resident per CU, two rounds), each slice a multiple of the staged tile
static void dw_split(int F0, int Fk, int64_t rows, int& n_chunks, int64_t& rows_per_split, int64_t& splits) {
    const int NT = ((Fk + 15) / 16) * F0;
    n_chunks = (NT + 4 * DW_TW - 1) / (4 * DW_TW);
    splits = dctr_ceil_div((int64_t)4 * n_cus(), (int64_t)n_chunks);
    const int64_t max_splits = dctr_ceil_div(rows, (int64_t)DW_TR);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    rows_per_split = dctr_ceil_div(dctr_ceil_div(rows, splits), (int64_t)DW_TR) * DW_TR;
    splits = dctr_ceil_div(rows, rows_per_split);
}

// floats of the partial-product buffer launch_dw_fused needs
int64_t dw_parts_floats(int F0, int Fk, int H, int64_t rows) {
    int n_chunks;
    int64_t rps, splits;
    dw_split(F0, Fk, rows, n_chunks, rps, splits);
    return splits * (int64_t)F0 * Fk * H;
}

// writes n_parts partial dW [F0*Fk, H] into `parts`; the caller adds their sum to dW
int launch_dw_fused(const float* dpre, const float* x0t, const float* xk, int64_t ldk, int F0, int Fk, int H, int64_t rows,
                    float* parts, int* n_parts, hipStream_t st) {
    constexpr int TW = DW_TW;
    DwParams p{};
    p.dpre = dpre; p.x0t = x0t; p.xk = xk; p.ldk = ldk; p.parts = parts; p.rows = rows; p.F0 = F0; p.Fk = Fk; p.H = H;
    const size_t lds = dw_lds_bytes(F0, Fk, H, p.sd, p.s0, p.sk);
    int64_t splits;
    dw_split(F0, Fk, rows, p.n_chunks, p.rows_per_split, splits);
    *n_parts = (int)splits;
    const int64_t blocks = splits * p.n_chunks;
    if (blocks > 0x7fffffffLL) return -1;
    const bool xk4 = Fk % 4 == 0 && ldk % 4 == 0 && dctr_aligned16(xk);
    if (!xk4 && Fk * DW_TR > 128 * DW_XKN) return -2;          // (the x_k rows must allow 16-B loads for Fk > 32)
#define DW_LAUNCH(NHV, X4)                                                                                                       \
    do {                                                                                                                         \
        if (lds > 64 * 1024) {                                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void*)cin_dw_fused_kernel<TW, NHV, X4>,                                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
            if (e != hipSuccess) return (int)e;                                                                                  \
        }                                                                                                                        \
        hipLaunchKernelGGL((cin_dw_fused_kernel<TW, NHV, X4>), dim3((unsigned)blocks), dim3(256), lds, st, p);                    \
        return 0;                                                                                                                \
    } while (0)
#define DW_CASE(NHV)                                                                                                             \
    case NHV:                                                                                                                    \
        if (xk4) DW_LAUNCH(NHV, true);                                                                                           \
        else DW_LAUNCH(NHV, false);
    switch (H / 16) {
        DW_CASE(1) DW_CASE(2) DW_CASE(3) DW_CASE(4) DW_CASE(5) DW_CASE(6) DW_CASE(7) DW_CASE(8)
        default: return -1;
    }
#undef DW_LAUNCH
#undef DW_CASE
}

}  // namespace dctr_cinbwd
