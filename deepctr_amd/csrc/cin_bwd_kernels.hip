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
        const int h4 = H >> 2;
        for (int idx = threadIdx.x; idx < RB * h4; idx += 256) {
            const int m = idx / h4, c = idx - m * h4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + m < p.rows) v = *reinterpret_cast<const float4*>(p.dpre + (r0 + m) * H + 4 * c);
            *reinterpret_cast<float4*>(dps + m * p.sd + 4 * c) = v;
        }
        for (int idx = threadIdx.x; idx < RB * F0; idx += 256) {
            const int m = idx / F0, i = idx - m * F0;
            x0s[m * p.s0 + i] = r0 + m < p.rows ? p.x0t[(r0 + m) * F0 + i] : 0.f;
            ax0[m * p.s0 + i] = 0.f;
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

static size_t lds_bytes(int F0, int Fk, int H, int& sd, int& s0, int& sk) {
    sd = H + 4;
    s0 = F0 | 1;
    sk = Fk | 1;
    return (size_t)RB * (sd + 2 * s0 + sk) * sizeof(float);
}

// shapes the fused kernel takes (else the caller keeps the GEMM + contraction pair)
bool dz_fused_ok(int F0, int Fk, int H, const float* dpre, const float* W) {
    int sd, s0, sk;
    return H % 16 == 0 && H >= 16 && H <= 16 * MAXKS && F0 >= 1 && Fk >= 1 && dctr_aligned16(dpre) && dctr_aligned16(W) &&
           lds_bytes(F0, Fk, H, sd, s0, sk) <= 160 * 1024;
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

}  // namespace dctr_cinbwd
