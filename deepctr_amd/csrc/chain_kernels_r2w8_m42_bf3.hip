// EXPLORATORY, never the default: the row-chained kernel with its MFMAs on v_mfma_f32_16x16x16_bf16 and every product split in three
// (hi hi + hi lo + lo hi, fp32 accumulation; dctr_mlp_args_t.precision == 1), DNN 256-128-64 only.  What it is for: with the matrix pipe
// ~5x faster the kernel shows what its gather side sustains (exact fp32 caps the whole forward at 0.126 of the HBM
// roofline).  Not bit-comparable with the fp32 kernels: |error| ~ 2^-16 per product; checked against the same 1e-4 oracle bar.
#include "chain_device.h"

namespace dctr_chain {

// W [K, N] fp32 (Keras layout) -> the packed images the BF3 kernel streams through its LDS ring (chain_device.h):
//   layer 0, k-block c (16 rows):  [k-slot g][M-group mg][M-tile mt][lane j] x 16 B = {hi(k0,k1), hi(k2,k3), lo(k0,k1), lo(k2,k3)} of
//                                  feature 64 mg + 4 j + mt, k_e = 16 c + 4 g + e (rows >= K: zeros)
//   layers >= 1, sub-block (mg, mg1): [g][input tile mt][output tile mt1][j] x 16 B of output feature 64 mg1 + 4 j + mt1,
//                                  k_e = 64 mg + 16 g + 4 e + mt   (the features register e of accumulator tile mt holds in lane group g)
__device__ __forceinline__ uint4 bf3_pack4(float k0, float k1, float k2, float k3) {
    uint2 h, l;
    bf3_split(f32x4{k0, k1, k2, k3}, h, l);
    return uint4{h.x, h.y, l.x, l.y};
}
__global__ __launch_bounds__(256) void chain_pack_bf3_kernel(const float* __restrict__ W, int K, int N, int layer0, uint4* __restrict__ out,
                                                             int64_t n_words) {
    const int M = N / 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * 256) {
        float k[4];
        if (layer0) {
            const int per = 4 * M * 4 * 16;                      // 16-B words per k-block
            const int c = (int)(i / per), r = (int)(i % per);
            const int j = r & 15, mt = (r >> 4) & 3, mg = (r >> 6) % M, g = (r >> 6) / M;
            const int n = 64 * mg + 4 * j + mt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * c + 4 * g + e;
                k[e] = row < K ? W[(int64_t)row * N + n] : 0.f;
            }
        } else {
            const int per = 4 * 4 * 4 * 16;                      // one 64 x 64 sub-block
            const int sbk = (int)(i / per), r = (int)(i % per);
            const int mg = sbk / M, mg1 = sbk % M;
            const int j = r & 15, mt1 = (r >> 4) & 3, mt = (r >> 6) & 3, g = r >> 8;
            const int n = 64 * mg1 + 4 * j + mt1;
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = W[(int64_t)(64 * mg + 16 * g + 4 * e + mt) * N + n];
        }
        out[i] = bf3_pack4(k[0], k[1], k[2], k[3]);
    }
}

// bytes of the packed images of a 256-128-64 DNN with `in_dim` inputs
size_t bf3_workspace_bytes(int in_dim) {
    const size_t nb = (size_t)(in_dim + 15) / 16;
    return nb * 16384 + (size_t)256 * 128 * 4 + (size_t)128 * 64 * 4;
}

template <int EB, bool I64>
static int launch_bf3_one(const ChainParams& p, unsigned blocks, hipStream_t stream) {
    const size_t lds = lds_bytes(2, 8, p.n_dense);
    static thread_local size_t granted[DCTR_MAX_DEVICES] = {0};
    hipError_t e = dctr_grant_lds((const void*)chain_kernel<2, 8, EB, I64, 4, 2, 1, true, true>, lds, granted);
    if (e != hipSuccess) {
        dctr_set_error("embed_mlp_fwd(chain, bf16x3): cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
        return (int)e;
    }
    DCTR_LAUNCH((chain_kernel<2, 8, EB, I64, 4, 2, 1, true, true>), dim3(blocks), dim3(512), lds, stream, p);
    return dctr_launch_status("dctr_embed_mlp_fwd(chain, bf16x3)");
}

// packs the three weight matrices into `ws` (bf3_workspace_bytes; skipped when the caller vouches `ws` still holds them: precision 2)
// and launches; p.W[] are the fp32 Keras matrices on entry
int launch_r2w8_m42_bf3(const ChainParams& p0, int E, void* ws, bool pack, unsigned blocks, hipStream_t stream) {
    ChainParams p = p0;
    const int nb = (p.in_dim + 15) / 16;
    char* w = static_cast<char*>(ws);
    const int64_t n0 = (int64_t)nb * 1024, n1 = 256 * 128 / 4, n2 = 128 * 64 / 4;      // 16-B words per layer
    auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)); };
    if (pack) {
        hipLaunchKernelGGL(chain_pack_bf3_kernel, grid(n0), dim3(256), 0, stream, p0.W[0], p.in_dim, 256, 1, reinterpret_cast<uint4*>(w), n0);
        hipLaunchKernelGGL(chain_pack_bf3_kernel, grid(n1), dim3(256), 0, stream, p0.W[1], 256, 128, 0, reinterpret_cast<uint4*>(w + n0 * 16), n1);
        hipLaunchKernelGGL(chain_pack_bf3_kernel, grid(n2), dim3(256), 0, stream, p0.W[2], 128, 64, 0, reinterpret_cast<uint4*>(w + (n0 + n1) * 16), n2);
    }
    p.W[0] = reinterpret_cast<const float*>(w);
    p.W[1] = reinterpret_cast<const float*>(w + n0 * 16);
    p.W[2] = reinterpret_cast<const float*>(w + (n0 + n1) * 16);
    if (E == 16) return p.ids_is_i64 ? launch_bf3_one<1, true>(p, blocks, stream) : launch_bf3_one<1, false>(p, blocks, stream);
    return p.ids_is_i64 ? launch_bf3_one<2, true>(p, blocks, stream) : launch_bf3_one<2, false>(p, blocks, stream);
}

}  // namespace dctr_chain
