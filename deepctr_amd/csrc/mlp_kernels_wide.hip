// dctr_mlp_fwd for a DNN input wider than the LDS tile of mlp_kernel holds (e.g. 25 fields of embedding_dim 64 = 1600+ columns in
// front of a 200-80 DNN — DNN.call of the reference takes any width: deepctr/layers/core.py:189-208).  16 batch rows per workgroup;
// the input row is walked in K chunks of `kc` columns staged from HBM into the first LDS tile, and every wave keeps the accumulators
// of ALL the layer-0 wave-tiles it owns (up to WIDE_MAXT of 16 columns) in registers across the chunks — the generalisation of
// mlp_kernel's two-half k_split.  Layer 0's epilogue, the further layers and the head are the tile kernel's own code (tile_epilogue,
// layer_dispatch); with no hidden layer the head's dot product is accumulated chunk by chunk.
// The same MFMAs on the same operands in the same k order as mlp_kernel: a model whose input fits either kernel gets the same bits.
#include "mlp_device.h"

namespace dctr_mlp {

constexpr int WIDE_MAXT = 8;      // wave-tiles of layer 0 per wave and pass: 8 waves x 8 x 16 = 1024 columns per pass over the input

template <int ACT>
__device__ __forceinline__ void wide_epilogue(const MlpParams& p, float* out, int N, int n_tiles, int t0, int wave,
                                              const dctr::f32x4 (&acc)[WIDE_MAXT][1][1]) {
#pragma unroll
    for (int t = 0; t < WIDE_MAXT; ++t) {
        const int wt = t0 + wave + t * NWAVE;
        if (wt < n_tiles) tile_epilogue<1, ACT, 1>(p, 0, out, N, 16 * wt, acc[t]);
    }
}

__global__ __launch_bounds__(NTHR, 1) void mlp_wide_kernel(MlpParams p, int kc) {
    constexpr int RT = 1, ROWS = 16, SD = StageDepth<1>::value;
    using dctr::f32x4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + ROWS * p.lda;
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kpad = pad64(p.in_dim);
    float* in = buf0;
    float* out = buf1;
    int K = p.in_dim;
    int l = 0;
    float head_acc = 0.f;                                     // (n_layers == 0: this lane's share of its row's head dot product)
    if (p.n_layers >= 1) {
        const int N = p.units[0], n_tiles = (N + 15) / 16;
        for (int t0 = 0; t0 < n_tiles; t0 += NWAVE * WIDE_MAXT) {
            f32x4 acc[WIDE_MAXT][1][1];
#pragma unroll
            for (int t = 0; t < WIDE_MAXT; ++t) acc[t][0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < kpad; c0 += kc) {
                const int cw = min(kc, kpad - c0);
                const Chunk ck{c0, cw / 4, 0, 0, c0 == 0, c0 + cw >= kpad};
                stage_x_chunk<RT>(p, buf0, b0, ck);           // (ends with a barrier)
                const int k_rows = min(p.in_dim - c0, cw);
                const float* W = p.W[0] + (size_t)c0 * N;
#pragma unroll
                for (int t = 0; t < WIDE_MAXT; ++t) {
                    const int wt = t0 + wave + t * NWAVE;
                    if (wt < n_tiles) tile_gemm_pipe<1, RT, SD>(buf0, p.lda, cw / 4, k_rows, W, N, 16 * wt, acc[t]);
                }
                __syncthreads();                              // buf0 is rebuilt by the next chunk
            }
            switch (p.activation) {
                case DCTR_ACT_RELU: wide_epilogue<DCTR_ACT_RELU>(p, buf1, N, n_tiles, t0, wave, acc); break;
                case DCTR_ACT_SIGMOID: wide_epilogue<DCTR_ACT_SIGMOID>(p, buf1, N, n_tiles, t0, wave, acc); break;
                case DCTR_ACT_TANH: wide_epilogue<DCTR_ACT_TANH>(p, buf1, N, n_tiles, t0, wave, acc); break;
                case DCTR_ACT_DICE: wide_epilogue<DCTR_ACT_DICE>(p, buf1, N, n_tiles, t0, wave, acc); break;
                default: wide_epilogue<DCTR_ACT_LINEAR>(p, buf1, N, n_tiles, t0, wave, acc); break;
            }
        }
        zero_k_padding<RT>(p, buf1, N);
        __syncthreads();
        in = buf1;
        out = buf0;
        K = N;
        l = 1;
    } else {
        // no hidden layer: y = x . head_w, the dot product taken chunk by chunk (16 lanes per row, as the head below)
        const int part = threadIdx.x & 15, row = threadIdx.x >> 4;
        for (int c0 = 0; c0 < kpad; c0 += kc) {
            const int cw = min(kc, kpad - c0);
            const Chunk ck{c0, cw / 4, 0, 0, c0 == 0, c0 + cw >= kpad};
            stage_x_chunk<RT>(p, buf0, b0, ck);
            const int k_rows = min(p.in_dim - c0, cw);
            if (row < ROWS)
                for (int n = part; n < k_rows; n += 16) head_acc = fmaf(buf0[row * p.lda + lds_pos(n, cw / 4)], p.head_w[c0 + n], head_acc);
            __syncthreads();
        }
    }
    for (; l < p.n_layers; ++l) {
        const int N = p.units[l];
        switch (p.activation) {
            case DCTR_ACT_RELU: layer_dispatch<DCTR_ACT_RELU, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_SIGMOID: layer_dispatch<DCTR_ACT_SIGMOID, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_TANH: layer_dispatch<DCTR_ACT_TANH, RT>(p, l, in, out, K, N); break;
            case DCTR_ACT_DICE: layer_dispatch<DCTR_ACT_DICE, RT>(p, l, in, out, K, N); break;
            default: layer_dispatch<DCTR_ACT_LINEAR, RT>(p, l, in, out, K, N); break;
        }
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        K = N;
    }
    if (p.has_head) {
        const int part = threadIdx.x & 15;
        const int KQh = pad64(K) / 4;
        for (int row = threadIdx.x >> 4; row < ROWS; row += NTHR / 16) {
            float acc = head_acc;
            if (p.n_layers >= 1)
                for (int n = part; n < K; n += 16) acc = fmaf(in[row * p.lda + lds_pos(n, KQh)], p.head_w[n], acc);
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
        }
    } else {
        for (int i = threadIdx.x; i < ROWS * K; i += NTHR) {
            const int r = i / K, c = i % K;
            const int64_t b = b0 + r;
            if (b < p.batch) p.y[b * p.y_stride + c] = in[r * p.lda + lds_pos(c, pad64(K) / 4)];
        }
    }
}

// kc: the chunk width (a multiple of 64); p.lda = pad64(max(kc, every layer width)) + 4
int launch_wide(const MlpParams& p, int kc, unsigned blocks, size_t lds, hipStream_t stream) {
    static thread_local size_t granted[DCTR_MAX_DEVICES] = {0};
    hipError_t e = dctr_grant_lds((const void*)mlp_wide_kernel, lds, granted);
    DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_fwd(wide): cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    DCTR_LAUNCH(mlp_wide_kernel, dim3(blocks), dim3(NTHR), lds, stream, p, kc);
    return dctr_launch_status("dctr_mlp_fwd(wide)");
}

}  // namespace dctr_mlp
