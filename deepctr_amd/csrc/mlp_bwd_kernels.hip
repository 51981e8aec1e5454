// Backward chain of the DNN (dctr_mlp_bwd; reference deepctr/layers/core.py:189-208 under Keras autodiff):
//     dH_{l-1} = dZ_l W_l^T,   dZ_{l-1} = dH_{l-1} .* act'(h_{l-1}),   ...,   dX = dZ_0 W_0^T
// is a multilayer perceptron over the row tile of dZ_{L-1} with the TRANSPOSED weights and the activation's derivative (from the
// forward's saved outputs) as the epilogue, so it runs on the forward's whole-MLP machinery (mlp_device.h: activations in LDS between
// layers, weights streamed through the three-stage register pipeline) as ONE launch: the dX GEMM + act' kernel pair per layer
// (2 L launches, ~25 + 7 us each on a 4096-row step whatever their size) is gone; every dZ_l also goes to HBM for dW_l = X_l^T dZ_l.
#include "mlp_device.h"

namespace dctr_mlp {

template <int RT>
__global__ __launch_bounds__(NTHR, RT <= 2 ? 4 : 2) void mlp_bwd_kernel(MlpParams p) {
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in = smem;
    float* out = smem + ROWS * p.lda;
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    const Chunk ck{0, pad64(p.in_dim) / 4, 0, 0, true, true};
    stage_x_chunk<RT>(p, in, b0, ck);
    int K = p.in_dim;
    for (int l = 0; l < p.n_layers; ++l) {
        const int N = p.units[l];
        layer_dispatch<ACT_BWD, RT>(p, l, in, out, K, N);
        __syncthreads();
        float* t = in;
        in = out;
        out = t;
        K = N;
    }
    if (p.y != nullptr) {
        for (int i = threadIdx.x; i < ROWS * K; i += NTHR) {
            const int r = i / K, c = i % K;
            const int64_t b = b0 + r;
            if (b < p.batch) p.y[b * p.y_stride + c] = in[r * p.lda + lds_pos(c, pad64(K) / 4)];
        }
    }
}

static int chain_lda(int in_dim, int n_layers, const int32_t* units_fwd) {
    int w = in_dim;
    for (int l = 0; l < n_layers; ++l) w = units_fwd[l] > w ? units_fwd[l] : w;
    return ((w + 63) & ~63) + 4;
}

bool bwd_chain_fits(int in_dim, int n_layers, const int32_t* units_fwd) {
    return n_layers >= 1 && n_layers <= MAX_LAYERS && (size_t)2 * 16 * chain_lda(in_dim, n_layers, units_fwd) * sizeof(float) <= 160 * 1024;
}

template <int RT>
static int launch_rt(const MlpParams& p, unsigned blocks, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_bwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    DCTR_LAUNCH(mlp_bwd_kernel<RT>, dim3(blocks), dim3(NTHR), lds, stream, p);
    return dctr_launch_status("dctr_mlp_bwd (chain)");
}

int launch_bwd_chain(hipStream_t stream, int64_t batch, int in_dim, int n_layers, const int32_t* units_fwd, const float* const* Wt,
                     const float* const* acts, int activation, const float* dz_in, float* const* dz_out, float* dx, int64_t dx_stride) {
    // chain layer j multiplies by W_{L-1-j}^T: input width units_fwd[L-1-j], output width units_fwd[L-2-j] (in_dim for the last)
    const int L = n_layers;
    const int n_chain = dx != nullptr ? L : L - 1;
    if (n_chain <= 0 || batch <= 0) return DCTR_OK;
    MlpParams p{};
    p.x = dz_in;
    p.batch = batch;
    p.x_stride = units_fwd[L - 1];
    p.in_dim = units_fwd[L - 1];
    p.n_layers = n_chain;
    p.deriv_act = activation;
    p.activation = ACT_BWD;
    for (int j = 0; j < n_chain; ++j) {
        const int lf = L - 1 - j;                       // forward layer whose weights this chain layer transposes
        p.units[j] = lf > 0 ? units_fwd[lf - 1] : in_dim;
        p.W[j] = Wt[lf];
        DCTR_REQUIRE(p.W[j] != nullptr && dctr_aligned16(p.W[j]), DCTR_E_ALIGN, "mlp_bwd chain: transposed kernel %d null / not 16-B aligned", lf);
        p.deriv_h[j] = lf > 0 ? acts[lf - 1] : nullptr;
        p.save[j] = lf > 0 ? dz_out[lf - 1] : nullptr;
    }
    p.y = dx;                                           // written by the last chain layer when the input gradient is wanted
    p.y_stride = dx_stride;
    p.lda = chain_lda(in_dim, n_layers, units_fwd);
    // rows per workgroup: 16 while that still gives every CU a workgroup, 32 / 64 for long batches (each weight fragment then
    // feeds 2 / 4 row tiles)
    int rt = batch > 65536 ? 4 : (batch > 16 * 2 * 256 ? 2 : 1);
    while (rt > 1 && (size_t)2 * 16 * rt * p.lda * sizeof(float) > 160 * 1024) rt >>= 1;
    const size_t lds = (size_t)2 * 16 * rt * p.lda * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "mlp_bwd chain: layer widths do not fit the LDS tile");
    const int64_t blocks = dctr_ceil_div(batch, (int64_t)(16 * rt));
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_bwd chain: batch too large");
    if (rt == 1) return launch_rt<1>(p, (unsigned)blocks, lds, stream);
    if (rt == 2) return launch_rt<2>(p, (unsigned)blocks, lds, stream);
    return launch_rt<4>(p, (unsigned)blocks, lds, stream);
}

}  // namespace dctr_mlp
