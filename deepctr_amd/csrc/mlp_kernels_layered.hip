// dctr_mlp_fwd, layer-by-layer form — DNN.call (reference deepctr/layers/core.py:189-208) for layer widths the one-launch kernels
// cannot hold: every one-launch form keeps a 16-row tile of the WIDEST layer in LDS twice (2 x 16 x pad64(width) x 4 B <= 160 KiB,
// i.e. <= 1,216 units); the reference's DNN takes any `hidden_units`.  Here a layer is
//     Z = X W          dctr_gemm::sgemm (gemm_kernels.hip: f32 MFMA, any m / n / k), rows in chunks that fit the caller's workspace
//     H = act(bn(Z + b))   one elementwise launch in place (BatchNormalization inference form, relu / sigmoid / tanh / linear / Dice
//                          with the stored statistics: activation.py:59-64), optionally copied to save_acts[l]
// and the head (Dense(1, no bias) + add[] + global bias + PredictionLayer) goes back through dctr_mlp_fwd with no hidden layer, whose
// 16-row workgroups walk an input row of any width in K chunks (mlp_kernels_wide.hip).  Same arithmetic as the one-launch kernels
// (exact fp32 products, fp32 accumulation); the summation order over k is the GEMM's.
#include "dctr_common.h"
#include "dctr_gemm.h"
#include "mlp_device.h"

namespace dctr_mlp {

namespace {

struct EpiParams {
    float* z;               // [rows, n] with row stride ldz: pre-activations in, activations out
    int64_t ldz;
    float* save;            // NULL or [rows, n] contiguous: a copy of the activations (training forward)
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    const float* dice_alpha;
    const float* dice_mean;
    const float* dice_var;
    float dice_eps;
    int64_t rows;
    int n;
    int act;
};

__global__ __launch_bounds__(256) void layer_epilogue_kernel(EpiParams p) {
    const int64_t total = p.rows * p.n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / p.n;
        const int c = (int)(i - r * p.n);
        float z = p.z[r * p.ldz + c] + (p.bias != nullptr ? p.bias[c] : 0.f);
        if (p.bn_scale != nullptr) z = fmaf(z, p.bn_scale[c], p.bn_shift[c]);        // keras: x * inv + (beta - mean * inv)
        float v;
        switch (p.act) {
            case DCTR_ACT_RELU: v = fmaxf(z, 0.f); break;
            case DCTR_ACT_SIGMOID: v = dctr::sigmoidf_(z); break;
            case DCTR_ACT_TANH: v = tanhf(z); break;
            case DCTR_ACT_DICE: v = dctr::dice_act(z, p.dice_alpha[c], p.dice_mean[c], p.dice_var[c], p.dice_eps); break;
            default: v = z; break;
        }
        p.z[r * p.ldz + c] = v;
        if (p.save != nullptr) p.save[i] = v;
    }
}

inline int pad4(int n) { return (n + 3) & ~3; }

}  // namespace

// widest hidden layer, padded to whole float4 rows
static int widest(const dctr_mlp_args_t* a) {
    int w = 1;
    for (int l = 0; l < a->n_layers; ++l) w = a->units[l] > w ? a->units[l] : w;
    return pad4(w);
}

// two ping-pong activation buffers of [rows, widest] floats; `rows` rows per chunk (the whole batch up to 65,536 rows — 1 GiB at a
// 2,048-wide layer — is what the query asks for; any multiple of 64 rows >= 64 works)
size_t layered_workspace_bytes(const dctr_mlp_args_t* a) {
    const int64_t rows = a->batch < 65536 ? ((a->batch + 63) & ~(int64_t)63) : 65536;
    return (size_t)2 * (size_t)(rows < 64 ? 64 : rows) * (size_t)widest(a) * sizeof(float);
}

// head: the caller's dctr_mlp_fwd with n_layers == 0 (mlp_kernels.hip passes its own launch function)
int launch_layered(const dctr_mlp_args_t* a, hipStream_t stream, int (*head)(const dctr_mlp_args_t*, void*)) {
    const int w = widest(a);
    DCTR_REQUIRE(a->workspace != nullptr && dctr_aligned16(a->workspace), DCTR_E_UNSUPPORTED,
                 "mlp_fwd: a layer of more than 1,216 units runs layer by layer and needs `workspace` (dctr_mlp_workspace_bytes() = %zu "
                 "bytes, 16-B aligned)", layered_workspace_bytes(a));
    int64_t chunk = (int64_t)(a->workspace_bytes / ((size_t)2 * w * sizeof(float)));
    chunk &= ~(int64_t)63;
    DCTR_REQUIRE(chunk >= 64, DCTR_E_UNSUPPORTED, "mlp_fwd: workspace of %zu bytes holds fewer than 64 rows of two %d-wide layers (%zu bytes)",
                 a->workspace_bytes, w, (size_t)2 * 64 * w * sizeof(float));
    DCTR_REQUIRE(a->has_head || a->y_stride >= a->units[a->n_layers - 1], DCTR_E_DIM, "mlp_fwd: y_stride %lld < units[last] %d",
                 (long long)a->y_stride, a->units[a->n_layers - 1]);
    float* buf[2] = {reinterpret_cast<float*>(a->workspace), reinterpret_cast<float*>(a->workspace) + chunk * w};
    for (int64_t r0 = 0; r0 < a->batch; r0 += chunk) {
        const int64_t rows = a->batch - r0 < chunk ? a->batch - r0 : chunk;
        DCTR_REQUIRE(rows <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: chunk too large");
        const float* in = a->x + r0 * a->x_stride;
        int64_t ld_in = a->x_stride;
        int K = a->in_dim;
        for (int l = 0; l < a->n_layers; ++l) {
            const int n = a->units[l];
            const bool to_y = !a->has_head && l == a->n_layers - 1;
            float* out = to_y ? a->y + r0 * a->y_stride : buf[l & 1];
            const int64_t ld_out = to_y ? a->y_stride : w;
            DCTR_REQUIRE(ld_in <= 0x7fffffffLL && ld_out <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: row stride too large");
            // row-major out [rows, n] = in [rows, K] W [K, n]  <=>  column-major out^T (n x rows) = W^T (n x K) in^T (K x rows)
            int rc = dctr_gemm::sgemm(stream, dctr_gemm::OP_N, dctr_gemm::OP_N, n, (int)rows, K, a->kernels[l], n, in, (int)ld_in, 0.f, out,
                                      (int)ld_out);
            if (rc != DCTR_OK) return rc;
            EpiParams e{};
            e.z = out;
            e.ldz = ld_out;
            e.save = (a->save_acts != nullptr && a->save_acts[l] != nullptr) ? a->save_acts[l] + r0 * n : nullptr;
            e.bias = a->biases[l];
            e.bn_scale = (a->bn_scale != nullptr) ? a->bn_scale[l] : nullptr;
            e.bn_shift = (a->bn_shift != nullptr) ? a->bn_shift[l] : nullptr;
            if (a->activation == DCTR_ACT_DICE) {
                e.dice_alpha = a->dice_alpha[l];
                e.dice_mean = a->dice_mean[l];
                e.dice_var = a->dice_var[l];
            }
            e.dice_eps = a->dice_eps;
            e.rows = rows;
            e.n = n;
            e.act = a->activation;
            const int64_t blocks = dctr_ceil_div(rows * n, (int64_t)(256 * 4));
            DCTR_LAUNCH(layer_epilogue_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0, stream, e);
            rc = dctr_launch_status("dctr_mlp_fwd(layered)");
            if (rc != DCTR_OK) return rc;
            in = out;
            ld_in = ld_out;
            K = n;
        }
        if (a->has_head) {
            dctr_mlp_args_t h = *a;
            h.x = in;
            h.x_stride = ld_in;
            h.in_dim = K;
            h.n_layers = 0;
            h.batch = rows;
            h.y = a->y + r0;
            h.save_acts = nullptr;
            h.bn_scale = nullptr;
            h.bn_shift = nullptr;
            h.workspace = nullptr;
            h.workspace_bytes = 0;
            h.tile_rows = 0;
            for (int i = 0; i < 4; ++i) h.add[i] = a->add[i] != nullptr ? a->add[i] + r0 : nullptr;
            const int rc = head(&h, (void*)stream);
            if (rc != DCTR_OK) return rc;
        }
    }
    return DCTR_OK;
}

}  // namespace dctr_mlp
