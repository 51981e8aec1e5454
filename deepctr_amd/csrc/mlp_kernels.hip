// adjacent to the hot path — DNN.call (+ Dense(1,use_bias=False) head + add_func + PredictionLayer.call)
// reference deepctr/layers/core.py:189-208, :250-259, layers/utils.py:328-333; and, fused in front of it
// (dctr_embed_mlp_fwd), the embedding gather + concat + linear + FM of embed_kernels.hip.
//
// One kernel runs the WHOLE multilayer perceptron for a tile of 16*RT batch rows: activations never leave LDS
// between layers, and the head (Dense(1) + linear/FM logits + global bias + sigmoid) is the epilogue.  Replaces,
// per layer, the reference's tensordot + bias_add + activation (+ Dice) kernels, then Dense, Add, bias_add, sigmoid.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact fp32, see mfma_tile.h).  A workgroup is 8 waves; every wave owns a
// 16*TPW-column slice of the layer output for all RT row tiles, walks K, and keeps its B (weight) fragments in a
// three-stage REGISTER pipeline fed by raw buffer loads (zero VALU per load, hardware bounds check for the K tail);
// A fragments come from the LDS tile, stored column-permuted so a lane reads its k-steps with ds_read_b128.
//
// What bounds it (scripts/mlp_lab.cpp, profiles/): the weights (603 KB for 429-256-128-64) are L2-resident, but a
// CU streams them at only ~16-21 B/clk, so with 16 rows per workgroup (RT = 1) the kernel is weight-stream-bound;
// RT = 2 halves the stream per row and brings layer 0 to ~80 % of the MFMA rate.  To still have two workgroups
// per CU at RT = 2 (one's gather / staging / barriers run under the other's MFMAs) the layer-0 input tile is built
// in two K-halves (k_split): half the LDS, accumulators persist in registers, <= 128 VGPRs.
// This file: argument checking, tile-shape selection, launch.  Kernel: mlp_device.h, instantiated per RT in
// mlp_kernels_rt{1,2,4}.hip.  fp32 MFMA floor for 429-256-128-64: 301.7 kFLOP/sample -> 7.9 us per 4096 rows.
#include "mlp_device.h"

using namespace dctr_mlp;

namespace dctr_stream {
int try_launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, int fm_used, int lin_used, bool forced,
               hipStream_t stream, int* rc);   // stream_kernels.hip
}
namespace dctr_chain {
int eligible(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, bool forced);   // chain_kernels.hip
int launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, int fm_used, int lin_used, int shape, hipStream_t stream);
int plan(int64_t batch, int shape, int64_t* rows, int32_t* rpw, int max);
}
namespace dctr_stream {
int eligible(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, bool forced);
}
namespace dctr_mlp {
size_t layered_workspace_bytes(const dctr_mlp_args_t* a);                               // mlp_kernels_layered.hip
int launch_layered(const dctr_mlp_args_t* a, hipStream_t stream, int (*head)(const dctr_mlp_args_t*, void*));
}

namespace {

// LDS row stride for a given layer-0 split (0 = none): pad64(widest tile) + 4
int mlp_lda(const dctr_mlp_args_t* a, int k_split) {
    const int kpad = (a->in_dim + 63) & ~63;
    int w = k_split > 0 ? (k_split > kpad - k_split ? k_split : kpad - k_split) : a->in_dim;
    for (int l = 0; l < a->n_layers; ++l) w = a->units[l] > w ? a->units[l] : w;
    return ((w + 63) & ~63) + 4;
}

// cross_bytes: the folded CrossNet's share ([CROSS_NV][rows] dot products + 8 constants + [CROSS_NV][pad64(in_dim)] vectors), 0 = none
size_t mlp_lds_bytes(int rows, int lda, size_t cross_bytes = 0) {
    return ((size_t)2 * rows * lda + 2 * rows) * sizeof(float) + cross_bytes;
}

constexpr size_t LDS_PER_CU = 160 * 1024;

// the 16-row workgroups of mlp_wide_kernel (the smallest tile any one-launch form has) hold two tiles of the widest layer
bool widest_layer_fits_lds(const dctr_mlp_args_t* a) {
    int wmax = 0;
    for (int l = 0; l < a->n_layers; ++l) wmax = a->units[l] > wmax ? a->units[l] : wmax;
    return mlp_lds_bytes(16, (((wmax > 64 ? wmax : 64) + 63) & ~63) + 4) <= LDS_PER_CU;
}

}  // namespace

// activations live in LDS — except in DNNs with a layer wider than the LDS tile holds (> 1,216 units), which run layer by layer through
// two activation buffers
extern "C" size_t dctr_mlp_workspace_bytes(const dctr_mlp_args_t* a) {
    if (a == nullptr || a->in_dim < 1) return 0;
    if (a->precision != 0 || a->n_layers < 1 || a->n_layers > MAX_LAYERS || a->units == nullptr) return 0;
    return widest_layer_fits_lds(a) ? 0 : dctr_mlp::layered_workspace_bytes(a);
}

static thread_local int g_last_fwd_kernel = -1;   // DCTR_FWD_KERNEL_* of this thread's last dctr_embed_mlp_fwd launch

static int mlp_launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* ga, int fm_used, int lin_used, void* stream, bool dry = false);
static int mlp_head_launch(const dctr_mlp_args_t* a, void* stream) { return mlp_launch(a, nullptr, 0, 0, stream); }

// dry: every check of a launch, no launch (dctr_mlp_fwd_supported / dctr_embed_mlp_fwd_supported)
static int mlp_launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* ga, int fm_used, int lin_used, void* stream, bool dry) {
    DCTR_REQUIRE(a == nullptr || a->precision == 0, DCTR_E_UNSUPPORTED,
                 "mlp_fwd: precision %d (0 = fp32 is the only arithmetic of this library)", a != nullptr ? a->precision : 0);
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "mlp_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->in_dim >= 1 && a->n_layers >= 0 && a->n_layers <= MAX_LAYERS, DCTR_E_DIM,
                 "mlp_fwd: bad sizes (batch=%lld in_dim=%d layers=%d, max %d layers)", (long long)a->batch, a->in_dim,
                 a->n_layers, MAX_LAYERS);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE((ga != nullptr || a->x) && a->y, DCTR_E_NULL, "mlp_fwd: null x / y");
    DCTR_REQUIRE(a->n_layers == 0 || (a->units && a->kernels && a->biases), DCTR_E_NULL, "mlp_fwd: null layer arrays");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_DICE, DCTR_E_ENUM, "mlp_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(!a->has_head || a->head_w, DCTR_E_NULL, "mlp_fwd: has_head without head_w");
    if (a->activation == DCTR_ACT_DICE && a->n_layers > 0)
        DCTR_REQUIRE(a->dice_alpha && a->dice_mean && a->dice_var, DCTR_E_NULL, "mlp_fwd: dice without parameters");
    MlpParams p{};
    p.x = a->x;
    p.batch = a->batch;
    p.x_stride = a->x_stride;
    p.in_dim = a->in_dim;
    p.n_layers = a->n_layers;
    for (int l = 0; l < a->n_layers; ++l) {
        DCTR_REQUIRE(a->units[l] >= 1, DCTR_E_DIM, "mlp_fwd: units[%d]=%d", l, a->units[l]);
        DCTR_REQUIRE(a->kernels[l] != nullptr, DCTR_E_NULL, "mlp_fwd: kernels[%d] null", l);
        DCTR_REQUIRE(dctr_aligned16(a->kernels[l]), DCTR_E_ALIGN, "mlp_fwd: kernels[%d] not 16-B aligned", l);
        p.units[l] = a->units[l];
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
        if (a->activation == DCTR_ACT_DICE) {
            DCTR_REQUIRE(a->dice_alpha[l] && a->dice_mean[l] && a->dice_var[l], DCTR_E_NULL, "mlp_fwd: dice[%d] null", l);
            p.dice_alpha[l] = a->dice_alpha[l];
            p.dice_mean[l] = a->dice_mean[l];
            p.dice_var[l] = a->dice_var[l];
        }
    }
    for (int l = 0; l < a->n_layers; ++l) p.save[l] = a->save_acts != nullptr ? a->save_acts[l] : nullptr;
    DCTR_REQUIRE((a->bn_scale == nullptr) == (a->bn_shift == nullptr), DCTR_E_NULL, "mlp_fwd: bn_scale and bn_shift go together");
    for (int l = 0; l < a->n_layers; ++l) {
        p.bn_scale[l] = a->bn_scale != nullptr ? a->bn_scale[l] : nullptr;
        p.bn_shift[l] = a->bn_shift != nullptr ? a->bn_shift[l] : nullptr;
        DCTR_REQUIRE((p.bn_scale[l] == nullptr) == (p.bn_shift[l] == nullptr), DCTR_E_NULL, "mlp_fwd: bn_scale[%d] / bn_shift[%d]", l, l);
    }
    p.probe = a->probe;
    p.dice_eps = a->dice_eps;
    p.activation = a->activation;
    p.has_head = a->has_head;
    p.sigmoid_out = a->sigmoid_out;
    p.head_w = a->head_w;
    for (int i = 0; i < 4; ++i) p.add[i] = a->add[i];
    p.global_bias = a->global_bias;
    p.y = a->y;
    p.y_stride = a->y_stride;
    const bool cross = a->cross_layers > 0;
    if (cross) {
        DCTR_REQUIRE(a->cross_layers <= CROSS_MAXL, DCTR_E_UNSUPPORTED, "mlp_fwd: cross_layers %d (the folded vector CrossNet takes <= %d)",
                     a->cross_layers, CROSS_MAXL);
        DCTR_REQUIRE(a->cross_w && a->cross_b && a->cross_head, DCTR_E_NULL, "mlp_fwd: cross_layers without cross_w / cross_b / cross_head");
        DCTR_REQUIRE(a->has_head, DCTR_E_DIM, "mlp_fwd: the folded CrossNet adds the cross branch's logit to the head (has_head)");
        DCTR_REQUIRE(a->save_acts == nullptr && a->precision == 0, DCTR_E_UNSUPPORTED, "mlp_fwd: cross_layers with save_acts");
        p.cross_w = a->cross_w;
        p.cross_b = a->cross_b;
        p.cross_head = a->cross_head;
        p.cross_const = a->cross_const;
        p.cross_layers = a->cross_layers;
    } else {
        DCTR_REQUIRE(a->cross_layers == 0, DCTR_E_DIM, "mlp_fwd: cross_layers %d", a->cross_layers);
    }
    FusedGather fg{};
    int lpr = 0;
    if (ga != nullptr) {
        DCTR_REQUIRE(ga->batch == a->batch, DCTR_E_DIM, "embed_mlp_fwd: gather batch %lld != mlp batch %lld",
                     (long long)ga->batch, (long long)a->batch);
        DCTR_REQUIRE(ga->n_fields >= 1 && ga->fields && ga->ids, DCTR_E_NULL, "embed_mlp_fwd: needs >= 1 gather field");
        DCTR_REQUIRE(ga->all_dim4 && ga->max_dim >= 1 && ga->max_dim <= 64, DCTR_E_UNSUPPORTED,
                     "embed_mlp_fwd: fused path needs every embedding_dim %% 4 == 0 and <= 64 (got max_dim %d, all_dim4 %d)",
                     ga->max_dim, ga->all_dim4);
        DCTR_REQUIRE(ga->n_dense == 0 || (ga->dense && ga->dense_stride >= ga->n_dense), DCTR_E_NULL,
                     "embed_mlp_fwd: bad dense matrix");
        DCTR_REQUIRE(ga->dense_copy_cols >= 0 && ga->dense_copy_cols <= ga->n_dense, DCTR_E_DIM,
                     "embed_mlp_fwd: dense_copy_cols outside [0, n_dense]");
        DCTR_REQUIRE(ga->split_col >= 0 && ga->split_col % 64 == 0 && ga->split_field >= 0 && ga->split_field <= ga->n_fields,
                     DCTR_E_DIM, "embed_mlp_fwd: split_col %d (multiple of 64) / split_field %d", ga->split_col, ga->split_field);
        static_cast<dctr_gather_fm_args_t&>(fg.g) = *ga;
        fg.g.fm_logit_used = fm_used;
        fg.g.lin_logit_used = lin_used;
        lpr = 4;                                           // lanes per sample row: 4 / 8 / 16 (dims <= 16 / 32 / 64)
        while (lpr * 4 < ga->max_dim) lpr <<= 1;
        fg.lpr = lpr;
    }
    // floats of the gather partial sums parked in the second activation tile
    auto red_floats = [lpr](int rows) -> size_t {
        if (lpr == 0) return 0;
        const int spw = 64 / lpr;
        return (size_t)NWAVE * (spw >= rows ? 1 : rows / spw) * 6 * 64;
    };

    // fused launches with >= 64 rows per CU of an eligible model: the row-chained kernel (chain_kernels.hip; tile_rows 256 /
    // 128 force one of its launch shapes, also for small launches)
    if (ga != nullptr && (a->tile_rows == 0 || a->tile_rows == 256 || a->tile_rows == 128)) {
        const bool forced = a->tile_rows != 0;
        const int ok = dctr_chain::eligible(a, ga, forced);
        DCTR_REQUIRE(!forced || ok, DCTR_E_UNSUPPORTED,
                     "embed_mlp_fwd: tile_rows %d (row-chained kernel) needs uniform embedding_dim 4 / 8 / 16 / 32 / 64, instantiated units, a head", a->tile_rows);
        if (ok) {
            if (dry) return DCTR_OK;
            const int rc = dctr_chain::launch(a, ga, fm_used, lin_used, a->tile_rows, (hipStream_t)stream);
            if (rc == DCTR_OK) g_last_fwd_kernel = DCTR_FWD_KERNEL_CHAIN;
            return rc;
        }
    }
    // sequences pooled inside the launch exist in the row-chained kernel only
    DCTR_REQUIRE(ga == nullptr || ga->n_pools == 0, DCTR_E_UNSUPPORTED,
                 "embed_mlp_fwd: n_pools %d: in-launch sequence pooling exists for row-chained launches (>= 64 rows per CU, uniform_dim 16, DNN of the "
                 "256-128-x family, sum / mean, pool_pieces + 2 <= SparseFeat fields); pre-pool with dctr_embed_pool (identity fields) otherwise",
                 ga != nullptr ? ga->n_pools : 0);
    // fused launches with >= 64 rows per CU (or tile_rows == 64): the streaming kernel, when the model is eligible
    if (ga != nullptr && (a->tile_rows == 0 || a->tile_rows == 64)) {
        int rc = DCTR_OK;
        if (dry && dctr_stream::eligible(a, ga, a->tile_rows == 64)) return DCTR_OK;
        if (!dry && dctr_stream::try_launch(a, ga, fm_used, lin_used, a->tile_rows == 64, (hipStream_t)stream, &rc)) {
            if (rc == DCTR_OK) g_last_fwd_kernel = DCTR_FWD_KERNEL_STREAM;
            return rc;
        }
    }

    // rows per workgroup.  auto: 16 while that still gives every CU a workgroup (latency), else 32
    int rt = a->tile_rows / 16;
    DCTR_REQUIRE(a->tile_rows == 0 || ((rt == 1 || rt == 2 || rt == 4) && a->tile_rows % 16 == 0), DCTR_E_DIM,
                 "mlp_fwd: tile_rows %d (0, 16, 32 or 64; 128 / 256 with dctr_embed_mlp_fwd)", a->tile_rows);
    if (rt == 0) rt = a->batch > 16 * 2 * 256 ? 2 : 1;
    if (ga != nullptr && rt > 2) rt = 2;
    if (ga != nullptr && lpr == 16) rt = 1;                  // embedding_dim > 32: see produce_chunk

    // layer-0 K split candidate: the caller's field boundary (fused) or the middle of the padded row (plain), usable
    // when every wave owns at most one wave-tile of layer 0 (its accumulators stay in registers across the halves)
    const int kpad = (a->in_dim + 63) & ~63;
    int split = 0;
    if (a->n_layers >= 1) {
        const int n0 = a->units[0];
        const bool wide = n0 % 32 == 0 && n0 >= 32 * NWAVE;
        const bool one_tile = (wide ? n0 / 32 : (n0 + 15) / 16) <= NWAVE;
        int cand = ga != nullptr ? ga->split_col : ((kpad / 2 + 63) & ~63);
        if (ga != nullptr && cand > 0 && ga->n_dense > 0 && ga->dense_out_offset >= 0 && ga->dense_copy_cols > 0 &&
            ga->dense_out_offset < cand)
            cand = 0;
        if (one_tile && cand > 0 && cand < kpad) split = cand;
    }
    // keep the split only where it raises the number of co-resident workgroups (max 2: 16 waves per CU), or makes
    // the tile fit at all; shrink the tile when nothing fits
    int lda = 0;
    for (;; rt >>= 1) {
        const int rows = 16 * rt;
        const int lda_full = mlp_lda(a, 0), lda_split = split ? mlp_lda(a, split) : lda_full;
        const size_t xb = cross ? ((size_t)CROSS_NV * rows + 8 + (size_t)CROSS_NV * ((a->in_dim + 63) & ~63)) * sizeof(float) : 0;
        auto fits = [&](int ld) { return mlp_lds_bytes(rows, ld, xb) <= LDS_PER_CU && red_floats(rows) <= (size_t)rows * ld; };
        auto per_cu = [&](int ld) { return fits(ld) ? (mlp_lds_bytes(rows, ld, xb) * 2 <= LDS_PER_CU ? 2 : 1) : 0; };
        const int full = per_cu(lda_full), half = split ? per_cu(lda_split) : 0;
        if (half > full) { p.k_split = split; lda = lda_split; break; }
        if (full > 0) { p.k_split = 0; lda = lda_full; break; }
        if (rt == 1) break;
    }
    if (lda == 0 && ga != nullptr && a->tile_rows == 0 && dctr_chain::eligible(a, ga, true)) {
        // a DNN input too wide for the LDS tile (e.g. 33 fields of embedding_dim 64: [16, 2128] x 2 floats) on a launch below 64 rows
        // per CU: the row-chained kernel streams layer 0 by k-block and holds no input tile — its tail phase takes the rows
        if (dry) return DCTR_OK;
        const int rc = dctr_chain::launch(a, ga, fm_used, lin_used, 0, (hipStream_t)stream);
        if (rc == DCTR_OK) g_last_fwd_kernel = DCTR_FWD_KERNEL_CHAIN;
        return rc;
    }
    if (lda == 0 && ga == nullptr && !cross && a->precision == 0 && (a->n_layers >= 1 || a->has_head)) {
        if (a->n_layers >= 1 && !widest_layer_fits_lds(a)) {
            // a layer too wide for any LDS tile (> 1,216 units): layer by layer through the caller's workspace (mlp_kernels_layered.hip)
            if (dry) return DCTR_OK;
            return dctr_mlp::launch_layered(a, (hipStream_t)stream, mlp_head_launch);
        }
        // an input row wider than the tile, read from HBM: 16-row workgroups that walk it in K chunks (mlp_kernels_wide.hip)
        if (dry) return DCTR_OK;
        int wmax = 0;
        for (int l = 0; l < a->n_layers; ++l) wmax = a->units[l] > wmax ? a->units[l] : wmax;
        int kc = 512;
        while (kc > 64 && mlp_lds_bytes(16, (((kc > wmax ? kc : wmax) + 63) & ~63) + 4) > LDS_PER_CU) kc >>= 1;
        const int wl = (((kc > wmax ? kc : wmax) + 63) & ~63) + 4;
        DCTR_REQUIRE(mlp_lds_bytes(16, wl) <= LDS_PER_CU, DCTR_E_UNSUPPORTED, "mlp_fwd: a layer of %d units does not fit the 160 KiB LDS tile", wmax);
        p.lda = wl;
        p.k_split = 0;
        const int64_t wblocks = dctr_ceil_div(a->batch, (int64_t)16);
        DCTR_REQUIRE(wblocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: batch too large");
        return launch_wide(p, kc, (unsigned)wblocks, mlp_lds_bytes(16, wl), (hipStream_t)stream);
    }
    DCTR_REQUIRE(lda > 0, DCTR_E_UNSUPPORTED,
                 "mlp_fwd: layer widths do not fit the 160 KiB LDS tile (or are too small for the gather partial sums)%s",
                 ga != nullptr ? "; dctr_embed_gather_fm + dctr_mlp_fwd take this shape" : "");
    if (dry) return DCTR_OK;
    p.lda = lda;
    const int rows = 16 * rt;
    const size_t lds = mlp_lds_bytes(rows, lda, cross ? ((size_t)CROSS_NV * rows + 8 + (size_t)CROSS_NV * ((a->in_dim + 63) & ~63)) * sizeof(float) : 0);
    const int64_t blocks = dctr_ceil_div(a->batch, (int64_t)rows);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "mlp_fwd: batch too large");
    if (ga != nullptr) g_last_fwd_kernel = DCTR_FWD_KERNEL_TILE;
    // 16-row workgroups that are alone on their CU (at most one per CU: a launch of <= 16 x CUs rows, or tile_rows 16): the kernel whose
    // waves pull their weight slices by LDS-DMA (mlp_device.h: mlp_ring_kernel) — every layer width a multiple of 16, the input tile
    // whole (no K split), tiles + 8 x 12 KiB of rings inside 160 KiB.  Same bits as the tile kernel.
    if (rt == 1 && p.k_split == 0 && a->n_layers >= 1 && (a->tile_rows == 16 || blocks <= (int64_t)dctr_n_cus())) {
        bool ok = true;
        for (int l = 0; l < a->n_layers; ++l) ok = ok && a->units[l] % 16 == 0;
        const size_t ring_off = (lds + 1023) & ~(size_t)1023;
        const size_t total = ring_off + (size_t)NWAVE * RING_WAVE_F * sizeof(float);
        if (ok && total <= LDS_PER_CU) return launch_rt1_ring(p, fg, (unsigned)blocks, total, (int)(ring_off / sizeof(float)), (hipStream_t)stream);
    }
    if (rt == 1) return launch_rt1(p, fg, (unsigned)blocks, lds, (hipStream_t)stream);
    if (rt == 2) return launch_rt2(p, fg, (unsigned)blocks, lds, (hipStream_t)stream);
    return launch_rt4(p, fg, (unsigned)blocks, lds, (hipStream_t)stream);
}

extern "C" int dctr_embed_mlp_fwd_last_kernel(void) { return g_last_fwd_kernel; }

// the row-independent constants of the folded vector CrossNet (dctr_mlp_args_t.cross_const): one wave
namespace {
__global__ __launch_bounds__(64) void cross_consts_kernel(const float* w, const float* b, const float* head, int L, int d, float* out) {
    float cst[CROSS_NV];
    cross_constants(w, b, head, L, d, (int)threadIdx.x, cst);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int v = 0; v < CROSS_NV; ++v) out[v] = cst[v];
    }
}
}  // namespace

extern "C" int dctr_crossnet_fold_consts(const float* cross_w, const float* cross_b, const float* cross_head, int32_t layers, int32_t dim,
                                         float* consts, void* stream) {
    DCTR_REQUIRE(cross_w && cross_b && cross_head && consts, DCTR_E_NULL, "crossnet_fold_consts: null pointer");
    DCTR_REQUIRE(layers >= 1 && layers <= CROSS_MAXL && dim >= 1, DCTR_E_DIM, "crossnet_fold_consts: layers %d (1 .. %d), dim %d", layers,
                 CROSS_MAXL, dim);
    DCTR_LAUNCH(cross_consts_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cross_w, cross_b, cross_head, layers, dim, consts);
    return dctr_launch_status("dctr_crossnet_fold_consts");
}

extern "C" int dctr_mlp_fwd(const dctr_mlp_args_t* a, void* stream) { return mlp_launch(a, nullptr, 0, 0, stream); }

// Would the launch be taken?  Every argument check and kernel-shape decision of dctr_mlp_fwd (g == NULL) / dctr_embed_mlp_fwd, no launch.
// 1 = yes (dctr_mlp_fwd: given the workspace dctr_mlp_workspace_bytes() asks for), 0 = no — dctr_last_error() says why.  The host asks
// instead of re-deriving the library's limits (LDS tile sizes, instantiated shapes).
extern "C" int dctr_mlp_fwd_supported(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int32_t add_fm_logit, int32_t add_lin_logit) {
    if (m == nullptr) return 0;
    if (g != nullptr && !(m->has_head || (!add_fm_logit && !add_lin_logit))) return 0;
    return mlp_launch(m, g, add_fm_logit ? 1 : 0, add_lin_logit ? 1 : 0, nullptr, true) == DCTR_OK ? 1 : 0;
}

extern "C" int dctr_embed_mlp_fwd(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int32_t add_fm_logit,
                                  int32_t add_lin_logit, void* stream) {
    DCTR_REQUIRE(g != nullptr && m != nullptr, DCTR_E_NULL, "embed_mlp_fwd: null args");
    DCTR_REQUIRE(m->has_head || (!add_fm_logit && !add_lin_logit), DCTR_E_DIM,
                 "embed_mlp_fwd: the gather logits can only be added by the fused head");
    return mlp_launch(m, g, add_fm_logit ? 1 : 0, add_lin_logit ? 1 : 0, stream);
}

extern "C" int dctr_embed_mlp_fwd_plan(const dctr_gather_fm_args_t* g, const dctr_mlp_args_t* m, int64_t* rows, int32_t* kernel,
                                       int32_t* rows_per_workgroup, int32_t max) {
    DCTR_REQUIRE(g != nullptr && m != nullptr && rows != nullptr && kernel != nullptr && rows_per_workgroup != nullptr && max >= 1,
                 DCTR_E_NULL, "embed_mlp_fwd_plan: null args");
    if (m->batch <= 0) return 0;
    if (m->tile_rows == 0 || m->tile_rows == 256 || m->tile_rows == 128) {
        if (dctr_chain::eligible(m, g, m->tile_rows != 0)) {
            const int n = dctr_chain::plan(m->batch, m->tile_rows, rows, rows_per_workgroup, max);
            for (int i = 0; i < n && i < max; ++i) kernel[i] = DCTR_FWD_KERNEL_CHAIN;
            return n;
        }
    }
    rows[0] = m->batch;
    if ((m->tile_rows == 0 || m->tile_rows == 64) && dctr_stream::eligible(m, g, m->tile_rows == 64)) {
        kernel[0] = DCTR_FWD_KERNEL_STREAM;
        rows_per_workgroup[0] = 64;
    } else {
        kernel[0] = DCTR_FWD_KERNEL_TILE;
        rows_per_workgroup[0] = m->tile_rows;                 // 0: 16 or 32, chosen from the batch size and the LDS the widths need
    }
    return 1;
}
