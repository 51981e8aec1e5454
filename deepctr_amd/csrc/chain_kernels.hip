// dctr_embed_mlp_fwd, row-chained form: eligibility, the split of a call into the kernel's main and tail phases, argument
// marshalling.  Kernel: chain_device.h, instantiated per launch shape and layer widths in chain_kernels_r{2w8,2w4}_m*.hip.
#include "chain_device.h"

namespace dctr_chain {

static int n_cus() { return dctr_n_cus(); }

// units -> (M0, M1, M2) of an instantiated kernel, or false
static bool widths(const dctr_mlp_args_t* a, int shape, int* M) {
    if (a->n_layers != 2 && a->n_layers != 3) return false;
    const int u0 = a->units[0], u1 = a->units[1], u2 = a->n_layers == 3 ? a->units[2] : 0;
    if ((u0 != 128 && u0 != 256) || (u1 != 64 && u1 != 128) || (u2 != 0 && u2 != 64 && u2 != 128)) return false;
    if (shape == 128 && !(u0 == 256 && u1 == 128 && u2 == 64)) return false;     // the diagnostic shape: units 256-128-64 only
    M[0] = u0 / 64;
    M[1] = u1 / 64;
    M[2] = u2 / 64;
    return true;
}

// 1: the row-chained kernel can take this call (all of its rows); 0: not eligible
int eligible(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, bool forced) {
    const int E = g->uniform_dim;
    if (E != 16 && E != 32 && E != 8 && E != 4 && E != 64) return 0;
    if (g->any_hash || !a->has_head || a->save_acts != nullptr) return 0;
    if (g->ids_stride_b != 1) return 0;                     // (rows of the id matrix contiguous: chain_device.h, request_pair_ids)
    const bool expact = a->activation == DCTR_ACT_SIGMOID || a->activation == DCTR_ACT_TANH;
    if (a->activation != DCTR_ACT_RELU && a->activation != DCTR_ACT_LINEAR && !expact) return 0;
    int M[3];
    if (!widths(a, forced ? a->tile_rows : 0, M)) return 0;
    // sigmoid / tanh DNNs: the m42 kernels, fp32, embedding_dim 16 / 32, no folded CrossNet
    if (expact && !(M[0] == 4 && M[1] == 2 && a->precision == 0 && a->tile_rows != 128 && (E == 16 || E == 32) && a->cross_layers == 0)) return 0;
    // embedding_dim 64 (four k-blocks per field): the m42 kernels, fp32, ReLU / linear, no folded CrossNet
    if (E == 64 && !(M[0] == 4 && M[1] == 2 && a->precision == 0 && a->tile_rows != 128 && a->cross_layers == 0)) return 0;
    if (a->precision != 0) return 0;                      // (fp32 only: the bf16x3 exploration was removed in round 6)
    // record-form tables (row_pitch 32, embedding_dim 16): the m42 kernels, fp32, ReLU / linear, no identity fields, no folded CrossNet
    if (g->any_pitch && !(E == 16 && M[0] == 4 && M[1] == 2 && a->precision == 0 && a->tile_rows != 128 && a->cross_layers == 0 && !expact &&
                          !g->any_identity))
        return 0;
    if (g->n_fields < 1 || g->n_fields > 64 || g->n_dense > 16 * MAX_DENSE_BLOCKS) return 0;
    // sequences pooled inside the pass (POOL): the m42 kernels, embedding_dim 16, ReLU / linear, plain tables, no identity fields; the
    // sequences' positions must fit the request slots of the SparseFeat k-blocks' steps (two per step, a pipeline of depth two)
    if (g->n_pools != 0 && !(g->n_pools >= 1 && g->n_pools <= 4 && g->pools != nullptr && E == 16 && M[0] == 4 && M[1] == 2 && a->tile_rows != 128 &&
                             a->cross_layers == 0 && !expact && !g->any_pitch && !g->any_identity && g->n_pools < g->n_fields &&
                             g->pool_pieces >= g->n_pools && g->pool_pieces + 2 <= g->n_fields - g->n_pools && g->pool_row0 >= 0 &&
                             g->pool_row0 + a->batch < (1LL << 31)))
        return 0;
    if (a->cross_layers > 0 && !(M[0] == 4 && M[1] == 2 && a->precision == 0 && a->tile_rows != 128 && E >= 16)) return 0;   // CROSS: the m42 kernels
    // embedding_dim 8 / 4 (several fields per k-block): the m42 kernels, fp32, no identity (pre-pooled) fields, <= 4 dense k-blocks
    if (E < 16 && !(M[0] == 4 && M[1] == 2 && a->precision == 0 && a->tile_rows != 128 && !g->any_identity)) return 0;
    if (lds_bytes(2, 8, g->n_dense > 0 ? g->n_dense : 0) + (a->cross_layers > 0 ? cross_lds_floats(a->in_dim) * sizeof(float) : 0) >
        160 * 1024)
        return 0;                                                                       // (dense staging area: <= 32 dense columns)
    if (g->n_dense > 0 && (g->dense_out_offset != g->n_fields * E || g->dense_copy_cols != g->n_dense)) return 0;
    if (a->in_dim != g->n_fields * E + (g->n_dense > 0 ? g->n_dense : 0)) return 0;
    if (g->ids_stride_f < 0 || g->ids_stride_b < 0) return 0;
    for (int l = 0; l < a->n_layers; ++l)
        if (!dctr_aligned16(a->kernels[l])) return 0;
    if (a->batch >= 0x7fffffffLL - 256 || (int64_t)a->in_dim * a->units[0] * 4 >= (1LL << 31)) return 0;
    // below 64 rows per CU the 32-row kernel (one workgroup per 32 rows, any number of them per CU) fills the chip better
    if (!forced && a->batch < (int64_t)64 * n_cus()) return 0;
    return 1;
}

// How a call of `batch` rows runs (shape 0 = auto): ONE launch of the <2, 8> kernel; its main phase takes the whole multiples of
// 256 rows x CUs (every CU the same number of 256-row passes), what is left — L < 256 x CUs rows — goes to the tail phase as 64-row
// units when that finishes sooner than one more (partly filled) round of 256-row passes.  Relative times (measured, C2): a 256-row
// pass = 1; a 64-row tail unit ~ TAIL_COST (one wave per SIMD: nothing hides a wave's request phase): L <= 64 x CUs x k rows cost
// k x TAIL_COST.  shape 256: everything in 256-row passes; shape 128: the <2, 4> kernel (its own launch).
constexpr double TAIL_COST = 0.35;      // measured: 55-57 us per unit round against 162 us per pass (profiles/r03_chain_lab_clock.log)
static void split(int64_t batch, int shape, int64_t* main_rows, int64_t* tail_rows) {
    const int64_t cus = n_cus();
    if (shape == 256 || shape == 128) {
        *main_rows = batch;
        *tail_rows = 0;
        return;
    }
    const int64_t full = batch / (256 * cus) * (256 * cus);
    const int64_t left = batch - full;
    const int64_t rounds = dctr_ceil_div(left, 64 * cus);
    if (left > 0 && TAIL_COST * (double)rounds < 1.0) {
        *main_rows = full;
        *tail_rows = left;
    } else {
        *main_rows = batch;
        *tail_rows = 0;
    }
}

// dctr_embed_mlp_fwd_plan: the phases of the ONE launch as (rows, batch rows per workgroup) entries — 256: main phase (or the
// forced <2, 8> shape), 128: the forced <2, 4> shape, 64: tail phase
int plan(int64_t batch, int shape, int64_t* rows, int32_t* rpw, int max) {
    int n = 0;
    auto push = [&](int64_t r, int w) {
        if (n < max) { rows[n] = r; rpw[n] = w; }
        ++n;
    };
    int64_t main_rows, tail_rows;
    split(batch, shape, &main_rows, &tail_rows);
    if (main_rows > 0) push(main_rows, shape == 128 ? 128 : 256);
    if (tail_rows > 0) push(tail_rows, 64);
    return n;
}

// The whole call on the row-chained kernel: one launch.
int launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, int fm_used, int lin_used, int shape, hipStream_t stream) {
    ChainParams p{};
    p.fields = g->fields;
    p.ids = g->ids;
    p.ids_stride_f = g->ids_stride_f;
    p.ids_stride_b = g->ids_stride_b;
    p.ids_is_i64 = g->ids_is_i64;
    p.n_fields = g->n_fields;
    p.n_dense = g->n_dense > 0 ? g->n_dense : 0;
    p.in_dim = a->in_dim;
    p.dense = g->dense;
    p.dense_stride = g->dense_stride;
    p.dense_lin_w = g->dense_lin_w;
    p.batch = a->batch;
    p.fm_logit = g->fm_logit;
    p.lin_logit = g->lin_logit;
    p.status = g->status;
    p.fm_used = fm_used;
    p.lin_used = lin_used;
    int M[3] = {0, 0, 0};
    if (!widths(a, shape, M)) {
        dctr_set_error("embed_mlp_fwd(chain): layer widths not instantiated");
        return DCTR_E_UNSUPPORTED;
    }
    for (int l = 0; l < a->n_layers; ++l) {
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
        p.bn_scale[l] = a->bn_scale != nullptr ? a->bn_scale[l] : nullptr;
        p.bn_shift[l] = (a->bn_scale != nullptr && a->bn_shift != nullptr) ? a->bn_shift[l] : nullptr;
    }
    p.activation = a->activation;
    p.sigmoid_out = a->sigmoid_out;
    p.head_w = a->head_w;
    for (int i = 0; i < 4; ++i) p.add[i] = a->add[i];
    p.global_bias = a->global_bias;
    p.y = a->y;
    p.probe = a->probe;
    p.cross_w = a->cross_w;
    p.cross_b = a->cross_b;
    p.cross_head = a->cross_head;
    p.cross_const = a->cross_const;
    p.cross_layers = a->cross_layers;
    p.xv_off = (int32_t)(lds_bytes(2, 8, p.n_dense) / sizeof(float));
    const int E = g->uniform_dim;
    int64_t main_rows, tail_rows;
    split(p.batch, shape, &main_rows, &tail_rows);
    const int prows = shape == 128 ? 128 : 256;
    p.main_rows = main_rows;
    p.n_pass = (int)dctr_ceil_div(main_rows, (int64_t)prows);
    p.n_tail = (int)dctr_ceil_div(tail_rows, (int64_t)64);
    const int64_t slots = n_cus();
    const int64_t want = p.n_pass > p.n_tail ? p.n_pass : p.n_tail;
    const unsigned blocks = (unsigned)(want < slots ? want : slots);
    if (g->n_pools > 0) {
        p.pool = g->pools;
        p.pool_row0 = g->pool_row0;
        p.n_pool = g->n_pools;
        p.pool_flags = g->pool_flags;
        return launch_r2w8_m42p(p, E, M[2], blocks, stream);
    }
    if (g->any_pitch) return launch_r2w8_m42r(p, E, M[2], blocks, stream);
    if (a->cross_layers > 0) return launch_r2w8_m42x(p, E, M[2], blocks, stream);
    if (a->activation == DCTR_ACT_SIGMOID || a->activation == DCTR_ACT_TANH) return launch_r2w8_m42t(p, E, M[2], blocks, stream);
    if (E < 16) return launch_r2w8_m42q(p, E, M[2], blocks, stream);
    if (E == 64) return launch_r2w8_m42w(p, E, M[2], blocks, stream);
    if (shape == 128) return launch_r2w4_m42(p, E, M[2], blocks, stream);
    if (M[0] == 4) return M[1] == 2 ? launch_r2w8_m42(p, E, M[2], blocks, stream) : launch_r2w8_m41(p, E, M[2], blocks, stream);
    return M[1] == 2 ? launch_r2w8_m22(p, E, M[2], blocks, stream) : launch_r2w8_m21(p, E, M[2], blocks, stream);
}

}  // namespace dctr_chain
