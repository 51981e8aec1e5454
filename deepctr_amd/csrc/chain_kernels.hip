// dctr_embed_mlp_fwd, row-chained form: eligibility, the split of a launch into launch shapes, argument marshalling.
// Kernel: chain_device.h, instantiated per launch shape in chain_kernels_r{2w8,2w4,1w4}.hip.
#include "chain_device.h"

namespace dctr_chain {

static int n_cus() { return dctr_n_cus(); }

// 1: the row-chained kernel can take this call (all of its rows); 0: not eligible
int eligible(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, bool forced) {
    const int E = g->uniform_dim;
    if (E != 16 && E != 32) return 0;
    if (g->any_hash || !a->has_head || a->save_acts != nullptr) return 0;
    if ((a->activation != DCTR_ACT_RELU && a->activation != DCTR_ACT_LINEAR) || a->bn_scale != nullptr) return 0;
    if (a->n_layers != 3 || a->units[0] != 256 || a->units[1] != 128 || a->units[2] != 64) return 0;
    if (g->n_fields < 1 || g->n_fields > 64 || g->n_dense > 16 * MAX_DENSE_BLOCKS) return 0;
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

static int launch_shape(int rt, int nw, const ChainParams& p, int E, hipStream_t stream) {
    const int64_t n_pass = dctr_ceil_div(p.batch, (int64_t)(16 * rt * nw));
    ChainParams q = p;
    q.n_pass = (int)n_pass;
#ifdef DCTR_CHAIN_W4X2
    const int64_t slots = (rt == 2 && nw == 4) ? 2 * (int64_t)n_cus() : n_cus();
#else
    const int64_t slots = n_cus();
#endif
    const unsigned blocks = (unsigned)(n_pass < slots ? n_pass : slots);
    if (rt == 2 && nw == 8) return launch_r2w8(q, E, blocks, stream);
    if (rt == 2 && nw == 4) return launch_r2w4(q, E, blocks, stream);
    return launch_r1w4(q, E, blocks, stream);
}

// rows [0, n) of p as their own launch (pointers advanced)
static ChainParams slice(const ChainParams& p, int64_t r0, int64_t n) {
    ChainParams q = p;
    q.ids = reinterpret_cast<const char*>(p.ids) + r0 * p.ids_stride_b * (p.ids_is_i64 ? 8 : 4);
    q.dense = p.dense != nullptr ? p.dense + r0 * p.dense_stride : nullptr;
    q.fm_logit = p.fm_logit != nullptr ? p.fm_logit + r0 : nullptr;
    q.lin_logit = p.lin_logit != nullptr ? p.lin_logit + r0 : nullptr;
    for (int i = 0; i < 4; ++i) q.add[i] = p.add[i] != nullptr ? p.add[i] + r0 : nullptr;
    q.y = p.y + r0;
    q.batch = n;
    return q;
}

// The launches of a call of `batch` rows: (rows, batch rows per workgroup = 256 / 128 / 64 <-> shapes <2,8> / <2,4> / <1,4>).
// shape 256 / 128: that shape for all rows (forced); 0: whole multiples of 256 rows x CUs as <2, 8> launches (every CU the same
// number of passes), the rest cut into the shapes that finish it soonest.  Relative pass times (measured, C2): a <2, 8> pass of
// 256 rows = 1 (162 us); a <2, 4> pass of 128 rows ~ 0.60 (one wave per SIMD: the matrix pipe is not shared, but nothing hides a
// wave's request phase either); a <1, 4> pass of 64 rows ~ 0.39.
int plan(int64_t batch, int shape, int64_t* rows, int32_t* rpw, int max) {
    int n = 0;
    auto push = [&](int64_t r, int w) {
        if (n < max) { rows[n] = r; rpw[n] = w; }
        ++n;
    };
    if (shape == 256 || shape == 128) {
        push(batch, shape);
        return n;
    }
    const int64_t cus = n_cus();
    const int64_t full = batch / (256 * cus) * (256 * cus);
    if (full > 0) push(full, 256);
    int64_t left = batch - full;
    while (left > 0) {
        // cost of finishing `left` rows with one shape (in <2, 8> pass times); the cheapest one takes as many whole rounds
        // over the CUs as it has, the loop goes on with what is left
        const double c28 = (double)dctr_ceil_div(left, 256 * cus);
        const double c24 = 0.60 * (double)dctr_ceil_div(left, 128 * cus);
        const double c14 = 0.39 * (double)dctr_ceil_div(left, 64 * cus);
        int w = 256;
        if (c24 < c28 && c24 <= c14) w = 128;
        else if (c14 < c28 && c14 < c24) w = 64;
        const int64_t round_rows = (int64_t)w * cus;
        int64_t take = left / round_rows * round_rows;
        if (take == 0) take = left;
        push(take, w);
        left -= take;
    }
    return n;
}

// The whole call on the row-chained kernel, launch by launch of plan().
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
    for (int l = 0; l < 3; ++l) {
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
    }
    p.activation = a->activation;
    p.sigmoid_out = a->sigmoid_out;
    p.head_w = a->head_w;
    for (int i = 0; i < 4; ++i) p.add[i] = a->add[i];
    p.global_bias = a->global_bias;
    p.y = a->y;
    p.probe = a->probe;
    const int E = g->uniform_dim;
    int64_t rows[16];
    int32_t rpw[16];
    const int n = plan(p.batch, shape, rows, rpw, 16);
    int64_t r0 = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = launch_shape(rpw[i] == 64 ? 1 : 2, rpw[i] == 256 ? 8 : 4, slice(p, r0, rows[i]), E, stream);
        if (rc != DCTR_OK) return rc;
        r0 += rows[i];
    }
    return DCTR_OK;
}

}  // namespace dctr_chain
