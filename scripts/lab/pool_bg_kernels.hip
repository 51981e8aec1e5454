// dctr_embed_pool_background — SequencePoolingLayer.call (reference deepctr/layers/sequence.py:76-106 over the rows of
// varlen_embedding_lookup, inputs.py:120-158) by a kernel that is built to run BESIDE a persistent MFMA kernel, not before it.
//
// Why.  The row-chained forward (chain_device.h) holds every CU with one workgroup of 8 waves x 234 (allocated: 240) registers: 480 of a
// SIMD lane's 512 registers, 147 of 160 KiB of LDS.  dctr_embed_pool's kernels (42 - 135 registers) cannot share a CU with it, so pooled
// VarLenSparseFeat ran as a pre-pass in FRONT of every fused launch: 2 x 22 us in front of 365 us at C2 + two T = 20 features,
// 131,072 rows per call (profiles/r06c_varlen_kernel_trace.csv) — the whole difference between 0.65 and 0.73 of the f32-MFMA peak.
// The pooling itself is bound by L2 request latency, not by anything the MFMA kernel uses up.  This kernel needs <= 32 registers per
// lane and no LDS: one of its waves fits into what the row-chained kernel leaves on every SIMD, so a launch of it on a SECOND stream
// pools the NEXT span's sequences while the current span's MFMAs run (host: engine.py, the span loop of predict).  It is slower
// than dctr_embed_pool when it has the part to itself (4 positions in flight per lane instead of 16) — it is not meant to.
//
// Same arithmetic as pool_fast_kernel / pool_kernel in the same order over t: acc += row[t] * mask[t], mean: / (len or count + 1e-8);
// max: max(acc, row[t] - (1 - mask[t]) * 1e9).  Masked positions of sum / mean are skipped instead of multiplied by 0: acc + (+-0) = acc
// for every finite row (acc is never -0: it starts at +0), so the results are bit-identical to dctr_embed_pool's.
// Takes: embedding_dim % 4 == 0 and <= 64, int32 / int64 ids, no per-position weights, no hashing; tables < 4 GiB.
#include "dctr_common.h"
#include "embed_device.h"

namespace {



// U = positions in flight per lane: 2 (ids: 2, offsets: 2, rows: 8, linear entries: 2, accumulators: 6, bookkeeping: ~8 registers)
template <int LPR, bool I64>
__global__ __launch_bounds__(256) void pool_bg_kernel(dctr_pool_args_t a) {
    constexpr int SPW = 64 / LPR, U = 2;
    const int lane = threadIdx.x & 63;
    const int s = lane / LPR, q = lane % LPR;
    const int T = a.maxlen;
    const bool by_len = a.length != nullptr;
    const bool is_max = a.combiner == DCTR_POOL_MAX;
    const bool colok = 4 * q < a.dim;
    const uint32_t vocab = (uint32_t)a.vocab, dim = (uint32_t)a.dim;
    const bool has_lin = a.lin_table != nullptr;
    int oor = 0;
    const int n_groups = (int)((a.batch + SPW - 1) / SPW);
    for (int grp = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); grp < n_groups; grp += (int)gridDim.x * 4) {
        const int b = grp * SPW + s;
        const bool valid = b < (int)a.batch;
        const int bb = valid ? b : (int)a.batch - 1;
        const int len = by_len ? a.length[bb] : 0;
        float acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = is_max ? -INFINITY : 0.f;
        float lacc = is_max ? -INFINITY : 0.f;
        float cnt = 0.f;
        const char* idp = reinterpret_cast<const char*>(a.idx) + (size_t)bb * (size_t)a.idx_stride * (I64 ? 8 : 4);
#pragma unroll 1
        for (int t0 = 0; t0 < T; t0 += U) {
            uint32_t row[U];
            bool hi_bad[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u < T ? t0 + u : T - 1;
                if constexpr (I64) {
                    const uint2 w = reinterpret_cast<const uint2*>(idp)[t];
                    row[u] = w.x;
                    hi_bad[u] = w.y != 0u;                    // negative or >= 2^32
                } else {
                    row[u] = reinterpret_cast<const uint32_t*>(idp)[t];
                    hi_bad[u] = (int32_t)row[u] < 0;
                }
            }
            float v[U][4], lv[U];
            bool mk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = valid && t0 + u < T;
                const bool ok = !hi_bad[u] && row[u] < vocab;
                if (live && !ok) oor = 1;
                mk[u] = live && (by_len ? (t0 + u < len) : (row[u] != 0u || hi_bad[u]));
                // sum / mean: a masked position adds +-0 (skipped); max: it enters as row - 1e9 (sequence.py:97) and must be read
                const bool need = live && ok && (is_max || mk[u]);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[u][c] = 0.f;
                lv[u] = 0.f;
                if (need && colok) load_vec<4>(a.table + (row[u] * dim + 4u * (uint32_t)q), v[u]);
                if (need && q == 0 && has_lin) lv[u] = a.lin_table[row[u]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!(valid && t0 + u < T)) continue;
                const float m = mk[u] ? 1.f : 0.f;
                cnt += m;
                if (is_max) {
                    const float pen = (1.f - m) * 1e9f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaxf(acc[c], v[u][c] * 1.f - pen);
                    lacc = fmaxf(lacc, lv[u] * 1.f - pen);
                } else if (mk[u]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] += v[u][c];
                    lacc += lv[u];
                }
            }
        }
        if (a.combiner == DCTR_POOL_MEAN) {
            const float denom = (by_len ? (float)len : cnt) + 1e-8f;      // layers/sequence.py:65,103
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = acc[c] / denom;
            lacc = lacc / denom;
        }
        if (valid) {
            if (colok) store_vec<4>(a.out + (size_t)b * (size_t)a.out_stride + 4 * q, acc);
            if (q == 0 && a.lin_out != nullptr) a.lin_out[b] = lacc;
        }
    }
    if (a.status != nullptr && __any(oor) && lane == 0) atomicOr(a.status, (int)DCTR_STATUS_INDEX_OOR);
}

}  // namespace

extern "C" int dctr_embed_pool_background_supported(const dctr_pool_args_t* a) {
    if (a == nullptr) return 0;
    return (a->dim >= 4 && a->dim <= 64 && a->dim % 4 == 0 && a->weight == nullptr && a->hash_mode == 0 && a->maxlen >= 1 &&
            a->vocab >= 1 && (uint64_t)a->vocab * (uint64_t)a->dim * 4ull < (1ull << 32) && a->out_stride % 4 == 0 &&
            a->combiner >= DCTR_POOL_SUM && a->combiner <= DCTR_POOL_MAX) ? 1 : 0;
}

extern "C" int dctr_embed_pool_background(const dctr_pool_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "embed_pool_background: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->maxlen >= 1 && a->dim >= 1, DCTR_E_DIM, "embed_pool_background: bad sizes B=%lld T=%d dim=%d",
                 (long long)a->batch, a->maxlen, a->dim);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->idx && a->table && a->out, DCTR_E_NULL, "embed_pool_background: null pointer");
    DCTR_REQUIRE(dctr_embed_pool_background_supported(a), DCTR_E_UNSUPPORTED,
                 "embed_pool_background: takes embedding_dim %% 4 == 0 and <= 64, no per-position weights, no hashing, tables < 4 GiB, "
                 "out_stride %% 4 == 0 (dctr_embed_pool takes the rest)");
    DCTR_REQUIRE(dctr_aligned16(a->table) && dctr_aligned16(a->out), DCTR_E_ALIGN, "embed_pool_background: table / out not 16-B aligned");
    int lpr = 4;
    while (lpr * 4 < a->dim) lpr <<= 1;
    // a persistent grid: at most four workgroups per CU are ever resident beside the MFMA kernel (one wave per SIMD), and a grid-stride
    // loop keeps the dispatcher from queueing tens of thousands of workgroups behind it
    const int64_t groups = dctr_ceil_div(a->batch, (int64_t)(64 / lpr));
    int64_t blocks = dctr_ceil_div(groups, (int64_t)4);
    const int64_t cap = (int64_t)dctr_n_cus() * 4;
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
#define CALL_BG(L)                                                                                   \
    do {                                                                                             \
        if (a->idx_is_i64) DCTR_LAUNCH((pool_bg_kernel<L, true>), dim3((unsigned)blocks), dim3(256), 0, st, *a);   \
        else DCTR_LAUNCH((pool_bg_kernel<L, false>), dim3((unsigned)blocks), dim3(256), 0, st, *a);                \
    } while (0)
    switch (lpr) {
        case 4: CALL_BG(4); break;
        case 8: CALL_BG(8); break;
        default: CALL_BG(16); break;
    }
#undef CALL_BG
    return dctr_launch_status("dctr_embed_pool_background");
}
