// a3-a8 — the HBM-bound core of the path: multi-table embedding gather fused with the DNN-input
// concat, the first-order (Linear) term and the FM second-order term; masked sequence pooling for
// VarLenSparseFeat; plain per-position lookup (DIN keys).
//
// Reference op sequence replaced (per batch, C2 = 26 sparse fields): 26 keras Embedding gathers
// (deepctr/inputs.py:101-117) + 26 one-wide gathers for the linear term (feature_column.py:171-210)
// + Concat/Flatten (layers/utils.py:336-346) + FM's 2 reductions over [B,F,E]
// (layers/interaction.py:588-604) + Linear's reduce_sum/tensordot (layers/utils.py:160-175).
//
// Lane layout (wave = 64): a table row of `dim` floats is read by LPR adjacent lanes, VEC floats
// each (VEC = 4 -> one global_load_dwordx4 per lane, a 64-B row = 4 lanes), so a wave-instruction
// fetches 64/LPR rows = 1 KiB of row data with every lane active.  Lane (s, q) owns chunk q of
// sample s and walks the FIELDS, so the FM sums over fields accumulate in registers with no
// cross-lane traffic; only the final sum over the embedding dimension crosses lanes (log2(LPR)
// DPP/shuffle steps).  Loads are issued in three phases per chunk of U fields (ids -> rows -> use)
// so U row reads per lane are in flight before the first use.
// Small batches are latency-, not bandwidth-bound (B=4096 x 26 rows = one 32-KiB burst per CU), so
// when the plain layout would give < 8 waves per CU the four waves of a workgroup split the
// FIELDS of the same 64/LPR samples (FSPLIT) and combine their partial sums through LDS.
#include <math.h>

#include "dctr_common.h"
#include "embed_device.h"

namespace {

// HASH = false builds carry no hashing code at all (the host knows whether any field has hash_mode != 0).
template <int VEC, int LPR, bool FSPLIT, bool HASH>
__global__ __launch_bounds__(256) void gather_fm_kernel(GatherParams p) {
    constexpr int SPW = 64 / LPR;  // samples per wave
    constexpr int DB = 4 * LPR;    // dense columns per pass (4 per lane)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane / LPR, q = lane % LPR;
    const int64_t b = FSPLIT ? (int64_t)blockIdx.x * SPW + s : ((int64_t)blockIdx.x * 4 + wave) * SPW + s;
    const bool valid = b < p.batch;
    const int64_t bb = valid ? b : 0;
    const int f_begin = FSPLIT ? wave : 0;
    const int f_step = FSPLIT ? 4 : 1;

    // dense block 0 is requested first (kernel-argument addresses only) and consumed last, so its
    // latency hides behind the id -> row chain.  With FSPLIT the dense passes go to the last waves,
    // which own the fewest fields.
    const bool dense0_mine = p.n_dense > 0 && (!FSPLIT || wave == 3);
    float dx[4], dw[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = q + m * LPR;
        const int kk = (dense0_mine && k < p.n_dense) ? k : 0;
        dx[m] = dense0_mine ? p.dense[bb * p.dense_stride + kk] : 0.f;
        dw[m] = (dense0_mine && p.dense_lin_w != nullptr) ? p.dense_lin_w[kk] : 0.f;
    }

    float sum[VEC], sq[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) sum[c] = sq[c] = 0.f;
    GatherAcc acc{0.f, 0};

    float* const out_row = p.dnn_in != nullptr ? p.dnn_in + (valid ? b : 0) * p.out_stride : nullptr;
    auto store = [out_row](int col, const float (&v)[VEC]) {
        if (out_row != nullptr) store_vec<VEC>(out_row + col, v);
    };
    gather_fields<VEC, LPR, HASH>(p, f_begin, f_step, p.n_fields, b, valid, q, sum, sq, acc, store);
    // embedding_dim > LPR * VEC (e.g. "auto" = 102 for a 1e5 vocabulary, feature_column.py:44-45): further passes over the fields for
    // elements [e0, e0 + LPR * VEC) of every row — FM is a sum over d, so a pass's share is closed before the next starts (host: no
    // FSPLIT then, a wave sees all the fields of its samples)
    float fm_more = 0.f;
    if constexpr (!FSPLIT) {
        for (int e0 = LPR * VEC; e0 < p.max_dim; e0 += LPR * VEC) {
            float s2[VEC], q2[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) s2[c] = q2[c] = 0.f;
            gather_fields<VEC, LPR, HASH>(p, f_begin, f_step, p.n_fields, b, valid, q, s2, q2, acc, store, e0);
#pragma unroll
            for (int c = 0; c < VEC; ++c) fm_more += s2[c] * s2[c] - q2[c];
        }
    }
    float lin = acc.lin;

    // dense features: passthrough into the concat + dense . Linear.kernel
    if (dense0_mine) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = q + m * LPR;
            if (valid && k < p.n_dense) {
                if (p.dnn_in != nullptr && p.dense_out_offset >= 0 && k < p.dense_copy_cols)
                    p.dnn_in[b * p.out_stride + p.dense_out_offset + k] = dx[m];
                lin = fmaf(dx[m], dw[m], lin);
            }
        }
    }
    for (int k0 = DB, kb = 1; k0 < p.n_dense; k0 += DB, ++kb) {
        if (FSPLIT && wave != 3 - (kb & 3)) continue;
        float x[4], w[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = k0 + q + m * LPR;
            const int kk = k < p.n_dense ? k : 0;
            x[m] = p.dense[bb * p.dense_stride + kk];
            w[m] = p.dense_lin_w != nullptr ? p.dense_lin_w[kk] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k = k0 + q + m * LPR;
            if (valid && k < p.n_dense) {
                if (p.dnn_in != nullptr && p.dense_out_offset >= 0 && k < p.dense_copy_cols)
                    p.dnn_in[b * p.out_stride + p.dense_out_offset + k] = x[m];
                lin = fmaf(x[m], w[m], lin);
            }
        }
    }

    if constexpr (FSPLIT) {
        __shared__ float red[3][2 * VEC + 1][64];
        if (wave > 0) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                red[wave - 1][c][lane] = sum[c];
                red[wave - 1][VEC + c][lane] = sq[c];
            }
            red[wave - 1][2 * VEC][lane] = lin;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    sum[c] += red[w][c][lane];
                    sq[c] += red[w][VEC + c][lane];
                }
                lin += red[w][2 * VEC][lane];
            }
        }
    }

    if (!FSPLIT || wave == 0) {
        float fm = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) fm += sum[c] * sum[c] - sq[c];
        fm = 0.5f * reduce_lpr<LPR>(fm + fm_more);
        lin = reduce_lpr<LPR>(lin);
        if (valid && q == 0) {
            if (p.fm_logit != nullptr) p.fm_logit[b] = fm;
            if (p.lin_logit != nullptr) p.lin_logit[b] = lin;
        }
    }
    if (p.status != nullptr && __any(acc.oor) && lane == 0) atomicOr(p.status, (int)DCTR_STATUS_INDEX_OOR);
}

// ---------------------------------------------------------------------------------------------------
// masked sequence pooling (VarLenSparseFeat)
// ---------------------------------------------------------------------------------------------------
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void pool_kernel(dctr_pool_args_t a) {
    constexpr int SPW = 64 / LPR;
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane / LPR, q = lane % LPR;
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * SPW + s;
    const bool valid = b < a.batch;
    const int T = a.maxlen;
    const bool by_len = a.length != nullptr;
    const int len = (by_len && valid) ? a.length[b] : 0;
    const float PAD = -4294967296.f;  // float32(-2**32 + 1), reference layers/sequence.py:171
    int oor = 0;

    // softmax statistics of the masked weights (WeightedSequenceLayer, weight_normalization=True)
    float wmax = -INFINITY, wden = 1.f;
    const bool wnorm = a.weight != nullptr && a.weight_norm;
    if (wnorm && valid) {
        for (int t = 0; t < T; ++t) {
            bool m;
            if (by_len) m = t < len;
            else m = resolve_row(read_id(a.idx, b * a.idx_stride + t, a.idx_is_i64), a.hash_mode, a.idx_is_i64,
                                 a.vocab) != 0;
            const float w = m ? a.weight[b * (int64_t)T + t] : PAD;
            wmax = fmaxf(wmax, w);
        }
        wden = 0.f;
        for (int t = 0; t < T; ++t) {
            bool m;
            if (by_len) m = t < len;
            else m = resolve_row(read_id(a.idx, b * a.idx_stride + t, a.idx_is_i64), a.hash_mode, a.idx_is_i64,
                                 a.vocab) != 0;
            const float w = m ? a.weight[b * (int64_t)T + t] : PAD;
            wden += expf(w - wmax);
        }
    }

    const bool is_max = a.combiner == DCTR_POOL_MAX;
    // rows wider than LPR * VEC elements (host: LPR = 64 then) are pooled in passes of that width
    for (int e0 = 0; e0 < a.dim; e0 += LPR * VEC) {
    const int qe = e0 + q * VEC;
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = is_max ? -INFINITY : 0.f;
    float lacc = is_max ? -INFINITY : 0.f;
    float cnt = 0.f;

    for (int t0 = 0; t0 < T; t0 += U) {
        int64_t row[U];
        bool ok[U];
        float mk[U], wt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            row[u] = 0;
            ok[u] = false;
            mk[u] = 0.f;
            wt[u] = 1.f;
            if (t < T && valid) {
                const int64_t r = resolve_row(read_id(a.idx, b * a.idx_stride + t, a.idx_is_i64), a.hash_mode,
                                              a.idx_is_i64, a.vocab);
                ok[u] = (uint64_t)r < (uint64_t)a.vocab;
                if (!ok[u]) oor = 1;
                row[u] = r;
                const bool m = by_len ? (t < len) : (r != 0);
                mk[u] = m ? 1.f : 0.f;
                if (a.weight != nullptr) {
                    const float w = a.weight[b * (int64_t)T + t];
                    wt[u] = wnorm ? expf((m ? w : PAD) - wmax) / wden : (m ? w : 0.f);
                }
            }
        }
        float v[U][VEC], lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[u][c] = 0.f;
            lv[u] = 0.f;
            if (ok[u] && qe < a.dim) load_vec<VEC>(a.table + row[u] * a.dim + qe, v[u]);
            if (ok[u] && qe == 0 && a.lin_table != nullptr) lv[u] = a.lin_table[row[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            if (t < T && valid) {
                cnt += mk[u];
                if (is_max) {
                    const float pen = (1.f - mk[u]) * 1e9f;  // layers/sequence.py:97
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc[c] = fmaxf(acc[c], v[u][c] * wt[u] - pen);
                    lacc = fmaxf(lacc, lv[u] * wt[u] - pen);
                } else {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc[c] += v[u][c] * wt[u] * mk[u];
                    lacc += lv[u] * wt[u] * mk[u];
                }
            }
        }
    }
    if (a.combiner == DCTR_POOL_MEAN) {
        const float denom = (by_len ? (float)len : cnt) + 1e-8f;  // layers/sequence.py:65,103
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = acc[c] / denom;
        lacc = lacc / denom;
    }
    if (valid) {
        if (qe < a.dim) store_vec<VEC>(a.out + b * a.out_stride + qe, acc);
        if (qe == 0 && a.lin_out != nullptr) a.lin_out[b] = lacc;
    }
    }
    if (a.status != nullptr && __any(oor) && lane == 0) atomicOr(a.status, (int)DCTR_STATUS_INDEX_OOR);
}

// The common case of the same pooling — int32 ids in rows of T <= 32 (T % 4 == 0, 16-B aligned), embedding_dim 16 or 32, no
// per-position weights, no hashing — with every memory round trip taken once: the T ids of a sample arrive as 16-B chunks spread
// over the sample's LPR lanes and reach the row loads through the LDS crossbar (the general kernel reads each id in each of the
// LPR lanes, 8 at a time), and up to 16 row loads per lane are in flight at once (8 there).  Same arithmetic in the same order.
template <int LPR, int TQ>
__global__ __launch_bounds__(256) void pool_fast_kernel(dctr_pool_args_t a) {
    constexpr int SPW = 64 / LPR;
    constexpr int NR = (TQ + LPR - 1) / LPR;       // 16-B id chunks per lane
    constexpr int TT = 4 * TQ;                     // = maxlen
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = lane / LPR, q = lane % LPR;
    const int64_t b = ((int64_t)blockIdx.x * 4 + wave) * SPW + s;
    const bool valid = b < a.batch;
    const int64_t bb = valid ? b : a.batch - 1;
    const bool by_len = a.length != nullptr;
    const int len = (by_len && valid) ? a.length[b] : 0;
    const int4* ids = reinterpret_cast<const int4*>(reinterpret_cast<const int32_t*>(a.idx) + bb * a.idx_stride);
    int4 chunk[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int c = r * LPR + q;
        chunk[r] = c < TQ ? ids[c] : int4{0, 0, 0, 0};
    }
    const bool is_max = a.combiner == DCTR_POOL_MAX;
    float acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = is_max ? -INFINITY : 0.f;
    float lacc = is_max ? -INFINITY : 0.f;
    float cnt = 0.f;
    int oor = 0;
#pragma unroll
    for (int h0 = 0; h0 < TT; h0 += 16) {
        constexpr int U = 16;
        int row[U];
        bool ok[U];
        float v[U][4], lv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = h0 + u;
            row[u] = 0;
            ok[u] = false;
            if (t < TT) {
                const int c = t >> 2;                                   // chunk c sits in lane (c % LPR) of the sample, register c / LPR
                const int4 ch = chunk[c / LPR];
                const int mine = (t & 3) == 0 ? ch.x : (t & 3) == 1 ? ch.y : (t & 3) == 2 ? ch.z : ch.w;
                row[u] = __shfl(mine, (lane - q) + (c % LPR), 64);
                ok[u] = valid && (uint64_t)(int64_t)row[u] < (uint64_t)a.vocab;
                if (valid && !ok[u]) oor = 1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = 0.f;
            lv[u] = 0.f;
            if (h0 + u < TT && ok[u]) {
                load_vec<4>(a.table + (int64_t)row[u] * a.dim + q * 4, v[u]);
                if (q == 0 && a.lin_table != nullptr) lv[u] = a.lin_table[row[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = h0 + u;
            if (t < TT && valid) {
                const float mk = (by_len ? (t < len) : (row[u] != 0)) ? 1.f : 0.f;
                cnt += mk;
                if (is_max) {
                    const float pen = (1.f - mk) * 1e9f;  // layers/sequence.py:97
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaxf(acc[c], v[u][c] * 1.f - pen);
                    lacc = fmaxf(lacc, lv[u] * 1.f - pen);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] += v[u][c] * 1.f * mk;
                    lacc += lv[u] * 1.f * mk;
                }
            }
        }
    }
    if (a.combiner == DCTR_POOL_MEAN) {
        const float denom = (by_len ? (float)len : cnt) + 1e-8f;  // layers/sequence.py:65,103
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = acc[c] / denom;
        lacc = lacc / denom;
    }
    if (valid) {
        store_vec<4>(a.out + b * a.out_stride + q * 4, acc);
        if (q == 0 && a.lin_out != nullptr) a.lin_out[b] = lacc;
    }
    if (a.status != nullptr && __any(oor) && lane == 0) atomicOr(a.status, (int)DCTR_STATUS_INDEX_OOR);
}

// ---------------------------------------------------------------------------------------------------
// plain lookup: n ids -> n rows (+ mask)
// ---------------------------------------------------------------------------------------------------
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void lookup_kernel(dctr_lookup_args_t a) {
    constexpr int RPB = 256 / LPR;  // rows per block pass
    const int r = threadIdx.x / LPR, q = threadIdx.x % LPR;
    int oor = 0;
    for (int64_t i = (int64_t)blockIdx.x * RPB + r; i < a.n; i += (int64_t)gridDim.x * RPB) {
        const int64_t row = resolve_row(read_id(a.idx, i, a.idx_is_i64), a.hash_mode, a.idx_is_i64, a.vocab);
        const bool ok = (uint64_t)row < (uint64_t)a.vocab;
        if (!ok) oor = 1;
        float v[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[c] = 0.f;
        for (int qe = q * VEC; qe < a.dim; qe += LPR * VEC) {          // (rows wider than LPR * VEC elements: several steps)
            if (ok) load_vec<VEC>(a.table + row * a.dim + qe, v);
            store_vec<VEC>(a.out + i * a.out_stride + qe, v);
        }
        if (q == 0 && a.mask != nullptr) a.mask[i] = row != 0 ? 1 : 0;
    }
    if (a.status != nullptr && __any(oor) && (threadIdx.x & 63) == 0) atomicOr(a.status, (int)DCTR_STATUS_INDEX_OOR);
}

// several lookups in ONE launch (DIN: query features and behaviour sequences): blockIdx.y selects the lookup, lanes
// beyond a lookup's width idle.  A lookup that writes a mask also ANDs in (id != 0) of up to four further id arrays of
// the same length: the conjunction of the mask_zero history features' masks the attention layer needs
// (models/sequence/din.py:68-76; a mask_zero Hash maps 0 -> 0 and nothing else to 0, so the raw id decides).
constexpr int LOOKUP_MULTI_MAX = 8, LOOKUP_EXTRA_MAX = 4;
struct LookupMulti {
    dctr_lookup_args_t a[LOOKUP_MULTI_MAX];
    const void* extra_idx[LOOKUP_EXTRA_MAX];
    int32_t extra_is_i64[LOOKUP_EXTRA_MAX];
    int32_t n_extra;
};

template <int VEC, int LPR>
__global__ __launch_bounds__(256) void lookup_multi_kernel(LookupMulti m) {
    constexpr int RPB = 256 / LPR;
    const dctr_lookup_args_t& a = m.a[blockIdx.y];
    const int r = threadIdx.x / LPR, q = threadIdx.x % LPR;
    int oor = 0;
    for (int64_t i = (int64_t)blockIdx.x * RPB + r; i < a.n; i += (int64_t)gridDim.x * RPB) {
        const int64_t row = resolve_row(read_id(a.idx, i, a.idx_is_i64), a.hash_mode, a.idx_is_i64, a.vocab);
        const bool ok = (uint64_t)row < (uint64_t)a.vocab;
        if (!ok) oor = 1;
        float v[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[c] = 0.f;
        for (int qe = q * VEC; qe < a.dim; qe += LPR * VEC) {          // (rows wider than LPR * VEC elements: several steps)
            if (ok) load_vec<VEC>(a.table + row * a.dim + qe, v);
            store_vec<VEC>(a.out + i * a.out_stride + qe, v);
        }
        if (q == 0 && a.mask != nullptr) {
            bool nz = row != 0;
            for (int e = 0; e < m.n_extra; ++e) nz = nz && read_id(m.extra_idx[e], i, m.extra_is_i64[e]) != 0;
            a.mask[i] = nz ? 1 : 0;
        }
    }
    if (a.status != nullptr && __any(oor) && (threadIdx.x & 63) == 0) atomicOr(a.status, (int)DCTR_STATUS_INDEX_OOR);
}

// ---------------------------------------------------------------------------------------------------
// stand-alone WeightedSequenceLayer.call over a materialised [B,T,E] tensor (reference layers/sequence.py:155-183)
// one wave per sample: masked weights -> (optional) softmax over T -> out[b,t,:] = seq[b,t,:] * w[b,t]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seq_weight_kernel(const float* __restrict__ seq, const float* __restrict__ weight,
                                                         const uint8_t* __restrict__ mask, const int32_t* __restrict__ length,
                                                         int64_t batch, int T, int E, int weight_norm,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const float PAD = -4294967296.f;
    const int len = length != nullptr ? length[b] : 0;
    float mx = -INFINITY, den = 1.f;
    if (weight_norm) {
        for (int t = lane; t < T; t += 64) {
            const bool m = length != nullptr ? (t < len) : (mask[b * T + t] != 0);
            mx = fmaxf(mx, m ? weight[b * T + t] : PAD);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        den = 0.f;
        for (int t = lane; t < T; t += 64) {
            const bool m = length != nullptr ? (t < len) : (mask[b * T + t] != 0);
            den += expf((m ? weight[b * T + t] : PAD) - mx);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) den += __shfl_xor(den, o, 64);
    }
    const int64_t n = (int64_t)T * E;
    for (int64_t i = lane; i < n; i += 64) {
        const int t = (int)(i / E);
        const bool m = length != nullptr ? (t < len) : (mask[b * T + t] != 0);
        const float wv = weight[b * T + t];
        const float w = weight_norm ? expf((m ? wv : PAD) - mx) / den : (m ? wv : 0.f);
        out[b * n + i] = seq[b * n + i] * w;
    }
}

// ---------------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------------
// lanes per row for (max_dim, VEC): smallest power of two with LPR * VEC >= max_dim
int lanes_per_row(int max_dim, int vec) {
    int l = 1;
    while (l * vec < max_dim && l < 64) l <<= 1;
    return l;
}

#define DCTR_DISPATCH_LPR(VECV, lpr, CALL)        \
    switch (lpr) {                                \
        case 1: CALL(VECV, 1); break;             \
        case 2: CALL(VECV, 2); break;             \
        case 4: CALL(VECV, 4); break;             \
        case 8: CALL(VECV, 8); break;             \
        case 16: CALL(VECV, 16); break;           \
        case 32: CALL(VECV, 32); break;           \
        case 64: CALL(VECV, 64); break;           \
        default: break;                           \
    }

}  // namespace

extern "C" int dctr_embed_gather_fm(const dctr_gather_fm_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "embed_gather_fm: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->n_fields >= 0 && a->n_dense >= 0, DCTR_E_DIM, "embed_gather_fm: negative size");
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->n_fields + a->n_dense > 0, DCTR_E_DIM, "embed_gather_fm: no fields");
    DCTR_REQUIRE(a->n_fields == 0 || (a->fields != nullptr && a->ids != nullptr), DCTR_E_NULL,
                 "embed_gather_fm: null field descriptors / id matrix");
    DCTR_REQUIRE(a->n_dense == 0 || a->dense != nullptr, DCTR_E_NULL, "embed_gather_fm: null dense matrix");
    DCTR_REQUIRE(a->n_dense == 0 || a->dense_stride >= a->n_dense, DCTR_E_DIM, "embed_gather_fm: dense_stride < n_dense");
    DCTR_REQUIRE(a->dense_copy_cols >= 0 && a->dense_copy_cols <= a->n_dense, DCTR_E_DIM,
                 "embed_gather_fm: dense_copy_cols outside [0, n_dense]");
    const int vec = (a->all_dim4 && a->n_fields > 0) ? 4 : 1;
    const int max_dim = a->max_dim > 0 ? a->max_dim : 1;
    DCTR_REQUIRE(max_dim <= 65536, DCTR_E_UNSUPPORTED, "embed_gather_fm: embedding_dim %d", max_dim);
    if (a->dnn_in != nullptr && vec == 4)
        DCTR_REQUIRE(dctr_aligned16(a->dnn_in) && a->out_stride % 4 == 0, DCTR_E_ALIGN,
                     "embed_gather_fm: dnn_in must be 16-B aligned with out_stride %% 4 == 0 when all_dim4");
    const int lpr = lanes_per_row(max_dim, vec);             // (<= 64: wider rows take several passes)
    const int spw = 64 / lpr;
    const int64_t waves_plain = dctr_ceil_div(a->batch, spw);
    const bool fsplit = waves_plain < 256 * 8 && a->n_fields >= 8 && max_dim <= 64 * vec;
    GatherParams p = *a;
    if (p.n_fields == 0) {  // dense-only call: keep the dummy-address reads valid
        p.fields = reinterpret_cast<const dctr_field_t*>(a->dense);
        p.ids = a->dense;
    }
    const int64_t blocks = fsplit ? waves_plain : dctr_ceil_div(waves_plain, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "embed_gather_fm: batch too large for one launch");
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_G(VECV, L, FS, HS) \
    DCTR_LAUNCH((gather_fm_kernel<VECV, L, FS, HS>), dim3((unsigned)blocks), dim3(256), 0, st, p)
#define CALL_G(VECV, L)                                                 \
    do {                                                                \
        if (fsplit) {                                                   \
            if (a->any_hash) LAUNCH_G(VECV, L, true, true);             \
            else LAUNCH_G(VECV, L, true, false);                        \
        } else {                                                        \
            if (a->any_hash) LAUNCH_G(VECV, L, false, true);            \
            else LAUNCH_G(VECV, L, false, false);                       \
        }                                                               \
    } while (0)
    if (vec == 4) { DCTR_DISPATCH_LPR(4, lpr, CALL_G) } else { DCTR_DISPATCH_LPR(1, lpr, CALL_G) }
#undef CALL_G
#undef LAUNCH_G
    return dctr_launch_status("dctr_embed_gather_fm");
}

extern "C" int dctr_embed_pool(const dctr_pool_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "embed_pool: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->maxlen >= 1 && a->dim >= 1, DCTR_E_DIM, "embed_pool: bad sizes B=%lld T=%d dim=%d",
                 (long long)a->batch, a->maxlen, a->dim);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->idx && a->table && a->out, DCTR_E_NULL, "embed_pool: null pointer");
    DCTR_REQUIRE(a->combiner >= DCTR_POOL_SUM && a->combiner <= DCTR_POOL_MAX, DCTR_E_ENUM, "embed_pool: combiner %d",
                 a->combiner);
    DCTR_REQUIRE(a->hash_mode == 0 || a->hash_mode == 2, DCTR_E_ENUM, "embed_pool: hash_mode %d", a->hash_mode);
    const bool v4 = a->dim % 4 == 0 && dctr_aligned16(a->table) && dctr_aligned16(a->out) && a->out_stride % 4 == 0;
    const int vec = v4 ? 4 : 1;
    DCTR_REQUIRE(a->dim <= 65536, DCTR_E_UNSUPPORTED, "embed_pool: embedding_dim %d too large", a->dim);
    const int lpr = lanes_per_row(a->dim, vec);
    const int64_t blocks = dctr_ceil_div(dctr_ceil_div(a->batch, 64 / lpr), 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "embed_pool: batch too large");
    hipStream_t st = (hipStream_t)stream;
    // the fast kernel: int32 ids in 16-B aligned rows of T <= 32 (T % 4 == 0), embedding_dim 16 / 32, no weights, no hashing
    if (v4 && (a->dim == 16 || a->dim == 32) && !a->idx_is_i64 && a->weight == nullptr && a->hash_mode == 0 && a->maxlen % 4 == 0 &&
        a->maxlen <= 32 && a->idx_stride % 4 == 0 && dctr_aligned16(a->idx)) {
#define CALL_PF(L, Q) DCTR_LAUNCH((pool_fast_kernel<L, Q>), dim3((unsigned)blocks), dim3(256), 0, st, *a)
#define CALL_PFQ(L)                                                                                     \
        switch (a->maxlen / 4) {                                                                        \
            case 1: CALL_PF(L, 1); break; case 2: CALL_PF(L, 2); break; case 3: CALL_PF(L, 3); break;   \
            case 4: CALL_PF(L, 4); break; case 5: CALL_PF(L, 5); break; case 6: CALL_PF(L, 6); break;   \
            case 7: CALL_PF(L, 7); break; default: CALL_PF(L, 8); break;                                \
        }
        if (a->dim == 16) { CALL_PFQ(4) } else { CALL_PFQ(8) }
#undef CALL_PFQ
#undef CALL_PF
        return dctr_launch_status("dctr_embed_pool");
    }
#define CALL_P(VECV, L) DCTR_LAUNCH((pool_kernel<VECV, L>), dim3((unsigned)blocks), dim3(256), 0, st, *a)
    if (vec == 4) { DCTR_DISPATCH_LPR(4, lpr, CALL_P) } else { DCTR_DISPATCH_LPR(1, lpr, CALL_P) }
#undef CALL_P
    return dctr_launch_status("dctr_embed_pool");
}

extern "C" int dctr_embed_lookup(const dctr_lookup_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "embed_lookup: null args");
    DCTR_REQUIRE(a->n >= 0 && a->dim >= 1, DCTR_E_DIM, "embed_lookup: bad sizes");
    if (a->n == 0) return DCTR_OK;
    DCTR_REQUIRE(a->idx && a->table && a->out, DCTR_E_NULL, "embed_lookup: null pointer");
    DCTR_REQUIRE(a->out_stride >= a->dim, DCTR_E_DIM, "embed_lookup: out_stride < dim");
    const bool v4 = a->dim % 4 == 0 && dctr_aligned16(a->table) && dctr_aligned16(a->out) && a->out_stride % 4 == 0;
    const int vec = v4 ? 4 : 1;
    DCTR_REQUIRE(a->dim <= 65536, DCTR_E_UNSUPPORTED, "embed_lookup: embedding_dim %d too large", a->dim);
    const int lpr = lanes_per_row(a->dim, vec);
    int64_t blocks = dctr_ceil_div(a->n, 256 / lpr);
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
#define CALL_L(VECV, L) DCTR_LAUNCH((lookup_kernel<VECV, L>), dim3((unsigned)blocks), dim3(256), 0, st, *a)
    if (vec == 4) { DCTR_DISPATCH_LPR(4, lpr, CALL_L) } else { DCTR_DISPATCH_LPR(1, lpr, CALL_L) }
#undef CALL_L
    return dctr_launch_status("dctr_embed_lookup");
}

extern "C" int dctr_seq_weight_fwd(const float* seq, const float* weight, const uint8_t* mask, const int32_t* length,
                                   int64_t batch, int32_t maxlen, int32_t dim, int32_t weight_norm, float* out,
                                   void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1 && dim >= 1, DCTR_E_DIM, "seq_weight_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(seq && weight && out && (mask || length), DCTR_E_NULL, "seq_weight_fwd: null pointer");
    const int64_t blocks = dctr_ceil_div(batch, 4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "seq_weight_fwd: batch too large");
    DCTR_LAUNCH(seq_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seq, weight, mask,
                       length, batch, maxlen, dim, weight_norm, out);
    return dctr_launch_status("dctr_seq_weight_fwd");
}

extern "C" int dctr_embed_lookup_multi(const dctr_lookup_args_t* args, int32_t n_lookups, const void* const* extra_mask_idx,
                                       const int32_t* extra_is_i64, int32_t n_extra, void* stream) {
    DCTR_REQUIRE(args != nullptr && n_lookups >= 0 && n_lookups <= LOOKUP_MULTI_MAX, DCTR_E_DIM,
                 "embed_lookup_multi: 0..%d lookups", LOOKUP_MULTI_MAX);
    DCTR_REQUIRE(n_extra >= 0 && n_extra <= LOOKUP_EXTRA_MAX && (n_extra == 0 || (extra_mask_idx && extra_is_i64)), DCTR_E_DIM,
                 "embed_lookup_multi: 0..%d extra mask id arrays", LOOKUP_EXTRA_MAX);
    if (n_lookups == 0) return DCTR_OK;
    LookupMulti m{};
    m.n_extra = n_extra;
    for (int e = 0; e < n_extra; ++e) {
        DCTR_REQUIRE(extra_mask_idx[e] != nullptr, DCTR_E_NULL, "embed_lookup_multi: extra_mask_idx[%d] null", e);
        m.extra_idx[e] = extra_mask_idx[e];
        m.extra_is_i64[e] = extra_is_i64[e];
    }
    bool v4 = true;
    int max_dim = 1;
    int64_t max_n = 0;
    for (int k = 0; k < n_lookups; ++k) {
        const dctr_lookup_args_t* a = &args[k];
        DCTR_REQUIRE(a->n >= 0 && a->dim >= 1, DCTR_E_DIM, "embed_lookup_multi[%d]: bad sizes", k);
        DCTR_REQUIRE(a->n == 0 || (a->idx && a->table && a->out), DCTR_E_NULL, "embed_lookup_multi[%d]: null pointer", k);
        DCTR_REQUIRE(a->out_stride >= a->dim, DCTR_E_DIM, "embed_lookup_multi[%d]: out_stride < dim", k);
        v4 = v4 && a->dim % 4 == 0 && dctr_aligned16(a->table) && dctr_aligned16(a->out) && a->out_stride % 4 == 0;
        max_dim = a->dim > max_dim ? a->dim : max_dim;
        max_n = a->n > max_n ? a->n : max_n;
        m.a[k] = *a;
    }
    if (max_n == 0) return DCTR_OK;
    const int vec = v4 ? 4 : 1;
    DCTR_REQUIRE(max_dim <= 65536, DCTR_E_UNSUPPORTED, "embed_lookup_multi: embedding_dim %d too large", max_dim);
    const int lpr = lanes_per_row(max_dim, vec);
    int64_t blocks = dctr_ceil_div(max_n, 256 / lpr);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
#define CALL_LM(VECV, L) \
    DCTR_LAUNCH((lookup_multi_kernel<VECV, L>), dim3((unsigned)blocks, (unsigned)n_lookups), dim3(256), 0, st, m)
    if (vec == 4) { DCTR_DISPATCH_LPR(4, lpr, CALL_LM) } else { DCTR_DISPATCH_LPR(1, lpr, CALL_LM) }
#undef CALL_LM
    return dctr_launch_status("dctr_embed_lookup_multi");
}
