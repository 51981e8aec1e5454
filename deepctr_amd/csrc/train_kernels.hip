// SURVEY.md §8(f) rank 1 — backward + optimizer for the hot path's ops, so that `fit()` runs on HIP kernels:
//   dctr_bce_grad             loss + d(loss)/d(logit) of PredictionLayer('binary') + binary_crossentropy / of mse
//   dctr_embed_gather_fm_bwd  backward of dctr_embed_gather_fm: embedding_lookup + concat (inputs.py:101-117,
//                             layers/utils.py:336-346), get_linear_logit (feature_column.py:171-210) and FM
//                             (layers/interaction.py:588-604): row gradients scatter-added into dense gradient tables
//   dctr_mlp_bwd              backward of DNN.call + Dense(1) head (layers/core.py:189-208): the two GEMMs per layer
//                             are plain GEMMs on the own MFMA kernel of gemm_kernels.hip (dctr_gemm; rocBLAS until round 3); masks, bias sums and the head are kernels here
//   dctr_adam_step            Keras Adam (non-lazy: every row of a table moves every step, as TF's
//                             _resource_apply_sparse does) with the reference's l2 regulariser folded in
// The reference has no code of its own for any of this (Keras autodiff + tf.keras.optimizers); the formulas are the
// derivatives of the forward expressions cited above.
#include <stdlib.h>

#include "dctr_gemm.h"

#include "dctr_common.h"
#include "embed_device.h"

namespace dctr_mlp {   // mlp_bwd_kernels.hip
int launch_bwd_chain(hipStream_t stream, int64_t batch, int in_dim, int n_layers, const int32_t* units_fwd, const float* const* Wt,
                     const float* const* acts, int activation, const float* dz_in, float* const* dz_out, float* dx, int64_t dx_stride);
bool bwd_chain_fits(int in_dim, int n_layers, const int32_t* units_fwd);
}

namespace {

// ---------------------------------------------------------------------------------------------------
// loss
// ---------------------------------------------------------------------------------------------------
// weight (optional): tf.keras' sample_weight — loss = sum_b w_b l_b / B (reduction SUM_OVER_BATCH_SIZE), so l_b and d l_b scale by w_b
__global__ __launch_bounds__(256) void bce_grad_kernel(const float* __restrict__ pred, const float* __restrict__ y,
                                                       const float* __restrict__ weight, int64_t batch, int task,
                                                       float* __restrict__ dlogit,
                                                       float* __restrict__ loss_sum, float* __restrict__ dlogit_sum) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f, ds = 0.f;
    if (b < batch) {
        const float p = pred[b], t = y[b];
        if (task == 0) {
            // d/dlogit of mean BCE(sigmoid(logit), y) = (p - y) / B; value with Keras' epsilon clip (backend.epsilon = 1e-7)
            const float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
            l = -(t * logf(pc) + (1.f - t) * logf(1.f - pc));
            ds = (p - t) / (float)batch;
        } else {
            l = (p - t) * (p - t);
            ds = 2.f * (p - t) / (float)batch;
        }
        if (weight != nullptr) {
            const float w = weight[b];
            l *= w;
            ds *= w;
        }
        dlogit[b] = ds;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        l += __shfl_xor(l, m, 64);
        ds += __shfl_xor(ds, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (loss_sum != nullptr) unsafeAtomicAdd(loss_sum, l);
        if (dlogit_sum != nullptr) unsafeAtomicAdd(dlogit_sum, ds);     // gradient of PredictionLayer's global_bias
    }
}

// ---------------------------------------------------------------------------------------------------
// embedding / linear / FM backward.  Lane (s, q) = 16-B chunk q of sample s, as in the forward gather.
//   d e_f = d dnn_in[b, off_f ..] + d_fm[b] * (S - e_f)     (S = sum over the FM fields of e: FM = 0.5 (S^2 - sum e^2))
//   d lin_f[row] += d_lin[b]
// ---------------------------------------------------------------------------------------------------
// One wave = 64 / LPR samples x LPR 16-B chunks; the workgroup's four waves split the FIELDS (wave w owns fields w, w + 4, ...):
// B = 4096, E = 16 -> 256 workgroups with 7 independent row reads per lane in flight (the first version walked all fields
// serially in 64 workgroups: 112 us).  S is summed per wave over its fields and combined through LDS.
// Measured and dropped (second half of round 2): 16 waves per workgroup (same 76 us: the 1.7 M float atomics run at the part's
// ~23 G distinct-address atomics/s) and a row census per table that lets rows hit once take a plain 16-B read-modify-write
// (62 us + two 6.6-us census launches = the same: then the ~640 k random sector accesses of the kernel are the limit, at the
// rate the forward gather reaches at this launch size).
// Any embedding width (round 6; the reference's embedding_dim is free, feature_column.py:44-45 gives e.g. 102 for "auto"): a lane owns
// the VEC-float chunks q, q + LPR, q + 2 LPR, ... of its sample's rows — ONE chunk when the widths fit 4 LPR <= 64 floats (the loop
// below then runs once: the kernel of rounds 2-5), several for wider rows; VEC = 1 (element per lane, scalar loads) when some width is
// not a multiple of 4 (rows are then not 16-B aligned).  The sum S is per chunk, so a chunk's share closes before the next starts.
template <int LPR, bool HASH, int VEC = 4>
__global__ __launch_bounds__(256) void gather_fm_bwd_kernel(dctr_gather_fm_args_t p, const dctr_field_grad_t* __restrict__ gr,
                                                            const float* __restrict__ d_in, int64_t d_stride,
                                                            const float* __restrict__ d_fm, const float* __restrict__ d_lin,
                                                            float* __restrict__ g_dense_lin_w,
                                                            const int32_t* __restrict__ dense_lin_rows) {
    constexpr int SPW = 64 / LPR, NWV = 4;
    __shared__ float s_part[NWV][64][VEC];
    // low-cardinality fields (DIN's `gender`: 2 rows for 2,048 samples = 1,024 atomics per address, ~90 ns each): their gradient
    // rows (and linear rows) accumulate in LDS and leave as one atomic per element and workgroup
    constexpr int SMALL_FLOATS = 2048, SMALL_FIELDS = 128, SMALL_VOCAB = 64;
    __shared__ float s_small[SMALL_FLOATS];
    __shared__ int s_soff[SMALL_FIELDS];
    __shared__ int s_total;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int j = 0; j < p.n_fields && j < SMALL_FIELDS; ++j) {
            const int64_t vocab = ((cfield_ptr)p.fields)[j].vocab;
            const int64_t need = vocab * (((cfield_ptr)p.fields)[j].dim + 1);
            const bool small = gr[j].g_table != nullptr && vocab <= SMALL_VOCAB && off + need <= SMALL_FLOATS;
            s_soff[j] = small ? off : -1;
            if (small) off += (int)need;
        }
        s_total = off;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < s_total; i += 256) s_small[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane / LPR, q0 = lane % LPR;
    const int64_t b = (int64_t)blockIdx.x * SPW + s;
    const bool valid = b < p.batch;
    cfield_ptr F = (cfield_ptr)p.fields;
    const float dfm = (valid && d_fm != nullptr) ? d_fm[b] : 0.f;
    const float dlin = (valid && d_lin != nullptr) ? d_lin[b] : 0.f;
    // the forward's dnn_in still holds every field's row of this sample (dctr_gather_fm_bwd_args_t.fwd->dnn_in; VEC = 4: 16-B aligned rows)
    const bool from_x = p.dnn_in != nullptr && (VEC == 1 || (p.out_stride % 4 == 0 && ((uintptr_t)p.dnn_in & 15) == 0));
    auto row_of = [&](const FieldRegs& f, int j) -> int64_t {
        int64_t r = f.identity ? b : read_id(p.ids, (int64_t)j * p.ids_stride_f + (valid ? b : 0) * p.ids_stride_b, p.ids_is_i64);
        if constexpr (HASH) {
            if (f.hash_mode != 0) r = resolve_row(r, f.hash_mode, p.ids_is_i64, f.vocab);
        }
        return r;
    };
    for (int q = q0; (q - q0) * VEC < p.max_dim; q += LPR) {      // (workgroup-uniform trip count: the barrier below is safe)
    // pass 1: S = sum over the FM fields of e_f (this wave's fields, then the four partial sums)
    float S[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) S[c] = 0.f;
    if (d_fm != nullptr) {
        if (q != q0) __syncthreads();                              // (the previous chunk's partial sums have been read)
        for (int j = wave; j < p.n_fields; j += NWV) {
            const FieldRegs f = load_field(F, j);
            if (!f.in_fm) continue;
            const int64_t r = row_of(f, j);
            const bool ok = valid && (uint64_t)r < (uint64_t)f.vocab && q * VEC < f.dim;
            float v[VEC];
            // (the forward left this row in the sample's dnn_in row: a coalesced read instead of a second random one — same bits)
            if (from_x && f.out_offset >= 0) load_vec<VEC>(p.dnn_in + (valid ? b : 0) * p.out_stride + f.out_offset + (ok ? q * VEC : 0), v);
            else load_vec<VEC>(f.table + (ok ? r : 0) * f.dim + (ok ? q * VEC : 0), v);
#pragma unroll
            for (int c = 0; c < VEC; ++c) S[c] += ok ? v[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) s_part[wave][lane][c] = S[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < VEC; ++c) S[c] = (s_part[0][lane][c] + s_part[1][lane][c]) + (s_part[2][lane][c] + s_part[3][lane][c]);
    }
    // pass 2: row gradients of this wave's fields
    for (int j = wave; j < p.n_fields; j += NWV) {
        const FieldRegs f = load_field(F, j);
        const int64_t r = row_of(f, j);
        const bool rok = valid && (uint64_t)r < (uint64_t)f.vocab;
        const bool ok = rok && q * VEC < f.dim;
        float* gt = gr[j].g_table;
        float* gl = gr[j].g_lin_table;
        const int so = j < SMALL_FIELDS ? s_soff[j] : -1;
        if (ok && gt != nullptr) {
            float g[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) g[c] = 0.f;
            if (f.out_offset >= 0 && d_in != nullptr) load_vec<VEC>(d_in + b * d_stride + f.out_offset + q * VEC, g);
            if (f.in_fm && d_fm != nullptr) {
                float v[VEC];
                if (from_x && f.out_offset >= 0) load_vec<VEC>(p.dnn_in + b * p.out_stride + f.out_offset + q * VEC, v);
                else load_vec<VEC>(f.table + r * f.dim + q * VEC, v);
#pragma unroll
                for (int c = 0; c < VEC; ++c) g[c] = fmaf(dfm, S[c] - v[c], g[c]);
            }
            if (gr[j].touched != nullptr) gr[j].touched[(r * f.dim + q * VEC) >> 2] = 1;  // (one byte per 16-B group: tables with dim % 4 == 0 only)
            if (so >= 0) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) atomicAdd(&s_small[so + (int)r * f.dim + q * VEC + c], g[c]);
            } else {
                float* dst = gt + r * f.dim + q * VEC;
#pragma unroll
                for (int c = 0; c < VEC; ++c) unsafeAtomicAdd(dst + c, g[c]);
            }
        }
        if (rok && q == 0 && gl != nullptr && f.lin_table != nullptr) {
            if (so >= 0) atomicAdd(&s_small[so + (int)f.vocab * f.dim + (int)r], dlin);
            else unsafeAtomicAdd(gl + r, dlin);
        }
    }
    }   // chunks
    const int q = q0;
    // dense . Linear.kernel: d w[k] += sum_b d_lin[b] * dense[b, k]  — dense column k by wave k % 4, the wave's samples summed
    // before the one atomic (4096 atomics on one address serialise)
    if (g_dense_lin_w != nullptr && p.n_dense > 0) {
        for (int k = wave; k < p.n_dense; k += NWV) {
            const int row = dense_lin_rows != nullptr ? dense_lin_rows[k] : k;     // dense column k -> row of Linear.kernel
            if (row < 0) continue;
            float t = (valid && q == 0) ? dlin * p.dense[b * p.dense_stride + k] : 0.f;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
            if (lane == 0) unsafeAtomicAdd(g_dense_lin_w + row, t);
        }
    }
    // the low-cardinality fields' LDS accumulators
    if (s_total > 0) {
        __syncthreads();
        for (int j = 0; j < p.n_fields && j < SMALL_FIELDS; ++j) {
            const int so = s_soff[j];
            if (so < 0) continue;
            const FieldRegs f = load_field(F, j);
            const int nt = (int)f.vocab * f.dim;
            for (int i = threadIdx.x; i < nt; i += 256) {
                const float v = s_small[so + i];
                if (v != 0.f) unsafeAtomicAdd(gr[j].g_table + i, v);
            }
            if (gr[j].g_lin_table != nullptr && f.lin_table != nullptr) {
                for (int i = threadIdx.x; i < (int)f.vocab; i += 256) {
                    const float v = s_small[so + nt + i];
                    if (v != 0.f) unsafeAtomicAdd(gr[j].g_lin_table + i, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward of dctr_embed_pool (inputs.py:120-158, layers/sequence.py:76-106, :155-183): position t of sample b
// contributed  e_t * wt_t * mk_t  (sum), the same / denom (mean), or was the per-dimension maximum (max); the
// gradient of the pooled vector is scattered back to the rows with those factors.  Per-position weights are inputs.
// ---------------------------------------------------------------------------------------------------
// (any embedding width, as gather_fm_bwd_kernel: a lane walks the VEC-float chunks q0, q0 + LPR, ... of the row; VEC = 1 when dim % 4 != 0)
template <int LPR, int VEC = 4>
__global__ __launch_bounds__(256) void pool_bwd_kernel(dctr_pool_args_t a, const float* __restrict__ d_out, int64_t d_stride,
                                                       const float* __restrict__ d_lin_out, float* __restrict__ g_table,
                                                       float* __restrict__ g_lin_table) {
    constexpr int SPB = 256 / LPR;
    const int s = threadIdx.x / LPR, q0 = threadIdx.x % LPR;
    const int64_t b = (int64_t)blockIdx.x * SPB + s;
    if (b >= a.batch) return;
    for (int q = q0; q * VEC < a.dim || q == q0; q += LPR) {
    const int T = a.maxlen;
    const bool by_len = a.length != nullptr;
    const int len = by_len ? a.length[b] : 0;
    const float PAD = -4294967296.f;
    const bool has_w = a.weight != nullptr, wnorm = has_w && a.weight_norm;
    const bool colok = q * VEC < a.dim;
    auto row_of = [&](int t) { return resolve_row(read_id(a.idx, b * a.idx_stride + t, a.idx_is_i64), a.hash_mode, a.idx_is_i64, a.vocab); };
    auto mask_of = [&](int t, int64_t r) { return by_len ? (t < len) : (r != 0); };
    float wmax = -INFINITY, wden = 1.f;
    if (wnorm) {
        for (int t = 0; t < T; ++t) wmax = fmaxf(wmax, mask_of(t, row_of(t)) ? a.weight[b * (int64_t)T + t] : PAD);
        wden = 0.f;
        for (int t = 0; t < T; ++t) wden += expf((mask_of(t, row_of(t)) ? a.weight[b * (int64_t)T + t] : PAD) - wmax);
    }
    auto weight_of = [&](int t, bool m) {
        if (!has_w) return 1.f;
        const float w = a.weight[b * (int64_t)T + t];
        return wnorm ? expf((m ? w : PAD) - wmax) / wden : (m ? w : 0.f);
    };
    float dv[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) dv[c] = 0.f;
    if (colok && d_out != nullptr) load_vec<VEC>(d_out + b * d_stride + q * VEC, dv);
    const float dl = (d_lin_out != nullptr && q == 0) ? d_lin_out[b] : 0.f;
    const bool is_max = a.combiner == DCTR_POOL_MAX;
    float denom = 1.f;
    if (a.combiner == DCTR_POOL_MEAN) {
        float cnt = 0.f;
        if (!by_len)
            for (int t = 0; t < T; ++t) cnt += mask_of(t, row_of(t)) ? 1.f : 0.f;
        denom = (by_len ? (float)len : cnt) + 1e-8f;
    }
    // max: forward maxima per dimension (and of the 1-wide linear term), then the first position that attains each
    float mx[VEC], lmx = -INFINITY;
#pragma unroll
    for (int c = 0; c < VEC; ++c) mx[c] = -INFINITY;
    if (is_max) {
        for (int t = 0; t < T; ++t) {
            const int64_t r = row_of(t);
            const bool ok = (uint64_t)r < (uint64_t)a.vocab, m = mask_of(t, r);
            const float wt = weight_of(t, m), pen = m ? 0.f : 1e9f;
            float v[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] = 0.f;
            if (ok && colok) load_vec<VEC>(a.table + r * a.dim + q * VEC, v);
#pragma unroll
            for (int c = 0; c < VEC; ++c) mx[c] = fmaxf(mx[c], v[c] * wt - pen);
            if (q == 0 && a.lin_table != nullptr) lmx = fmaxf(lmx, (ok ? a.lin_table[r] : 0.f) * wt - pen);
        }
    }
    bool taken[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) taken[c] = false;
    bool ltaken = false;
    for (int t = 0; t < T; ++t) {
        const int64_t r = row_of(t);
        if ((uint64_t)r >= (uint64_t)a.vocab) continue;
        const bool m = mask_of(t, r);
        const float wt = weight_of(t, m);
        if (is_max) {
            const float pen = m ? 0.f : 1e9f;
            float v[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) v[c] = 0.f;
            if (colok) load_vec<VEC>(a.table + r * a.dim + q * VEC, v);
            if (colok && g_table != nullptr) {
#pragma unroll
                for (int c = 0; c < VEC; ++c)
                    if (!taken[c] && v[c] * wt - pen == mx[c]) {
                        taken[c] = true;
                        unsafeAtomicAdd(g_table + r * a.dim + q * VEC + c, dv[c] * wt);
                    }
            }
            if (q == 0 && g_lin_table != nullptr && a.lin_table != nullptr && !ltaken && a.lin_table[r] * wt - pen == lmx) {
                ltaken = true;
                unsafeAtomicAdd(g_lin_table + r, dl * wt);
            }
        } else {
            const float f = wt * (m ? 1.f : 0.f) / denom;
            if (f != 0.f) {
                if (colok && g_table != nullptr) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) unsafeAtomicAdd(g_table + r * a.dim + q * VEC + c, dv[c] * f);
                }
                if (q == 0 && g_lin_table != nullptr && a.lin_table != nullptr) unsafeAtomicAdd(g_lin_table + r, dl * f);
            }
        }
    }
    }   // chunks
}

// ---------------------------------------------------------------------------------------------------
// DNN backward helpers
// ---------------------------------------------------------------------------------------------------
constexpr int BWD_ROWS = 16;     // batch rows per iteration of the generic Dice passes further down (the 16-B forms take over when N % 4 == 0)

// dZ[b, n] = dlogit[b] * head_w[n] * act'(h[b, n]);   d_head_w[n] += sum_b dlogit[b] * h[b, n]
// (any N / strides: thread = one column (blockIdx.y * 256 + threadIdx.x), rows grid-strided over blockIdx.x; the 16-B forms below
// take over when the shapes allow)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogit, const float* __restrict__ head_w,
                                                       const float* __restrict__ h, int64_t h_stride, int64_t batch, int N,
                                                       int act, float* __restrict__ dz, int64_t dz_stride,
                                                       float* __restrict__ d_head_w) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    const float hw = head_w[n];
    float acc = 0.f;
#pragma unroll 4
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        const float hv = h[b * h_stride + n], dl = dlogit[b];
        float d = dl * hw;
        if (act == DCTR_ACT_RELU) d = hv > 0.f ? d : 0.f;
        else if (act == DCTR_ACT_SIGMOID) d *= hv * (1.f - hv);
        else if (act == DCTR_ACT_TANH) d *= 1.f - hv * hv;
        dz[b * dz_stride + n] = d;
        acc = fmaf(dl, hv, acc);
    }
    unsafeAtomicAdd(d_head_w + n, acc);
}

// in place: dh[b, n] *= act'(h[b, n]);  db[n] += sum_b dz[b, n]   (also used with h == NULL: only the column sums)
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t batch,
                                                             int N, int act, float* __restrict__ db) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
#pragma unroll 4
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
        float d = dh[b * N + n];
        if (h != nullptr) {
            const float hv = h[b * N + n];
            if (act == DCTR_ACT_RELU) d = hv > 0.f ? d : 0.f;
            else if (act == DCTR_ACT_SIGMOID) d *= hv * (1.f - hv);
            else if (act == DCTR_ACT_TANH) d *= 1.f - hv * hv;
            dh[b * N + n] = d;
        }
        acc += d;
    }
    if (db != nullptr) unsafeAtomicAdd(db + n, acc);
}

// The two helpers above with 16-B accesses and the rows spread over the workgroup: thread (rl, c) owns columns 4c .. 4c+3 of rows
// rl, rl + RL, ... (RL = 256 / (N/4) row lanes; a workgroup sweeps RL x N contiguous floats per iteration, grid-stride), the column
// sums meet in LDS and leave as ONE atomic per column and workgroup (N % 4 == 0, N <= 1024, strides % 4 == 0, 16-B aligned).
constexpr int COLSUM_MAX_WG = 64;      // workgroups of a pass that ends in one atomic per column and workgroup

// workgroups of a column-sum pass over `rows` rows: enough of them to keep the loads in flight, few enough that the final atomics
// (one per column and workgroup, ~90 ns each on one address) stay a short tail
static unsigned colsum_grid(int64_t rows, int RL) {
    int64_t g = dctr_ceil_div(rows, (int64_t)RL * 4);
    const int64_t cap = rows >= 32768 ? 128 : COLSUM_MAX_WG;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// grid of the BWD_ROWS-rows-per-iteration kernels (grid-stride over row blocks)
static unsigned rows_grid(int64_t rows, bool column_sums) {
    const int64_t nb = dctr_ceil_div(rows, (int64_t)BWD_ROWS), cap = column_sums ? COLSUM_MAX_WG : 16384;
    return (unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

__device__ __forceinline__ float4 act_grad4(float4 d, float4 hv, int act) {
    if (act == DCTR_ACT_RELU) {
        d.x = hv.x > 0.f ? d.x : 0.f; d.y = hv.y > 0.f ? d.y : 0.f; d.z = hv.z > 0.f ? d.z : 0.f; d.w = hv.w > 0.f ? d.w : 0.f;
    } else if (act == DCTR_ACT_SIGMOID) {
        d.x *= hv.x * (1.f - hv.x); d.y *= hv.y * (1.f - hv.y); d.z *= hv.z * (1.f - hv.z); d.w *= hv.w * (1.f - hv.w);
    } else if (act == DCTR_ACT_TANH) {
        d.x *= 1.f - hv.x * hv.x; d.y *= 1.f - hv.y * hv.y; d.z *= 1.f - hv.z * hv.z; d.w *= 1.f - hv.w * hv.w;
    }
    return d;
}

__device__ __forceinline__ void colsum4_finish(float4 acc, int N4, int RL, int c, float* __restrict__ out) {
    __shared__ float4 red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (out != nullptr && (int)threadIdx.x < N4) {
        float4 t = red[c];
        for (int r = 1; r < RL; ++r) {
            const float4 u = red[r * N4 + c];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        unsafeAtomicAdd(out + 4 * c + 0, t.x);
        unsafeAtomicAdd(out + 4 * c + 1, t.y);
        unsafeAtomicAdd(out + 4 * c + 2, t.z);
        unsafeAtomicAdd(out + 4 * c + 3, t.w);
    }
}

__global__ __launch_bounds__(256) void head_bwd4_kernel(const float* __restrict__ dlogit, const float* __restrict__ head_w,
                                                        const float* __restrict__ h, int64_t h_stride, int64_t batch, int N,
                                                        int act, float* __restrict__ dz, int64_t dz_stride,
                                                        float* __restrict__ d_head_w) {
    const int N4 = N >> 2, RL = 256 / N4;
    const int c = threadIdx.x % N4, rl = threadIdx.x / N4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < RL) {
        const float4 hw = *reinterpret_cast<const float4*>(head_w + 4 * c);
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < batch; r += (int64_t)gridDim.x * RL) {
            const float4 hv = *reinterpret_cast<const float4*>(h + r * h_stride + 4 * c);
            const float dl = dlogit[r];
            const float4 d = act_grad4(make_float4(dl * hw.x, dl * hw.y, dl * hw.z, dl * hw.w), hv, act);
            *reinterpret_cast<float4*>(dz + r * dz_stride + 4 * c) = d;
            acc.x = fmaf(dl, hv.x, acc.x); acc.y = fmaf(dl, hv.y, acc.y); acc.z = fmaf(dl, hv.z, acc.z); acc.w = fmaf(dl, hv.w, acc.w);
        }
    }
    colsum4_finish(acc, N4, RL, c, d_head_w);
}

__global__ __launch_bounds__(256) void act_bwd_colsum4_kernel(float* __restrict__ dh, const float* __restrict__ h, int64_t batch,
                                                              int N, int act, float* __restrict__ db) {
    const int N4 = N >> 2, RL = 256 / N4;
    const int c = threadIdx.x % N4, rl = threadIdx.x / N4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < RL) {
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < batch; r += (int64_t)gridDim.x * RL) {
            float4 d = *reinterpret_cast<const float4*>(dh + r * N + 4 * c);
            if (h != nullptr) {
                d = act_grad4(d, *reinterpret_cast<const float4*>(h + r * N + 4 * c), act);
                *reinterpret_cast<float4*>(dh + r * N + 4 * c) = d;
            }
            acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
    }
    colsum4_finish(acc, N4, RL, c, db);
}

// launch either form: the 16-B one when the shapes allow it
static void launch_act_bwd_colsum(hipStream_t st, float* dh, const float* h, int64_t batch, int N, int act, float* db) {
    if (N % 4 == 0 && N <= 1024 && dctr_aligned16(dh) && (h == nullptr || dctr_aligned16(h))) {
        const int RL = 256 / (N / 4);
        int64_t g = dctr_ceil_div(batch, (int64_t)RL * 4);           // >= 4 rows per thread where the batch has them
        // atomics on ONE address serialise at ~100 ns each on this part (measured: 256 workgroups x 256 columns = 27 us of a
        // 4-MB pass): with column sums wanted, few workgroups with long row loops; without, as many as the rows give
        const int64_t cap = db != nullptr ? (int64_t)colsum_grid(batch, RL) : 1024;
        g = g < 1 ? 1 : (g > cap ? cap : g);
        hipLaunchKernelGGL(act_bwd_colsum4_kernel, dim3((unsigned)g), dim3(256), 0, st, dh, h, batch, N, act, db);
    } else {
        const int64_t gx = db != nullptr ? COLSUM_MAX_WG : 1024;
        hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3((unsigned)(batch < gx ? batch : gx), (unsigned)((N + 255) / 256)), dim3(256), 0, st,
                           dh, h, batch, N, act, db);
    }
}

static void launch_head_bwd(hipStream_t st, const float* dlogit, const float* head_w, const float* h, int64_t h_stride, int64_t batch,
                            int N, int act, float* dz, int64_t dz_stride, float* d_head_w) {
    if (N % 4 == 0 && N <= 1024 && h_stride % 4 == 0 && dz_stride % 4 == 0 && dctr_aligned16(h) && dctr_aligned16(dz) &&
        dctr_aligned16(head_w)) {
        const int RL = 256 / (N / 4);
        int64_t g = dctr_ceil_div(batch, (int64_t)RL * 4);
        g = colsum_grid(batch, RL);
        hipLaunchKernelGGL(head_bwd4_kernel, dim3((unsigned)g), dim3(256), 0, st, dlogit, head_w, h, h_stride, batch, N, act, dz,
                           dz_stride, d_head_w);
    } else {
        hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)(batch < COLSUM_MAX_WG ? batch : COLSUM_MAX_WG), (unsigned)((N + 255) / 256)),
                           dim3(256), 0, st, dlogit, head_w, h, h_stride, batch, N, act, dz, dz_stride, d_head_w);
    }
}

// ---------------------------------------------------------------------------------------------------
// Adam (Keras): m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  w -= alpha * m / (sqrt(v) + eps),
// alpha = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the caller;  g includes 2*l2*w;  the gradient buffer is cleared.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ g, int64_t n, float alpha, float b1, float b2, float eps,
                                                   float l2, int zero_grad) {
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 wv = reinterpret_cast<float4*>(w)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<float4*>(g)[i];
        float* wp = &wv.x; float* mp = &mv.x; float* vp = &vv.x; const float* gp = &gv.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gg = fmaf(2.f * l2, wp[c], gp[c]);
            mp[c] = fmaf(b1, mp[c], (1.f - b1) * gg);
            vp[c] = fmaf(b2, vp[c], (1.f - b2) * gg * gg);
            wp[c] -= alpha * mp[c] / (sqrtf(vp[c]) + eps);
        }
        reinterpret_cast<float4*>(w)[i] = wv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gg = fmaf(2.f * l2, w[i], g[i]);
        m[i] = fmaf(b1, m[i], (1.f - b1) * gg);
        v[i] = fmaf(b2, v[i], (1.f - b2) * gg * gg);
        w[i] -= alpha * m[i] / (sqrtf(v[i]) + eps);
        if (zero_grad) g[i] = 0.f;
    }
}

// one optimizer update of a single element (tf.keras formulas, optimizer_v2/{adam,adagrad,rmsprop,gradient_descent}.py)
template <int KIND>
__device__ __forceinline__ void opt_update(float& w, float& m, float& v, float g, float lr, float b1, float b2, float eps) {
    if constexpr (KIND == DCTR_OPT_ADAM) {            // lr = lr0 * sqrt(1 - b2^t) / (1 - b1^t) from the caller
        m = fmaf(b1, m, (1.f - b1) * g);
        v = fmaf(b2, v, (1.f - b2) * g * g);
        w -= lr * m / (sqrtf(v) + eps);
    } else if constexpr (KIND == DCTR_OPT_ADAGRAD) {  // v = accumulator (initial_accumulator_value set by the caller)
        v = fmaf(g, g, v);
        w -= lr * g / (sqrtf(v) + eps);
    } else if constexpr (KIND == DCTR_OPT_RMSPROP) {  // b2 = rho; no momentum, not centered
        v = fmaf(b2, v, (1.f - b2) * g * g);
        w -= lr * g / (sqrtf(v) + eps);
    } else {                                          // SGD without momentum
        w -= lr * g;
    }
}

// all parameters of a model in ONE launch: blockIdx.y = segment, blockIdx.x walks the segment (blocks past its end exit).
// sg.touched (embedding tables): one byte per 16-B group of the gradient table, set by the backward kernels where they added
// something — groups with a clear byte have a ZERO gradient, which is then neither read nor cleared (a 4096-row batch touches
// < 4 % of a 1e5-row table: 24 instead of 32 bytes of traffic per element; the update itself stays non-lazy, every element moves).
// Two 16-B groups per thread and trip, all their loads issued before the first use.
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float4* p) {
    if constexpr (NT) {
        const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void st4(float4* p, const float4& v) {
    if constexpr (NT) __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
    else *p = v;
}

// PEN: also adds  pen_scale * sum_segments l2 * sum(w^2)  of the weights BEFORE the update to *pen — the l2 penalties tf.keras adds to
// that batch's loss (regularizers: inputs.py:22, layers/core.py:170, interaction.py:100,258,387); the weights pass through registers anyway
template <int KIND, int U = 2, bool NT = false, bool PEN = false>
__global__ __launch_bounds__(256) void opt_multi_kernel(const dctr_adam_seg_t* __restrict__ segs, float lr, float b1, float b2,
                                                        float eps, int zero_grad, double* __restrict__ pen, float pen_scale) {
    const dctr_adam_seg_t sg = segs[blockIdx.y];
    float4* __restrict__ w = reinterpret_cast<float4*>(sg.w);
    float4* __restrict__ m = reinterpret_cast<float4*>(sg.m);
    float4* __restrict__ v = reinterpret_cast<float4*>(sg.v);
    float4* __restrict__ g = reinterpret_cast<float4*>(sg.g);
    uint8_t* __restrict__ tch = sg.touched;
    const float l2 = sg.l2;
    const int64_t n = sg.n, n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    constexpr bool USE_M = KIND == DCTR_OPT_ADAM, USE_V = KIND != DCTR_OPT_SGD;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float psum = 0.f;                  // (PEN: this thread's share of sum(w^2): a few hundred terms at most)
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
        bool in[U], has[U];
        float4 wv[U], mv[U], vv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            in[u] = i < n4;
            has[u] = in[u] && (tch == nullptr || tch[i] != 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = in[u] ? i0 + u * stride : i0;
            wv[u] = ld4<NT>(w + i);
            mv[u] = USE_M ? ld4<NT>(m + i) : z4;
            vv[u] = USE_V ? ld4<NT>(v + i) : z4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) gv[u] = has[u] ? g[i0 + u * stride] : z4;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!in[u]) continue;
            const int64_t i = i0 + u * stride;
            float* wp = &wv[u].x; float* mp = &mv[u].x; float* vp = &vv[u].x; const float* gp = &gv[u].x;
            if constexpr (PEN) psum += (wp[0] * wp[0] + wp[1] * wp[1]) + (wp[2] * wp[2] + wp[3] * wp[3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) opt_update<KIND>(wp[c], mp[c], vp[c], fmaf(2.f * l2, wp[c], gp[c]), lr, b1, b2, eps);
            st4<NT>(w + i, wv[u]);
            if (USE_M) st4<NT>(m + i, mv[u]);
            if (USE_V) st4<NT>(v + i, vv[u]);
            if (zero_grad && has[u]) {
                g[i] = z4;
                if (tch != nullptr) tch[i] = 0;
            }
        }
    }
    if (blockIdx.x == 0) {
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            float mm = USE_M ? sg.m[i] : 0.f, vv = USE_V ? sg.v[i] : 0.f;
            if constexpr (PEN) psum = fmaf(sg.w[i], sg.w[i], psum);
            opt_update<KIND>(sg.w[i], mm, vv, fmaf(2.f * l2, sg.w[i], sg.g[i]), lr, b1, b2, eps);
            if (USE_M) sg.m[i] = mm;
            if (USE_V) sg.v[i] = vv;
            if (zero_grad) sg.g[i] = 0.f;
        }
    }
    if constexpr (PEN) {
        if (l2 != 0.f) {               // (uniform per workgroup: a segment's l2)
            __shared__ double pw[4];
            double d = (double)psum;
#pragma unroll
            for (int mk = 32; mk >= 1; mk >>= 1) d += __shfl_xor(d, mk, 64);
            if ((threadIdx.x & 63) == 0) pw[threadIdx.x >> 6] = d;
            __syncthreads();
            if (threadIdx.x == 0) {
                const double t = (pw[0] + pw[1]) + (pw[2] + pw[3]);
                if (t != 0.0) unsafeAtomicAdd(pen, t * (double)l2 * (double)pen_scale);
            }
        }
    }
}

// touched bytes of the rows a scatter kernel adds to (dctr_embed_lookup_bwd / dctr_embed_pool_bwd): every valid row an id
// resolves to — a superset of the rows that receive a non-zero gradient, which only costs the optimizer a read
__global__ __launch_bounds__(256) void mark_rows_kernel(const void* __restrict__ ids, int64_t n_rows, int n_cols, int64_t row_stride,
                                                        int is_i64, int hash_mode, int64_t vocab, int dim4, uint8_t* __restrict__ touched) {
    const int64_t total = n_rows * n_cols * dim4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t e = t / dim4;
        const int q = (int)(t - e * dim4);
        const int64_t rr = e / n_cols;
        const int cc = (int)(e - rr * n_cols);
        const int64_t r = resolve_row(read_id(ids, rr * row_stride + cc, is_i64), hash_mode, is_i64, vocab);
        if ((uint64_t)r < (uint64_t)vocab) touched[r * dim4 + q] = 1;
    }
}
static void launch_mark_rows(hipStream_t st, const void* ids, int64_t n_rows, int n_cols, int64_t row_stride, int is_i64, int hash_mode,
                             int64_t vocab, int dim, uint8_t* touched) {
    const int dim4 = dim / 4;
    int64_t blocks = dctr_ceil_div(n_rows * n_cols * dim4, (int64_t)256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ids, n_rows, n_cols, row_stride, is_i64, hash_mode, vocab,
                       dim4, touched);
}

// ---------------------------------------------------------------------------------------------------
// CrossNet backward (interaction.py:405-424).  No activations are saved by the forward: both forms recompute x_l.
//   vector: x_{l+1} = x0 * s_l + b_l + x_l,  s_l = x_l . w_l
//     ds = g . x0;  d b_l += g;  d w_l += ds * x_l;  d x0 += g * s_l;  g += ds * w_l      (l = L-1 .. 0),  d x0 += g
//   One wave walks CROSS_SPW samples (x_l per layer in LDS), accumulating d w / d b in LDS, flushed once with atomics.
// ---------------------------------------------------------------------------------------------------
constexpr int CROSS_SPW = 4;       // (16 at first: 64 workgroups, 174 us at B = 4096; the accumulators are combined per workgroup now)

__global__ __launch_bounds__(256) void cross_vector_bwd_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int d,
                                                               int L, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const float* __restrict__ dy, int64_t dy_stride,
                                                               float* __restrict__ dw, float* __restrict__ db,
                                                               float* __restrict__ dx, int64_t dx_stride, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = smem + (size_t)wave * (3 * L * d + L);      // [L][d] x_l
    float* aw = xs + (size_t)L * d;                           // [L][d] d w accumulators
    float* ab = aw + (size_t)L * d;                           // [L][d] d b accumulators
    float* sl = ab + (size_t)L * d;                           // [L] s_l
    for (int i = lane; i < 2 * L * d; i += 64) aw[i] = 0.f;
    const int64_t b_first = ((int64_t)blockIdx.x * 4 + wave) * CROSS_SPW;
    for (int it = 0; it < CROSS_SPW; ++it) {
        const int64_t b = b_first + it;
        if (b >= batch) break;
        const float* x0 = x + b * x_stride;
        // forward, keeping x_l
        for (int i = lane; i < d; i += 64) xs[i] = x0[i];
        for (int l = 0; l < L; ++l) {
            const float* xl = xs + (size_t)l * d;
            float dot = 0.f;
            for (int i = lane; i < d; i += 64) dot = fmaf(xl[i], w[(size_t)l * d + i], dot);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
            if (lane == 0) sl[l] = dot;
            if (l + 1 < L)
                for (int i = lane; i < d; i += 64) xs[(size_t)(l + 1) * d + i] = x0[i] * dot + bias[(size_t)l * d + i] + xl[i];
        }
        // backward: g lives in dx (or a register-free walk over LDS): reuse ab? no — keep g in the output row
        float* gout = dx + b * dx_stride;
        const float* gin = dy + b * dy_stride;
        // g is held in registers in chunks of 64 columns: up to 32 chunks (d <= 2048)
        float g[32], dx0[32];
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int i = lane + 64 * r;
            g[r] = i < d ? gin[i] : 0.f;
            dx0[r] = 0.f;
        }
        for (int l = L - 1; l >= 0; --l) {
            const float* xl = xs + (size_t)l * d;
            const float s = sl[l];
            float ds = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = lane + 64 * r;
                if (i < d) ds = fmaf(g[r], x0[i], ds);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) ds += __shfl_xor(ds, m, 64);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int i = lane + 64 * r;
                if (i < d) {
                    ab[(size_t)l * d + i] += g[r];
                    aw[(size_t)l * d + i] = fmaf(ds, xl[i], aw[(size_t)l * d + i]);
                    dx0[r] = fmaf(g[r], s, dx0[r]);
                    g[r] = fmaf(ds, w[(size_t)l * d + i], g[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int i = lane + 64 * r;
            if (i < d) gout[i] = (accumulate ? gout[i] : 0.f) + dx0[r] + g[r];
        }
    }
    // the four waves' accumulators meet here: one atomic per element and WORKGROUP (they serialise per address)
    __syncthreads();
    const size_t region = (size_t)3 * L * d + L;
    for (int i = threadIdx.x; i < L * d; i += 256) {
        const float* r0 = smem + (size_t)L * d + i;
        unsafeAtomicAdd(dw + i, (r0[0] + r0[region]) + (r0[2 * region] + r0[3 * region]));
        const float* r1 = r0 + (size_t)L * d;
        unsafeAtomicAdd(db + i, (r1[0] + r1[region]) + (r1[2 * region] + r1[3 * region]));
    }
}

// vector form, any width (the kernel above keeps a wave's x_l, d w, d b in LDS and a row's gradient in registers: <= 2048 columns and
// 48 L d bytes of LDS): the same recurrence layer by layer over the whole batch, x_l / s_l / g / d x0 in the caller's workspace —
// memory-bound passes, no limit on d.  Forward step: s_l[b] = x_l[b] . w_l;  x_{l+1}[b] = x0[b] * s_l[b] + b_l + x_l[b]  (one wave per sample)
__global__ __launch_bounds__(256) void cross_vec_fwd_step_kernel(const float* __restrict__ x0, int64_t x0_stride, const float* __restrict__ xl,
                                                                 int64_t xl_stride, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 int64_t batch, int d, float* __restrict__ s, float* __restrict__ xn) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    const float* r0 = x0 + b * x0_stride;
    const float* rl = xl + b * xl_stride;
    float dot = 0.f;
    for (int i = lane; i < d; i += 64) dot = fmaf(rl[i], w[i], dot);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (lane == 0) s[b] = dot;
    if (xn != nullptr)
        for (int i = lane; i < d; i += 64) xn[b * d + i] = r0[i] * dot + bias[i] + rl[i];
}

// ds[b] = g[b] . x0[b]
__global__ __launch_bounds__(256) void cross_vec_dot_kernel(const float* __restrict__ g, const float* __restrict__ x0, int64_t x0_stride,
                                                            int64_t batch, int d, float* __restrict__ ds) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= batch) return;
    const float* gr = g + b * d;
    const float* r0 = x0 + b * x0_stride;
    float dot = 0.f;
    for (int i = lane; i < d; i += 64) dot = fmaf(gr[i], r0[i], dot);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    if (lane == 0) ds[b] = dot;
}

// d b_l[i] += sum_b g[b, i];  d w_l[i] += sum_b ds[b] * x_l[b, i]: a thread owns a column over a slice of the rows (coalesced across the
// workgroup), one atomic per column and slice
__global__ __launch_bounds__(256) void cross_vec_colsum_kernel(const float* __restrict__ g, const float* __restrict__ ds,
                                                               const float* __restrict__ xl, int64_t xl_stride, int64_t batch, int d,
                                                               float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const int64_t per = (batch + gridDim.y - 1) / gridDim.y;
    const int64_t b0 = (int64_t)blockIdx.y * per, b1 = b0 + per < batch ? b0 + per : batch;
    float aw = 0.f, ab = 0.f;
    for (int64_t b = b0; b < b1; ++b) {
        ab += g[b * d + i];
        aw = fmaf(ds[b], xl[b * xl_stride + i], aw);
    }
    if (b1 > b0) {
        unsafeAtomicAdd(dw + i, aw);
        unsafeAtomicAdd(db + i, ab);
    }
}

// d x0[b, i] += g[b, i] * s_l[b];  g[b, i] += ds[b] * w_l[i]
__global__ __launch_bounds__(256) void cross_vec_update_kernel(float* __restrict__ g, float* __restrict__ dx0, const float* __restrict__ ds,
                                                               const float* __restrict__ s, const float* __restrict__ w, int64_t batch, int d) {
    const int64_t total = batch * d;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / d;
        const int i = (int)(e - b * d);
        const float gv = g[e];
        dx0[e] = fmaf(gv, s[b], dx0[e]);
        g[e] = fmaf(ds[b], w[i], gv);
    }
}

// matrix form, elementwise parts (the GEMMs are dctr_gemm): forward  x_next = x0 .* (u + b) + x_l
__global__ __launch_bounds__(256) void cross_matrix_fwd_elem_kernel(const float* __restrict__ x0, int64_t x_stride,
                                                                    const float* __restrict__ xl, int64_t xl_stride,
                                                                    const float* __restrict__ u, const float* __restrict__ bias,
                                                                    int64_t batch, int d, float* __restrict__ xn) {
    const int64_t total = batch * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / d;
        const int c = (int)(i - b * d);
        xn[i] = x0[b * x_stride + c] * (u[i] + bias[c]) + xl[b * xl_stride + c];
    }
}

// backward: du = g .* x0;  dx0 (+)= g .* (u + b)
__global__ __launch_bounds__(256) void cross_matrix_bwd_elem_kernel(const float* __restrict__ x0, int64_t x_stride,
                                                                    const float* __restrict__ g, const float* __restrict__ u,
                                                                    const float* __restrict__ bias, int64_t batch, int d,
                                                                    float* __restrict__ du, float* __restrict__ dx0) {
    const int64_t total = batch * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / d;
        const int c = (int)(i - b * d);
        const float gv = g[i];
        du[i] = gv * x0[b * x_stride + c];
        dx0[i] += gv * (u[i] + bias[c]);
    }
}

// out[b, c] (+)= src[b, c]   (strided rows)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ src, int64_t src_stride, int64_t batch, int d,
                                                       float* __restrict__ out, int64_t out_stride, int accumulate) {
    const int64_t total = batch * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / d;
        const int c = (int)(i - b * d);
        out[b * out_stride + c] = (accumulate ? out[b * out_stride + c] : 0.f) + src[b * src_stride + c];
    }
}

// ---------------------------------------------------------------------------------------------------
// CIN backward (interaction.py:277-325), first version: the reference's own formulation — z materialised per layer,
// 1x1 conv = GEMM — in the (b,d)-major row layout R = B*D:  X0t [R,F0],  X_k = Y_{k-1}[:, :Hn] [R,F_k],
//   z_k[r, i*F_k+j] = X0t[r,i] X_k[r,j];   Y_k = act(z_k W_k + b_k) [R,H_k];   out[b, .] = sum_d of the direct maps.
// The GEMMs (forward recompute, dW = z^T dpre, dz = dpre W^T) are dctr_gemm; the rest are the kernels below.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cin_to_rows_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F0,
                                                          int D, float* __restrict__ xt) {
    const int64_t total = batch * D * F0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int i = (int)(o % F0);
        const int64_t r = o / F0;
        const int64_t b = r / D;
        const int d = (int)(r - b * D);
        xt[o] = x[b * x_stride + (int64_t)i * D + d];
    }
}

__global__ __launch_bounds__(256) void cin_from_rows_kernel(const float* __restrict__ dxt, int64_t batch, int F0, int D,
                                                            float* __restrict__ dx, int64_t dx_stride, int accumulate) {
    const int64_t total = batch * F0 * D;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int d = (int)(o % D);
        const int64_t t = o / D;
        const int i = (int)(t % F0);
        const int64_t b = t / F0;
        float* dst = dx + b * dx_stride + (int64_t)i * D + d;
        *dst = (accumulate ? *dst : 0.f) + dxt[(b * D + d) * F0 + i];
    }
}

__global__ __launch_bounds__(256) void cin_outer_kernel(const float* __restrict__ x0t, int F0, const float* __restrict__ xk,
                                                        int64_t ldk, int Fk, int64_t rows, float* __restrict__ z) {
    const int K = F0 * Fk;
    const int64_t total = rows * K;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t r = o / K;
        const int c = (int)(o - r * K);
        const int i = c / Fk, j = c - i * Fk;
        z[o] = x0t[r * F0 + i] * xk[r * ldk + j];
    }
}

// the same with 16-B stores: thread = four consecutive j of one (row, i)   (Fk % 4 == 0, ldk % 4 == 0, 16-B aligned xk / z)
__global__ __launch_bounds__(256) void cin_outer4_kernel(const float* __restrict__ x0t, int F0, const float* __restrict__ xk,
                                                         int64_t ldk, int Fk, int64_t rows, float* __restrict__ z) {
    const int Fk4 = Fk >> 2, K4 = F0 * Fk4;
    const int64_t total = rows * K4;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t r = o / K4;
        const int c = (int)(o - r * K4);
        const int i = c / Fk4, j4 = c - i * Fk4;
        const float a = x0t[r * F0 + i];
        const float4 b = *reinterpret_cast<const float4*>(xk + r * ldk + 4 * j4);
        *reinterpret_cast<float4*>(z + r * (int64_t)(F0 * Fk) + i * Fk + 4 * j4) = make_float4(a * b.x, a * b.y, a * b.z, a * b.w);
    }
}

// dpre[r,h] = (hidden part: dxnext[r,h] for h < Hn) + (direct part: d_out[b, off + h - d0] for h >= d0), times act'(Y)
__global__ __launch_bounds__(256) void cin_dpre_kernel(const float* __restrict__ y, const float* __restrict__ dxnext, int64_t ldn,
                                                       int Hn, const float* __restrict__ d_out, int64_t out_dim, int off, int d0,
                                                       int64_t rows, int H, int D, int act, float* __restrict__ dpre) {
    const int64_t total = rows * H;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t r = o / H;
        const int h = (int)(o - r * H);
        float g = 0.f;
        if (dxnext != nullptr && h < Hn) g += dxnext[r * ldn + h];
        if (h >= d0) g += d_out[(r / D) * out_dim + off + (h - d0)];
        const float v = y[o];
        if (act == DCTR_ACT_RELU) g = v > 0.f ? g : 0.f;
        else if (act == DCTR_ACT_SIGMOID) g *= v * (1.f - v);
        else if (act == DCTR_ACT_TANH) g *= 1.f - v * v;
        dpre[o] = g;
    }
}

// from dz [R, F0*Fk]:  dX0t[r,i] += sum_j dz[r, i*Fk+j] Xk[r,j];   dXk[r,j] = sum_i dz[r, i*Fk+j] X0t[r,i]
// One wave per row: the row of dz (F0*Fk floats) is staged once in LDS (coalesced) and both sums read it from there,
// so dz crosses HBM once (two independent thread-per-output passes read it twice: 860 us per step at C3).
__global__ __launch_bounds__(256) void cin_outer_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x0t, int F0,
                                                            const float* __restrict__ xk, int64_t ldk, int Fk, int64_t rows,
                                                            float* __restrict__ dx0t, float* __restrict__ dxk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int K = F0 * Fk;
    float* zr = smem + (size_t)wave * (K + F0 + Fk);        // [K] dz row, [F0] x0t row, [Fk] xk row
    float* x0r = zr + K;
    float* xkr = x0r + F0;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const float* dzr = dz + r * K;
        for (int c = lane; c < K; c += 64) zr[c] = dzr[c];
        for (int i = lane; i < F0; i += 64) x0r[i] = x0t[r * F0 + i];
        for (int j = lane; j < Fk; j += 64) xkr[j] = xk[r * ldk + j];
        // (same wave: LDS operations complete in order)
        for (int i = lane; i < F0; i += 64) {
            float acc = 0.f;
            for (int j = 0; j < Fk; ++j) acc = fmaf(zr[i * Fk + j], xkr[j], acc);
            dx0t[r * F0 + i] += acc;
        }
        if (dxk != nullptr) {
            for (int j = lane; j < Fk; j += 64) {
                float acc = 0.f;
                for (int i = 0; i < F0; ++i) acc = fmaf(zr[i * Fk + j], x0r[i], acc);
                dxk[r * Fk + j] = acc;
            }
        }
    }
}

// out[i] += sum_s parts[s * n + i]
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, int64_t n, int n_parts,
                                                        float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // four independent chains: the loads of 4 slices in flight per step
        int s = 0;
        for (; s + 4 <= n_parts; s += 4) {
            a0 += parts[(int64_t)s * n + i];
            a1 += parts[(int64_t)(s + 1) * n + i];
            a2 += parts[(int64_t)(s + 2) * n + i];
            a3 += parts[(int64_t)(s + 3) * n + i];
        }
        for (; s < n_parts; ++s) a0 += parts[(int64_t)s * n + i];
        out[i] += (a0 + a1) + (a2 + a3);
    }
}

// ---------------------------------------------------------------------------------------------------
// sibling interaction layers: backward of BiInteractionPooling (NFM) and InnerProductLayer(reduce_sum) (PNN).
// One thread per (row, e); the F embeddings of the row are re-read from the forward's input (HBM/L2, F*E*4 B per row).
// ---------------------------------------------------------------------------------------------------
// y[b,e] = 0.5((sum_f x)^2 - sum_f x^2)  =>  dx[b,f,e] = dy[b,e] * (sum_f' x[b,f',e] - x[b,f,e])
// (dy_estride 1: dy [B, E] — BiInteractionPooling; 0: dy [B] broadcast over e — FM, whose logit is that pooling summed over e)
__global__ __launch_bounds__(256) void bi_interaction_bwd_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                                 int E, const float* __restrict__ dy, int64_t dy_stride,
                                                                 float* __restrict__ dx, int64_t dx_stride, int accumulate,
                                                                 int dy_estride = 1) {
    const int64_t total = batch * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / E;
        const int e = (int)(o - b * E);
        const float* xb = x + b * x_stride + e;
        float s = 0.f;
        for (int f = 0; f < F; ++f) s += xb[(int64_t)f * E];
        const float g = dy[b * dy_stride + (int64_t)e * dy_estride];
        float* db = dx + b * dx_stride + e;
        for (int f = 0; f < F; ++f) {
            const float v = g * (s - xb[(int64_t)f * E]);
            db[(int64_t)f * E] = accumulate ? db[(int64_t)f * E] + v : v;
        }
    }
}

// y[b,p(i,j)] = <x_i, x_j> (i<j, pairs ordered by i then j)  =>  dx[b,i,e] = sum_{j != i} dy[b,p(min,max)] * x[b,j,e]
__global__ __launch_bounds__(256) void inner_product_bwd_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F,
                                                                int E, const float* __restrict__ dy, int64_t dy_stride,
                                                                float* __restrict__ dx, int64_t dx_stride, int accumulate) {
    const int64_t total = batch * F * E;
    const int FE = F * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / FE;
        const int c = (int)(o - b * FE);
        const int i = c / E, e = c - i * E;
        const float* xb = x + b * x_stride + e;
        const float* gb = dy + b * dy_stride;
        float acc = 0.f;
        for (int j = 0; j < i; ++j) acc = fmaf(gb[j * (2 * F - j - 1) / 2 + (i - j - 1)], xb[(int64_t)j * E], acc);
        const int base = i * (2 * F - i - 1) / 2 - i - 1;
        for (int j = i + 1; j < F; ++j) acc = fmaf(gb[base + j], xb[(int64_t)j * E], acc);
        float* d = dx + b * dx_stride + c;
        *d = accumulate ? *d + acc : acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward of AFMLayer (interaction.py:116-146).  Same shape as the forward kernel: one wave per sample, the sample's [F,E]
// tile and the layer's weights in LDS, lanes walk the F(F-1)/2 pairs; nothing was saved by the forward, so the attention
// logits and the softmax are recomputed.  With bi_p = x_i * x_j, pre_pa = b_a + sum_e bi_pe W_ea, s_p = sum_a relu(pre_pa) h_a,
// alpha = softmax_p(s), t_p = bi_p . proj_p, y = sum_p alpha_p t_p and g = dy:
//     ds_p = alpha_p (g t_p - g y),  d pre_pa = ds_p h_a [pre_pa > 0],  d bi_pe = g alpha_p proj_p[e] + sum_a d pre_pa W_ea,
//     d x_i += d bi_p * x_j,  d x_j += d bi_p * x_i;   weight gradients are summed in LDS per workgroup, then one atomic each.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void afm_pair_ij(int p, int F, int& i, int& j) {
    int ii = 0, rem = p;
    while (rem >= F - 1 - ii) {
        rem -= F - 1 - ii;
        ++ii;
    }
    i = ii;
    j = ii + 1 + rem;
}

__global__ __launch_bounds__(256) void afm_bwd_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int F, int E,
                                                      const float* __restrict__ att_w, const float* __restrict__ att_b,
                                                      const float* __restrict__ proj_h, const float* __restrict__ proj_p, int A,
                                                      const float* __restrict__ dy, float* __restrict__ dx, int64_t dx_stride,
                                                      int accumulate, float* __restrict__ g_w, float* __restrict__ g_b,
                                                      float* __restrict__ g_h, float* __restrict__ g_p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = F * (F - 1) / 2;
    const int NW = E * A + 2 * A + E;
    float* wsh = smem;                       // [E*A] attention_W, then b[A], h[A], p[E]
    float* bsh = wsh + E * A;
    float* hsh = bsh + A;
    float* psh = hsh + A;
    float* gsh = psh + E;                    // gradients in the same order: W, b, h, p
    float* per_wave = gsh + NW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xs = per_wave + wave * (2 * F * E + P);   // [F*E] sample tile
    float* dxs = xs + F * E;                         // [F*E] its gradient
    float* alpha = dxs + F * E;                      // [P]
    for (int i = threadIdx.x; i < E * A; i += 256) wsh[i] = att_w[i];
    for (int i = threadIdx.x; i < A; i += 256) {
        bsh[i] = att_b[i];
        hsh[i] = proj_h[i];
    }
    for (int i = threadIdx.x; i < E; i += 256) psh[i] = proj_p[i];
    for (int i = threadIdx.x; i < NW; i += 256) gsh[i] = 0.f;
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    const bool valid = b < batch;
    if (valid)
        for (int i = lane; i < F * E; i += 64) {
            xs[i] = x[b * x_stride + i];
            dxs[i] = 0.f;
        }
    __syncthreads();
    if (valid) {
        // forward recompute: logits, softmax, y
        float mx = -INFINITY;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            afm_pair_ij(p, F, i, j);
            float lg = 0.f;
            for (int a = 0; a < A; ++a) {
                float t = bsh[a];
                for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e] * xs[j * E + e], wsh[e * A + a], t);
                lg = fmaf(fmaxf(t, 0.f), hsh[a], lg);
            }
            alpha[p] = lg;
            mx = fmaxf(mx, lg);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        float den = 0.f;
        for (int p = lane; p < P; p += 64) {
            const float e_ = expf(alpha[p] - mx);
            alpha[p] = e_;
            den += e_;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) den += __shfl_xor(den, m, 64);
        float yv = 0.f;
        for (int p = lane; p < P; p += 64) {
            int i, j;
            afm_pair_ij(p, F, i, j);
            const float sc = alpha[p] / den;
            alpha[p] = sc;
            float t = 0.f;
            for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e] * xs[j * E + e], psh[e], t);
            yv = fmaf(sc, t, yv);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) yv += __shfl_xor(yv, m, 64);
        // backward per pair
        const float g = dy[b];
        for (int p = lane; p < P; p += 64) {
            int i, j;
            afm_pair_ij(p, F, i, j);
            const float sc = alpha[p];
            float t = 0.f;
            for (int e = 0; e < E; ++e) t = fmaf(xs[i * E + e] * xs[j * E + e], psh[e], t);
            const float ds = sc * g * (t - yv);
            const float gsc = g * sc;
            for (int e = 0; e < E; ++e) {
                const float bi = xs[i * E + e] * xs[j * E + e];
                atomicAdd(&gsh[E * A + 2 * A + e], gsc * bi);                        // d proj_p
                const float dbi = gsc * psh[e];
                atomicAdd(&dxs[i * E + e], dbi * xs[j * E + e]);
                atomicAdd(&dxs[j * E + e], dbi * xs[i * E + e]);
            }
            for (int a = 0; a < A; ++a) {
                float pre = bsh[a];
                for (int e = 0; e < E; ++e) pre = fmaf(xs[i * E + e] * xs[j * E + e], wsh[e * A + a], pre);
                if (pre > 0.f) {
                    atomicAdd(&gsh[E * A + A + a], ds * pre);                        // d proj_h
                    const float dpre = ds * hsh[a];
                    atomicAdd(&gsh[E * A + a], dpre);                                // d attention_b
                    for (int e = 0; e < E; ++e) {
                        const float xi = xs[i * E + e], xj = xs[j * E + e];
                        atomicAdd(&gsh[e * A + a], dpre * xi * xj);                  // d attention_W
                        const float dbi = dpre * wsh[e * A + a];
                        atomicAdd(&dxs[i * E + e], dbi * xj);
                        atomicAdd(&dxs[j * E + e], dbi * xi);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (valid) {
        float* d = dx + b * dx_stride;
        for (int i = lane; i < F * E; i += 64) d[i] = accumulate ? d[i] + dxs[i] : dxs[i];
    }
    for (int i = threadIdx.x; i < NW; i += 256) {
        const float v = gsh[i];
        if (v != 0.f) {
            float* dst = i < E * A ? g_w + i : i < E * A + A ? g_b + (i - E * A) : i < E * A + 2 * A ? g_h + (i - E * A - A)
                                                                                                   : g_p + (i - E * A - 2 * A);
            unsafeAtomicAdd(dst, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Dice (layers/activation.py:59-64, inference statistics):  y = z (alpha + (1 - alpha) p),  p = sigmoid((z - mean) / sqrt(var + eps))
//   dy/dz = alpha + (1 - alpha) p + z (1 - alpha) p (1 - p) / sqrt(var + eps);   dy/dalpha = z (1 - p)
// z is not recoverable from y, so dctr_mlp_bwd recomputes Z = X W with one more GEMM and this kernel adds the bias.
// in place: dh[b,n] (gradient w.r.t. y) -> gradient w.r.t. z;  d_alpha[n] += sum_b dh[b,n] z (1 - p)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_bwd_kernel(float* __restrict__ dh, const float* __restrict__ zw, const float* __restrict__ bias,
                                                       const float* __restrict__ alpha, const float* __restrict__ mean,
                                                       const float* __restrict__ var, float eps, int64_t batch, int N,
                                                       float* __restrict__ d_alpha) {
    for (int n = threadIdx.x; n < N; n += 256) {
        const float bn = bias != nullptr ? bias[n] : 0.f, al = alpha[n], mu = mean[n];
        const float rs = 1.f / sqrtf(var[n] + eps);
        float acc = 0.f;
        for (int64_t r0 = (int64_t)blockIdx.x * BWD_ROWS; r0 < batch; r0 += (int64_t)gridDim.x * BWD_ROWS)     // grid-stride: few workgroups when column sums follow
        for (int r = 0; r < BWD_ROWS; ++r) {
            const int64_t b = r0 + r;
            if (b >= batch) break;
            const float z = zw[b * N + n] + bn;
            const float p = 1.f / (1.f + expf(-(z - mu) * rs));
            const float d = dh[b * N + n];
            acc = fmaf(d, z * (1.f - p), acc);
            dh[b * N + n] = d * (al + (1.f - al) * p + z * (1.f - al) * p * (1.f - p) * rs);
        }
        if (d_alpha != nullptr) unsafeAtomicAdd(d_alpha + n, acc);
    }
}

// ---------------------------------------------------------------------------------------------------
// Dice under training=True (layers/activation.py:51-64: BatchNormalization(center=False, scale=False, epsilon=1e-9) called with
// the training flag): the statistics are those of THIS batch over every row (biased variance), gradients flow through them,
// and the stored statistics move towards them with the layer's momentum.
//   xn = (z - mu) rs,  p = sigmoid(xn),  y = z (alpha + (1 - alpha) p)
//   dy/dz (direct) = alpha + (1 - alpha) p;   dxn = dy * z (1 - alpha) * p (1 - p);   dalpha = sum dy * z (1 - p)
//   dz = dy * direct + rs * (dxn - mean_rows(dxn) - xn * mean_rows(dxn * xn))          (BatchNormalization backward)
// Column statistics: two passes over z (sum -> mean, then sum of squared deviations): E[z^2] - E[z]^2 would cancel in fp32.
// ---------------------------------------------------------------------------------------------------
// pass kind 0: acc[n] += sum_b (z + bias);  kind 1: acc[n] += sum_b (z + bias - mean)^2     (mean = acc0 / rows)
__global__ __launch_bounds__(256) void dice_colstat_kernel(const float* __restrict__ z, int64_t z_stride, const float* __restrict__ bias,
                                                          int64_t rows, int N, int kind, const float* __restrict__ sum0,
                                                          float* __restrict__ acc) {
    const float inv = 1.f / (float)rows;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float bn = bias != nullptr ? bias[n] : 0.f;
        const float mu = kind ? sum0[n] * inv : 0.f;
        float a = 0.f;
        for (int64_t r0 = (int64_t)blockIdx.x * BWD_ROWS; r0 < rows; r0 += (int64_t)gridDim.x * BWD_ROWS)     // grid-stride: few workgroups when column sums follow
        for (int r = 0; r < BWD_ROWS; ++r) {
            const int64_t b = r0 + r;
            if (b >= rows) break;
            const float v = z[b * z_stride + n] + bn - mu;
            a += kind ? v * v : v;
        }
        unsafeAtomicAdd(acc + n, a);
    }
}

// sums -> batch mean / biased variance (in place), stored statistics moved towards them
__global__ void dice_stat_finish_kernel(float* __restrict__ mean_sum, float* __restrict__ var_sum, int64_t rows, int N, float momentum,
                                        float* __restrict__ moving_mean, float* __restrict__ moving_var) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float inv = 1.f / (float)rows;
    const float m = mean_sum[n] * inv, v = var_sum[n] * inv;
    mean_sum[n] = m;
    var_sum[n] = v;
    if (moving_mean != nullptr) moving_mean[n] = moving_mean[n] * momentum + m * (1.f - momentum);
    if (moving_var != nullptr) moving_var[n] = moving_var[n] * momentum + v * (1.f - momentum);
}

__global__ __launch_bounds__(256) void dice_apply_kernel(const float* __restrict__ z, int64_t z_stride, const float* __restrict__ bias,
                                                        const float* __restrict__ alpha, const float* __restrict__ mean,
                                                        const float* __restrict__ var, float eps, int64_t rows, int N,
                                                        float* __restrict__ h, int64_t h_stride) {
    const int64_t total = rows * N;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / N;
        const int n = (int)(o - b * N);
        const float zz = z[b * z_stride + n] + (bias != nullptr ? bias[n] : 0.f);
        const float p = 1.f / (1.f + expf(-(zz - mean[n]) / sqrtf(var[n] + eps)));
        h[b * h_stride + n] = zz * (alpha[n] + (1.f - alpha[n]) * p);
    }
}

// backward pass 1: S1[n] += sum dxn, S2[n] += sum dxn xn, d_alpha[n] += sum dy z (1 - p)      (zw = X W without the bias)
__global__ __launch_bounds__(256) void dice_train_bwd_reduce_kernel(const float* __restrict__ dh, const float* __restrict__ zw,
                                                                   const float* __restrict__ bias, const float* __restrict__ alpha,
                                                                   const float* __restrict__ mean, const float* __restrict__ var,
                                                                   float eps, int64_t batch, int N, float* __restrict__ s12,
                                                                   float* __restrict__ d_alpha) {
    for (int n = threadIdx.x; n < N; n += 256) {
        const float bn = bias != nullptr ? bias[n] : 0.f, al = alpha[n], mu = mean[n];
        const float rs = 1.f / sqrtf(var[n] + eps);
        float a1 = 0.f, a2 = 0.f, aa = 0.f;
        for (int64_t r0 = (int64_t)blockIdx.x * BWD_ROWS; r0 < batch; r0 += (int64_t)gridDim.x * BWD_ROWS)     // grid-stride: few workgroups when column sums follow
        for (int r = 0; r < BWD_ROWS; ++r) {
            const int64_t b = r0 + r;
            if (b >= batch) break;
            const float z = zw[b * N + n] + bn;
            const float xn = (z - mu) * rs;
            const float p = 1.f / (1.f + expf(-xn));
            const float d = dh[b * N + n];
            const float dxn = d * z * (1.f - al) * p * (1.f - p);
            a1 += dxn;
            a2 = fmaf(dxn, xn, a2);
            aa = fmaf(d, z * (1.f - p), aa);
        }
        unsafeAtomicAdd(s12 + n, a1);
        unsafeAtomicAdd(s12 + N + n, a2);
        if (d_alpha != nullptr) unsafeAtomicAdd(d_alpha + n, aa);
    }
}

// backward pass 2, in place: dh (gradient w.r.t. y) -> gradient w.r.t. z
__global__ __launch_bounds__(256) void dice_train_bwd_apply_kernel(float* __restrict__ dh, const float* __restrict__ zw,
                                                                  const float* __restrict__ bias, const float* __restrict__ alpha,
                                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                                  float eps, int64_t batch, int N, const float* __restrict__ s12) {
    const float inv = 1.f / (float)batch;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float bn = bias != nullptr ? bias[n] : 0.f, al = alpha[n], mu = mean[n];
        const float rs = 1.f / sqrtf(var[n] + eps);
        const float m1 = s12[n] * inv, m2 = s12[N + n] * inv;
        for (int64_t r0 = (int64_t)blockIdx.x * BWD_ROWS; r0 < batch; r0 += (int64_t)gridDim.x * BWD_ROWS)     // grid-stride: few workgroups when column sums follow
        for (int r = 0; r < BWD_ROWS; ++r) {
            const int64_t b = r0 + r;
            if (b >= batch) break;
            const float z = zw[b * N + n] + bn;
            const float xn = (z - mu) * rs;
            const float p = 1.f / (1.f + expf(-xn));
            const float d = dh[b * N + n];
            const float dxn = d * z * (1.f - al) * p * (1.f - p);
            dh[b * N + n] = d * (al + (1.f - al) * p) + rs * (dxn - m1 - xn * m2);
        }
    }
}

// The three Dice passes that end in column sums, in the 16-B / row-lane layout of act_bwd_colsum4_kernel (N % 4 == 0, N <= 1024):
// DIN runs them on B*T = 102,400 rows x 80 / 40 columns, where one workgroup per 16 rows meant 6,400 atomics per column address
// (140-190 us per pass, all of it the serialised atomics).
template <int K>
__device__ __forceinline__ void colsum4_finish_k(const float (&acc)[K][4], int N4, int RL, int c, float* const (&out)[K]) {
    __shared__ float red[K][256][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[k][threadIdx.x][j] = acc[k][j];
    __syncthreads();
    if ((int)threadIdx.x < N4) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (out[k] == nullptr) continue;
            float t[4] = {red[k][c][0], red[k][c][1], red[k][c][2], red[k][c][3]};
            for (int r = 1; r < RL; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] += red[k][r * N4 + c][j];
#pragma unroll
            for (int j = 0; j < 4; ++j) unsafeAtomicAdd(out[k] + 4 * c + j, t[j]);
        }
    }
}

struct DiceCols {            // per-thread constants of its four columns
    float bn[4], al[4], mu[4], rs[4];
};
__device__ __forceinline__ DiceCols dice_cols(const float* bias, const float* alpha, const float* mean, const float* var, float eps, int c) {
    DiceCols k;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 4 * c + j;
        k.bn[j] = bias != nullptr ? bias[n] : 0.f;
        k.al[j] = alpha != nullptr ? alpha[n] : 0.f;
        k.mu[j] = mean != nullptr ? mean[n] : 0.f;
        k.rs[j] = var != nullptr ? 1.f / sqrtf(var[n] + eps) : 1.f;
    }
    return k;
}

__global__ __launch_bounds__(256) void dice_colstat4_kernel(const float* __restrict__ z, int64_t z_stride, const float* __restrict__ bias,
                                                           int64_t rows, int N, int kind, const float* __restrict__ sum0,
                                                           float* __restrict__ acc_out) {
    const int N4 = N >> 2, RL = 256 / N4;
    const int c = threadIdx.x % N4, rl = threadIdx.x / N4;
    float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
    if (rl < RL) {
        const float inv = 1.f / (float)rows;
        float sh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[j] = (bias != nullptr ? bias[4 * c + j] : 0.f) - (kind ? sum0[4 * c + j] * inv : 0.f);
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < rows; r += (int64_t)gridDim.x * RL) {
            const float4 t = *reinterpret_cast<const float4*>(z + r * z_stride + 4 * c);
            const float v[4] = {t.x + sh[0], t.y + sh[1], t.z + sh[2], t.w + sh[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[0][j] += kind ? v[j] * v[j] : v[j];
        }
    }
    float* const outs[1] = {acc_out};
    colsum4_finish_k<1>(acc, N4, RL, c, outs);
}

__global__ __launch_bounds__(256) void dice_bwd4_kernel(float* __restrict__ dh, const float* __restrict__ zw, const float* __restrict__ bias,
                                                        const float* __restrict__ alpha, const float* __restrict__ mean,
                                                        const float* __restrict__ var, float eps, int64_t batch, int N,
                                                        float* __restrict__ d_alpha) {
    const int N4 = N >> 2, RL = 256 / N4;
    const int c = threadIdx.x % N4, rl = threadIdx.x / N4;
    float acc[1][4] = {{0.f, 0.f, 0.f, 0.f}};
    if (rl < RL) {
        const DiceCols k = dice_cols(bias, alpha, mean, var, eps, c);
#pragma unroll 2
        for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < batch; r += (int64_t)gridDim.x * RL) {
            const float4 zt = *reinterpret_cast<const float4*>(zw + r * N + 4 * c);
            const float4 dt = *reinterpret_cast<const float4*>(dh + r * N + 4 * c);
            const float zz[4] = {zt.x, zt.y, zt.z, zt.w}, dd[4] = {dt.x, dt.y, dt.z, dt.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = zz[j] + k.bn[j];
                const float p = 1.f / (1.f + expf(-(z - k.mu[j]) * k.rs[j]));
                acc[0][j] = fmaf(dd[j], z * (1.f - p), acc[0][j]);
                o[j] = dd[j] * (k.al[j] + (1.f - k.al[j]) * p + z * (1.f - k.al[j]) * p * (1.f - p) * k.rs[j]);
            }
            *reinterpret_cast<float4*>(dh + r * N + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    float* const outs[1] = {d_alpha};
    colsum4_finish_k<1>(acc, N4, RL, c, outs);
}

__global__ __launch_bounds__(256) void dice_train_bwd_reduce4_kernel(const float* __restrict__ dh, const float* __restrict__ zw,
                                                                    const float* __restrict__ bias, const float* __restrict__ alpha,
                                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                                    float eps, int64_t batch, int N, float* __restrict__ s12,
                                                                    float* __restrict__ d_alpha) {
    const int N4 = N >> 2, RL = 256 / N4;
    const int c = threadIdx.x % N4, rl = threadIdx.x / N4;
    float acc[3][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (rl < RL) {
        const DiceCols k = dice_cols(bias, alpha, mean, var, eps, c);
#pragma unroll 2
        for (int64_t r = (int64_t)blockIdx.x * RL + rl; r < batch; r += (int64_t)gridDim.x * RL) {
            const float4 zt = *reinterpret_cast<const float4*>(zw + r * N + 4 * c);
            const float4 dt = *reinterpret_cast<const float4*>(dh + r * N + 4 * c);
            const float zz[4] = {zt.x, zt.y, zt.z, zt.w}, dd[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float z = zz[j] + k.bn[j];
                const float xn = (z - k.mu[j]) * k.rs[j];
                const float p = 1.f / (1.f + expf(-xn));
                const float dxn = dd[j] * z * (1.f - k.al[j]) * p * (1.f - p);
                acc[0][j] += dxn;
                acc[1][j] = fmaf(dxn, xn, acc[1][j]);
                acc[2][j] = fmaf(dd[j], z * (1.f - p), acc[2][j]);
            }
        }
    }
    float* const outs[3] = {s12, s12 + N, d_alpha};
    colsum4_finish_k<3>(acc, N4, RL, c, outs);
}

static bool rowlane4_ok(int N, const void* a, const void* b, int64_t stride) {
    return N % 4 == 0 && N >= 4 && N <= 1024 && stride % 4 == 0 && dctr_aligned16(a) && (b == nullptr || dctr_aligned16(b));
}

extern "C" int dctr_dice_train_fwd(const float* z, int64_t z_stride, const float* bias, int64_t rows, int32_t n, const float* alpha,
                                   float eps, float momentum, float* moving_mean, float* moving_var, float* batch_mean,
                                   float* batch_var, float* h, int64_t h_stride, void* stream) {
    DCTR_REQUIRE(z && alpha && batch_mean && batch_var && h, DCTR_E_NULL, "dice_train_fwd: null pointer");
    DCTR_REQUIRE(rows >= 1 && n >= 1 && z_stride >= n && h_stride >= n, DCTR_E_DIM, "dice_train_fwd: bad sizes (rows=%lld n=%d)",
                 (long long)rows, n);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(batch_mean, 0, (size_t)n * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(batch_var, 0, (size_t)n * sizeof(float), st);
    DCTR_REQUIRE(e == hipSuccess, (int)e, "dice_train_fwd: memset failed: %s", hipGetErrorString(e));
    if (rowlane4_ok((int)n, z, nullptr, z_stride)) {
        const unsigned g = colsum_grid(rows, 256 / ((int)n / 4));
        hipLaunchKernelGGL(dice_colstat4_kernel, dim3(g), dim3(256), 0, st, z, z_stride, bias, rows, (int)n, 0, (const float*)nullptr, batch_mean);
        hipLaunchKernelGGL(dice_colstat4_kernel, dim3(g), dim3(256), 0, st, z, z_stride, bias, rows, (int)n, 1, (const float*)batch_mean, batch_var);
    } else {
        const unsigned rb = rows_grid(rows, true);
        hipLaunchKernelGGL(dice_colstat_kernel, dim3(rb), dim3(256), 0, st, z, z_stride, bias, rows, (int)n, 0, (const float*)nullptr, batch_mean);
        hipLaunchKernelGGL(dice_colstat_kernel, dim3(rb), dim3(256), 0, st, z, z_stride, bias, rows, (int)n, 1, (const float*)batch_mean, batch_var);
    }
    hipLaunchKernelGGL(dice_stat_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, st, batch_mean, batch_var, rows, (int)n, momentum,
                       moving_mean, moving_var);
    const int64_t total = rows * n;
    const unsigned blocks = (unsigned)(dctr_ceil_div(total, (int64_t)256) < 4096 ? dctr_ceil_div(total, (int64_t)256) : 4096);
    hipLaunchKernelGGL(dice_apply_kernel, dim3(blocks), dim3(256), 0, st, z, z_stride, bias, alpha, (const float*)batch_mean,
                       (const float*)batch_var, eps, rows, (int)n, h, h_stride);
    return dctr_launch_status("dctr_dice_train_fwd");
}

// ---------------------------------------------------------------------------------------------------
// DNN layer under training=True with BatchNormalization / Dropout (include/dctr.h: dctr_dnn_train_layer_t; reference
// layers/core.py:196-208).  Elementwise passes: thread = column, rows grid-strided over blockIdx.x (coalesced along n); the column
// sums of the BatchNormalization backward leave as one atomic per column and workgroup (few workgroups, long row loops).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, float rate) {
    unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed;                    // splitmix64 finaliser over (seed, element)
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)(unsigned)(x >> 40) * (1.f / 16777216.f) >= rate;               // 24 uniform bits
}

struct TrainLayerCol {
    float mu, rs, g, bt;
};
__device__ __forceinline__ TrainLayerCol train_layer_col(const dctr_dnn_train_layer_t& a, int n) {
    TrainLayerCol c{0.f, 1.f, 1.f, 0.f};
    if (a.use_bn) {
        c.mu = a.bn_batch_mean[n];
        c.rs = 1.f / sqrtf(a.bn_batch_var[n] + a.bn_eps);
        c.g = a.bn_gamma != nullptr ? a.bn_gamma[n] : 1.f;
        c.bt = a.bn_beta != nullptr ? a.bn_beta[n] : 0.f;
    }
    return c;
}
__device__ __forceinline__ float act_value(float y, int act) {
    if (act == DCTR_ACT_RELU) return fmaxf(y, 0.f);
    if (act == DCTR_ACT_SIGMOID) return 1.f / (1.f + expf(-y));
    if (act == DCTR_ACT_TANH) return tanhf(y);
    return y;
}
__device__ __forceinline__ float act_deriv(float y, int act) {
    if (act == DCTR_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == DCTR_ACT_SIGMOID) { const float s = 1.f / (1.f + expf(-y)); return s * (1.f - s); }
    if (act == DCTR_ACT_TANH) { const float t = tanhf(y); return 1.f - t * t; }
    return 1.f;
}

__global__ __launch_bounds__(256) void train_layer_fwd_kernel(dctr_dnn_train_layer_t a) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= a.n) return;
    const TrainLayerCol c = train_layer_col(a, n);
    const float scale = a.dropout_rate > 0.f ? 1.f / (1.f - a.dropout_rate) : 1.f;
    for (int64_t b = blockIdx.x; b < a.rows; b += gridDim.x) {
        const float z = a.z[b * a.z_stride + n];
        const float y = a.use_bn ? c.g * ((z - c.mu) * c.rs) + c.bt : z;
        float h = act_value(y, a.activation);
        if (a.dropout_rate > 0.f) h = dropout_keep(a.dropout_seed, (unsigned long long)b * a.n + n, a.dropout_rate) ? h * scale : 0.f;
        a.h[b * a.h_stride + n] = h;
    }
}

// dy of one element (shared by the two backward passes)
__device__ __forceinline__ float train_layer_dy(const dctr_dnn_train_layer_t& a, const TrainLayerCol& c, int64_t b, int n, float scale,
                                                float& xh) {
    const float z = a.z[b * a.z_stride + n];
    xh = (z - c.mu) * c.rs;
    const float y = a.use_bn ? c.g * xh + c.bt : z;
    float d = a.dh[b * a.dh_stride + n];
    if (a.dropout_rate > 0.f) d = dropout_keep(a.dropout_seed, (unsigned long long)b * a.n + n, a.dropout_rate) ? d * scale : 0.f;
    return d * act_deriv(y, a.activation);
}

// BatchNormalization backward, pass 1: ws[n] += sum_b dy, ws[N + n] += sum_b dy xhat
__global__ __launch_bounds__(256) void train_layer_bwd_reduce_kernel(dctr_dnn_train_layer_t a) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= a.n) return;
    const TrainLayerCol c = train_layer_col(a, n);
    const float scale = a.dropout_rate > 0.f ? 1.f / (1.f - a.dropout_rate) : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t b = blockIdx.x; b < a.rows; b += gridDim.x) {
        float xh;
        const float dy = train_layer_dy(a, c, b, n, scale, xh);
        s1 += dy;
        s2 = fmaf(dy, xh, s2);
    }
    unsafeAtomicAdd(a.workspace + n, s1);
    unsafeAtomicAdd(a.workspace + a.n + n, s2);
}

// pass 2 (the only pass without BatchNormalization): dz; blockIdx.x == 0 also adds the sums to d_beta / d_gamma
__global__ __launch_bounds__(256) void train_layer_bwd_apply_kernel(dctr_dnn_train_layer_t a) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= a.n) return;
    const TrainLayerCol c = train_layer_col(a, n);
    const float scale = a.dropout_rate > 0.f ? 1.f / (1.f - a.dropout_rate) : 1.f;
    float m1 = 0.f, m2 = 0.f;
    if (a.use_bn) {
        const float s1 = a.workspace[n], s2 = a.workspace[a.n + n];
        m1 = s1 / (float)a.rows;
        m2 = s2 / (float)a.rows;
        if (blockIdx.x == 0) {
            if (a.d_beta != nullptr) a.d_beta[n] += s1;
            if (a.d_gamma != nullptr) a.d_gamma[n] += s2;
        }
    }
    for (int64_t b = blockIdx.x; b < a.rows; b += gridDim.x) {
        float xh;
        const float dy = train_layer_dy(a, c, b, n, scale, xh);
        a.dz[b * a.n + n] = a.use_bn ? c.g * c.rs * (dy - m1 - xh * m2) : dy;
    }
}

static int train_layer_check(const dctr_dnn_train_layer_t* a, const char* what) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "%s: null args", what);
    DCTR_REQUIRE(a->rows >= 0 && a->n >= 1 && a->z_stride >= a->n, DCTR_E_DIM, "%s: bad sizes (rows=%lld n=%d z_stride=%lld)", what,
                 (long long)a->rows, a->n, (long long)a->z_stride);
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_TANH, DCTR_E_ENUM, "%s: activation %d", what, a->activation);
    DCTR_REQUIRE(a->dropout_rate >= 0.f && a->dropout_rate < 1.f, DCTR_E_DIM, "%s: dropout_rate %g outside [0, 1)", what, (double)a->dropout_rate);
    DCTR_REQUIRE(a->z != nullptr, DCTR_E_NULL, "%s: null z", what);
    DCTR_REQUIRE(!a->use_bn || (a->bn_batch_mean != nullptr && a->bn_batch_var != nullptr), DCTR_E_NULL,
                 "%s: use_bn needs bn_batch_mean / bn_batch_var", what);
    return DCTR_OK;
}

extern "C" int dctr_dnn_train_layer_fwd(const dctr_dnn_train_layer_t* a, void* stream) {
    const int rc = train_layer_check(a, "dnn_train_layer_fwd");
    if (rc != DCTR_OK) return rc;
    if (a->rows == 0) return DCTR_OK;
    DCTR_REQUIRE(a->h != nullptr && a->h_stride >= a->n, DCTR_E_NULL, "dnn_train_layer_fwd: h missing or h_stride < n");
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    if (a->use_bn) {
        // this batch's statistics: two passes (sum -> mean, then squared deviations), then the stored statistics move towards them
        hipError_t e = hipMemsetAsync(a->bn_batch_mean, 0, (size_t)n * sizeof(float), st);
        if (e == hipSuccess) e = hipMemsetAsync(a->bn_batch_var, 0, (size_t)n * sizeof(float), st);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "dnn_train_layer_fwd: memset failed: %s", hipGetErrorString(e));
        if (rowlane4_ok(n, a->z, nullptr, a->z_stride)) {
            const unsigned g = colsum_grid(a->rows, 256 / (n / 4));
            hipLaunchKernelGGL(dice_colstat4_kernel, dim3(g), dim3(256), 0, st, a->z, a->z_stride, (const float*)nullptr, a->rows, n, 0, (const float*)nullptr, a->bn_batch_mean);
            hipLaunchKernelGGL(dice_colstat4_kernel, dim3(g), dim3(256), 0, st, a->z, a->z_stride, (const float*)nullptr, a->rows, n, 1, (const float*)a->bn_batch_mean, a->bn_batch_var);
        } else {
            const unsigned rb = rows_grid(a->rows, true);
            hipLaunchKernelGGL(dice_colstat_kernel, dim3(rb), dim3(256), 0, st, a->z, a->z_stride, (const float*)nullptr, a->rows, n, 0, (const float*)nullptr, a->bn_batch_mean);
            hipLaunchKernelGGL(dice_colstat_kernel, dim3(rb), dim3(256), 0, st, a->z, a->z_stride, (const float*)nullptr, a->rows, n, 1, (const float*)a->bn_batch_mean, a->bn_batch_var);
        }
        hipLaunchKernelGGL(dice_stat_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a->bn_batch_mean, a->bn_batch_var, a->rows, n,
                           a->bn_momentum, a->bn_moving_mean, a->bn_moving_var);
    }
    const unsigned gx = (unsigned)(a->rows < 2048 ? a->rows : 2048);
    hipLaunchKernelGGL(train_layer_fwd_kernel, dim3(gx, (unsigned)((n + 255) / 256)), dim3(256), 0, st, *a);
    return dctr_launch_status("dctr_dnn_train_layer_fwd");
}

extern "C" int dctr_dnn_train_layer_bwd(const dctr_dnn_train_layer_t* a, void* stream) {
    const int rc = train_layer_check(a, "dnn_train_layer_bwd");
    if (rc != DCTR_OK) return rc;
    if (a->rows == 0) return DCTR_OK;
    DCTR_REQUIRE(a->dh != nullptr && a->dz != nullptr && a->dh_stride >= a->n, DCTR_E_NULL, "dnn_train_layer_bwd: dh / dz missing or dh_stride < n");
    DCTR_REQUIRE(a->dz != a->dh || a->dh_stride == a->n, DCTR_E_DIM, "dnn_train_layer_bwd: in place needs dh_stride == n");
    DCTR_REQUIRE(!a->use_bn || a->workspace != nullptr, DCTR_E_NULL, "dnn_train_layer_bwd: use_bn needs a workspace of 2 n floats");
    hipStream_t st = (hipStream_t)stream;
    const int n = a->n;
    const unsigned gy = (unsigned)((n + 255) / 256);
    if (a->use_bn) {
        hipError_t e = hipMemsetAsync(a->workspace, 0, (size_t)2 * n * sizeof(float), st);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "dnn_train_layer_bwd: memset failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(train_layer_bwd_reduce_kernel, dim3((unsigned)(a->rows < COLSUM_MAX_WG ? a->rows : COLSUM_MAX_WG), gy), dim3(256), 0, st, *a);
    }
    const unsigned gx = (unsigned)(a->rows < 2048 ? a->rows : 2048);
    hipLaunchKernelGGL(train_layer_bwd_apply_kernel, dim3(gx, gy), dim3(256), 0, st, *a);
    return dctr_launch_status("dctr_dnn_train_layer_bwd");
}

// ---------------------------------------------------------------------------------------------------
// DIN's LocalActivationUnit as a training step (layers/core.py:94-108, layers/sequence.py:261-298): the attention input
// [q, k, q - k, q * k] is materialised once per batch ([B*T, 4E]) so that the attention MLP runs through dctr_mlp_fwd /
// dctr_mlp_bwd with saved activations; the masked weighted sum and the scatter of the key gradients are kernels here.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void din_att_in_kernel(const float* __restrict__ q, const float* __restrict__ k, int64_t rows, int T,
                                                         int E, float* __restrict__ a) {
    // one thread per (row = b*T + t, e)
    const int64_t total = rows * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t r = o / E;
        const int e = (int)(o - r * E);
        const float qv = q[(r / T) * E + e], kv = k[o];
        float* ar = a + r * 4 * E;
        ar[e] = qv;
        ar[E + e] = kv;
        ar[2 * E + e] = qv - kv;
        ar[3 * E + e] = qv * kv;
    }
}

// out[b, e] = sum_t (mask ? score : 0) k[b,t,e]                         (weight_normalization=False, sequence.py:286-296)
__global__ __launch_bounds__(256) void din_wsum_kernel(const float* __restrict__ score, const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ k, int64_t batch, int T, int E,
                                                       float* __restrict__ out, int64_t out_stride) {
    const int64_t total = batch * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / E;
        const int e = (int)(o - b * E);
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            const float s = mask[b * T + t] ? score[b * T + t] : 0.f;
            acc = fmaf(s, k[(b * T + t) * E + e], acc);
        }
        out[b * out_stride + e] = acc;
    }
}

// d_score[b,t] = mask ? <d_out[b,:], k[b,t,:]> : 0;  dk[b,t,:] = (mask ? score : 0) d_out[b,:] (written);
// d_bias += sum d_score (the bias of the unit's final Dense(1));  one wave per (b, t)
__global__ __launch_bounds__(256) void din_wsum_bwd_kernel(const float* __restrict__ d_out, int64_t d_stride,
                                                           const float* __restrict__ score, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ k, int64_t batch, int T, int E,
                                                           float* __restrict__ d_score, float* __restrict__ dk,
                                                           float* __restrict__ d_bias) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = batch * T;
    float bsum = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t b = r / T;
        const bool m = mask[r] != 0;
        const float s = m ? score[r] : 0.f;
        float dot = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float g = d_out[b * d_stride + e];
            dot = fmaf(g, k[r * E + e], dot);
            dk[r * E + e] = s * g;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 64);
        const float ds = m ? dot : 0.f;
        if (lane == 0) d_score[r] = ds;
        bsum += ds;
    }
    // one atomic per workgroup (B*T / 4 of them on ONE address serialised to 0.5 ms at C4)
    __shared__ float wsum[4];
    if (lane == 0) wsum[threadIdx.x >> 6] = bsum;
    __syncthreads();
    if (d_bias != nullptr && threadIdx.x == 0) {
        const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t != 0.f) unsafeAtomicAdd(d_bias, t);
    }
}

// out[0] += sum_i v[i]: grid-stride, wave sums, one atomic per workgroup
__global__ __launch_bounds__(256) void sum_vec_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ out) {
    __shared__ float ws4[4];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += v[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) ws4[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (ws4[0] + ws4[1]) + (ws4[2] + ws4[3]);
        if (t != 0.f) unsafeAtomicAdd(out, t);
    }
}

// da [B*T, 4E] -> dq[b,e] = sum_t (d0 + d2 + d3 k), added into dx[b, qcol[e]];  dk[b,t,e] += d1 - d2 + d3 q
__global__ __launch_bounds__(256) void din_att_in_bwd_kernel(const float* __restrict__ da, const float* __restrict__ q,
                                                             const float* __restrict__ k, int64_t batch, int T, int E,
                                                             float* __restrict__ dk, float* __restrict__ dx, int64_t dx_stride,
                                                             const int32_t* __restrict__ qcol) {
    const int64_t total = batch * E;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / E;
        const int e = (int)(o - b * E);
        const float qv = q[o];
        float dq = 0.f;
        for (int t = 0; t < T; ++t) {
            const int64_t r = b * T + t;
            const float* ar = da + r * 4 * E;
            const float d0 = ar[e], d1 = ar[E + e], d2 = ar[2 * E + e], d3 = ar[3 * E + e];
            const float kv = k[r * E + e];
            dq += d0 + d2 + d3 * kv;
            dk[r * E + e] += d1 - d2 + d3 * qv;
        }
        dx[b * dx_stride + qcol[e]] += dq;
    }
}

// backward of dctr_embed_lookup: g_table[row(idx[i]), :] += d_out[i, :dim]   (rows out of range are skipped, as the forward
// zero-fills them and raises the status flag)
__global__ __launch_bounds__(256) void lookup_bwd_kernel(dctr_lookup_args_t a, const float* __restrict__ d_out, int64_t d_stride,
                                                         float* __restrict__ g_table) {
    const int64_t total = a.n * a.dim;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t i = o / a.dim;
        const int c = (int)(o - i * a.dim);
        const float g = d_out[i * d_stride + c];
        if (g == 0.f) continue;
        const int64_t row = resolve_row(read_id(a.idx, i, a.idx_is_i64), a.hash_mode, a.idx_is_i64, a.vocab);
        if ((uint64_t)row < (uint64_t)a.vocab) unsafeAtomicAdd(g_table + row * a.dim + c, g);
    }
}

// The same scatter for skewed ids (DIN's behaviour sequences: half of the B*T positions are padding id 0, and under training-mode
// Dice the padded keys DO receive gradient through the batch statistics — 51,000 x 32 atomics on ONE row took 1.6 ms per table):
// a workgroup sorts a tile of LB_TILE positions by row in LDS (bitonic), then 64 / dimP walkers per wave run over contiguous
// stretches of the sorted tile, summing runs of equal rows in registers: one atomic per (run, column) instead of one per position.
constexpr int LB_TILE = 1024;
__global__ __launch_bounds__(256) void lookup_bwd_tile_kernel(dctr_lookup_args_t a, const float* __restrict__ d_out, int64_t d_stride,
                                                              float* __restrict__ g_table, int dimP) {
    __shared__ uint32_t key[LB_TILE];
    __shared__ uint16_t pos[LB_TILE];
    constexpr uint32_t NONE = 0xffffffffu;
    const int64_t base = (int64_t)blockIdx.x * LB_TILE;
    for (int j = threadIdx.x; j < LB_TILE; j += 256) {
        const int64_t i = base + j;
        uint32_t k = NONE;
        if (i < a.n) {
            const int64_t row = resolve_row(read_id(a.idx, i, a.idx_is_i64), a.hash_mode, a.idx_is_i64, a.vocab);
            if ((uint64_t)row < (uint64_t)a.vocab) k = (uint32_t)row;
        }
        key[j] = k;
        pos[j] = (uint16_t)j;
    }
    __syncthreads();
    for (int size = 2; size <= LB_TILE; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < LB_TILE / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t k0 = key[lo], k1 = key[hi];
                if ((k0 > k1) == up) {
                    key[lo] = k1; key[hi] = k0;
                    const uint16_t p0 = pos[lo]; pos[lo] = pos[hi]; pos[hi] = p0;
                }
            }
            __syncthreads();
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub_n = 64 / dimP, sub = lane / dimP, c = lane % dimP;
    const int per = LB_TILE / (4 * sub_n);
    const int j0 = (wave * sub_n + sub) * per;
    const bool col = c < a.dim;
    uint32_t cur = NONE;
    float acc = 0.f;
    constexpr int U = 8;
    for (int j = j0; j < j0 + per; j += U) {
        uint32_t kk[U];
        float gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kk[u] = key[j + u];
            const int64_t i = base + pos[j + u];
            gg[u] = (kk[u] != NONE && col) ? d_out[i * d_stride + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kk[u] != cur) {
                if (cur != NONE && col && acc != 0.f) unsafeAtomicAdd(g_table + (int64_t)cur * a.dim + c, acc);
                cur = kk[u];
                acc = gg[u];
            } else {
                acc += gg[u];
            }
        }
    }
    if (cur != NONE && col && acc != 0.f) unsafeAtomicAdd(g_table + (int64_t)cur * a.dim + c, acc);
}

}  // namespace

extern "C" int dctr_din_att_in_fwd(const float* q, const float* k, int64_t batch, int32_t maxlen, int32_t dim, float* a, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1 && dim >= 1, DCTR_E_DIM, "din_att_in_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(q && k && a, DCTR_E_NULL, "din_att_in_fwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * maxlen * dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(din_att_in_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, q, k, batch * maxlen, (int)maxlen,
                       (int)dim, a);
    return dctr_launch_status("dctr_din_att_in_fwd");
}

extern "C" int dctr_din_wsum_fwd(const float* score, const uint8_t* mask, const float* k, int64_t batch, int32_t maxlen, int32_t dim,
                                 float* out, int64_t out_stride, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1 && dim >= 1 && out_stride >= dim, DCTR_E_DIM, "din_wsum_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(score && mask && k && out, DCTR_E_NULL, "din_wsum_fwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(din_wsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, score, mask, k, batch, (int)maxlen,
                       (int)dim, out, out_stride);
    return dctr_launch_status("dctr_din_wsum_fwd");
}

extern "C" int dctr_din_wsum_bwd(const float* d_out, int64_t d_stride, const float* score, const uint8_t* mask, const float* k,
                                 int64_t batch, int32_t maxlen, int32_t dim, float* d_score, float* dk, float* d_bias, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1 && dim >= 1 && d_stride >= dim, DCTR_E_DIM, "din_wsum_bwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(d_out && score && mask && k && d_score && dk, DCTR_E_NULL, "din_wsum_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * maxlen, (int64_t)4);
    if (blocks > 16384) blocks = 16384;
    // the bias gradient (sum of d_score) is taken by a second small kernel: as one atomic per workgroup of this one it either
    // serialised thousands of atomics on one address (0.5 ms at C4) or capped the grid at 256 workgroups (86 us)
    hipLaunchKernelGGL(din_wsum_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_out, d_stride, score, mask, k,
                       batch, (int)maxlen, (int)dim, d_score, dk, (float*)nullptr);
    if (d_bias != nullptr) {
        const int64_t n = batch * maxlen;
        int64_t g = dctr_ceil_div(n, (int64_t)256 * 8);
        g = g < 1 ? 1 : (g > COLSUM_MAX_WG ? COLSUM_MAX_WG : g);
        hipLaunchKernelGGL(sum_vec_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)d_score, n, d_bias);
    }
    return dctr_launch_status("dctr_din_wsum_bwd");
}

// att_weight_normalization=True (layers/sequence.py:283-289): p = softmax over ALL T positions of where(mask, score, -2^32 + 1) — a
// row without valid positions gets the uniform 1/T, as tf.nn.softmax gives it — and the weighted sum then runs over every position.
// One wave per row.  Backward: ds = p (dp - <p, dp>), kept where the mask is set (a padded position's input is the constant).
__global__ __launch_bounds__(256) void din_softmax_kernel(const float* __restrict__ score, const uint8_t* __restrict__ mask, int64_t batch,
                                                          int T, float* __restrict__ p) {
    const int lane = threadIdx.x & 63;
    const float pad = -4294967295.f;                       // float(-2 ** 32 + 1)
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < batch; b += (int64_t)gridDim.x * 4) {
        float mx = -__builtin_inff();
        for (int t = lane; t < T; t += 64) mx = fmaxf(mx, mask[b * T + t] ? score[b * T + t] : pad);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int t = lane; t < T; t += 64) sum += expf((mask[b * T + t] ? score[b * T + t] : pad) - mx);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        for (int t = lane; t < T; t += 64) p[b * T + t] = expf((mask[b * T + t] ? score[b * T + t] : pad) - mx) / sum;
    }
}
__global__ __launch_bounds__(256) void din_softmax_bwd_kernel(const float* __restrict__ p, const uint8_t* __restrict__ mask,
                                                              const float* __restrict__ dp, int64_t batch, int T, float* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < batch; b += (int64_t)gridDim.x * 4) {
        float dot = 0.f;
        for (int t = lane; t < T; t += 64) dot = fmaf(p[b * T + t], dp[b * T + t], dot);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 64);
        for (int t = lane; t < T; t += 64) ds[b * T + t] = mask[b * T + t] ? p[b * T + t] * (dp[b * T + t] - dot) : 0.f;
    }
}

extern "C" int dctr_din_softmax_fwd(const float* score, const uint8_t* mask, int64_t batch, int32_t maxlen, float* p, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1, DCTR_E_DIM, "din_softmax_fwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(score && mask && p, DCTR_E_NULL, "din_softmax_fwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch, (int64_t)4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(din_softmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, score, mask, batch, (int)maxlen, p);
    return dctr_launch_status("dctr_din_softmax_fwd");
}

// d_score (may alias dp) = softmax backward masked; d_bias (NULL ok) += sum d_score
extern "C" int dctr_din_softmax_bwd(const float* p, const uint8_t* mask, const float* dp, int64_t batch, int32_t maxlen, float* d_score,
                                    float* d_bias, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1, DCTR_E_DIM, "din_softmax_bwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(p && mask && dp && d_score, DCTR_E_NULL, "din_softmax_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch, (int64_t)4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(din_softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, mask, dp, batch, (int)maxlen, d_score);
    if (d_bias != nullptr) {
        const int64_t n = batch * maxlen;
        int64_t g = dctr_ceil_div(n, (int64_t)256 * 8);
        g = g < 1 ? 1 : (g > COLSUM_MAX_WG ? COLSUM_MAX_WG : g);
        hipLaunchKernelGGL(sum_vec_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const float*)d_score, n, d_bias);
    }
    return dctr_launch_status("dctr_din_softmax_bwd");
}

extern "C" int dctr_din_att_in_bwd(const float* da, const float* q, const float* k, int64_t batch, int32_t maxlen, int32_t dim,
                                   float* dk, float* dx, int64_t dx_stride, const int32_t* qcol, void* stream) {
    DCTR_REQUIRE(batch >= 0 && maxlen >= 1 && dim >= 1, DCTR_E_DIM, "din_att_in_bwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(da && q && k && dk && dx && qcol, DCTR_E_NULL, "din_att_in_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(din_att_in_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, da, q, k, batch, (int)maxlen,
                       (int)dim, dk, dx, dx_stride, qcol);
    return dctr_launch_status("dctr_din_att_in_bwd");
}

extern "C" int dctr_embed_lookup_bwd(const dctr_lookup_args_t* fwd, const float* d_out, int64_t d_stride, float* g_table, uint8_t* touched,
                                     void* stream) {
    DCTR_REQUIRE(fwd != nullptr, DCTR_E_NULL, "embed_lookup_bwd: null args");
    DCTR_REQUIRE(fwd->n >= 0 && fwd->dim >= 1 && d_stride >= fwd->dim && fwd->vocab >= 1, DCTR_E_DIM, "embed_lookup_bwd: bad sizes");
    if (fwd->n == 0) return DCTR_OK;
    DCTR_REQUIRE(fwd->idx && d_out && g_table, DCTR_E_NULL, "embed_lookup_bwd: null pointer");
    if (touched != nullptr) {
        DCTR_REQUIRE(fwd->dim % 4 == 0, DCTR_E_DIM, "embed_lookup_bwd: touched bytes need dim %% 4 == 0 (dim %d)", fwd->dim);
        launch_mark_rows((hipStream_t)stream, fwd->idx, fwd->n, 1, 1, fwd->idx_is_i64, fwd->hash_mode, fwd->vocab, fwd->dim, touched);
    }
    // (dim >= 3: a walker's stretch of LB_TILE / (4 * 64 / dimP) sorted entries must cover its 8-entry load batches)
    if (fwd->dim >= 3 && fwd->dim <= 64 && fwd->vocab < 0xffffffffLL && fwd->n >= LB_TILE) {
        int dimP = 1;
        while (dimP < fwd->dim) dimP <<= 1;
        const int64_t tiles = dctr_ceil_div(fwd->n, (int64_t)LB_TILE);
        DCTR_REQUIRE(tiles <= 0x7fffffffLL, DCTR_E_DIM, "embed_lookup_bwd: n too large");
        hipLaunchKernelGGL(lookup_bwd_tile_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, *fwd, d_out, d_stride,
                           g_table, dimP);
        return dctr_launch_status("dctr_embed_lookup_bwd");
    }
    int64_t blocks = dctr_ceil_div(fwd->n * fwd->dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(lookup_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *fwd, d_out, d_stride, g_table);
    return dctr_launch_status("dctr_embed_lookup_bwd");
}

extern "C" int dctr_afm_bwd(const dctr_afm_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "afm_bwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->fields >= 2 && a->dim >= 1 && a->att_factor >= 1, DCTR_E_DIM, "afm_bwd: bad sizes");
    DCTR_REQUIRE(a->x_stride >= (int64_t)a->fields * a->dim && a->dx_stride >= (int64_t)a->fields * a->dim, DCTR_E_DIM,
                 "afm_bwd: stride smaller than fields*dim");
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->att_w && a->att_b && a->proj_h && a->proj_p && a->dy && a->dx, DCTR_E_NULL, "afm_bwd: null pointer");
    DCTR_REQUIRE(a->d_att_w && a->d_att_b && a->d_proj_h && a->d_proj_p, DCTR_E_NULL, "afm_bwd: null gradient pointer");
    const int P = a->fields * (a->fields - 1) / 2;
    const size_t nw = (size_t)a->dim * a->att_factor + 2 * a->att_factor + a->dim;
    const size_t lds = (2 * nw + 4 * (2 * (size_t)a->fields * a->dim + P)) * sizeof(float);
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "afm_bwd: needs %zu B of LDS", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)afm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "afm_bwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, (int64_t)4);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "afm_bwd: batch too large");
    hipLaunchKernelGGL(afm_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a->x, a->x_stride, a->batch,
                       (int)a->fields, (int)a->dim, a->att_w, a->att_b, a->proj_h, a->proj_p, (int)a->att_factor, a->dy, a->dx,
                       a->dx_stride, (int)a->dx_accumulate, a->d_att_w, a->d_att_b, a->d_proj_h, a->d_proj_p);
    return dctr_launch_status("dctr_afm_bwd");
}

extern "C" int dctr_bi_interaction_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dy,
                                       int64_t dy_stride, float* dx, int64_t dx_stride, int32_t accumulate, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 1 && dim >= 1, DCTR_E_DIM, "bi_interaction_bwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim && dx_stride >= (int64_t)fields * dim && dy_stride >= dim, DCTR_E_DIM,
                 "bi_interaction_bwd: stride smaller than a row");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && dy && dx, DCTR_E_NULL, "bi_interaction_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bi_interaction_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch,
                       (int)fields, (int)dim, dy, dy_stride, dx, dx_stride, (int)accumulate);
    return dctr_launch_status("dctr_bi_interaction_bwd");
}

// FM.call backward (interaction.py:588-604) on a strided [B, >= F*E] buffer: dx[b,f,:] (+)= dlogit[b] * (sum_f' x[b,f',:] - x[b,f,:])
extern "C" int dctr_fm_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dlogit, float* dx,
                           int64_t dx_stride, int32_t accumulate, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 1 && dim >= 1, DCTR_E_DIM, "fm_bwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim && dx_stride >= (int64_t)fields * dim, DCTR_E_DIM, "fm_bwd: stride < fields*dim");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && dlogit && dx, DCTR_E_NULL, "fm_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bi_interaction_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch, (int)fields,
                       (int)dim, dlogit, (int64_t)1, dx, dx_stride, (int)accumulate, 0);
    return dctr_launch_status("dctr_fm_bwd");
}

extern "C" int dctr_inner_product_bwd(const float* x, int64_t batch, int64_t x_stride, int32_t fields, int32_t dim, const float* dy,
                                      int64_t dy_stride, float* dx, int64_t dx_stride, int32_t accumulate, void* stream) {
    DCTR_REQUIRE(batch >= 0 && fields >= 2 && fields <= 1024 && dim >= 1, DCTR_E_DIM, "inner_product_bwd: bad sizes");
    DCTR_REQUIRE(x_stride >= (int64_t)fields * dim && dx_stride >= (int64_t)fields * dim &&
                     dy_stride >= (int64_t)fields * (fields - 1) / 2,
                 DCTR_E_DIM, "inner_product_bwd: stride smaller than a row");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && dy && dx, DCTR_E_NULL, "inner_product_bwd: null pointer");
    int64_t blocks = dctr_ceil_div(batch * fields * dim, (int64_t)256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(inner_product_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_stride, batch,
                       (int)fields, (int)dim, dy, dy_stride, dx, dx_stride, (int)accumulate);
    return dctr_launch_status("dctr_inner_product_bwd");
}

extern "C" int dctr_bce_grad_w(const float* pred, const float* y, const float* weight, int64_t batch, int32_t task, float* dlogit,
                               float* loss_sum, float* dlogit_sum, void* stream) {
    DCTR_REQUIRE(batch >= 0 && (task == 0 || task == 1), DCTR_E_DIM, "bce_grad: bad batch / task");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(pred && y && dlogit, DCTR_E_NULL, "bce_grad: null pointer");
    hipLaunchKernelGGL(bce_grad_kernel, dim3((unsigned)dctr_ceil_div(batch, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, pred,
                       y, weight, batch, (int)task, dlogit, loss_sum, dlogit_sum);
    return dctr_launch_status("dctr_bce_grad");
}

extern "C" int dctr_bce_grad(const float* pred, const float* y, int64_t batch, int32_t task, float* dlogit, float* loss_sum,
                             float* dlogit_sum, void* stream) {
    return dctr_bce_grad_w(pred, y, nullptr, batch, task, dlogit, loss_sum, dlogit_sum, stream);
}

extern "C" int dctr_embed_gather_fm_bwd(const dctr_gather_fm_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr && a->fwd != nullptr, DCTR_E_NULL, "embed_gather_fm_bwd: null args");
    const dctr_gather_fm_args_t* f = a->fwd;
    DCTR_REQUIRE(f->batch >= 0 && f->n_fields >= 0, DCTR_E_DIM, "embed_gather_fm_bwd: bad sizes");
    if (f->batch == 0 || f->n_fields == 0) return DCTR_OK;
    DCTR_REQUIRE(f->fields && f->ids && a->grads, DCTR_E_NULL, "embed_gather_fm_bwd: null fields / ids / grads");
    DCTR_REQUIRE(!f->any_pitch && f->max_dim >= 1, DCTR_E_UNSUPPORTED, "embed_gather_fm_bwd: plain (not record-form) tables, max_dim >= 1");
    DCTR_REQUIRE(a->d_dnn_in == nullptr || !f->all_dim4 || (a->d_stride % 4 == 0 && dctr_aligned16(a->d_dnn_in)), DCTR_E_ALIGN,
                 "embed_gather_fm_bwd: d_dnn_in must be 16-B aligned with a stride %% 4 == 0");
    hipStream_t st = (hipStream_t)stream;
    if (!f->all_dim4) {
        // some width is not a multiple of 4 (rows / DNN-input columns not 16-B aligned): element per lane, 16 lanes per sample
        const int64_t blocks1 = dctr_ceil_div(f->batch, (int64_t)4);
        DCTR_REQUIRE(blocks1 <= 0x7fffffffLL, DCTR_E_DIM, "embed_gather_fm_bwd: batch too large");
        if (f->any_hash)
            hipLaunchKernelGGL((gather_fm_bwd_kernel<16, true, 1>), dim3((unsigned)blocks1), dim3(256), 0, st, *f, a->grads, a->d_dnn_in,
                               a->d_stride, a->d_fm, a->d_lin, a->g_dense_lin_w, a->dense_lin_rows);
        else
            hipLaunchKernelGGL((gather_fm_bwd_kernel<16, false, 1>), dim3((unsigned)blocks1), dim3(256), 0, st, *f, a->grads, a->d_dnn_in,
                               a->d_stride, a->d_fm, a->d_lin, a->g_dense_lin_w, a->dense_lin_rows);
        return dctr_launch_status("dctr_embed_gather_fm_bwd");
    }
    int lpr = 1;
    while (lpr * 4 < f->max_dim && lpr < 16) lpr <<= 1;                          // (rows wider than 64 floats: several chunks per lane)
    const int64_t blocks = dctr_ceil_div(f->batch, (int64_t)(64 / lpr));        // one wave's worth of samples per workgroup
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "embed_gather_fm_bwd: batch too large");
#define CALL_BWD(L)                                                                                                       \
    do {                                                                                                                  \
        if (f->any_hash)                                                                                                  \
            hipLaunchKernelGGL((gather_fm_bwd_kernel<L, true>), dim3((unsigned)blocks), dim3(256), 0, st, *f, a->grads,   \
                               a->d_dnn_in, a->d_stride, a->d_fm, a->d_lin, a->g_dense_lin_w, a->dense_lin_rows);        \
        else                                                                                                              \
            hipLaunchKernelGGL((gather_fm_bwd_kernel<L, false>), dim3((unsigned)blocks), dim3(256), 0, st, *f, a->grads,  \
                               a->d_dnn_in, a->d_stride, a->d_fm, a->d_lin, a->g_dense_lin_w, a->dense_lin_rows);        \
    } while (0)
    switch (lpr) {
        case 1: CALL_BWD(1); break;
        case 2: CALL_BWD(2); break;
        case 4: CALL_BWD(4); break;
        case 8: CALL_BWD(8); break;
        default: CALL_BWD(16); break;
    }
#undef CALL_BWD
    return dctr_launch_status("dctr_embed_gather_fm_bwd");
}

extern "C" int dctr_embed_pool_bwd(const dctr_pool_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr && a->fwd != nullptr, DCTR_E_NULL, "embed_pool_bwd: null args");
    const dctr_pool_args_t* f = a->fwd;
    DCTR_REQUIRE(f->batch >= 0 && f->maxlen >= 1 && f->dim >= 1, DCTR_E_DIM, "embed_pool_bwd: bad sizes");
    if (f->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(f->idx && f->table, DCTR_E_NULL, "embed_pool_bwd: null idx / table");
    const bool vec4 = f->dim % 4 == 0;
    DCTR_REQUIRE(a->d_out == nullptr || !vec4 || (a->d_stride % 4 == 0 && dctr_aligned16(a->d_out)), DCTR_E_ALIGN,
                 "embed_pool_bwd: d_out must be 16-B aligned with a stride %% 4 == 0");
    DCTR_REQUIRE(a->touched == nullptr || vec4, DCTR_E_DIM, "embed_pool_bwd: touched bytes need dim %% 4 == 0 (dim %d)", f->dim);
    int lpr = 1;
    while (lpr * 4 < f->dim && lpr < 16) lpr <<= 1;          // (rows wider than 64 floats: several chunks per lane)
    if (!vec4) lpr = 16;                                     // element per lane
    const int64_t blocks = dctr_ceil_div(f->batch, (int64_t)(256 / lpr));
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "embed_pool_bwd: batch too large");
    hipStream_t st = (hipStream_t)stream;
    if (a->touched != nullptr && a->g_table != nullptr)
        launch_mark_rows(st, f->idx, f->batch, f->maxlen, f->idx_stride, f->idx_is_i64, f->hash_mode, f->vocab, f->dim, a->touched);
    if (!vec4) {
        hipLaunchKernelGGL((pool_bwd_kernel<16, 1>), dim3((unsigned)blocks), dim3(256), 0, st, *f, a->d_out, a->d_stride, a->d_lin_out,
                           a->g_table, a->g_lin_table);
        return dctr_launch_status("dctr_embed_pool_bwd");
    }
#define CALL_PB(L) \
    hipLaunchKernelGGL((pool_bwd_kernel<L>), dim3((unsigned)blocks), dim3(256), 0, st, *f, a->d_out, a->d_stride, a->d_lin_out, \
                       a->g_table, a->g_lin_table)
    switch (lpr) {
        case 1: CALL_PB(1); break;
        case 2: CALL_PB(2); break;
        case 4: CALL_PB(4); break;
        case 8: CALL_PB(8); break;
        default: CALL_PB(16); break;
    }
#undef CALL_PB
    return dctr_launch_status("dctr_embed_pool_bwd");
}

// ---------------------------------------------------------------------------------------------------
// dctr_mlp_bwd, chained form (relu / linear / sigmoid / tanh): head -> dZ_{L-1};  W_l^T of every layer (one launch);  the backward
// chain (mlp_bwd_kernels.hip: every dZ_l and dX in ONE launch);  dW_l (+ d_bias_l as one more output row) of every layer as row
// slices in ONE grouped GEMM launch;  one launch that sums the slices into the gradient buffers.  Five launches per step where the
// layer-by-layer form below took 4 L + 1 (a 3-layer DNN at B = 4096: ~60 us against ~210).
// ---------------------------------------------------------------------------------------------------
struct TransposeGroups {
    const float* src[8];
    float* dst[8];
    int rows[8], cols[8];       // src is [rows, cols] row-major, dst [cols, rows]
};
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeGroups tg) {
    __shared__ float tile[32][33];
    const int gi = blockIdx.y;
    const int R = tg.rows[gi], C = tg.cols[gi];
    const int tc = (C + 31) / 32, n_tiles = ((R + 31) / 32) * tc;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            tile[ty + 8 * i][tx] = (r < R && c < C) ? tg.src[gi][(int64_t)r * C + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;
            if (c < C && r < R) tg.dst[gi][(int64_t)c * R + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

// dst_w[i] += sum_s parts[s * n + i] (i < n_w);  dst_b[i - n_w] += the same for the bias row behind the kernel's rows
struct SumGroups {
    const float* parts[8];
    float* dst_w[8];
    float* dst_b[8];
    int64_t n[8], n_w[8];
    int n_parts[8];
};
__global__ __launch_bounds__(256) void sum_parts_multi_kernel(SumGroups sg) {
    const int gi = blockIdx.y;
    const float* __restrict__ parts = sg.parts[gi];
    const int64_t n = sg.n[gi], n_w = sg.n_w[gi];
    const int n_parts = sg.n_parts[gi];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 4 <= n_parts; s += 4) {
            a0 += parts[(int64_t)s * n + i];
            a1 += parts[(int64_t)(s + 1) * n + i];
            a2 += parts[(int64_t)(s + 2) * n + i];
            a3 += parts[(int64_t)(s + 3) * n + i];
        }
        for (; s < n_parts; ++s) a0 += parts[(int64_t)s * n + i];
        const float v = (a0 + a1) + (a2 + a3);
        if (i < n_w) sg.dst_w[gi][i] += v;
        else sg.dst_b[gi][i - n_w] += v;
    }
}

// rows per dW slice of the chained form (DCTR_DW_ROWS overrides: scripts/gemm_lab.py)
static int chain_dw_rows() {
    static const int v = [] {
        const char* e = dctr_lab_env("DCTR_DW_ROWS");
        const int x = e != nullptr ? atoi(e) : 0;
        return (x >= 32 && x <= 4096 && x % 32 == 0) ? x : 256;      // (whole 32-row tiles, bounded: the value sizes a workspace)
    }();
    return v;
}
static bool chain_disabled() {
    static const bool v = [] { const char* e = dctr_lab_env("DCTR_MLP_BWD_CHAIN"); return e != nullptr && atoi(e) == 0; }();
    return v;
}
static bool mlp_bwd_chained(const dctr_mlp_bwd_args_t* a) {
    return !chain_disabled() && a->activation != DCTR_ACT_DICE && a->n_layers <= 8 && dctr_mlp::bwd_chain_fits(a->in_dim, a->n_layers, a->units);
}
// workspace of the chained form, in floats: dZ_l [B, units[l]] | W_l^T | dW slices [(K_l + 1) units[l]] x slices   (each 4-float aligned)
static size_t mlp_bwd_chain_floats(const dctr_mlp_bwd_args_t* a) {
    size_t f = 0;
    const int slices = dctr_gemm::k_slices((int)a->batch, chain_dw_rows());
    for (int l = 0; l < a->n_layers; ++l) {
        const size_t K = l == 0 ? a->in_dim : a->units[l - 1], N = a->units[l];
        f += ((size_t)a->batch * N + 3) & ~(size_t)3;
        f += (K * N + 3) & ~(size_t)3;
        f += ((size_t)slices * (K + 1) * N + 3) & ~(size_t)3;
    }
    return f;
}

// dW = X^T dZ has a small output and a reduction as long as the batch: it runs as a strided batch of row slices into partial
// products + a sum (deterministic), 512 rows per slice from 1024 rows on, at most 32 slices (as ONE gemm a 429 x 256 output is 28
// workgroups walking the whole batch: 64 us per layer at B = 4096; round 2's rocBLAS call split K by itself at small batches and took
// 1.78 ms per layer at B = 65,536)
// (a SMALL output under a very long reduction — DIN's attention unit: 256 x 80 over 102,400 rows — gets up to 128 slices: as 32 slices
//  of 3,200 rows on 10 tiles it ran 268 us per layer on 320 workgroups)
static int mlp_dw_parts(int64_t batch, int64_t out_elems) {
    if (batch < 1024) return 1;
    const int64_t tiles = out_elems / 4096 > 1 ? out_elems / 4096 : 1;
    int64_t cap = 1024 / tiles;
    cap = cap < 32 ? 32 : (cap > 128 ? 128 : cap);
    int parts = (int)(batch / 512 > cap ? cap : batch / 512);
    while (parts > 1 && batch % parts != 0) --parts;
    return parts;
}

static size_t mlp_bwd_main_floats(const dctr_mlp_bwd_args_t* a, int w) {
    // two ping-pong buffers [B, widest layer]; Dice needs a third for the recomputed pre-activations
    // ... and, with batch statistics, 2 x widest for the two column sums of the BatchNormalization backward
    return ((size_t)(a->activation == DCTR_ACT_DICE ? 3 : 2) * a->batch * w + (a->activation == DCTR_ACT_DICE ? 2 * (size_t)w : 0) + 3) &
           ~(size_t)3;
}

extern "C" size_t dctr_mlp_bwd_workspace_bytes(const dctr_mlp_bwd_args_t* a) {
    if (a == nullptr || a->batch <= 0 || a->n_layers < 1 || a->units == nullptr) return 0;
    int w = a->in_dim;
    size_t kn = 0;
    for (int l = 0; l < a->n_layers; ++l) {
        const size_t k = l == 0 ? a->in_dim : a->units[l - 1];
        kn = k * a->units[l] > kn ? k * a->units[l] : kn;
        w = a->units[l] > w ? a->units[l] : w;
    }
    size_t parts_floats = 0;
    for (int l = 0; l < a->n_layers; ++l) {
        const size_t k = l == 0 ? a->in_dim : a->units[l - 1], knl = k * a->units[l];
        const int parts = mlp_dw_parts(a->batch, (int64_t)knl);
        if (parts > 1 && (size_t)parts * knl > parts_floats) parts_floats = (size_t)parts * knl;
    }
    const size_t layered = mlp_bwd_main_floats(a, w) + parts_floats;
    const size_t chained = (a->n_layers <= 8 && a->in_dim >= 1 && a->batch < 0x7fffffffLL && mlp_bwd_chained(a)) ? mlp_bwd_chain_floats(a) : 0;
    return (layered > chained ? layered : chained) * sizeof(float);
}

extern "C" int dctr_mlp_bwd(const dctr_mlp_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "mlp_bwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->in_dim >= 1 && a->n_layers >= 1 && a->n_layers <= 8, DCTR_E_DIM, "mlp_bwd: bad sizes");
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->units && a->kernels && a->acts && a->d_kernels, DCTR_E_NULL, "mlp_bwd: null pointer");
    DCTR_REQUIRE((a->head_w != nullptr && a->dlogit != nullptr && a->d_head_w != nullptr) ||
                     (a->head_w == nullptr && a->d_out != nullptr && a->d_out_stride >= a->units[a->n_layers - 1]),
                 DCTR_E_NULL, "mlp_bwd: needs either (head_w, dlogit, d_head_w) or d_out");
    const bool dice = a->activation == DCTR_ACT_DICE;
    DCTR_REQUIRE((a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_TANH) || dice, DCTR_E_UNSUPPORTED,
                 "mlp_bwd: activation %d has no backward (linear, relu, sigmoid, tanh, dice)", a->activation);
    DCTR_REQUIRE(!dice || (a->dice_alpha && a->dice_mean && a->dice_var), DCTR_E_NULL, "mlp_bwd: dice needs alpha / mean / var");
    DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= dctr_mlp_bwd_workspace_bytes(a), DCTR_E_NULL,
                 "mlp_bwd: needs a workspace of dctr_mlp_bwd_workspace_bytes() bytes");
    DCTR_REQUIRE(a->batch < 0x7fffffffLL, DCTR_E_DIM, "mlp_bwd: batch too large");
    hipStream_t st = (hipStream_t)stream;
    if (mlp_bwd_chained(a)) {
        const int L = a->n_layers, B = (int)a->batch;
        const int slices = dctr_gemm::k_slices(B, chain_dw_rows());
        float* dz[8];
        float* wt[8];
        float* parts[8];
        float* ws = static_cast<float*>(a->workspace);
        for (int l = 0; l < L; ++l) {
            const size_t K = l == 0 ? a->in_dim : a->units[l - 1], N = a->units[l];
            dz[l] = ws; ws += ((size_t)a->batch * N + 3) & ~(size_t)3;
            wt[l] = ws; ws += (K * N + 3) & ~(size_t)3;
            parts[l] = ws; ws += ((size_t)slices * (K + 1) * N + 3) & ~(size_t)3;
        }
        // head: dZ_last = dlogit (x) head_w .* act'(h_last);  d_head_w = h_last^T dlogit   (headless: d_out .* act'(h_last))
        const int NL = a->units[L - 1];
        if (a->head_w != nullptr) {
            launch_head_bwd(st, a->dlogit, a->head_w, a->acts[L - 1], (int64_t)NL, a->batch, NL, (int)a->activation, dz[L - 1], (int64_t)NL,
                            a->d_head_w);
        } else {
            hipError_t ce = hipMemcpy2DAsync(dz[L - 1], (size_t)NL * sizeof(float), a->d_out, (size_t)a->d_out_stride * sizeof(float),
                                             (size_t)NL * sizeof(float), (size_t)a->batch, hipMemcpyDeviceToDevice, st);
            DCTR_REQUIRE(ce == hipSuccess, (int)ce, "mlp_bwd: copy of d_out failed: %s", hipGetErrorString(ce));
            if (a->activation != DCTR_ACT_LINEAR)
                launch_act_bwd_colsum(st, dz[L - 1], a->acts[L - 1], a->batch, NL, (int)a->activation, (float*)nullptr);
        }
        const int l_first = a->dx != nullptr ? 0 : 1;       // layers whose transposed kernel the chain multiplies by
        if (l_first < L) {
            TransposeGroups tg{};
            int ng = 0, max_tiles = 1;
            for (int l = l_first; l < L; ++l, ++ng) {
                const int K = l == 0 ? a->in_dim : a->units[l - 1], N = a->units[l];
                tg.src[ng] = a->kernels[l];
                tg.dst[ng] = wt[l];
                tg.rows[ng] = K;
                tg.cols[ng] = N;
                const int t = ((K + 31) / 32) * ((N + 31) / 32);
                max_tiles = t > max_tiles ? t : max_tiles;
            }
            hipLaunchKernelGGL(transpose_multi_kernel, dim3((unsigned)max_tiles, (unsigned)ng), dim3(256), 0, st, tg);
            const int rc = dctr_mlp::launch_bwd_chain(st, a->batch, a->in_dim, L, a->units, wt, a->acts, (int)a->activation, dz[L - 1], dz,
                                                      a->dx, a->dx_stride);
            if (rc != DCTR_OK) return rc;
        }
        // dW_l[K, N] (row-major) += X_l^T dZ_l, d_bias_l += colsum(dZ_l): column-major  dW'(N x (K + 1)) = dZ'(N x rows) * [X' ; 1](.. x rows)^T
        dctr_gemm::GroupDesc gd[8];
        SumGroups sg{};
        int64_t max_n = 0;
        for (int l = 0; l < L; ++l) {
            const int N = a->units[l], K = l == 0 ? a->in_dim : a->units[l - 1];
            const bool bias = a->d_biases != nullptr && a->d_biases[l] != nullptr;
            dctr_gemm::GroupDesc& d = gd[l];
            d = dctr_gemm::GroupDesc{};
            d.op_a = dctr_gemm::OP_N;
            d.op_b = dctr_gemm::OP_T;
            d.m = N;
            d.n = K + (bias ? 1 : 0);
            d.k = B;
            d.A = dz[l];
            d.lda = N;
            d.B = l == 0 ? a->x : a->acts[l - 1];
            d.ldb = l == 0 ? (int)a->x_stride : K;
            d.C = parts[l];
            d.ldc = N;
            d.batch = 1;
            d.ones_last = bias ? 1 : 0;
            d.k_slices = slices;
            d.slice_stride_c = (int64_t)d.n * N;            // (sum_parts_multi reads slice s at parts + s * n)
            sg.parts[l] = parts[l];
            sg.dst_w[l] = a->d_kernels[l];
            sg.dst_b[l] = bias ? a->d_biases[l] : nullptr;
            sg.n[l] = (int64_t)d.n * N;
            sg.n_w[l] = (int64_t)K * N;
            sg.n_parts[l] = slices;
            max_n = sg.n[l] > max_n ? sg.n[l] : max_n;
        }
        // the weight-gradient half: on the caller's second stream when it gave one (behind an event on `stream`; the caller joins)
        hipStream_t dws = st;
        if (a->dw_stream != nullptr && (hipStream_t)a->dw_stream != st) {
            // (one event per device and calling thread, alive as long as the process: hipEventRecord re-arms it on every call)
            static thread_local hipEvent_t ev[DCTR_MAX_DEVICES] = {};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            DCTR_REQUIRE(e == hipSuccess && dev >= 0 && dev < DCTR_MAX_DEVICES, DCTR_E_UNSUPPORTED, "mlp_bwd: dw_stream on device %d", dev);
            if (ev[dev] == nullptr) {
                e = hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming);
                DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_bwd: hipEventCreate failed: %s", hipGetErrorString(e));
            }
            dws = (hipStream_t)a->dw_stream;
            e = hipEventRecord(ev[dev], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(dws, ev[dev], 0);
            DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_bwd: cannot order dw_stream behind stream: %s", hipGetErrorString(e));
        }
        const int rg = dctr_gemm::sgemm_grouped(dws, gd, L);
        DCTR_REQUIRE(rg == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm_grouped(dW) failed (%d)", rg);
        int64_t gx = dctr_ceil_div(max_n, (int64_t)256);
        hipLaunchKernelGGL(sum_parts_multi_kernel, dim3((unsigned)(gx > 512 ? 512 : gx), (unsigned)L), dim3(256), 0, dws, sg);
        return dctr_launch_status("dctr_mlp_bwd");
    }
    int w = a->in_dim;
    for (int l = 0; l < a->n_layers; ++l) w = a->units[l] > w ? a->units[l] : w;
    float* bufA = static_cast<float*>(a->workspace);
    float* bufB = bufA + (size_t)a->batch * w;
    float* bufZ = bufB + (size_t)a->batch * w;                 // dice only
    float* dw_parts = bufA + mlp_bwd_main_floats(a, w);        // partial dW of the row slices (batch >= 1024)
    const int B = (int)a->batch;
    const unsigned rb = (unsigned)dctr_ceil_div(a->batch, (int64_t)BWD_ROWS);
    const int L = a->n_layers;
    // Dice: dH of layer l (in `buf`, in place) -> dZ.  Z_l = X_l W_l is recomputed (column-major Z'(N x B) = W'(N x K) X'(K x B)).
    auto dice_bwd = [&](int l, float* buf) -> int {
        const int N = a->units[l], K = l == 0 ? a->in_dim : a->units[l - 1];
        const float* xin = l == 0 ? a->x : a->acts[l - 1];
        const int ldx = l == 0 ? (int)a->x_stride : K;
        // (saved_z: the forward's own pre-activations, bias included — no recompute, no bias in the kernels below)
        const float* zsrc = bufZ;
        const float* bl = a->biases != nullptr ? a->biases[l] : nullptr;
        if (a->saved_z != nullptr && a->saved_z[l] != nullptr) {
            zsrc = a->saved_z[l];
            bl = nullptr;
        } else {
            int rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_N, N, B, K, a->kernels[l], N, xin, ldx, 0.f, bufZ, N);
            if (rs != 0) return (int)rs;
        }
        float* dal = a->d_dice_alpha != nullptr ? a->d_dice_alpha[l] : nullptr;
        if (a->dice_batch_mean != nullptr && a->dice_batch_var != nullptr && a->dice_batch_mean[l] != nullptr) {
            // training-mode Dice: this batch's statistics, gradients through them
            float* s12 = bufZ + (size_t)a->batch * w;
            if (hipMemsetAsync(s12, 0, (size_t)2 * N * sizeof(float), st) != hipSuccess) return -1;
            if (rowlane4_ok(N, buf, zsrc, N))
                hipLaunchKernelGGL(dice_train_bwd_reduce4_kernel, dim3(colsum_grid(a->batch, 256 / (N / 4))), dim3(256), 0, st,
                                   (const float*)buf, zsrc, bl, a->dice_alpha[l], a->dice_batch_mean[l],
                                   a->dice_batch_var[l], a->dice_eps, a->batch, N, s12, dal);
            else
                hipLaunchKernelGGL(dice_train_bwd_reduce_kernel, dim3(rows_grid(a->batch, true)), dim3(256), 0, st, (const float*)buf, zsrc, bl,
                                   a->dice_alpha[l], a->dice_batch_mean[l], a->dice_batch_var[l], a->dice_eps, a->batch, N, s12, dal);
            hipLaunchKernelGGL(dice_train_bwd_apply_kernel, dim3(rb), dim3(256), 0, st, buf, zsrc, bl, a->dice_alpha[l],
                               a->dice_batch_mean[l], a->dice_batch_var[l], a->dice_eps, a->batch, N, (const float*)s12);
            return 0;
        }
        if (rowlane4_ok(N, buf, zsrc, N))
            hipLaunchKernelGGL(dice_bwd4_kernel, dim3(colsum_grid(a->batch, 256 / (N / 4))), dim3(256), 0, st, buf, zsrc, bl,
                               a->dice_alpha[l], a->dice_mean[l], a->dice_var[l], a->dice_eps, a->batch, N, dal);
        else
            hipLaunchKernelGGL(dice_bwd_kernel, dim3(rows_grid(a->batch, true)), dim3(256), 0, st, buf, zsrc, bl, a->dice_alpha[l], a->dice_mean[l], a->dice_var[l],
                               a->dice_eps, a->batch, N, dal);
        return 0;
    };
    // head: dZ_last = dlogit (x) head_w .* act'(h_last);  d_head_w = h_last^T dlogit
    const int NL = a->units[L - 1];
    if (a->head_w != nullptr) {
        launch_head_bwd(st, a->dlogit, a->head_w, a->acts[L - 1], (int64_t)NL, a->batch, NL,
                        dice ? (int)DCTR_ACT_LINEAR : (int)a->activation, bufA, (int64_t)NL, a->d_head_w);
    } else {
        // headless (the DNN branch of DCN): the caller hands d(loss)/d(h_last); dZ_last = d_out .* act'(h_last)
        hipError_t ce = hipMemcpy2DAsync(bufA, (size_t)NL * sizeof(float), a->d_out, (size_t)a->d_out_stride * sizeof(float),
                                         (size_t)NL * sizeof(float), (size_t)a->batch, hipMemcpyDeviceToDevice, st);
        DCTR_REQUIRE(ce == hipSuccess, (int)ce, "mlp_bwd: copy of d_out failed: %s", hipGetErrorString(ce));
        if (a->activation != DCTR_ACT_LINEAR && !dice)
            launch_act_bwd_colsum(st, bufA, a->acts[L - 1], a->batch, NL, (int)a->activation, (float*)nullptr);
    }
    if (dice) {
        const int rc = dice_bwd(L - 1, bufA);
        DCTR_REQUIRE(rc == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm(Z) failed (%d)", rc);
    }
    float* dz = bufA;
    float* other = bufB;
    for (int l = L - 1; l >= 0; --l) {
        const int N = a->units[l], K = l == 0 ? a->in_dim : a->units[l - 1];
        const float* xin = l == 0 ? a->x : a->acts[l - 1];
        const int ldx = l == 0 ? (int)a->x_stride : K;
        // d_bias[n] += sum_b dZ[b, n]   (dZ is final here: the head / the previous iteration applied act')
        if (a->d_biases != nullptr && a->d_biases[l] != nullptr)
            launch_act_bwd_colsum(st, dz, (const float*)nullptr, a->batch, N, 0, a->d_biases[l]);
        // dW[K, N] (row-major) += X^T dZ:  column-major  dW'(N x K) = dZ'(N x B) * X'(K x B)^T
        int rs;
        const int n_parts = mlp_dw_parts(a->batch, (int64_t)K * N);
        const bool to_dx = l == 0;
        const bool want_dx = !(to_dx && a->dx == nullptr);
        float* dst = to_dx ? a->dx : other;
        const int ldd = to_dx ? (int)a->dx_stride : K;
        if (n_parts > 1) {
            // dW slices and dH_prev[B, K] = dZ W^T (column-major dH'(K x B) = W'(N x K)^T * dZ'(N x B)) read dZ and nothing of each
            // other: ONE grouped launch (two latency-bound problems side by side: DCN-matrix's pair went from 2 x 35 to 45 us)
            const int rs_ = B / n_parts;
            const int64_t kn = (int64_t)K * N;
            dctr_gemm::GroupDesc gd[2];
            gd[0] = dctr_gemm::GroupDesc{};
            gd[0].op_a = dctr_gemm::OP_N; gd[0].op_b = dctr_gemm::OP_T; gd[0].m = N; gd[0].n = K; gd[0].k = rs_;
            gd[0].A = dz; gd[0].lda = N; gd[0].stride_a = (int64_t)rs_ * N;
            gd[0].B = xin; gd[0].ldb = ldx; gd[0].stride_b = (int64_t)rs_ * ldx;
            gd[0].C = dw_parts; gd[0].ldc = N; gd[0].stride_c = kn; gd[0].batch = n_parts;
            gd[1] = dctr_gemm::GroupDesc{};
            gd[1].op_a = dctr_gemm::OP_T; gd[1].op_b = dctr_gemm::OP_N; gd[1].m = K; gd[1].n = B; gd[1].k = N;
            gd[1].A = a->kernels[l]; gd[1].lda = N; gd[1].B = dz; gd[1].ldb = N; gd[1].C = dst; gd[1].ldc = ldd; gd[1].batch = 1;
            rs = dctr_gemm::sgemm_grouped(st, gd, want_dx ? 2 : 1);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm_grouped(dW, dX) failed (%d)", (int)rs);
            int64_t g = dctr_ceil_div(kn, (int64_t)256);
            hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, (const float*)dw_parts, kn, n_parts,
                               a->d_kernels[l]);
            if (!want_dx) break;
        } else {
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_T, N, K, B, dz, N, xin, ldx, 1.f, a->d_kernels[l], N);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm(dW) failed (%d)", (int)rs);
            if (!want_dx) break;
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_T, dctr_gemm::OP_N, K, B, N, a->kernels[l], N, dz, N, 0.f, dst, ldd);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm(dX) failed (%d)", (int)rs);
        }
        if (!to_dx) {
            // dZ_prev = dH_prev .* act'(h_prev)
            if (dice) {
                const int rc = dice_bwd(l - 1, other);
                DCTR_REQUIRE(rc == 0, DCTR_E_UNSUPPORTED, "mlp_bwd: sgemm(Z) failed (%d)", rc);
            } else if (a->activation != DCTR_ACT_LINEAR) {
                launch_act_bwd_colsum(st, other, a->acts[l - 1], a->batch, K, (int)a->activation, (float*)nullptr);
            }
            float* t = dz;
            dz = other;
            other = t;
        }
    }
    return dctr_launch_status("dctr_mlp_bwd");
}

extern "C" int dctr_adam_step(float* w, float* m, float* v, float* g, int64_t n, float alpha, float beta1, float beta2,
                              float eps, float l2, int32_t zero_grad, void* stream) {
    DCTR_REQUIRE(n >= 0, DCTR_E_DIM, "adam_step: n < 0");
    if (n == 0) return DCTR_OK;
    DCTR_REQUIRE(w && m && v && g, DCTR_E_NULL, "adam_step: null pointer");
    DCTR_REQUIRE(dctr_aligned16(w) && dctr_aligned16(m) && dctr_aligned16(v) && dctr_aligned16(g), DCTR_E_ALIGN,
                 "adam_step: buffers must be 16-B aligned");
    int64_t blocks = dctr_ceil_div(n / 4 + 1, (int64_t)256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, m, v, g, n, alpha, beta1, beta2,
                       eps, l2, (int)zero_grad);
    return dctr_launch_status("dctr_adam_step");
}

extern "C" int dctr_opt_multi(int32_t kind, const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float lr, float beta1,
                              float beta2, float eps, int32_t zero_grad, void* stream) {
    return dctr_opt_multi_l2(kind, segs, n_segs, max_n, lr, beta1, beta2, eps, zero_grad, nullptr, 0.f, stream);
}

// ABI 13 — the same step; with l2_penalty != NULL the launch also adds penalty_scale * sum over the segments of l2 * sum(w^2), taken on the
// weights BEFORE the update, to *l2_penalty (a DEVICE double): what tf.keras adds to that batch's reported loss
extern "C" int dctr_opt_multi_l2(int32_t kind, const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float lr, float beta1,
                                 float beta2, float eps, int32_t zero_grad, double* l2_penalty, float penalty_scale, void* stream) {
    DCTR_REQUIRE(kind >= DCTR_OPT_ADAM && kind <= DCTR_OPT_SGD, DCTR_E_ENUM, "opt_multi: optimizer kind %d", kind);
    DCTR_REQUIRE(n_segs >= 0 && n_segs <= 65535 && max_n >= 0, DCTR_E_DIM, "opt_multi: bad n_segs / max_n");
    if (n_segs == 0 || max_n == 0) return DCTR_OK;
    DCTR_REQUIRE(segs != nullptr, DCTR_E_NULL, "opt_multi: null segment array");
    // Two 16-B groups per trip, non-temporal loads / stores, four groups per thread in the largest segment: 175 us for the C2 DeepFM
    // parameter set with touched bytes (6.15 TB/s) against 215 us with default-policy accesses and 205 us with four groups per trip
    // (profiles/r03c_opt_lab.log).  DCTR_OPT_VARIANT / DCTR_OPT_F4 re-run that lab (scripts/opt_lab.py).
    static const int f4 = [] { const char* e = dctr_lab_env("DCTR_OPT_F4"); return e != nullptr && atoi(e) >= 1 ? atoi(e) : 4; }();
    static const int variant = [] { const char* e = dctr_lab_env("DCTR_OPT_VARIANT"); return e != nullptr ? atoi(e) : 2; }();
    int64_t bx = dctr_ceil_div(max_n / 4 + 1, (int64_t)(256 * f4));
    if (bx > 4096) bx = 4096;
    const dim3 grid((unsigned)bx, (unsigned)n_segs);
    hipStream_t st = (hipStream_t)stream;
#define DCTR_OPT_LAUNCH(K)                                                                                                                     \
    do {                                                                                                                                       \
        if (pen != nullptr) hipLaunchKernelGGL((opt_multi_kernel<K, 2, true, true>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, penalty_scale); \
        else if (variant == 0) hipLaunchKernelGGL((opt_multi_kernel<K, 2, false>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, 0.f);      \
        else if (variant == 1) hipLaunchKernelGGL((opt_multi_kernel<K, 4, false>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, 0.f); \
        else if (variant == 3) hipLaunchKernelGGL((opt_multi_kernel<K, 4, true>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, 0.f);  \
        else hipLaunchKernelGGL((opt_multi_kernel<K, 2, true>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, 0.f);                    \
    } while (0)
#define DCTR_OPT_LAUNCH2(K)                                                                                                                    \
    do {                                                                                                                                       \
        if (pen != nullptr) hipLaunchKernelGGL((opt_multi_kernel<K, 2, true, true>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, penalty_scale); \
        else hipLaunchKernelGGL((opt_multi_kernel<K, 2, true>), grid, dim3(256), 0, st, segs, lr, beta1, beta2, eps, (int)zero_grad, pen, 0.f);  \
    } while (0)
    double* pen = l2_penalty;
    switch (kind) {
        case DCTR_OPT_ADAM: DCTR_OPT_LAUNCH(DCTR_OPT_ADAM); break;
        case DCTR_OPT_ADAGRAD: DCTR_OPT_LAUNCH2(DCTR_OPT_ADAGRAD); break;
        case DCTR_OPT_RMSPROP: DCTR_OPT_LAUNCH2(DCTR_OPT_RMSPROP); break;
        default: DCTR_OPT_LAUNCH2(DCTR_OPT_SGD); break;
    }
#undef DCTR_OPT_LAUNCH
#undef DCTR_OPT_LAUNCH2
    return dctr_launch_status("dctr_opt_multi");
}

extern "C" int dctr_adam_multi(const dctr_adam_seg_t* segs, int32_t n_segs, int64_t max_n, float alpha, float beta1, float beta2,
                               float eps, int32_t zero_grad, void* stream) {
    return dctr_opt_multi(DCTR_OPT_ADAM, segs, n_segs, max_n, alpha, beta1, beta2, eps, zero_grad, stream);
}

// Dense(1, use_bias=False) on an arbitrary [B, N] input with row stride (DCN's head over the [cross, deep] stack):
// dx[b, n] = dlogit[b] * w[n];  d_w[n] += sum_b dlogit[b] * x[b, n]
extern "C" int dctr_dense1_bwd(const float* x, int64_t x_stride, int64_t batch, int32_t n, const float* w, const float* dlogit,
                               float* dx, int64_t dx_stride, float* d_w, void* stream) {
    DCTR_REQUIRE(batch >= 0 && n >= 1 && x_stride >= n && dx_stride >= n, DCTR_E_DIM, "dense1_bwd: bad sizes");
    if (batch == 0) return DCTR_OK;
    DCTR_REQUIRE(x && w && dlogit && dx && d_w, DCTR_E_NULL, "dense1_bwd: null pointer");
    launch_head_bwd((hipStream_t)stream, dlogit, w, x, x_stride, batch, (int)n, (int)DCTR_ACT_LINEAR, dx, dx_stride, d_w);
    return dctr_launch_status("dctr_dense1_bwd");
}

// vector form: does the one-kernel backward (x_l of a wave's sample + its d w / d b accumulators in LDS, the row's gradient in registers) hold it?
static bool cross_vector_bwd_on_chip(int d, int L) { return d <= 2048 && (size_t)4 * (3 * (size_t)L * d + L) * sizeof(float) <= 160 * 1024; }

extern "C" size_t dctr_crossnet_bwd_workspace_bytes(const dctr_crossnet_bwd_args_t* a) {
    if (a == nullptr || a->batch <= 0 || a->layers <= 0) return 0;
    if (a->mode == DCTR_CROSS_VECTOR) {
        // the layer-by-layer form of wide inputs: x_1 .. x_{L-1}, g, d x0 [B, d] each; s_0 .. s_{L-1}, ds [B] each
        if (a->dim < 1 || cross_vector_bwd_on_chip(a->dim, a->layers)) return 0;
        return ((size_t)(a->layers + 1) * a->batch * a->dim + (size_t)(a->layers + 1) * a->batch) * sizeof(float);
    }
    if (a->mode != DCTR_CROSS_MATRIX) return 0;
    // x_1 .. x_{L-1}, u_0 .. u_{L-1}, g, du, dx0; from 8192 rows on the partial dW of the row slices (as in dctr_mlp_bwd)
    const int parts = mlp_dw_parts(a->batch, (int64_t)a->dim * a->dim);
    return ((size_t)(2 * a->layers + 2) * a->batch * a->dim + (parts > 1 ? (size_t)parts * a->dim * a->dim : 0)) * sizeof(float);
}

extern "C" int dctr_crossnet_bwd(const dctr_crossnet_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "crossnet_bwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->dim >= 1 && a->layers >= 0, DCTR_E_DIM, "crossnet_bwd: bad sizes");
    DCTR_REQUIRE(a->mode == DCTR_CROSS_VECTOR || a->mode == DCTR_CROSS_MATRIX, DCTR_E_ENUM, "crossnet_bwd: mode %d", a->mode);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->dy && a->dx && a->x_stride >= a->dim && a->dy_stride >= a->dim && a->dx_stride >= a->dim, DCTR_E_NULL,
                 "crossnet_bwd: null pointer or stride < dim");
    hipStream_t st = (hipStream_t)stream;
    const int d = a->dim, L = a->layers;
    const unsigned eb = (unsigned)(dctr_ceil_div(a->batch * d, (int64_t)256) > 4096 ? 4096 : dctr_ceil_div(a->batch * d, (int64_t)256));
    if (L == 0) {
        hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, a->dy, a->dy_stride, a->batch, d, a->dx, a->dx_stride,
                           (int)a->dx_accumulate);
        return dctr_launch_status("dctr_crossnet_bwd");
    }
    DCTR_REQUIRE(a->kernels && a->bias && a->d_kernels && a->d_bias, DCTR_E_NULL, "crossnet_bwd: null weights / gradients");
    if (a->mode == DCTR_CROSS_VECTOR && !cross_vector_bwd_on_chip(d, L)) {
        // any width (round 6; the reference's CrossNet has none, interaction.py:405-424): layer by layer over the batch
        DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= dctr_crossnet_bwd_workspace_bytes(a), DCTR_E_NULL,
                     "crossnet_bwd(vector): %d layers x dim %d need a workspace of dctr_crossnet_bwd_workspace_bytes() bytes", L, d);
        const size_t bd = (size_t)a->batch * d;
        float* ws = static_cast<float*>(a->workspace);
        float* xsave = ws;                                   // x_1 .. x_{L-1}
        float* g = ws + (size_t)(L - 1) * bd;
        float* dx0 = g + bd;
        float* sl = dx0 + bd;                                // [L][B]
        float* ds = sl + (size_t)L * a->batch;               // [B]
        const unsigned wb = (unsigned)dctr_ceil_div(a->batch, (int64_t)4);
        DCTR_REQUIRE(dctr_ceil_div(a->batch, (int64_t)4) <= 0x7fffffffLL, DCTR_E_DIM, "crossnet_bwd: batch too large");
        auto xl_of = [&](int l, const float*& p, int64_t& ld) {
            if (l == 0) { p = a->x; ld = a->x_stride; }
            else { p = xsave + (size_t)(l - 1) * bd; ld = d; }
        };
        for (int l = 0; l < L; ++l) {
            const float* xl; int64_t ldx;
            xl_of(l, xl, ldx);
            hipLaunchKernelGGL(cross_vec_fwd_step_kernel, dim3(wb), dim3(256), 0, st, a->x, a->x_stride, xl, ldx, a->kernels + (size_t)l * d,
                               a->bias + (size_t)l * d, a->batch, d, sl + (size_t)l * a->batch, l + 1 < L ? xsave + (size_t)l * bd : (float*)nullptr);
        }
        hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, a->dy, a->dy_stride, a->batch, d, g, (int64_t)d, 0);
        hipError_t me = hipMemsetAsync(dx0, 0, bd * sizeof(float), st);
        DCTR_REQUIRE(me == hipSuccess, (int)me, "crossnet_bwd: memset failed: %s", hipGetErrorString(me));
        int64_t slices = dctr_ceil_div(a->batch, (int64_t)64);
        if (slices > 64) slices = 64;
        for (int l = L - 1; l >= 0; --l) {
            const float* xl; int64_t ldx;
            xl_of(l, xl, ldx);
            hipLaunchKernelGGL(cross_vec_dot_kernel, dim3(wb), dim3(256), 0, st, (const float*)g, a->x, a->x_stride, a->batch, d, ds);
            hipLaunchKernelGGL(cross_vec_colsum_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)slices), dim3(256), 0, st, (const float*)g,
                               (const float*)ds, xl, ldx, a->batch, d, a->d_kernels + (size_t)l * d, a->d_bias + (size_t)l * d);
            hipLaunchKernelGGL(cross_vec_update_kernel, dim3(eb), dim3(256), 0, st, g, dx0, (const float*)ds, (const float*)(sl + (size_t)l * a->batch),
                               a->kernels + (size_t)l * d, a->batch, d);
        }
        hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, (const float*)g, (int64_t)d, a->batch, d, dx0, (int64_t)d, 1);
        hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, (const float*)dx0, (int64_t)d, a->batch, d, a->dx, a->dx_stride,
                           (int)a->dx_accumulate);
        return dctr_launch_status("dctr_crossnet_bwd");
    }
    if (a->mode == DCTR_CROSS_VECTOR) {
        const size_t lds = (size_t)4 * (3 * (size_t)L * d + L) * sizeof(float);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)cross_vector_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DCTR_REQUIRE(e == hipSuccess, (int)e, "crossnet_bwd: cannot raise dynamic LDS: %s", hipGetErrorString(e));
        }
        const int64_t blocks = dctr_ceil_div(a->batch, (int64_t)(4 * CROSS_SPW));
        hipLaunchKernelGGL(cross_vector_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a->x, a->x_stride, a->batch, d, L,
                           a->kernels, a->bias, a->dy, a->dy_stride, a->d_kernels, a->d_bias, a->dx, a->dx_stride,
                           (int)a->dx_accumulate);
        return dctr_launch_status("dctr_crossnet_bwd");
    }
    // matrix: dctr_gemm GEMMs + elementwise kernels, intermediates in the workspace
    DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= dctr_crossnet_bwd_workspace_bytes(a) && a->batch < 0x7fffffffLL,
                 DCTR_E_NULL, "crossnet_bwd(matrix): needs a workspace of dctr_crossnet_bwd_workspace_bytes() bytes");
    const int B = (int)a->batch;
    const size_t bd = (size_t)a->batch * d;
    float* ws = static_cast<float*>(a->workspace);
    float* xsave = ws;                       // x_1 .. x_{L-1}  (x_0 = a->x)
    float* us = ws + (size_t)(L - 1) * bd;   // u_0 .. u_{L-1}
    float* g = us + (size_t)L * bd;
    float* du = g + bd;
    float* dx0 = du + bd;
    float* dw_parts = dx0 + bd;
    const int n_parts = mlp_dw_parts(a->batch, (int64_t)a->dim * a->dim);
    // (saved_u / saved_x: the forward kernel's own u_l and x_l — no recompute)
    const bool saved = a->saved_u != nullptr && (L == 1 || a->saved_x != nullptr);
    const float* us_r = saved ? a->saved_u : us;
    const float* xs_r = saved ? a->saved_x : xsave;
    auto xl_of = [&](int l, const float*& p, int& ld) {
        if (l == 0) { p = a->x; ld = (int)a->x_stride; }
        else { p = xs_r + (size_t)(l - 1) * bd; ld = d; }
    };
    for (int l = 0; l < L && !saved; ++l) {  // forward recompute: u_l = x_l W_l^T;  x_{l+1} = x0 .* (u_l + b_l) + x_l
        const float* xl; int ldx;
        xl_of(l, xl, ldx);
        const float* W = a->kernels + (size_t)l * d * d;
        int rs = dctr_gemm::sgemm(st, dctr_gemm::OP_T, dctr_gemm::OP_N, d, B, d, W, d, xl, ldx, 0.f, us + (size_t)l * bd, d);
        DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "crossnet_bwd: sgemm(u) failed (%d)", (int)rs);
        if (l + 1 < L)
            hipLaunchKernelGGL(cross_matrix_fwd_elem_kernel, dim3(eb), dim3(256), 0, st, a->x, a->x_stride, xl, (int64_t)ldx,
                               us + (size_t)l * bd, a->bias + (size_t)l * d, a->batch, d, xsave + (size_t)l * bd);
    }
    hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, a->dy, a->dy_stride, a->batch, d, g, (int64_t)d, 0);
    hipError_t me = hipMemsetAsync(dx0, 0, bd * sizeof(float), st);
    DCTR_REQUIRE(me == hipSuccess, (int)me, "crossnet_bwd: memset failed: %s", hipGetErrorString(me));
    for (int l = L - 1; l >= 0; --l) {
        const float* xl; int ldx;
        xl_of(l, xl, ldx);
        const float* W = a->kernels + (size_t)l * d * d;
        hipLaunchKernelGGL(cross_matrix_bwd_elem_kernel, dim3(eb), dim3(256), 0, st, a->x, a->x_stride, g, us_r + (size_t)l * bd,
                           a->bias + (size_t)l * d, a->batch, d, du, dx0);
        launch_act_bwd_colsum(st, du, (const float*)nullptr, a->batch, d, 0, a->d_bias + (size_t)l * d);
        // dW[n][k] += sum_b du[b][n] x_l[b][k]:  column-major  dW'(k x n) = X'(k x B) * du'(n x B)^T
        // g[b][k] += sum_n du[b][n] W[n][k]:  column-major  g'(k x B) += W'(k x n) * du'(n x B)
        // both read du and nothing of each other: ONE grouped launch (the dW reduction over the batch as row slices + a sum)
        int rs;
        if (n_parts > 1) {
            const int rs_ = B / n_parts;
            const int64_t dd = (int64_t)d * d;
            dctr_gemm::GroupDesc gd[2];
            gd[0] = dctr_gemm::GroupDesc{};
            gd[0].op_a = dctr_gemm::OP_N; gd[0].op_b = dctr_gemm::OP_T; gd[0].m = d; gd[0].n = d; gd[0].k = rs_;
            gd[0].A = xl; gd[0].lda = ldx; gd[0].stride_a = (int64_t)rs_ * ldx;
            gd[0].B = du; gd[0].ldb = d; gd[0].stride_b = (int64_t)rs_ * d;
            gd[0].C = dw_parts; gd[0].ldc = d; gd[0].stride_c = dd; gd[0].batch = n_parts;
            gd[1] = dctr_gemm::GroupDesc{};
            gd[1].op_a = dctr_gemm::OP_N; gd[1].op_b = dctr_gemm::OP_N; gd[1].m = d; gd[1].n = B; gd[1].k = d;
            gd[1].A = W; gd[1].lda = d; gd[1].B = du; gd[1].ldb = d; gd[1].C = g; gd[1].ldc = d; gd[1].batch = 1; gd[1].accumulate = 1;
            rs = dctr_gemm::sgemm_grouped(st, gd, 2);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "crossnet_bwd: sgemm_grouped(dW, g) failed (%d)", (int)rs);
            int64_t gp = dctr_ceil_div(dd, (int64_t)256);
            hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)(gp > 8192 ? 8192 : gp)), dim3(256), 0, st, (const float*)dw_parts, dd, n_parts,
                               a->d_kernels + (size_t)l * d * d);
        } else {
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_T, d, d, B, xl, ldx, du, d, 1.f, a->d_kernels + (size_t)l * d * d, d);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "crossnet_bwd: sgemm(dW) failed (%d)", (int)rs);
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_N, d, B, d, W, d, du, d, 1.f, g, d);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "crossnet_bwd: sgemm(g) failed (%d)", (int)rs);
        }
    }
    // d x0 = dx0 + g  (x_0 is also the first x_l)
    hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, g, (int64_t)d, a->batch, d, dx0, (int64_t)d, 1);
    hipLaunchKernelGGL(add_rows_kernel, dim3(eb), dim3(256), 0, st, dx0, (int64_t)d, a->batch, d, a->dx, a->dx_stride,
                       (int)a->dx_accumulate);
    return dctr_launch_status("dctr_crossnet_bwd");
}

// ---------------------------------------------------------------------------------------------------
// CIN.call, layer by layer (interaction.py:277-325 as the reference writes it: z materialised per layer, 1x1 conv = GEMM) — the route of
// dctr_cin_fwd for layer sizes no LDS tile of the one-kernel form holds (a layer of more than ~480 maps: cin_kernels.hip refuses it for
// every tile height).  Rows r = (b, d); samples in chunks of what the workspace holds; per chunk and layer: z = x_0 (outer) x_k
// (cin_outer_kernel), y = z W_k on dctr_gemm, bias + activation in place, the direct maps summed over d into `out`; y goes straight to
// save_y[k] when the caller asked for the activations (the same [B * D, H_k] rows).  Not part of the ABI: called by cin_kernels.hip.
// ---------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void cin_bias_act_kernel(float* __restrict__ y, int64_t rows, int H, const float* __restrict__ bias, int act) {
    const int64_t total = rows * H;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
        y[o] = act_value(y[o] + bias[(int)(o % H)], act);
}
// out[b, off + j] = sum_d y[(b D + d), d0 + j]   (j < nd): deterministic serial sum over d, as the one-kernel form's LDS sum
__global__ __launch_bounds__(256) void cin_sum_d_kernel(const float* __restrict__ y, int H, int64_t batch, int D, int d0, int nd,
                                                        float* __restrict__ out, int64_t out_dim, int off) {
    const int64_t total = batch * nd;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / nd;
        const int j = (int)(o - b * nd);
        const float* yp = y + (b * D) * H + d0 + j;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += yp[(int64_t)d * H];
        out[b * out_dim + off + j] = acc;
    }
}
}  // namespace

// floats of workspace per SAMPLE of the layered route: x0t [D, F0] + z [D, max F0 F_k] + two y buffers [D, max H]
size_t dctr_cin_layered_sample_floats(const dctr_cin_args_t* a) {
    size_t kmax = 0, hmax = 0;
    int fk = a->fields;
    for (int k = 0; k < a->n_layers; ++k) {
        const int H = a->layer_size[k];
        const bool last = k == a->n_layers - 1;
        kmax = (size_t)a->fields * fk > kmax ? (size_t)a->fields * fk : kmax;
        hmax = (size_t)H > hmax ? (size_t)H : hmax;
        fk = last ? 0 : (a->split_half ? H / 2 : H);
    }
    return (size_t)a->dim * ((((size_t)a->fields + 3) & ~(size_t)3) + ((kmax + 3) & ~(size_t)3) + 2 * ((hmax + 3) & ~(size_t)3));
}

int dctr_cin_fwd_layered(const dctr_cin_args_t* a, void* workspace, size_t workspace_bytes, void* stream) {
    const int F0 = a->fields, D = a->dim, L = a->n_layers;
    const size_t per = dctr_cin_layered_sample_floats(a) * sizeof(float);
    DCTR_REQUIRE(workspace != nullptr && dctr_aligned16(workspace) && workspace_bytes >= 16 * per, DCTR_E_NULL,
                 "cin_fwd: these layer sizes run layer by layer and need a 16-B aligned workspace (dctr_cin_workspace_bytes; at least %zu B)", 16 * per);
    DCTR_REQUIRE(a->x != nullptr && a->out != nullptr && a->filters != nullptr && a->bias != nullptr, DCTR_E_NULL, "cin_fwd: null pointer");
    int64_t cap = (int64_t)(workspace_bytes / per);
    cap = cap > 65536 ? 65536 : cap & ~(int64_t)3;
    size_t kmax = 0, hmax = 0;
    int out_dim = 0;
    {
        int fk = F0;
        for (int k = 0; k < L; ++k) {
            const int H = a->layer_size[k];
            const bool last = k == L - 1;
            DCTR_REQUIRE(a->filters[k] && a->bias[k], DCTR_E_NULL, "cin_fwd: filters/bias[%d] null", k);
            kmax = (size_t)F0 * fk > kmax ? (size_t)F0 * fk : kmax;
            hmax = (size_t)H > hmax ? (size_t)H : hmax;
            out_dim += a->split_half ? (last ? H : H - H / 2) : H;
            fk = last ? 0 : (a->split_half ? H / 2 : H);
        }
    }
    // (the GEMM's sizes and element offsets are ints: a chunk's z stays below 2^31 elements)
    while (cap > 16 && (int64_t)cap * D * (int64_t)(kmax > hmax ? kmax : hmax) >= 0x7fffffffLL) cap = (cap >> 1) & ~(int64_t)3;
    DCTR_REQUIRE((int64_t)cap * D * (int64_t)(kmax > hmax ? kmax : hmax) < 0x7fffffffLL, DCTR_E_DIM, "cin_fwd: layer too large (%zu products per row)", kmax);
    float* ws = static_cast<float*>(workspace);
    const size_t f0p = ((size_t)F0 + 3) & ~(size_t)3, kp = (kmax + 3) & ~(size_t)3, hp = (hmax + 3) & ~(size_t)3;
    float* x0t = ws;
    float* z = x0t + (size_t)cap * D * f0p;
    float* ybuf[2] = {z + (size_t)cap * D * kp, z + (size_t)cap * D * kp + (size_t)cap * D * hp};
    hipStream_t st = (hipStream_t)stream;
    auto grid = [](int64_t n) { int64_t b = dctr_ceil_div(n, (int64_t)256); return dim3((unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b))); };
    for (int64_t r0 = 0; r0 < a->batch; r0 += cap) {
        const int64_t nb = a->batch - r0 < cap ? a->batch - r0 : cap;
        const int64_t R = nb * D;
        hipLaunchKernelGGL(cin_to_rows_kernel, grid(R * F0), dim3(256), 0, st, a->x + r0 * a->x_stride, a->x_stride, nb, F0, D, x0t);
        const float* xk = x0t;
        int64_t ldk = F0;
        int fk = F0, off = 0;
        for (int k = 0; k < L; ++k) {
            const int H = a->layer_size[k];
            const bool last = k == L - 1;
            const int Hn = last ? 0 : (a->split_half ? H / 2 : H), d0 = a->split_half ? (last ? 0 : H / 2) : 0;
            const int K = F0 * fk;
            hipLaunchKernelGGL(cin_outer_kernel, grid(R * K), dim3(256), 0, st, (const float*)x0t, F0, xk, ldk, fk, R, z);
            float* y = (a->save_y != nullptr && a->save_y[k] != nullptr) ? a->save_y[k] + (size_t)r0 * D * H : ybuf[k & 1];
            // row-major y [R, H] = z [R, K] W [K, H]  <=>  column-major y' (H x R) = W' (H x K) z' (K x R)
            const int rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_N, H, (int)R, K, a->filters[k], H, z, K, 0.f, y, H);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "cin_fwd: sgemm(layer %d) failed (%d)", k, rs);
            hipLaunchKernelGGL(cin_bias_act_kernel, grid(R * H), dim3(256), 0, st, y, R, H, a->bias[k], (int)a->activation);
            hipLaunchKernelGGL(cin_sum_d_kernel, grid(nb * (H - d0)), dim3(256), 0, st, (const float*)y, H, nb, D, d0, H - d0, a->out + r0 * out_dim,
                               (int64_t)out_dim, off);
            off += H - d0;
            xk = y;
            ldk = H;
            fk = Hn;
        }
    }
    return dctr_launch_status("dctr_cin_fwd");
}

namespace dctr_cinbwd {      // cin_bwd_kernels.hip: the z-free backward of one CIN layer (filter gradient; input gradients)
bool fused_shape_ok(int F0, int Fk, int H);
int64_t dw_parts_floats(int F0, int Fk, int H, int64_t rows);
int launch_dw_fused(const float* dpre, const float* x0t, const float* xk, int64_t ldk, int F0, int Fk, int H, int64_t rows,
                    float* parts, int* n_parts, hipStream_t st);
int launch_dz_fused(const float* dpre, const float* W, const float* x0t, const float* xk, int64_t ldk, int F0, int Fk, int H,
                    int64_t rows, float* dx0t, float* dxk, hipStream_t st);
}  // namespace dctr_cinbwd

namespace {
struct CinPlan {
    int L, F0, D;
    int H[8], Fk[8], Hn[8], d0[8], off[8];
    bool fused[8];         // layer runs on the z-free kernels of cin_bwd_kernels.hip: no z / dz for it
    int64_t R;
    size_t x0t, y[8], z[8], dpre, dz, dx0t, dxk[2], fwd_out, fwd_ws, fwd_ws_bytes, parts, total;
    int out_dim;
};
bool cin_plan(const dctr_cin_args_t* f, CinPlan& p) {
    p.L = f->n_layers;
    p.F0 = f->fields;
    p.D = f->dim;
    p.R = f->batch * (int64_t)f->dim;
    if (p.L < 1 || p.L > 8) return false;
    int fk = p.F0, off = 0;
    size_t cur = 0;
    auto take = [&](size_t n) { size_t o = cur; cur += (n + 3) & ~(size_t)3; return o; };
    p.x0t = take((size_t)p.R * p.F0);
    size_t zmax = 0, hmax = 0, fkmax = 0, pmax = 0;
    for (int k = 0; k < p.L; ++k) {
        const int H = f->layer_size[k];
        const bool last = k == p.L - 1;
        p.H[k] = H;
        p.Fk[k] = fk;
        p.Hn[k] = last ? 0 : (f->split_half ? H / 2 : H);
        p.d0[k] = f->split_half ? (last ? 0 : H / 2) : 0;
        p.off[k] = off;
        off += H - p.d0[k];
        p.y[k] = take((size_t)p.R * H);
        p.fused[k] = dctr_cinbwd::fused_shape_ok(p.F0, fk, H);
        p.z[k] = take(p.fused[k] ? 0 : (size_t)p.R * p.F0 * fk);
        if (!p.fused[k]) zmax = (size_t)p.R * p.F0 * fk > zmax ? (size_t)p.R * p.F0 * fk : zmax;
        else {
            const size_t pf = (size_t)dctr_cinbwd::dw_parts_floats(p.F0, fk, H, p.R);
            pmax = pf > pmax ? pf : pmax;
        }
        hmax = (size_t)H > hmax ? H : hmax;
        fkmax = (size_t)fk > fkmax ? fk : fkmax;
        fk = p.Hn[k];
    }
    p.out_dim = off;
    p.fwd_out = take((size_t)f->batch * off);          // the forward kernel's [B, featuremap_num] output when it is re-run for y_k
    p.dpre = take((size_t)p.R * hmax);
    p.dz = take(zmax);
    p.parts = take(pmax);                              // the z-free dW kernel's per-row-slice partial products
    p.dx0t = take((size_t)p.R * p.F0);
    p.dxk[0] = take((size_t)p.R * fkmax);
    p.dxk[1] = take((size_t)p.R * fkmax);
    // the re-run forward's own workspace (layer 0's fold; REQUIRED by the sliced / layer-by-layer routes of wide samples / layers)
    p.fwd_ws_bytes = dctr_cin_workspace_bytes(f);
    p.fwd_ws = take((p.fwd_ws_bytes + 3) / 4);
    p.total = cur;
    return true;
}
}  // namespace

extern "C" size_t dctr_cin_bwd_workspace_bytes(const dctr_cin_bwd_args_t* a) {
    CinPlan p;
    if (a == nullptr || a->fwd == nullptr || a->fwd->batch <= 0 || !cin_plan(a->fwd, p)) return 0;
    return p.total * sizeof(float);
}

extern "C" int dctr_cin_bwd(const dctr_cin_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr && a->fwd != nullptr, DCTR_E_NULL, "cin_bwd: null args");
    const dctr_cin_args_t* f = a->fwd;
    DCTR_REQUIRE(f->batch >= 0 && f->fields >= 1 && f->dim >= 1, DCTR_E_DIM, "cin_bwd: bad sizes");
    if (f->batch == 0) return DCTR_OK;
    CinPlan p;
    DCTR_REQUIRE(cin_plan(f, p), DCTR_E_DIM, "cin_bwd: 1..8 layers");
    DCTR_REQUIRE(f->x && f->layer_size && f->filters && f->bias && a->d_out && a->d_filters && a->d_bias, DCTR_E_NULL,
                 "cin_bwd: null pointer");
    DCTR_REQUIRE(f->activation >= DCTR_ACT_LINEAR && f->activation <= DCTR_ACT_TANH, DCTR_E_ENUM, "cin_bwd: activation %d",
                 f->activation);
    DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= p.total * sizeof(float), DCTR_E_NULL,
                 "cin_bwd: needs a workspace of dctr_cin_bwd_workspace_bytes() bytes");
    DCTR_REQUIRE(p.R < 0x7fffffffLL, DCTR_E_DIM, "cin_bwd: batch * dim too large");
    hipStream_t st = (hipStream_t)stream;
    float* ws = static_cast<float*>(a->workspace);
    const int R = (int)p.R, F0 = p.F0, D = p.D;
    auto grid = [](int64_t n) { int64_t b = dctr_ceil_div(n, (int64_t)256); return dim3((unsigned)(b > 8192 ? 8192 : b)); };
    float* x0t = ws + p.x0t;
    hipLaunchKernelGGL(cin_to_rows_kernel, grid(p.R * F0), dim3(256), 0, st, f->x, f->x_stride, f->batch, F0, D, x0t);
    // the activations y_k: written by the forward call (saved_y), else the forward kernel is re-run here with the workspace as its
    // save_y (one launch; the first version recomputed them as z W with a GEMM per layer)
    const float* yk[8];
    bool rerun = false;
    for (int k = 0; k < p.L; ++k) {
        yk[k] = (a->saved_y != nullptr && a->saved_y[k] != nullptr) ? a->saved_y[k] : ws + p.y[k];
        rerun = rerun || yk[k] == ws + p.y[k];
    }
    if (rerun) {
        dctr_cin_args_t fa = *f;
        float* sv[8];
        for (int k = 0; k < p.L; ++k) sv[k] = yk[k] == ws + p.y[k] ? ws + p.y[k] : nullptr;
        fa.save_y = sv;
        fa.out = ws + p.fwd_out;
        if (p.fwd_ws_bytes > 0 && (fa.workspace == nullptr || fa.workspace_bytes < p.fwd_ws_bytes)) {   // (fwd->workspace is documented unused here)
            fa.workspace = ws + p.fwd_ws;
            fa.workspace_bytes = p.fwd_ws_bytes;
            fa.workspace_ready = 0;
        }
        const int rc = dctr_cin_fwd(&fa, stream);
        if (rc != 0) return rc;
    }
    // z only for layers outside the z-free kernels' shapes (the dW GEMM's operand)
    for (int k = 0; k < p.L; ++k) {
        if (p.fused[k]) continue;
        const int Fk = p.Fk[k], K = F0 * Fk;
        const float* xk = k == 0 ? x0t : yk[k - 1];
        const int64_t ldk = k == 0 ? F0 : p.H[k - 1];
        float* z = ws + p.z[k];
        if (Fk % 4 == 0 && ldk % 4 == 0 && dctr_aligned16(xk) && dctr_aligned16(z))
            hipLaunchKernelGGL(cin_outer4_kernel, grid(p.R * (K / 4)), dim3(256), 0, st, x0t, F0, xk, ldk, Fk, p.R, z);
        else
            hipLaunchKernelGGL(cin_outer_kernel, grid(p.R * K), dim3(256), 0, st, x0t, F0, xk, ldk, Fk, p.R, z);
    }
    hipError_t me = hipMemsetAsync(ws + p.dx0t, 0, (size_t)p.R * F0 * sizeof(float), st);
    DCTR_REQUIRE(me == hipSuccess, (int)me, "cin_bwd: memset failed: %s", hipGetErrorString(me));
    const float* dxnext = nullptr;
    int64_t ldn = 0;
    for (int k = p.L - 1; k >= 0; --k) {
        const int Fk = p.Fk[k], H = p.H[k], K = F0 * Fk;
        const float* xk = k == 0 ? x0t : yk[k - 1];
        const int64_t ldk = k == 0 ? F0 : p.H[k - 1];
        float* dpre = ws + p.dpre;
        float* dz = ws + p.dz;
        hipLaunchKernelGGL(cin_dpre_kernel, grid(p.R * H), dim3(256), 0, st, yk[k], dxnext, ldn, p.Hn[k], a->d_out,
                           (int64_t)a->out_dim, p.off[k], p.d0[k], p.R, H, D, (int)f->activation, dpre);
        launch_act_bwd_colsum(st, dpre, (const float*)nullptr, p.R, H, 0, a->d_bias[k]);
        // dW[K,H] += z^T dpre:  column-major  dW'(H x K) = dpre'(H x R) z'(K x R)^T.  The output is small (H x K) and the
        // reduction long (R = B*D): as ONE gemm it runs on ~18 workgroups (1.8 ms at C3); split the rows into `parts`
        // slices computed as a strided batch of partial products in the (now free) dz buffer, then summed.
        const int64_t hk = (int64_t)H * K;
        int parts = (int)((size_t)p.R * K / (size_t)hk);                 // partials fit the dz buffer: parts*H*K <= R*K
        if (parts > 32) parts = 32;
        while (parts > 1 && R % parts != 0) --parts;
        int rs = 0;
        if (p.fused[k]) {
            // z-free: x0[r,i] xk[r,j] formed in registers as the MFMA A operand, rows = the K dimension (cin_bwd_kernels.hip)
            DCTR_REQUIRE(dctr_aligned16(dpre) && dctr_aligned16(f->filters[k]), DCTR_E_ALIGN,
                         "cin_bwd: workspace / filters[%d] must be 16-B aligned", k);
            int n_parts = 0;
            const int rc = dctr_cinbwd::launch_dw_fused(dpre, x0t, xk, ldk, F0, Fk, H, p.R, ws + p.parts, &n_parts, st);
            DCTR_REQUIRE(rc == 0, rc, "cin_bwd: cannot launch the fused dW kernel (%d)", rc);
            hipLaunchKernelGGL(sum_parts_kernel, grid(hk), dim3(256), 0, st, (const float*)(ws + p.parts), hk, n_parts, a->d_filters[k]);
        } else if (parts > 1) {
            const int rs_ = R / parts;
            rs = dctr_gemm::sgemm_strided_batched(st, dctr_gemm::OP_N, dctr_gemm::OP_T, H, K, rs_, dpre, H, (int64_t)rs_ * H, ws + p.z[k], K, (int64_t)rs_ * K, 0.f, dz, H, (int64_t)hk, parts);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "cin_bwd: sgemm_strided_batched(dW) failed (%d)", (int)rs);
            hipLaunchKernelGGL(sum_parts_kernel, grid(hk), dim3(256), 0, st, dz, hk, parts, a->d_filters[k]);
        } else {
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_N, dctr_gemm::OP_T, H, K, R, dpre, H, ws + p.z[k], K, 1.f, a->d_filters[k], H);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "cin_bwd: sgemm(dW) failed (%d)", (int)rs);
        }
        float* dxk = ws + p.dxk[k & 1];          // layer 0: x_0 is also its x_k; that second-factor gradient lands in dxk[0]
        if (p.fused[k]) {
            // dz = dpre W^T is formed tile by tile on the matrix cores and contracted with x_0 / x_k at once (never stored)
            const int rc = dctr_cinbwd::launch_dz_fused(dpre, f->filters[k], x0t, xk, ldk, F0, Fk, H, p.R, ws + p.dx0t, dxk, st);
            DCTR_REQUIRE(rc == 0, rc, "cin_bwd: cannot launch the fused dz kernel (%d)", rc);
        } else {
            // dz[R,K] = dpre[R,H] W^T:  column-major  dz'(K x R) = W'(H x K)^T dpre'(H x R)
            rs = dctr_gemm::sgemm(st, dctr_gemm::OP_T, dctr_gemm::OP_N, K, R, H, f->filters[k], H, dpre, H, 0.f, dz, K);
            DCTR_REQUIRE(rs == 0, DCTR_E_UNSUPPORTED, "cin_bwd: sgemm(dz) failed (%d)", (int)rs);
            const size_t lds = (size_t)4 * (K + F0 + Fk) * sizeof(float);
            DCTR_REQUIRE(lds <= 64 * 1024, DCTR_E_UNSUPPORTED, "cin_bwd: F0*Fk = %d too large for the row-staging kernel", K);
            int64_t nb = dctr_ceil_div(p.R, (int64_t)4);
            if (nb > 256 * 16) nb = 256 * 16;
            hipLaunchKernelGGL(cin_outer_bwd_kernel, dim3((unsigned)nb), dim3(256), lds, st, dz, x0t, F0, xk, ldk, Fk, p.R,
                               ws + p.dx0t, dxk);
        }
        dxnext = dxk;
        ldn = Fk;
    }
    if (a->dx != nullptr) {
        // d x0 = dX0t (first factor, all layers) + layer 0's second factor (in dxk[0], [R, F0])
        hipLaunchKernelGGL(add_rows_kernel, grid(p.R * F0), dim3(256), 0, st, ws + p.dxk[0], (int64_t)F0, p.R, F0, ws + p.dx0t,
                           (int64_t)F0, 1);
        hipLaunchKernelGGL(cin_from_rows_kernel, grid(p.R * F0), dim3(256), 0, st, ws + p.dx0t, f->batch, F0, D, a->dx, a->dx_stride,
                           (int)a->dx_accumulate);
    }
    return dctr_launch_status("dctr_cin_bwd");
}

// ---------------------------------------------------------------------------------------------------
// CrossNetMix backward (reference forward: deepctr/layers/interaction.py:511-549, DCNMix):
//   per layer l, expert e:  v1 = tanh(x_l V_e),  v2 = tanh(v1 C_e^T),  out_e = x_0 .* (v2 U_e^T + b_l),  s_e = x_l . g_e
//   p = softmax_e(s),  x_{l+1} = sum_e p_e out_e + x_l
// Nothing is saved by the forward kernel: the intermediates (x_l, v1, v2, uv = v2 U^T, p) are recomputed here into the
// workspace — the low-rank projections as plain GEMMs (dctr_gemm), the gating / softmax / combination as one row kernel — and the
// backward walks the layers in reverse:
//   dp_e = g . out_e,  ds_e = p_e (dp_e - sum_e' p_e' dp_e'),  t_e = p_e g .* x_0   (gradient of  v2 U^T + b)
//   d b += sum_b g .* x_0 (sum_e p_e = 1),  d U_e += t_e^T v2,  d v2 = t_e U_e,  a2 = d v2 .* (1 - v2^2),  d C_e += a2^T v1,
//   d v1 = a2 C_e,  a1 = d v1 .* (1 - v1^2),  d V_e += x_l^T a1,  d g_e += x_l^T ds_e,
//   d x_l = g + sum_e (ds_e (x) g_e + a1 V_e^T),   d x_0 += sum_e p_e g .* (uv_e + b)
// ---------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void mix_copy_rows_kernel(const float* __restrict__ x, int64_t x_stride, int64_t batch, int d,
                                                           float* __restrict__ y) {
    const int64_t total = batch * d;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / d;
        y[o] = x[b * x_stride + (o - b * d)];
    }
}

__global__ __launch_bounds__(256) void mix_tanh_kernel(float* __restrict__ v, int64_t n) {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) v[o] = tanhf(v[o]);
}

// a = d .* (1 - v^2), in place over d
__global__ __launch_bounds__(256) void mix_tanh_bwd_kernel(float* __restrict__ dv, const float* __restrict__ v, int64_t n) {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) dv[o] *= 1.f - v[o] * v[o];
}

__device__ __forceinline__ float mix_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

constexpr int MIX_MAX_EXPERTS = 16;

// forward combination, one wave per row: p = softmax_e(x_l . g_e),  x_next = sum_e p_e x_0 .* (uv_e + b) + x_l
__global__ __launch_bounds__(256) void mix_combine_kernel(const float* __restrict__ x0, const float* __restrict__ xl,
                                                         const float* __restrict__ uv, int64_t uv_expert_stride,
                                                         const float* __restrict__ gating, const float* __restrict__ bias,
                                                         int64_t batch, int d, int ne, float* __restrict__ p_out,
                                                         float* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    float s[MIX_MAX_EXPERTS];
    for (int e = 0; e < ne; ++e) {
        float a = 0.f;
        for (int i = lane; i < d; i += 64) a = fmaf(xl[b * d + i], gating[(int64_t)e * d + i], a);
        s[e] = mix_wave_sum(a);
    }
    float mx = s[0];
    for (int e = 1; e < ne; ++e) mx = fmaxf(mx, s[e]);
    float den = 0.f;
    for (int e = 0; e < ne; ++e) {
        s[e] = expf(s[e] - mx);
        den += s[e];
    }
    for (int e = 0; e < ne; ++e) {
        s[e] /= den;
        if (lane == 0) p_out[b * ne + e] = s[e];
    }
    for (int i = lane; i < d; i += 64) {
        float acc = 0.f;
        for (int e = 0; e < ne; ++e) acc = fmaf(s[e], uv[e * uv_expert_stride + b * d + i], acc);
        xn[b * d + i] = fmaf(x0[b * d + i], acc + bias[i], xl[b * d + i]);        // sum_e p_e = 1: the bias comes out of the sum
    }
}

// backward gate part, one wave per row.  In: g (gradient w.r.t. x_{l+1}), x_0, uv_e, p.  Out: ds [B, ne];
// gx = g + sum_e ds_e g_e;  dx0 += sum_e p_e g .* (uv_e + b)      (d_bias[i] += sum_b g x_0: colsum_prod_kernel)
__global__ __launch_bounds__(256) void mix_bwd_gate_kernel(const float* __restrict__ g, const float* __restrict__ x0,
                                                          const float* __restrict__ uv, int64_t uv_expert_stride,
                                                          const float* __restrict__ p, const float* __restrict__ gating,
                                                          const float* __restrict__ bias, int64_t batch, int d, int ne,
                                                          float* __restrict__ ds, float* __restrict__ gx, float* __restrict__ dx0,
                                                          float* __restrict__ d_bias) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    float pe[MIX_MAX_EXPERTS], dp[MIX_MAX_EXPERTS];
    float mean = 0.f;
    for (int e = 0; e < ne; ++e) {
        pe[e] = p[b * ne + e];
        float a = 0.f;
        for (int i = lane; i < d; i += 64) a = fmaf(g[b * d + i], x0[b * d + i] * (uv[e * uv_expert_stride + b * d + i] + bias[i]), a);
        dp[e] = mix_wave_sum(a);
        mean = fmaf(pe[e], dp[e], mean);
    }
    for (int e = 0; e < ne; ++e) {
        dp[e] = pe[e] * (dp[e] - mean);                                           // now ds_e
        if (lane == 0) ds[b * ne + e] = dp[e];
    }
    for (int i = lane; i < d; i += 64) {
        const float gi = g[b * d + i];
        float a = gi, u = 0.f;
        for (int e = 0; e < ne; ++e) {
            a = fmaf(dp[e], gating[(int64_t)e * d + i], a);
            u = fmaf(pe[e], uv[e * uv_expert_stride + b * d + i], u);
        }
        gx[b * d + i] = a;
        dx0[b * d + i] = fmaf(gi, u + bias[i], dx0[b * d + i]);
    }
}

// out[n] += sum_b a[b, n] * c[b, n]   (the bias gradient of a CrossNetMix layer: one atomic per row and column from the gate kernel
// put B atomics on each address — 214 us per layer at B = 4096): rows strided over few workgroups, one atomic per column each
__global__ __launch_bounds__(256) void colsum_prod_kernel(const float* __restrict__ a, const float* __restrict__ c, int64_t batch, int d,
                                                         float* __restrict__ out) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= d) return;
    float acc = 0.f;
#pragma unroll 4
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) acc = fmaf(a[b * d + n], c[b * d + n], acc);
    unsafeAtomicAdd(out + n, acc);
}

// t[e][b, i] = p[b, e] g[b, i] x0[b, i] for every expert e
__global__ __launch_bounds__(256) void mix_t_all_kernel(const float* __restrict__ g, const float* __restrict__ x0, const float* __restrict__ p,
                                                       int ne, int64_t batch, int d, float* __restrict__ t) {
    const int64_t bd = batch * d, total = bd * ne;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int e = (int)(o / bd);
        const int64_t rem = o - (int64_t)e * bd;
        t[o] = p[(rem / d) * ne + e] * g[rem] * x0[rem];
    }
}

// dx[b, :] (+)= g[b, :] + dx0[b, :];  also used as a strided copy of dy into g
__global__ __launch_bounds__(256) void mix_out_kernel(const float* __restrict__ g, const float* __restrict__ dx0, int64_t batch, int d,
                                                     float* __restrict__ dx, int64_t dx_stride, int accumulate) {
    const int64_t total = batch * d;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / d;
        const float v = g[o] + (dx0 != nullptr ? dx0[o] : 0.f);
        float* dst = dx + b * dx_stride + (o - b * d);
        *dst = accumulate ? *dst + v : v;
    }
}

struct MixPlan {
    size_t X, V1, V2, UV, P, G, GX, T, DX0, DV2, DV1, DS, total;
};
MixPlan mix_plan(int64_t B, int d, int L, int ne, int r) {
    MixPlan m{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 3) & ~(size_t)3; return at; };
    m.X = take((size_t)L * B * d);
    m.V1 = take((size_t)L * ne * B * r);
    m.V2 = take((size_t)L * ne * B * r);
    m.UV = take((size_t)L * ne * B * d);
    m.P = take((size_t)L * B * ne);
    m.G = take((size_t)B * d);
    m.GX = take((size_t)B * d);
    m.T = take((size_t)ne * B * d);          // per expert: the GEMMs of a layer run as strided batches over the experts
    m.DX0 = take((size_t)B * d);
    m.DV2 = take((size_t)ne * B * r);
    m.DV1 = take((size_t)ne * B * r);
    m.DS = take((size_t)B * ne);
    m.total = o;
    return m;
}

}  // namespace

extern "C" size_t dctr_crossnet_mix_bwd_workspace_bytes(const dctr_crossnet_mix_bwd_args_t* a) {
    if (a == nullptr || a->batch <= 0 || a->dim <= 0 || a->layers <= 0 || a->experts <= 0 || a->low_rank <= 0) return 0;
    return mix_plan(a->batch, a->dim, a->layers, a->experts, a->low_rank).total * sizeof(float);
}

extern "C" int dctr_crossnet_mix_bwd(const dctr_crossnet_mix_bwd_args_t* a, void* stream) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "crossnet_mix_bwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->dim >= 1 && a->layers >= 0 && a->experts >= 1 && a->experts <= MIX_MAX_EXPERTS && a->low_rank >= 1,
                 DCTR_E_DIM, "crossnet_mix_bwd: bad sizes (batch=%lld dim=%d layers=%d experts=%d (max %d) low_rank=%d)",
                 (long long)a->batch, a->dim, a->layers, a->experts, MIX_MAX_EXPERTS, a->low_rank);
    if (a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x && a->dy && a->dx && a->x_stride >= a->dim && a->dy_stride >= a->dim && a->dx_stride >= a->dim, DCTR_E_NULL,
                 "crossnet_mix_bwd: null x / dy / dx or a stride < dim");
    hipStream_t st = (hipStream_t)stream;
    const int64_t B = a->batch;
    const int d = a->dim, L = a->layers, ne = a->experts, r = a->low_rank;
    auto grid = [](int64_t n) { int64_t g = dctr_ceil_div(n, (int64_t)256); return dim3((unsigned)(g < 8192 ? g : 8192)); };
    if (L == 0) {                                                  // identity: dx (+)= dy
        DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= (size_t)B * d * sizeof(float), DCTR_E_NULL,
                     "crossnet_mix_bwd: layers == 0 needs a workspace of batch * dim floats");
        hipLaunchKernelGGL(mix_copy_rows_kernel, grid(B * d), dim3(256), 0, st, a->dy, a->dy_stride, B, d, (float*)a->workspace);
        hipLaunchKernelGGL(mix_out_kernel, grid(B * d), dim3(256), 0, st, (const float*)a->workspace, (const float*)nullptr, B, d, a->dx,
                           a->dx_stride, (int)a->dx_accumulate);
        return dctr_launch_status("dctr_crossnet_mix_bwd");
    }
    DCTR_REQUIRE(a->U && a->V && a->C && a->gating && a->bias && a->dU && a->dV && a->dC && a->dgating && a->dbias, DCTR_E_NULL,
                 "crossnet_mix_bwd: null weights / gradients");
    const MixPlan m = mix_plan(B, d, L, ne, r);
    DCTR_REQUIRE(a->workspace != nullptr && a->workspace_bytes >= m.total * sizeof(float), DCTR_E_NULL,
                 "crossnet_mix_bwd: needs a workspace of dctr_crossnet_mix_bwd_workspace_bytes() bytes");
    DCTR_REQUIRE(B < 0x7fffffffLL / (d > r ? d : r), DCTR_E_DIM, "crossnet_mix_bwd: batch too large");
    float* ws = static_cast<float*>(a->workspace);
    const float one = 1.f, zero = 0.f;
    const int Bi = (int)B;
    const int64_t Br = B * r, Bd = B * d;
    const unsigned row_blocks = (unsigned)dctr_ceil_div(B, (int64_t)4);
#define MIX_GEMM(ta, tb, mm, nn, kk, A_, lda, B_, ldb, beta, C_, ldc)                                                      \
    do {                                                                                                                   \
        int rs_ = dctr_gemm::sgemm(st, ta, tb, mm, nn, kk, A_, lda, B_, ldb, *(beta), C_, ldc);                  \
        DCTR_REQUIRE(rs_ == 0, DCTR_E_UNSUPPORTED, "crossnet_mix_bwd: sgemm failed (%d)", (int)rs_); \
    } while (0)
#define MIX_BGEMM(ta, tb, mm, nn, kk, A_, lda, sa, B_, ldb, sb, beta, C_, ldc, sc)                                            \
    do {                                                                                                                   \
        int rs_ = dctr_gemm::sgemm_strided_batched(st, ta, tb, mm, nn, kk, A_, lda, (int64_t)(sa), B_, ldb, \
                                                           (int64_t)(sb), *(beta), C_, ldc, (int64_t)(sc), ne);     \
        DCTR_REQUIRE(rs_ == 0, DCTR_E_UNSUPPORTED, "crossnet_mix_bwd: sgemm_strided_batched failed (%d)", (int)rs_); \
    } while (0)
    const dctr_gemm::Op N_ = dctr_gemm::OP_N, T_ = dctr_gemm::OP_T;
    // ---- forward recompute.  X[0] = x_0 (contiguous copy), X[l+1] needed only for l + 1 < L
    hipLaunchKernelGGL(mix_copy_rows_kernel, grid(Bd), dim3(256), 0, st, a->x, a->x_stride, B, d, ws + m.X);
    const float* x0 = ws + m.X;
    for (int l = 0; l < L; ++l) {
        const float* xl = ws + m.X + (size_t)l * Bd;
        {
            // the experts of a layer as ONE strided batch per GEMM (4 experts x 3 GEMMs x 2 tanh launches per layer before: the
            // step spent 1.0 of its 1.74 ms in 106 small rocBLAS calls)
            const size_t l0 = (size_t)l * ne;
            const float* Ul = a->U + l0 * d * r;
            const float* Vl = a->V + l0 * d * r;
            const float* Cl = a->C + l0 * r * r;
            float* v1 = ws + m.V1 + l0 * Br;
            float* v2 = ws + m.V2 + l0 * Br;
            float* uv = ws + m.UV + l0 * Bd;
            MIX_BGEMM(N_, N_, r, Bi, d, Vl, r, (int64_t)d * r, xl, d, 0, &zero, v1, r, Br);      // v1'(r x B) = V'(r x d) x_l'(d x B)
            hipLaunchKernelGGL(mix_tanh_kernel, grid(ne * Br), dim3(256), 0, st, v1, ne * Br);
            MIX_BGEMM(T_, N_, r, Bi, r, Cl, r, (int64_t)r * r, (const float*)v1, r, Br, &zero, v2, r, Br);   // v2[b, j] = sum_k C[j][k] v1[b, k]
            hipLaunchKernelGGL(mix_tanh_kernel, grid(ne * Br), dim3(256), 0, st, v2, ne * Br);
            MIX_BGEMM(T_, N_, d, Bi, r, Ul, r, (int64_t)d * r, (const float*)v2, r, Br, &zero, uv, d, Bd);   // uv[b, i] = sum_j U[i][j] v2[b, j]
        }
        // the last layer's output is not needed (its gradient comes in as dy); the kernel still needs somewhere to write: GX
        float* xn = l + 1 < L ? ws + m.X + (size_t)(l + 1) * Bd : ws + m.GX;
        hipLaunchKernelGGL(mix_combine_kernel, dim3(row_blocks), dim3(256), 0, st, x0, xl, (const float*)(ws + m.UV + (size_t)l * ne * Bd),
                           Bd, a->gating, a->bias + (size_t)l * d, B, d, ne, ws + m.P + (size_t)l * B * ne, xn);
    }
    // ---- backward
    float* g = ws + m.G;
    float* gx = ws + m.GX;
    hipLaunchKernelGGL(mix_copy_rows_kernel, grid(Bd), dim3(256), 0, st, a->dy, a->dy_stride, B, d, g);
    hipError_t me = hipMemsetAsync(ws + m.DX0, 0, (size_t)Bd * sizeof(float), st);
    DCTR_REQUIRE(me == hipSuccess, (int)me, "crossnet_mix_bwd: memset failed: %s", hipGetErrorString(me));
    for (int l = L - 1; l >= 0; --l) {
        const float* xl = ws + m.X + (size_t)l * Bd;
        const float* p = ws + m.P + (size_t)l * B * ne;
        const float* uvl = ws + m.UV + (size_t)l * ne * Bd;
        hipLaunchKernelGGL(mix_bwd_gate_kernel, dim3(row_blocks), dim3(256), 0, st, (const float*)g, x0, uvl, Bd, p, a->gating,
                           a->bias + (size_t)l * d, B, d, ne, ws + m.DS, gx, ws + m.DX0, a->dbias + (size_t)l * d);
        hipLaunchKernelGGL(colsum_prod_kernel, dim3((unsigned)(B < COLSUM_MAX_WG ? B : COLSUM_MAX_WG), (unsigned)((d + 255) / 256)), dim3(256),
                           0, st, (const float*)g, x0, B, d, a->dbias + (size_t)l * d);
        {
            // Independent products share a launch (dctr_gemm::sgemm_grouped): {d gating, dU, dv2}, {dC, dv1}, {dV, gx of expert 0} —
            // 13 GEMM launches per layer were 6 + ne; the small outputs under the batch-long reduction (d gating, dU, dC, dV) add
            // their k slices with float atomics as dctr_gemm::sgemm does for such shapes
            const size_t l0 = (size_t)l * ne;
            const float* Ul = a->U + l0 * d * r;
            const float* Vl = a->V + l0 * d * r;
            const float* Cl = a->C + l0 * r * r;
            const float* v1 = ws + m.V1 + l0 * Br;
            const float* v2 = ws + m.V2 + l0 * Br;
            float* t = ws + m.T;
            float* dv2 = ws + m.DV2;
            float* dv1 = ws + m.DV1;
            const int ks = dctr_gemm::k_slices(Bi, 256) > 16 ? 16 : dctr_gemm::k_slices(Bi, 256);
            auto desc = [&](dctr_gemm::Op ta, dctr_gemm::Op tb, int mm, int nn, int kk, const float* A_, int lda, int64_t sa, const float* B_,
                            int ldb, int64_t sb, float* C_, int ldc, int64_t sc, int batch, bool acc, bool long_k) {
                dctr_gemm::GroupDesc gd_ = dctr_gemm::GroupDesc{};
                gd_.op_a = ta; gd_.op_b = tb; gd_.m = mm; gd_.n = nn; gd_.k = kk; gd_.A = A_; gd_.lda = lda; gd_.stride_a = sa;
                gd_.B = B_; gd_.ldb = ldb; gd_.stride_b = sb; gd_.C = C_; gd_.ldc = ldc; gd_.stride_c = sc; gd_.batch = batch;
                gd_.accumulate = acc ? 1 : 0;
                if (long_k && acc) gd_.k_slices = dctr_gemm::k_slices(kk, (kk + ks - 1) / ks);
                return gd_;
            };
            hipLaunchKernelGGL(mix_t_all_kernel, grid(ne * Bd), dim3(256), 0, st, (const float*)g, x0, p, ne, B, d, t);
            dctr_gemm::GroupDesc g1[3] = {
                // d gating[e][i] += sum_b ds[b, e] x_l[b, i]:  column-major dG'(d x ne) += x_l'(d x B) ds'(ne x B)^T
                desc(N_, T_, d, ne, Bi, xl, d, 0, (const float*)(ws + m.DS), ne, 0, a->dgating, d, 0, 1, true, true),
                // dU[i][j] += sum_b t[b,i] v2[b,j]
                desc(N_, T_, r, d, Bi, v2, r, Br, (const float*)t, d, Bd, a->dU + l0 * d * r, r, (int64_t)d * r, ne, true, true),
                // dv2[b,j] = sum_i t[b,i] U[i][j]
                desc(N_, N_, r, Bi, d, Ul, r, (int64_t)d * r, (const float*)t, d, Bd, dv2, r, Br, ne, false, false)};
            int rsg = dctr_gemm::sgemm_grouped(st, g1, 3);
            DCTR_REQUIRE(rsg == 0, DCTR_E_UNSUPPORTED, "crossnet_mix_bwd: sgemm_grouped(1) failed (%d)", rsg);
            hipLaunchKernelGGL(mix_tanh_bwd_kernel, grid(ne * Br), dim3(256), 0, st, dv2, v2, ne * Br);                    // a2
            dctr_gemm::GroupDesc g2[2] = {
                // dC[j][k] += sum_b a2[b,j] v1[b,k]
                desc(N_, T_, r, r, Bi, v1, r, Br, (const float*)dv2, r, Br, a->dC + l0 * r * r, r, (int64_t)r * r, ne, true, true),
                // dv1[b,k] = sum_j a2[b,j] C[j][k]
                desc(N_, N_, r, Bi, r, Cl, r, (int64_t)r * r, (const float*)dv2, r, Br, dv1, r, Br, ne, false, false)};
            rsg = dctr_gemm::sgemm_grouped(st, g2, 2);
            DCTR_REQUIRE(rsg == 0, DCTR_E_UNSUPPORTED, "crossnet_mix_bwd: sgemm_grouped(2) failed (%d)", rsg);
            hipLaunchKernelGGL(mix_tanh_bwd_kernel, grid(ne * Br), dim3(256), 0, st, dv1, v1, ne * Br);                    // a1
            dctr_gemm::GroupDesc g3[2] = {
                // dV[i][j] += sum_b x_l[b,i] a1[b,j]
                desc(N_, T_, r, d, Bi, (const float*)dv1, r, Br, xl, d, 0, a->dV + l0 * d * r, r, (int64_t)d * r, ne, true, true),
                // gx[b,i] += sum_j a1[b,j] V[i][j]: one output, expert by expert — expert 0 here, the others behind it
                desc(T_, N_, d, Bi, r, Vl, r, 0, (const float*)dv1, r, 0, gx, d, 0, 1, true, false)};
            rsg = dctr_gemm::sgemm_grouped(st, g3, 2);
            DCTR_REQUIRE(rsg == 0, DCTR_E_UNSUPPORTED, "crossnet_mix_bwd: sgemm_grouped(3) failed (%d)", rsg);
            for (int e = 1; e < ne; ++e)
                MIX_GEMM(T_, N_, d, Bi, r, Vl + (size_t)e * d * r, r, (const float*)(dv1 + (size_t)e * Br), r, &one, gx, d);
        }
        float* tmp = g;
        g = gx;
        gx = tmp;
    }
#undef MIX_GEMM
#undef MIX_BGEMM
    hipLaunchKernelGGL(mix_out_kernel, grid(Bd), dim3(256), 0, st, (const float*)g, (const float*)(ws + m.DX0), B, d, a->dx, a->dx_stride,
                       (int)a->dx_accumulate);
    return dctr_launch_status("dctr_crossnet_mix_bwd");
}

// The join dctr_mlp_bwd_args_t.dw_stream asks of the caller, as one call: `stream` waits for everything issued to `dw_stream` so far.
extern "C" int dctr_mlp_bwd_join(void* stream, void* dw_stream) {
    if (dw_stream == nullptr || dw_stream == stream) return DCTR_OK;
    static thread_local hipEvent_t ev[DCTR_MAX_DEVICES] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    DCTR_REQUIRE(e == hipSuccess && dev >= 0 && dev < DCTR_MAX_DEVICES, DCTR_E_UNSUPPORTED, "mlp_bwd_join: device %d", dev);
    if (ev[dev] == nullptr) {
        e = hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_bwd_join: hipEventCreate failed: %s", hipGetErrorString(e));
    }
    e = hipEventRecord(ev[dev], (hipStream_t)dw_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev[dev], 0);
    DCTR_REQUIRE(e == hipSuccess, (int)e, "mlp_bwd_join: %s", hipGetErrorString(e));
    return DCTR_OK;
}
