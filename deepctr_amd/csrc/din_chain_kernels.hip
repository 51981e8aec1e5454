// a13 — LocalActivationUnit.call over all B*T (sample, position) rows (reference deepctr/layers/core.py:94-108, Dice
// layers/activation.py:59-64), row-chained form: the attention MLP computed TRANSPOSED, as chain_device.h does for the DNN.
//
// din_score_kernel (din_kernels.hip) keeps a wave's activations in a wave-private LDS tile between the layers and reads the
// weights as the MFMA B operand, one ds_read_b32 per MFMA: 0.46 of the f32-MFMA rate.  Here
//   * out^T[n, row] = sum_k W'[k, n] x[row, k]: the WEIGHTS are the MFMA A operand (M = output features), a wave's 32 rows are
//     two 16-wide N tiles.  v_mfma_f32_16x16x4_f32 leaves C[m = 4g + r][n = j] in lane (g, j), register r, and wants
//     B[k = g][n = j] from lane (g, j): layer 0's accumulators ARE layer 1's B operand, register for register (k-slot g of
//     k-step (tile, r) is whatever output feature that register holds — a K permutation, matched by the weight row each lane
//     reads).  No activation ever touches LDS;
//   * all weights stay resident in LDS (the attention MLP is small: 3E x 80 + 80 x 40 floats = 74 KB at E = 64): no ring, no
//     DMA, NO barrier after the prologue — every wave runs its 32-row units on its own, four waves per SIMD hide each other's
//     global-load and LDS latencies;
//   * A fragments: the first 64 output features of layer 0 are read with ONE ds_read_b128 per k-step (lane (g, j) takes floats
//     4j .. 4j + 3 of weight row k: M-tile mt = feature 4 i + mt, four M-tiles x two N-tiles = 8 MFMAs per LDS instruction),
//     the remaining features (80 = 64 + 16) and layer 1 as 16-feature tiles of one ds_read_b32 each (2 MFMAs per instruction):
//     exact tile counts, no padding of 80 to 128;
//   * att_input = [q, k, q - k, q * k] W is evaluated as q (Wq + Wd) + k (Wk - Wd) + (q * k) Wp (K = 3E, as din_score_kernel
//     does; the two sums are formed while the weights are copied to LDS).  Lane (g, j) loads the 16-B piece g of row j's key /
//     query block with one global_load_dwordx4 — which IS the B operand of that block's four k-steps; the block after next is
//     requested before the current block's 120 MFMAs.
// Per row: raw score = Dense(1)(act(act(x W0 + b0) W1 + b1)); masking / softmax / the weighted sum of the keys stay in
// din_pool_kernel.  Eligibility (host, below): two layers, units[0] = 64 a + 16 b (a <= 1, b <= 3), units[1] <= 64,
// embedding_dim in {16, 32, 64}.  Everything else keeps din_score_kernel.
//
// Masked positions are SKIPPED (round 4): the reference scores every (sample, position) row and then replaces the masked scores
// (sequence.py:280-285: where(mask, score, 0 or -2^32 + 1)), so a masked row's MLP never reaches the output.  din_compact_kernel
// writes the list of rows that count — per CHUNK of `chunk_rows` consecutive rows: the chunk's valid rows, in row order, at
// pos[chunk * chunk_rows ..) and their number in cnt[chunk]; no atomics, the same list every run — and the score kernel walks 32-row
// units of that list (a chunk's last unit may be partly filled).  With behaviour sequences padded to maxlen (C4: lengths uniform on
// [1, 50]) that is half the rows.  A unit's rows are independent MFMA columns: a row's score does not depend on its place in the list.
#include <math.h>

#include "dctr_common.h"
#include "mfma_tile.h"

namespace dctr_din_chain {

using dctr::f32x4;

constexpr int NW = 16;                 // waves per workgroup: four per SIMD (<= 128 VGPRs)
constexpr int RT = 2;                  // 16-row N tiles per wave
constexpr int UROWS = 16 * RT;         // rows of a wave's unit

struct Params {
    const float* query;                // [B, E]
    const float* keys;                 // [B * T, E]
    int64_t rows;                      // B * T
    int32_t T, E, activation;
    int32_t n0, n1;                    // units
    const float* W0;                   // [4E, n0]
    const float* W1;                   // [n0, n1]
    const float* bias[2];
    const float* dice_alpha[2];
    const float* dice_mean[2];
    const float* dice_var[2];
    float dice_eps;
    const float* out_kernel;           // [n1]
    const float* out_bias;             // [1]
    float* raw;                        // [B * T]
    int64_t n_units;
    // GATHER instantiations (dctr_din_attn_gather_fwd): query / keys are NULL, the rows come straight from the embedding tables —
    // key row (b, t) = concat_h table_h[hist_ids_h[b, t]], query row b = concat_h qtable_h[query_ids_h[b]]; every feature E / NF wide
    int32_t nf, ids_i64;
    const void* hist_ids[2];
    const void* query_ids[2];
    int64_t hist_stride, query_stride;
    const float* hist_table[2];
    const float* query_table[2];
    int64_t hist_vocab[2], query_vocab[2];
    int32_t* status;
    // compacted row list (din_compact_kernel) or NULL: rows pos[c * chunk_rows + i], i < cnt[c], of chunk c < n_chunks
    const int32_t* pos;
    const int32_t* cnt;
    int32_t chunk_rows, n_chunks;
    // the workgroups' LDS image (weights folded and laid out, biases, Dice constants: fill_image) written once per launch by
    // din_prep_kernel, or NULL: every workgroup folds the weights itself (≈ 20 us of prologue)
    float* image;
};

constexpr int CMP_MAX_CHUNKS = 1024;   // (the score kernel keeps the chunks' unit prefix in LDS)
constexpr size_t IMAGE_BYTES = 160 * 1024;
constexpr int CMP_THREADS = 1024;

// what decides whether a row counts: the key mask bytes (dctr_din_attn_pool_fwd) or the history ids (dctr_din_attn_gather_fwd)
struct CompactSrc {
    const uint8_t* key_mask;
    int32_t nf, ids_i64, T;
    const void* hist_ids[2];
    int64_t hist_stride;
    int32_t mask_zero[2];
    int64_t hist_vocab[2];       // ids are range-checked here at EVERY position (the score kernel only sees the positions that count)
    int32_t* status;
};

typedef const __attribute__((address_space(1))) f32x4* gbl_f4_t;

#define DC_SB __builtin_amdgcn_sched_barrier(0)

// EB = E / 16; NB0 = 64-feature groups of layer 0 (ds_read_b128), NS0 = further 16-feature tiles of layer 0, NS1 = 16-feature
// tiles of layer 1
template <int EB, int NB0, int NS0, int NS1>
struct Lay {
    static constexpr int E = 16 * EB;
    static constexpr int K0 = 3 * E;                       // rows of the folded layer-0 weights
    static constexpr int NT0 = 4 * NB0 + NS0;              // accumulator tiles of layer 0
    static constexpr int N0P = 64 * NB0 + 16 * NS0;        // features of layer 0 incl. zero padding = K of layer 1
    static constexpr int S0A = 64 * NB0;                   // row stride of the b128 part (0 mod 64 floats: conflict-free)
    static constexpr int S0B = NS0 > 0 ? 16 * NS0 + 4 : 0; // row stride of the b32 part (4 mod 8: the four k-slots on distinct banks)
    static constexpr int N1P = 16 * NS1;
    static constexpr int S1 = N1P + 4;
    // the small arrays first: their reads carry compile-time offsets from the LDS base, and a ds_read offset field holds 64 KiB
    // (behind 63 KB of layer-0 weights every such offset became a per-lane address register, hoisted out of the unit loop: spills)
    static constexpr int PB0 = 0;                          // layer 0: bias [N0P], then the Dice constants of a feature as ONE 16-B
                                                           // group {alpha, 1 - alpha, -inv log2e, -shift log2e} [N0P][4]
    static constexpr int PB1 = PB0 + 5 * N0P;              // ... of layer 1: [N1P] + [N1P][4]
    static constexpr int OK = PB1 + 5 * N1P;               // out_kernel [N1P], out_bias
    static constexpr int W1 = OK + N1P + 4;
    static constexpr int W0B = W1 + N0P * S1;
    static constexpr int W0A = ((W0B + K0 * S0B + 63) / 64) * 64;     // (256-B aligned rows for the b128 reads)
    static constexpr int END = W0A + K0 * S0A;
};

constexpr float LOG2E = 1.4426950408889634f;

// Dice, inference form (layers/activation.py:59-64: x_p = sigmoid((x - mean) / sqrt(var + eps)), y = alpha (1 - x_p) x + x_p x) from
// the feature's 16-B constant group c = {alpha, 1 - alpha, -inv log2e, -shift log2e}:  x_p = 1 / (1 + 2^(x c.z + c.w)),
// y = x (alpha + x_p (1 - alpha)) — six instructions, two of them on the transcendental units.  (Round 4: as a per-element
// `dice ? .. : switch (activation)` the epilogues were ~20 VALU instructions and a handful of branches per element — 1,750 VALU
// instructions per 32-row unit beside its 600 MFMAs, and fp32 MFMAs share the vector lanes: 63 % matrix-pipe utilisation.)
__device__ __forceinline__ float dice_c(float x, const f32x4& c) {
    const float xp = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(x, c[2], c[3])));
    return x * fmaf(xp, c[1], c[0]);
}

// the other activations of an accumulator register pair, `act` uniform: one branch per call site, none per element
template <int RTN>
__device__ __forceinline__ void act_plain(int act, float (&x)[RTN]) {
    if (act == DCTR_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < RTN; ++i) x[i] = fmaxf(x[i], 0.f);
    } else if (act == DCTR_ACT_SIGMOID) {
#pragma unroll
        for (int i = 0; i < RTN; ++i) x[i] = dctr::sigmoidf_(x[i]);
    } else if (act == DCTR_ACT_TANH) {
#pragma unroll
        for (int i = 0; i < RTN; ++i) x[i] = dctr::tanh_fast(x[i]);
    }
}

__device__ __forceinline__ int64_t load_any_id(const void* base, int64_t idx, int i64) {
    return i64 ? reinterpret_cast<const int64_t*>(base)[idx] : (int64_t)reinterpret_cast<const int32_t*>(base)[idx];
}

// one workgroup per chunk: tiles of 1024 rows, four tiles' masks in flight; a row's slot = valid rows before it in the chunk
__device__ __forceinline__ void compact_chunk(const CompactSrc& s, int64_t rows, int32_t chunk_rows, int32_t* __restrict__ pos,
                                              int32_t* __restrict__ cnt, int (*wave_cnt)[CMP_THREADS / 64]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * chunk_rows;
    const int64_t r1 = r0 + chunk_rows < rows ? r0 + chunk_rows : rows;
    int32_t* out = pos + r0;
    int base = 0;
    for (int64_t t0 = r0; t0 < r1; t0 += 4 * CMP_THREADS) {
        bool m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t R = t0 + u * CMP_THREADS + threadIdx.x;
            m[u] = false;
            if (R < r1) {
                if (s.key_mask != nullptr) {
                    m[u] = s.key_mask[R] != 0;
                } else {
                    const uint32_t b = (uint32_t)R / (uint32_t)s.T, t = (uint32_t)R - b * (uint32_t)s.T;     // host: rows < 2^31
                    bool ok = true, bad = false;
                    for (int h = 0; h < s.nf; ++h) {
                        const int64_t id = load_any_id(s.hist_ids[h], (int64_t)b * s.hist_stride + t, s.ids_i64);
                        if (s.mask_zero[h]) ok = ok && id != 0;
                        bad = bad || (uint64_t)id >= (uint64_t)s.hist_vocab[h];
                    }
                    if (bad && s.status != nullptr) atomicOr(s.status, (int)DCTR_STATUS_INDEX_OOR);
                    m[u] = ok;
                }
            }
        }
        int before[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t bal = __ballot(m[u]);
            before[u] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[u][wave] = __popcll(bal);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int off = base, tot = 0;
#pragma unroll
            for (int w = 0; w < CMP_THREADS / 64; ++w) {
                const int cw = wave_cnt[u][w];
                off += w < wave ? cw : 0;
                tot += cw;
            }
            if (m[u]) out[off + before[u]] = (int32_t)(t0 + u * CMP_THREADS + threadIdx.x);
            base += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[blockIdx.x] = base;
}


// The LDS image of a workgroup (Lay<>): dst = the workgroup's LDS (thread tid of nthr) or the launch's image in HBM.
// Layer-0 row k of the folded matrix: k < E: Wq + Wd; k < 2E: Wk - Wd; else Wp
template <int EB, int NB0, int NS0, int NS1>
__device__ __forceinline__ void fill_image(const Params& p, float* dst, int tid, int nthr) {
    typedef Lay<EB, NB0, NS0, NS1> L;
    constexpr int E = L::E, N0P = L::N0P;
    for (int idx = tid; idx < L::K0 * N0P; idx += nthr) {
        const int k = idx / N0P, n = idx - k * N0P;
        float v = 0.f;
        if (n < p.n0) {
            int r1 = k, r2 = k;
            float sg = 0.f;
            if (k < E) { r2 = 2 * E + k; sg = 1.f; }
            else if (k < 2 * E) { r2 = E + k; sg = -1.f; }
            else { r1 = E + k; r2 = r1; }
            v = p.W0[(size_t)r1 * p.n0 + n] + sg * p.W0[(size_t)r2 * p.n0 + n];
        }
        if (n < L::S0A) dst[L::W0A + k * L::S0A + n] = v;
        else dst[L::W0B + k * L::S0B + (n - L::S0A)] = v;
    }
    for (int idx = tid; idx < N0P * L::N1P; idx += nthr) {
        const int k = idx / L::N1P, n = idx - k * L::N1P;
        dst[L::W1 + k * L::S1 + n] = (k < p.n0 && n < p.n1) ? p.W1[(size_t)k * p.n1 + n] : 0.f;
    }
    const bool dice = p.activation == DCTR_ACT_DICE;
    for (int n = tid; n < N0P; n += nthr) {
        const bool in = n < p.n0;
        float al = 0.f, inv = 0.f, sh = 0.f;
        if (in && dice) {
            al = p.dice_alpha[0][n];
            inv = 1.f / sqrtf(p.dice_var[0][n] + p.dice_eps);
            sh = -p.dice_mean[0][n] * inv;
        }
        dst[L::PB0 + n] = (in && p.bias[0] != nullptr) ? p.bias[0][n] : 0.f;
        dst[L::PB0 + N0P + 4 * n] = al;
        dst[L::PB0 + N0P + 4 * n + 1] = 1.f - al;
        dst[L::PB0 + N0P + 4 * n + 2] = -inv * LOG2E;
        dst[L::PB0 + N0P + 4 * n + 3] = -sh * LOG2E;
    }
    for (int n = tid; n < L::N1P; n += nthr) {
        const bool in = n < p.n1;
        float al = 0.f, inv = 0.f, sh = 0.f;
        if (in && dice) {
            al = p.dice_alpha[1][n];
            inv = 1.f / sqrtf(p.dice_var[1][n] + p.dice_eps);
            sh = -p.dice_mean[1][n] * inv;
        }
        dst[L::PB1 + n] = (in && p.bias[1] != nullptr) ? p.bias[1][n] : 0.f;
        dst[L::PB1 + L::N1P + 4 * n] = al;
        dst[L::PB1 + L::N1P + 4 * n + 1] = 1.f - al;
        dst[L::PB1 + L::N1P + 4 * n + 2] = -inv * LOG2E;
        dst[L::PB1 + L::N1P + 4 * n + 3] = -sh * LOG2E;
        dst[L::OK + n] = in ? p.out_kernel[n] : 0.f;
    }
    if (tid == 0) dst[L::OK + L::N1P] = p.out_bias[0];
}

// one launch in front of the score kernel: workgroups [0, n_chunks) compact their chunk of rows (when p.pos), the other
// PREP_IMAGE_BLOCKS write the LDS image (when p.image)
constexpr int PREP_IMAGE_BLOCKS = 16;
template <int EB, int NB0, int NS0, int NS1>
__global__ __launch_bounds__(CMP_THREADS) void din_prep_kernel(Params p, CompactSrc s) {
    __shared__ int wave_cnt[4][CMP_THREADS / 64];
    const int nc = p.pos != nullptr ? p.n_chunks : 0;
    if ((int)blockIdx.x < nc) {
        compact_chunk(s, p.rows, p.chunk_rows, const_cast<int32_t*>(p.pos), const_cast<int32_t*>(p.cnt), wave_cnt);
    } else if (p.image != nullptr) {
        fill_image<EB, NB0, NS0, NS1>(p, p.image, ((int)blockIdx.x - nc) * CMP_THREADS + (int)threadIdx.x, PREP_IMAGE_BLOCKS * CMP_THREADS);
    }
}

template <int EB, int NB0, int NS0, int NS1, bool GATHER = false>
__global__ __launch_bounds__(64 * NW) void din_chain_kernel(Params p) {
    typedef Lay<EB, NB0, NS0, NS1> L;
    constexpr int E = L::E, NT0 = L::NT0, N0P = L::N0P;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTHR = 64 * NW;
    // ---- once per workgroup: the LDS image — copied from the launch's image (din_prep_kernel), else folded here
    if (p.image != nullptr) {
        constexpr int N4 = L::END / 4, IT = (N4 + NTHR - 1) / NTHR;
        static_assert(L::END % 4 == 0, "the image is copied in 16-B pieces");
        const f32x4* src = reinterpret_cast<const f32x4*>(p.image);
        f32x4 v[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) v[i] = src[min(i * NTHR + (int)threadIdx.x, N4 - 1)];
#pragma unroll
        for (int i = 0; i < IT; ++i)
            if (i * NTHR + (int)threadIdx.x < N4) reinterpret_cast<f32x4*>(smem)[i * NTHR + threadIdx.x] = v[i];
    } else {
        fill_image<EB, NB0, NS0, NS1>(p, smem, (int)threadIdx.x, NTHR);
    }
    const bool dice = p.activation == DCTR_ACT_DICE;
    // compacted rows: upre[c] = 32-row units of the chunks before c (upre[n_chunks] = all units); one wave, <= 16 chunks per lane
    int* const upre = reinterpret_cast<int*>(smem + L::END);
    if (p.pos != nullptr && threadIdx.x < 64) {
        const int ln = threadIdx.x, per = (p.n_chunks + 63) >> 6;
        int local = 0;
        for (int i = 0; i < per; ++i) {
            const int c = ln * per + i;
            local += c < p.n_chunks ? (p.cnt[c] + UROWS - 1) / UROWS : 0;
        }
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            incl += ln >= d ? v : 0;
        }
        int run = incl - local;
        for (int i = 0; i < per; ++i) {
            const int c = ln * per + i;
            if (c < p.n_chunks) {
                upre[c] = run;
                run += (p.cnt[c] + UROWS - 1) / UROWS;
            }
        }
        if (ln == 63) upre[p.n_chunks] = incl;
    }
    __syncthreads();                                       // the only barrier

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    // lane bases of the A-operand reads: k-step t of a 16-row k-block reads weight row 4g + t (k-slot g)
    const float* a0A = smem + L::W0A + (4 * g) * L::S0A + 4 * j;       // + row * S0A
    const float* a0B = smem + L::W0B + (4 * g) * L::S0B + j;           // + row * S0B + 16 * s
    // layer 1: k-slot g of k-step (tile, r): feature 16g + 4r + mt for a b128 tile mt of group 0, 64 NB0 + 16 s + 4g + r else
    const float* a1G = smem + L::W1 + (16 * g) * L::S1 + j;            // + (4r + mt) * S1 + 16 * s1
    const float* a1S = smem + L::W1 + (64 * NB0 + 4 * g) * L::S1 + j;  // + (16 s + r) * S1 + 16 * s1

    // units go round the workgroups first (unit u: workgroup u mod grid, wave u / grid): a launch with fewer units than wave slots
    // still spreads over every CU and every SIMD
    const int64_t n_units = p.pos != nullptr ? (int64_t)upre[p.n_chunks] : p.n_units;
    int cc = 0;                                            // compacted: the chunk of the current unit (units of a wave ascend)
    for (int64_t unit = (int64_t)wave * gridDim.x + blockIdx.x; unit < n_units; unit += (int64_t)gridDim.x * NW) {
        // rows [R0, Rlim) of the row list (compacted: of the chunk's part of pos) are this unit's
        int64_t R0 = unit * UROWS, Rlim = p.rows;
        if (p.pos != nullptr) {
            while (upre[cc + 1] <= (int)unit) ++cc;
            R0 = (int64_t)cc * p.chunk_rows + (int64_t)((int)unit - upre[cc]) * UROWS;
            Rlim = (int64_t)cc * p.chunk_rows + p.cnt[cc];
        }
        // rows of this lane's two N tiles; query rows
        const float* kp[RT];
        const float* qp[RT];
        // GATHER: the table rows of this lane's (sample, position) rows per feature (as float offsets from the table base; an id
        // outside the vocabulary reads row 0 and raises the status flag)
        uint32_t kro[RT][2], qro[RT][2];                    // (table ROWS: vocabularies < 2^31, host; the float offset is formed per load)
        const int EH = GATHER ? E / max(p.nf, 1) : E;       // embedding_dim of one history feature
        const int EBH = EH / 16;                            // its 16-wide blocks
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            const int64_t Rl = min(R0 + 16 * nt + j, Rlim - 1);
            const int64_t R = p.pos != nullptr ? (int64_t)p.pos[Rl] : Rl;
            const int64_t b = (int64_t)((uint32_t)R / (uint32_t)p.T);      // host: rows < 2^31
            if constexpr (GATHER) {
                const int64_t t = R - b * p.T;
                bool bad = false;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h < p.nf) {
                        const int64_t ik = load_any_id(p.hist_ids[h], b * p.hist_stride + t, p.ids_i64);
                        const int64_t iq = load_any_id(p.query_ids[h], b * p.query_stride, p.ids_i64);
                        const bool okk = (uint64_t)ik < (uint64_t)p.hist_vocab[h], okq = (uint64_t)iq < (uint64_t)p.query_vocab[h];
                        bad = bad || !okk || !okq;
                        kro[nt][h] = okk ? (uint32_t)ik : 0u;
                        qro[nt][h] = okq ? (uint32_t)iq : 0u;
                    } else {
                        kro[nt][h] = qro[nt][h] = 0u;
                    }
                }
                if (bad && p.status != nullptr && R0 + 16 * nt + j < Rlim) atomicOr(p.status, (int)DCTR_STATUS_INDEX_OOR);
                kp[nt] = qp[nt] = nullptr;
            } else {
                kp[nt] = p.keys + R * E + 4 * g;
                qp[nt] = p.query + b * E + 4 * g;
            }
        }
        // the 16-B piece of block c of this lane's key / query row
        auto load_k = [&](int nt, int c) -> f32x4 {
            if constexpr (GATHER) {
                const int h = c >= EBH ? 1 : 0;
                return *(gbl_f4_t)(p.hist_table[h] + ((int64_t)kro[nt][h] * EH + 4 * g + 16 * (c - h * EBH)));
            } else {
                return *(gbl_f4_t)(kp[nt] + 16 * c);
            }
        };
        auto load_q = [&](int nt, int c) -> f32x4 {
            if constexpr (GATHER) {
                const int h = c >= EBH ? 1 : 0;
                return *(gbl_f4_t)(p.query_table[h] + ((int64_t)qro[nt][h] * EH + 4 * g + 16 * (c - h * EBH)));
            } else {
                return *(gbl_f4_t)(qp[nt] + 16 * c);
            }
        };
        // ---- layer 0.  Accumulators start at the bias: tile tt < 4 NB0 (b128 tile mt of group 0): feature 16g + 4r + mt;
        // tile 4 NB0 + s: feature 64 NB0 + 16 s + 4g + r
        f32x4 acc0[NT0][RT];
#pragma unroll
        for (int tt = 0; tt < NT0; ++tt) {
            f32x4 bv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = tt < 4 * NB0 ? 16 * g + 4 * r + tt : 64 * NB0 + 16 * (tt - 4 * NB0) + 4 * g + r;
                bv[r] = smem[L::PB0 + f];
            }
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) acc0[tt][nt] = bv;
        }
        f32x4 qc[RT], kc[RT], qn[RT], kn[RT];
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            qc[nt] = load_q(nt, 0);
            kc[nt] = load_k(nt, 0);
            qn[nt] = qc[nt];
            kn[nt] = kc[nt];
        }
        f32x4 fa, fan;                                     // A fragments of the b128 group: current / next k-step
        float fb[NS0 > 0 ? NS0 : 1], fbn[NS0 > 0 ? NS0 : 1];
        // (block pointers advance with the rolled block loop: every row offset inside a block is an immediate)
        const float* pA = a0A;
        const float* pB = a0B;
        auto read_a0 = [&](int row, f32x4& a, float (&b)[NS0 > 0 ? NS0 : 1]) {
            if constexpr (NB0 > 0) a = *reinterpret_cast<const f32x4*>(pA + row * L::S0A);
#pragma unroll
            for (int s = 0; s < NS0; ++s) b[s] = pB[row * L::S0B + 16 * s];
        };
        read_a0(0, fa, fb);
        // (the block loop stays rolled: unrolled over E = 64 hipcc keeps the raw rows of several blocks in flight and spills)
#pragma unroll 1
        for (int c = 0; c < EB; ++c) {
            const int c1 = min(c + 1, EB - 1);             // raw rows of the next block: in flight during this block's MFMAs
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) {
                qn[nt] = load_q(nt, c1);
                kn[nt] = load_k(nt, c1);
            }
#pragma unroll
            for (int part = 0; part < 3; ++part) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // next k-step's fragments first (its row: same block / next part / next block; past the last block: a
                    // redundant read of row 0, dropped)
                    const int nrel = part * 4 + t + 1;     // within this block: 1 .. 12
                    const int nrow = nrel < 12 ? (nrel / 4) * E + (nrel % 4) : 16;      // relative to the block's first row
                    read_a0(nrow, fan, fbn);
                    DC_SB;
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) {
                        const float bq = qc[nt][t], bk = kc[nt][t];
                        const float b = part == 0 ? bq : part == 1 ? bk : bq * bk;
                        if constexpr (NB0 > 0) {
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                                acc0[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mt], b, acc0[mt][nt], 0, 0, 0);
                        }
#pragma unroll
                        for (int s = 0; s < NS0; ++s)
                            acc0[4 * NB0 + s][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s], b, acc0[4 * NB0 + s][nt], 0, 0, 0);
                    }
                    DC_SB;
                    fa = fan;
#pragma unroll
                    for (int s = 0; s < NS0; ++s) fb[s] = fbn[s];
                }
            }
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) {
                qc[nt] = qn[nt];
                kc[nt] = kn[nt];
            }
            pA += 16 * L::S0A;
            pB += 16 * L::S0B;
        }
        // ---- activation of layer 0 in place (padded features: weights and bias 0 -> the activation of 0, taken out again by
        // the zero rows of W1).  ONE uniform branch for the whole accumulator set
        // (the constants' lane bases come from an OPAQUE copy of g: with the known-zero low bits of g << k hipcc turns base + constant
        //  into base | constant, materialises one address register per constant, hoists all of them out of the unit loop and spills)
        int go = g;
        asm volatile("" : "+v"(go));
        const float* const cA = smem + L::PB0 + N0P + 64 * go;                  // b128 tiles: feature 16 g + (4 r + tt)
        const float* const cB = smem + L::PB0 + N0P + 256 * NB0 + 16 * go;      // b32 tiles: feature 64 NB0 + 16 s + 4 g + r
        if (dice) {
#pragma unroll
            for (int tt = 0; tt < NT0; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(tt < 4 * NB0 ? cA + 4 * (4 * r + tt) : cB + 64 * (tt - 4 * NB0) + 4 * r);
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) acc0[tt][nt][r] = dice_c(acc0[tt][nt][r], c);
                    if (r == 3) DC_SB;                     // (a tile at a time: hipcc otherwise keeps every tile's constants in flight)
                }
        } else if (p.activation != DCTR_ACT_LINEAR) {
#pragma unroll
            for (int tt = 0; tt < NT0; ++tt)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt) {
                    float x[4] = {acc0[tt][nt][0], acc0[tt][nt][1], acc0[tt][nt][2], acc0[tt][nt][3]};
                    act_plain<4>(p.activation, x);
                    acc0[tt][nt] = f32x4{x[0], x[1], x[2], x[3]};
                    DC_SB;
                }
        }
        // ---- layer 1: k-steps (tile tt, r) of layer 0's accumulators; A = W1 rows of the features those registers hold
        f32x4 acc1[NS1][RT];
#pragma unroll
        for (int s1 = 0; s1 < NS1; ++s1) {
            f32x4 bv;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = smem[L::PB1 + 16 * s1 + 4 * g + r];
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) acc1[s1][nt] = bv;
        }
        float w1c[NS1], w1n[NS1];
        auto read_a1 = [&](int ks, float (&w)[NS1]) {      // k-step ks = 4 tt + r
            const int tt = ks >> 2, r = ks & 3;
            const float* src = tt < 4 * NB0 ? a1G + (4 * r + tt) * L::S1 : a1S + (16 * (tt - 4 * NB0) + r) * L::S1;
#pragma unroll
            for (int s1 = 0; s1 < NS1; ++s1) w[s1] = src[16 * s1];
        };
        read_a1(0, w1c);
#pragma unroll
        for (int ks = 0; ks < 4 * NT0; ++ks) {
            if (ks + 1 < 4 * NT0) read_a1(ks + 1, w1n);
            DC_SB;
#pragma unroll
            for (int nt = 0; nt < RT; ++nt)
#pragma unroll
                for (int s1 = 0; s1 < NS1; ++s1)
                    acc1[s1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1c[s1], acc0[ks >> 2][nt][ks & 3], acc1[s1][nt], 0, 0, 0);
            DC_SB;
#pragma unroll
            for (int s1 = 0; s1 < NS1; ++s1) w1c[s1] = w1n[s1];
        }
        // ---- activation of layer 1, Dense(1): this lane's features 16 s1 + 4g + r of rows 16 nt + j, then over g
        float part[RT];
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) part[nt] = 0.f;
        int go1 = g;
        asm volatile("" : "+v"(go1));
        const float* const c1 = smem + L::PB1 + L::N1P + 16 * go1;             // feature 16 s1 + 4 g + r
        const float* const ok1 = smem + L::OK + 4 * go1;
        if (dice) {
#pragma unroll
            for (int s1 = 0; s1 < NS1; ++s1)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(c1 + 64 * s1 + 4 * r);
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) acc1[s1][nt][r] = dice_c(acc1[s1][nt][r], c);
                    if (r == 3) DC_SB;
                }
        } else if (p.activation != DCTR_ACT_LINEAR) {
#pragma unroll
            for (int s1 = 0; s1 < NS1; ++s1)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt) {
                    float x[4] = {acc1[s1][nt][0], acc1[s1][nt][1], acc1[s1][nt][2], acc1[s1][nt][3]};
                    act_plain<4>(p.activation, x);
                    acc1[s1][nt] = f32x4{x[0], x[1], x[2], x[3]};
                    DC_SB;
                }
        }
        // (padded features f >= n1: out_kernel is 0 there and the activation of 0 is finite)
#pragma unroll
        for (int s1 = 0; s1 < NS1; ++s1) {
            const f32x4 okv = *reinterpret_cast<const f32x4*>(ok1 + 16 * s1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt) part[nt] = fmaf(acc1[s1][nt][r], okv[r], part[nt]);
        }
        const float ob = smem[L::OK + L::N1P];
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            float v = part[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int64_t Rl = R0 + 16 * nt + j;
            if (g == 0 && Rl < Rlim) p.raw[p.pos != nullptr ? (int64_t)p.pos[Rl] : Rl] = v + ob;
        }
    }
}

template <int EB, int NB0, int NS0, int NS1, bool GATHER>
static int launch_one_g(const Params& p, const CompactSrc& cs, hipStream_t stream) {
    typedef Lay<EB, NB0, NS0, NS1> L;
    const size_t lds = (size_t)L::END * sizeof(float) + (p.pos != nullptr ? (size_t)(p.n_chunks + 1) * sizeof(int) : 0);
    if (lds > 160 * 1024) return 0;
    static_assert((size_t)L::END * sizeof(float) <= IMAGE_BYTES, "the image area of the workspace holds a workgroup's LDS image");
    static thread_local size_t granted[DCTR_MAX_DEVICES] = {0};
    if (dctr_grant_lds((const void*)din_chain_kernel<EB, NB0, NS0, NS1, GATHER>, lds, granted) != hipSuccess) return 0;
    if (p.pos != nullptr || p.image != nullptr) {
        const unsigned blocks = (unsigned)((p.pos != nullptr ? p.n_chunks : 0) + (p.image != nullptr ? PREP_IMAGE_BLOCKS : 0));
        hipLaunchKernelGGL((din_prep_kernel<EB, NB0, NS0, NS1>), dim3(blocks), dim3(CMP_THREADS), 0, stream, p, cs);
    }
    // units go round the workgroups (unit u: workgroup u mod grid): one workgroup per CU as soon as there are that many units —
    // 1,632 units of a compacted C4 call are 6.4 per CU, two per SIMD, where 16 per workgroup would fill 102 CUs
    const int64_t cus = dctr_n_cus();
    const int64_t units_max = p.n_units + p.n_chunks;                       // (compacted: an upper bound, a partly filled unit per chunk)
    int64_t grid = units_max < cus ? units_max : cus;
    DCTR_LAUNCH((din_chain_kernel<EB, NB0, NS0, NS1, GATHER>), dim3((unsigned)grid), dim3(64 * NW), lds, stream, p);
    return 1;
}

template <int EB, int NB0, int NS0, int NS1>
static int launch_one(const Params& p, const CompactSrc& cs, hipStream_t stream) {
    return p.nf > 0 ? launch_one_g<EB, NB0, NS0, NS1, true>(p, cs, stream) : launch_one_g<EB, NB0, NS0, NS1, false>(p, cs, stream);
}

template <int EB>
static int launch_e(const Params& p, const CompactSrc& cs, int nb0, int ns0, int ns1, hipStream_t stream) {
#define DC_CASE(A, B, C) if (nb0 == A && ns0 == B && ns1 == C) return launch_one<EB, A, B, C>(p, cs, stream)
    DC_CASE(1, 1, 3);      // 80-40 (DIN's att_hidden_size default)
    DC_CASE(1, 0, 2);      // 64-32 (LocalActivationUnit's default)
    DC_CASE(1, 0, 1);      // 64-16
    DC_CASE(0, 2, 1);      // 32-16
    DC_CASE(1, 0, 4);      // 64-64
#undef DC_CASE
    return 0;
}

// rows of a compaction chunk: 4096, doubled until the launch has <= CMP_MAX_CHUNKS of them
static int32_t compact_chunk_rows(int64_t rows) {
    int64_t c = 4 * CMP_THREADS;
    while (dctr_ceil_div(rows, c) > CMP_MAX_CHUNKS) c *= 2;
    return (int32_t)c;
}

// bytes of the row list + the chunk counts behind the [rows] raw scores in the attention workspace
size_t compact_bytes(int64_t rows) {
    if (rows <= 0 || rows >= 0x7fffffffLL) return 0;
    const int64_t chunk = compact_chunk_rows(rows);
    return (size_t)(dctr_ceil_div(rows, chunk) * chunk + CMP_MAX_CHUNKS) * sizeof(int32_t);
}

size_t image_bytes() { return IMAGE_BYTES; }

// 1: the chained kernel was launched (raw scores of the rows that count -> raw; all rows without `compact_ws`); 0: shape not covered,
// nothing launched.  compact_ws: compact_bytes(rows) bytes, 4-B aligned (NULL: every row is scored); key_mask: what counts on the
// lookup route (the gather route reads the history ids; without a mask_zero feature every row counts and nothing is compacted);
// image_ws: image_bytes() bytes, 16-B aligned, for the workgroups' LDS image (NULL: every workgroup folds the weights itself)
int try_launch(const float* query, const float* keys, int64_t batch, int T, int E, int n_layers, const int32_t* units,
               const float* const* kernels, const float* const* biases, int activation, const float* const* dice_alpha,
               const float* const* dice_mean, const float* const* dice_var, float dice_eps, const float* out_kernel,
               const float* out_bias, float* raw, hipStream_t stream, const dctr_din_gather_t* gd, const uint8_t* key_mask,
               void* compact_ws, void* image_ws) {
    if (n_layers != 2 || (E != 16 && E != 32 && E != 64)) return 0;
    if (gd != nullptr && (gd->n_feats < 1 || gd->n_feats > 2 || E % (16 * gd->n_feats) != 0)) return 0;
    const int n0 = units[0], n1 = units[1];
    if (n0 < 1 || n1 < 1 || n1 > 64) return 0;
    const int nb0 = n0 >= 64 ? 1 : 0;
    const int ns0 = (n0 - 64 * nb0 + 15) / 16;
    const int ns1 = (n1 + 15) / 16;
    if (n0 > 64 + 48) return 0;
    Params p{};
    p.query = query;
    p.keys = keys;
    p.rows = batch * (int64_t)T;
    p.T = T;
    p.E = E;
    p.activation = activation;
    p.n0 = n0;
    p.n1 = n1;
    p.W0 = kernels[0];
    p.W1 = kernels[1];
    for (int l = 0; l < 2; ++l) {
        p.bias[l] = biases[l];
        p.dice_alpha[l] = dice_alpha != nullptr ? dice_alpha[l] : nullptr;
        p.dice_mean[l] = dice_mean != nullptr ? dice_mean[l] : nullptr;
        p.dice_var[l] = dice_var != nullptr ? dice_var[l] : nullptr;
    }
    p.dice_eps = dice_eps;
    p.out_kernel = out_kernel;
    p.out_bias = out_bias;
    p.raw = raw;
    if (gd != nullptr) {
        p.nf = gd->n_feats;
        p.ids_i64 = gd->ids_is_i64;
        p.hist_stride = gd->hist_stride;
        p.query_stride = gd->query_stride;
        p.status = gd->status;
        for (int h = 0; h < gd->n_feats; ++h) {
            p.hist_ids[h] = gd->hist_ids[h];
            p.query_ids[h] = gd->query_ids[h];
            p.hist_table[h] = gd->hist_table[h];
            p.query_table[h] = gd->query_table[h];
            p.hist_vocab[h] = gd->hist_vocab[h];
            p.query_vocab[h] = gd->query_vocab[h];
        }
    }
    p.n_units = dctr_ceil_div(p.rows, (int64_t)UROWS);
    const bool masked = gd != nullptr ? (gd->mask_zero[0] || (gd->n_feats > 1 && gd->mask_zero[1])) : key_mask != nullptr;
    CompactSrc cs{};
    p.image = static_cast<float*>(image_ws);
    if (compact_ws != nullptr && masked && p.rows < 0x7fffffffLL) {
        {
            cs.key_mask = gd != nullptr ? nullptr : key_mask;
            cs.T = T;
            if (gd != nullptr) {
                cs.nf = gd->n_feats;
                cs.ids_i64 = gd->ids_is_i64;
                cs.hist_stride = gd->hist_stride;
                cs.status = gd->status;
                for (int h = 0; h < gd->n_feats; ++h) {
                    cs.hist_ids[h] = gd->hist_ids[h];
                    cs.mask_zero[h] = gd->mask_zero[h];
                    cs.hist_vocab[h] = gd->hist_vocab[h];
                }
            }
            p.chunk_rows = compact_chunk_rows(p.rows);
            p.n_chunks = (int32_t)dctr_ceil_div(p.rows, (int64_t)p.chunk_rows);
            int32_t* pos = static_cast<int32_t*>(compact_ws);
            int32_t* cnt = pos + (int64_t)p.n_chunks * p.chunk_rows;
            p.pos = pos;
            p.cnt = cnt;
        }
    }
    if (E == 16) return launch_e<1>(p, cs, nb0, ns0, ns1, stream);
    if (E == 32) return launch_e<2>(p, cs, nb0, ns0, ns1, stream);
    return launch_e<4>(p, cs, nb0, ns0, ns1, stream);
}

}  // namespace dctr_din_chain
