// dctr_embed_mlp_fwd, row-chained form — the throughput kernel of the DeepFM-family forward (reference
// deepctr/inputs.py:101-117 embedding_lookup, feature_column.py:171-210 linear logit, layers/interaction.py:588-604 FM,
// layers/core.py:189-208 DNN, :250-259 PredictionLayer) for launches that give every CU >= 256 rows.
//
// Why a third kernel.  stream_kernel (stream_kernels.hip) splits the layer OUTPUT columns over its 8 MFMA waves: every
// layer ends in an epilogue that stores the activations to LDS, a barrier of the 8 waves, and a cold start of the next
// layer's operand pipeline; its MFMA waves alone reach 0.72 of the f32-MFMA rate (profiles/r02_stream_lab_ablation.log),
// 0.55-0.62 with the loader waves beside them.  Here a wave owns BATCH ROWS end to end and the MLP is computed transposed:
//   * out^T[n, b] = sum_k W[k, n] * x[b, k]: the WEIGHTS are the MFMA A operand (M = output features), the wave's 32 batch
//     rows are the N dimension (two 16-wide N tiles).  v_mfma_f32_16x16x4_f32 leaves C[m = 4g + r][n = j] in lane (g, j),
//     register r — and wants B[k = g][n = j] from lane (g, j): a layer's accumulators ARE the next layer's B operand,
//     register for register (k-slot g of k-step (M-tile, r) is output feature 4g + r of that M-tile; a K permutation, which
//     an fp32 fmaf chain does not care about as long as the weight rows are fetched in the same order).  Activations never
//     leave the register file: no LDS round trip, no epilogue stores, no barrier between layers;
//   * the 8 waves of a workgroup walk the SAME weight sequence (layer 0 in 16-row k-blocks, layers >= 1 in 64 x 64
//     sub-blocks), so the weights go L2 -> LDS once per 256 rows (603 KB per 300k cycles = 2 B/clk/CU against the 16 B/clk
//     the 32-row kernel pulls) through LDS-DMA (global_load_lds_dwordx4, every wave moves 1/8 of a 16-KiB chunk) into a
//     ring of three chunks; one s_barrier per chunk both publishes the chunk after next and retires the previous one;
//   * A fragments are ds_read_b128: with M-tile mt of an M-group holding output features 64*mg + 4*i + mt (i = MFMA row),
//     lane (g, j) reads floats 64*mg + 4j .. + 3 of weight row k: one read feeds FOUR M-tiles, 16 lanes of a group cover
//     256 contiguous bytes (conflict-free for every lane grouping of ds_read_b128), and the rows need no repacking: the
//     DMA image of a chunk is the Keras [K, N] rows as they lie in memory;
//   * the embedding gather goes HBM/MALL -> REGISTERS: lane (g, j) of a wave loads the 16-B piece g of row j's embedding of
//     one field with one global_load_dwordx4 — which is exactly the B operand of that field's four k-steps.  FM sums and
//     the linear terms are lane-local adds (reduced over g once per pass).  Rows are requested one k-block (8k cycles)
//     ahead of their MFMAs, ids two.
// 8 waves x 32 rows = 256 rows per pass and CU; registers: 128 (layer-0 accumulators: 16 M-tiles x 2 N-tiles) + 64
// (layer 1) + operand staging -> two waves per SIMD.
//
// One launch = a MAIN phase (256-row passes, 8 waves x 32 rows, persistent workgroups) + a TAIL phase in the same kernel: what
// is left when the rows do not fill every CU with 256 goes out in 64-row units (waves 0-3 x 16 rows; waves 4-7 leave — the
// hardware barrier counts the surviving waves only), each workgroup taking its units round-robin.  No second launch, no launch
// gap, the parameters are already in LDS.  Both phases walk k in the same order: a row's bits do not depend on its phase.
//
// Eligibility (host, chain_kernels.hip): uniform embedding_dim 16 or 32 (fixed-length SparseFeat, or fields pre-pooled by
// dctr_embed_pool = identity fields, whose row is the sample index), no in-kernel hashing, dense columns right behind the
// embeddings, two or three layers with units[0] in {128, 256}, units[1] in {64, 128}, units[2] in {64, 128}, ReLU or linear
// activation, a head, optional BatchNormalization scale / shift (inference form), no saved activations, and at least 64 rows
// per CU (or tile_rows 128 / 256).  Everything else takes stream_kernel / mlp_kernel.
// Same arithmetic as those: v_mfma_f32_16x16x4_f32 = exact fp32; only the summation order over k differs.
#pragma once
#include "mlp_device.h"

// two schedule constants of a layer-0 step (A/B history: DESIGN.md §4)
#ifndef CHAIN_SPREAD
#define CHAIN_SPREAD 1                         // 1: the request phase of a layer-0 step is spread over its micro-steps (0: one block)
#endif
#ifndef CHAIN_DMA_LATE
#define CHAIN_DMA_LATE 8                       // (even) micro-step behind which waves 4-7 issue their DMA share (waves 0-3: 0)
#endif

namespace dctr_chain {

using dctr::f32x4;
using namespace dctr_mlp;

// A launch shape = (RT 16-row N tiles per wave, NW waves per workgroup): 16 * RT * NW batch rows per pass.  <2, 8> is the
// throughput shape (256 rows, two waves per SIMD); <2, 4> (128 rows) and <1, 4> (64 rows, one wave per SIMD) take what is
// left of a launch when the rows do not fill every CU with 256 — every shape walks k in the same order, so a row's result
// does not depend on the shape (or the position) it was computed in.
constexpr int NSLOT = 3;
constexpr int SLOT_F = 4096;                   // floats per ring chunk (16 KiB)
constexpr int CPAR_OFF = 0;                    // biases of every layer, head weights, BatchNormalization scale / shift (<= 2048 floats)
constexpr int POOLD_OFF = 2016;                // [n_pool <= 4][8 dwords]: the POOL kernels' sequence descriptors (the tail of the parameter area)
constexpr int FDESC_OFF = 2048;                // [n_fields <= 64][12 dwords]
constexpr int DLW_OFF = 2816;                  // dense_lin_w (<= 256 floats, zeros when absent)
constexpr int RING_OFF = 3072;
constexpr int DENSE_OFF = RING_OFF + NSLOT * SLOT_F;   // [rows of a pass][16 * dense k-blocks] zero-padded dense values of the pass
constexpr int MAX_DENSE_BLOCKS = 4;
// + [NW waves][RT N tiles][64 lanes] shares of dense . dense_lin_w of the staged pass
// + [NW waves][4 RT quads][64 lanes] parked accumulators (layer-0 M-group M0 - 1 while layer 1 works on the others)
static inline size_t lds_bytes(int rt, int nw, int n_dense) {
    const int park = nw * 4 * rt * 256;
    return (size_t)(DENSE_OFF + nw * 16 * rt * ((n_dense + 15) & ~15) + nw * rt * 64 + park) * sizeof(float);
}

// CROSS kernels: [CROSS_NV][16 NB] zero-padded cross vectors + the CROSS_NV constants, behind everything else in LDS
static inline int cross_lds_floats(int in_dim) { return dctr_mlp::CROSS_NV * 16 * ((in_dim + 15) >> 4) + 8; }

struct ChainParams {
    const dctr_field_t* fields;
    const void* ids;
    int64_t ids_stride_f, ids_stride_b;
    int32_t ids_is_i64, n_fields, n_dense, in_dim;
    const float* dense;
    int64_t dense_stride;
    const float* dense_lin_w;
    int64_t batch;
    float* fm_logit;
    float* lin_logit;
    int32_t* status;
    int32_t fm_used, lin_used;
    const float* W[3];
    const float* bias[3];
    int32_t activation, sigmoid_out;
    const float* head_w;
    const float* add[4];
    const float* global_bias;
    float* y;
    unsigned long long* probe;
    int32_t n_pass;                // main phase: passes of 16 * RT * NW rows over rows [0, main_rows)
    int32_t n_tail;                // tail phase (kernels instantiated with TAIL): 64-row units over rows [main_rows, batch)
    int64_t main_rows;
    const float* bn_scale[3];      // DNN(use_bn=True), inference form: act((x W + b) * bn_scale + bn_shift); NULL = none
    const float* bn_shift[3];
    // CrossNet, vector parameterization, folded into the pass (kernels instantiated with CROSS; mlp_device.h: cross_logit): the
    // L + 1 vectors w_0 .. w_{L-1}, k_c lie zero-padded in LDS at xv_off ([CROSS_NV][16 NB] floats + the CROSS_NV constants)
    // the same 32 bytes serve the CROSS kernels' vectors or the POOL kernels' sequences (a kernel is instantiated with at most one of the
    // two; the struct — the kernel argument — keeps its size and layout, so the other instantiations' code does not depend on either)
    union {
        struct {
            const float* cross_w;
            const float* cross_b;
            const float* cross_head;
            const float* cross_const;  // [CROSS_NV] precomputed (dctr_crossnet_fold_consts) or NULL: wave 0 computes them
        };
        // VarLenSparseFeat pooled INSIDE the pass (kernels instantiated with POOL): the LAST n_pool fields of `fields` are sequences; their
        // descriptors carry table / lin_table / vocab / in_fm as for a SparseFeat, pool[i] (DEVICE array, dctr_pool_seq_t) the id
        // matrix of the i-th of them from the staged data's row 0; the launch's rows are rows pool_row0 ... of those matrices
        struct {
            const dctr_pool_seq_t* pool;
            int64_t pool_row0;
            int32_t n_pool;
            int32_t pool_flags;        // bit i: sequence i is mean-pooled; bit 4 + i: it has a length vector (dctr_gather_fm_args_t.pool_flags)
            int32_t pool_pad_[2];
        };
    };
    int32_t cross_layers;
    int32_t xv_off;
};
static_assert(sizeof(dctr_pool_seq_t) == 32, "dctr_pool_seq_t: 32 bytes (two scalar loads)");

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// loads through pointers that come out of LDS (field descriptors) would be FLAT instructions (vmcnt AND lgkmcnt, slower
// address path): the global address space is stated explicitly
typedef const __attribute__((address_space(1))) f32x4* gbl_f4_t;
typedef const __attribute__((address_space(1))) float* gbl_f_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x2* gbl_u2_t;
typedef const __attribute__((address_space(1))) uint32_t* gbl_u_t;
typedef const __attribute__((address_space(1))) int32_t* gbl_i_t;

// field descriptor words as they lie in LDS (copied once per launch); decoded to scalars where they are used
struct FieldRaw {
    uint4 a;       // table, lin_table
    uint2 b;       // vocab
    uint32_t c;    // in_fm
};
__device__ __forceinline__ FieldRaw field_raw(const float* fdesc, int f) {
    FieldRaw r;
    r.a = *reinterpret_cast<const uint4*>(fdesc + 12 * f);
    r.b = *reinterpret_cast<const uint2*>(fdesc + 12 * f + 4);
    r.c = *reinterpret_cast<const uint32_t*>(fdesc + 12 * f + 8);
    return r;
}
__device__ __forceinline__ uint64_t sgpr64(uint32_t lo, uint32_t hi) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
}

// The MFMA goes out as inline asm with the accumulator as a read-write operand: D = C in place, always.  Left to hipcc, the
// 48 accumulator quads of a pass are renamed from MFMA to MFMA (D != C), the rotation needs spare quads, and with 192 of 256
// registers holding accumulators the allocator tips into spilling accumulators inside the k-loop (scratch reloads whose
// vmcnt waits then sit behind the DMA of the step).  What hipcc's hazard recognizer no longer sees is covered by
// construction: dependent MFMAs on one accumulator are >= 8 MFMAs apart; operands come from LDS / global loads (waitcnt
// by hipcc, the operands are visible) or from accumulators finished a layer earlier; non-MFMA reads of accumulators come
// after mfma_drain().
__device__ __forceinline__ void mfma_ip(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// (layers >= 1 keep the builtin: their B operand is an ELEMENT of a previous layer's accumulator quad, and an inline-asm
// operand cannot be a sub-register — every such operand would be copied out first)
__device__ __forceinline__ void mfma_bi(f32x4& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// activation of a whole accumulator set in place.  ReLU / linear in the plain instantiations; sigmoid / tanh DNNs (reference
// layers/activation.py:75-85 -> tf.keras.layers.Activation) take the EXPACT instantiations (chain_kernels_r2w8_m42_t.hip) — their own
// kernels: libm code unrolled over 128 accumulator registers made hipcc spill around the layer boundaries on the ReLU path as well.
// Their forms run on the transcendental units (v_exp_f32, v_rcp_f32: a few fp32 ulp): sigmoid = 1 / (1 + e^{-x}); tanh = its odd
// series through x^9 for |x| < 1/4 (next term < 1e-8 of the result), else 1 - 2 / (1 + e^{2|x|}) with the sign put back
template <int NM, int RT, bool EXPACT = false>
__device__ __forceinline__ void act_block(int act, f32x4 (&acc)[NM][RT]) {
    if constexpr (EXPACT) {
        // one accumulator quad at a time (sched_barrier: hipcc otherwise starts every element's v_exp early and keeps hundreds of
        // temporaries alive over the 128 registers — scratch)
        const bool sig = act == DCTR_ACT_SIGMOID;
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) {
                f32x4 v = acc[m][nt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // sigmoid(x) = 1 / (1 + e^{-x});  tanh(x) = 2 sigmoid(2x) - 1 away from 0, its odd series near 0
                    const float x = v[r];
                    const float ax = fabsf(x), x2 = x * x;
                    const float e = __expf(sig ? -x : 2.f * ax);
                    const float q = __builtin_amdgcn_rcpf(1.f + e);
                    const float small = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
                    const float th = ax < 0.25f ? small : copysignf(fmaf(-2.f, q, 1.f), x);
                    v[r] = sig ? q : th;
                }
                acc[m][nt] = v;
                __builtin_amdgcn_sched_barrier(0);
            }
        return;
    }
    const float floor_ = act == DCTR_ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int nt = 0; nt < RT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][nt][r] = fmaxf(acc[m][nt][r], floor_);
}

// BatchNormalization (inference form) of a whole accumulator set in place, between bias_add and the activation (reference
// layers/core.py:200-201): acc = acc * scale + shift per output feature; register r of M-tile mt of M-group mg holds feature
// 64 mg + 16 g + 4 r + mt (the layout the biases are loaded in)
template <int NM, int RT>
__device__ __forceinline__ void bn_block(const float* sc, const float* sh, int g, f32x4 (&acc)[NM][RT]) {
#pragma unroll
    for (int mg = 0; mg < NM / 4; ++mg) {
        float sv[16], tv[16];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4 a = *reinterpret_cast<const float4*>(sc + 64 * mg + 16 * g + 4 * qq);
            const float4 b = *reinterpret_cast<const float4*>(sh + 64 * mg + 16 * g + 4 * qq);
            sv[4 * qq] = a.x; sv[4 * qq + 1] = a.y; sv[4 * qq + 2] = a.z; sv[4 * qq + 3] = a.w;
            tv[4 * qq] = b.x; tv[4 * qq + 1] = b.y; tv[4 * qq + 2] = b.z; tv[4 * qq + 3] = b.w;
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < RT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[4 * mg + mt][nt][r] = fmaf(acc[4 * mg + mt][nt][r], sv[4 * r + mt], tv[4 * r + mt]);
    }
}

// the gathered operand of one k-block for this lane: 16-B piece g of row j's embedding per N tile
template <int RT>
struct XBlkT {
    f32x4 x[RT];
};

// RT, NW: launch shape (above); EB = embedding_dim / 16 k-blocks per field; I64: int64 ids; M0 / M1 / M2 = units[l] / 64
// (M2 == 0: two layers)
// parameter-area offsets (floats) shared by the kernel's once-per-launch loads and the passes
template <int M0, int M1, int M2>
struct ChainOff {
    static constexpr int ML = M2 > 0 ? M2 : M1;
    static constexpr int B1 = 64 * M0, B2 = B1 + 64 * M1, HW = B2 + 64 * M2, GB = HW + 64 * ML;
    static constexpr int BN_S = GB + 16;                           // scale of layer 0, 1, 2, then the shifts
    static constexpr int BN_T = BN_S + 64 * (M0 + M1 + M2);
    static constexpr int END = BN_T + 64 * (M0 + M1 + M2);
};

// The passes [first, first + stride, ...) < n_pass of one phase: pass q covers rows row_base + q * (16 RT NW) ... of the launch
// (absolute row numbers; rows >= row_end do not exist).  Called by all NW waves of the phase, with the launch parameters in LDS.
// FPB > 1 (embedding_dim 16 / FPB = 8 or 4, the reference's default is 4): SEVERAL fields share a 16-wide k-block — lane group g
// gathers field FPB b + fb (fb = g for E = 4, g >> 1 for E = 8), the 16-B piece g & 1 of its row for E = 8.  A block then needs the ids
// of FPB / 2 field PAIRS (two id registers for E = 4), per-lane table pointers (read from the descriptors in LDS), the FM sums reduced
// over the lane groups BEFORE squaring, and dense k-blocks addressed from row n_fields E of W0 (the embedding part need not end on a
// k-block boundary: the slots past the last field are zeroed and meet finite weight rows).  Every block: its pairs' linear entries, the
// next block's range check, the ids of the block after next.  Same arithmetic, same k order as the tile kernels' layer 0.
// REC (dctr_field_t.row_pitch, the RECORD form of embedding_dim-16 tables: a row's linear weight lies behind it, records 32 floats apart):
// row r of a field lies at table + 128 r bytes and its linear entry at lin_table + 128 r — compile-time shifts, no instruction more
template <int RT, int NW, int EB, bool I64, int M0, int M1, int M2, bool CROSS = false, int FPB = 1, bool EXPACT = false, bool REC = false, bool POOL = false>
__device__ __forceinline__ void chain_passes(const ChainParams& p, float* smem, const int wave, const int lane, const int row_base,
                                             const int row_end, const int first, const int stride, const int n_pass, int& oor) {
    if (first >= n_pass) return;                   // (workgroup-uniform: every wave of the phase skips it)
    constexpr int WROWS = 16 * RT;                 // batch rows of a wave
    constexpr int PROWS = NW * WROWS;              // batch rows per pass
    typedef XBlkT<RT> XBlk;
    constexpr int S1 = M0 * M1, S2 = M1 * M2;      // chunks (= steps) of layers 1 and 2
    constexpr int SL = S1 + S2;
    static_assert(SL >= 2, "the next pass's gather prologue needs two steps behind layer 0");
    static_assert(FPB == 1 || (EB == 1 && (FPB == 2 || FPB == 4) && !CROSS), "several fields per k-block: E = 8 / 4, plain fp32 kernels");
    static_assert(!REC || (EB == 1 && FPB == 1 && !CROSS && !EXPACT), "record-form tables: embedding_dim 16, plain fp32 kernels");
    static_assert(!POOL || (EB == 1 && FPB == 1 && !CROSS && !EXPACT && !REC && M0 == 4), "in-pass sequence pooling: embedding_dim 16, plain fp32 kernels, units[0] = 256");
    constexpr int E = FPB > 1 ? 16 / FPB : 16 * EB;
    constexpr int PPB = FPB / 2;                   // field pairs per k-block (FPB > 1)
    constexpr int PAIR = 2 * EB;                   // layer-0 steps per field pair (FPB == 1)
    typedef ChainOff<M0, M1, M2> Off;
    constexpr int B1_OFF = Off::B1, B2_OFF = Off::B2, HW_OFF = Off::HW, GB_OFF = Off::GB;
    static_assert(Off::END <= (POOL ? POOLD_OFF : FDESC_OFF), "biases + head weights + global bias + BatchNormalization must fit the parameter area");
    float* cpar = smem + CPAR_OFF;
    float* fdesc = smem + FDESC_OFF;
    float* dlw = smem + DLW_OFF;
    float* ring = smem + RING_OFF;
    const int g = lane >> 4, j = lane & 15;
    // per-lane constants of cold or once-per-pair code are rebuilt from an opaque copy of the lane index: as invariants of the
    // k-loop they would be hoisted, kept live across it, and push accumulators into scratch
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };

    const int NBE = FPB > 1 ? (p.n_fields + FPB - 1) / FPB : p.n_fields * EB;         // embedding k-blocks
    const int NB = FPB > 1 ? NBE + ((p.n_dense + 15) >> 4) : (p.in_dim + 15) >> 4;     // k-blocks of the DNN input (= steps of layer 0)
    const int NDB = NB - NBE;                      // dense k-blocks (0 .. MAX_DENSE_BLOCKS)
    // POOL: the last n_pool fields are sequences pooled inside the pass — the pair machinery below (ids, range checks, linear entries, row
    // requests) covers the first NSF fields = NBS k-blocks only; their k-blocks [NBS, NBE) take their operand from the pooled vectors
    const int NSF = POOL ? p.n_fields - p.n_pool : p.n_fields;
    const int NBS = POOL ? NSF * EB : NBE;
    const int emb_rows = p.n_fields * E;           // rows of W0 the embedding part takes (FPB > 1: not necessarily whole k-blocks)
    const int STEPS = NB + SL;
    const int k_last = p.in_dim - 1;
    float* dreg = smem + DENSE_OFF + (WROWS * wave) * (16 * NDB);     // this wave's rows of the dense staging area
    // Layer 1 starts with 128 + 64 accumulator registers live (all of layer 0's outputs, its own) and uses layer 0's M-groups
    // one after the other: the LAST M-group of layer 0 waits in LDS until the first ones are dead (8 x 1 KiB per wave, written
    // once and read once per pass) — 32 registers less at the peak, which is what keeps hipcc from spilling accumulator
    // elements to scratch there (reloads whose vmcnt(0) sits behind the DMA of the step)
    auto park_ptr = [&]() -> f32x4* {
        return reinterpret_cast<f32x4*>(smem + DENSE_OFF + PROWS * 16 * NDB + NW * RT * 64 + wave * (4 * RT * 256)) + opaque_lane();
    };
    auto dlacc_ptr = [&]() -> float* { return smem + DENSE_OFF + PROWS * 16 * NDB + wave * (RT * 64) + opaque_lane(); };   // [nt * 64]

    // ---- weight chunks.  Chunk ci of a pass: ci < NB: rows 16*ci .. + 15 of W0 (all 64*M0 columns); then the 64 x 64
    // sub-blocks (mg, mg1) of W1, mg-major; then those of W2; ci >= STEPS wraps to the next pass.  A chunk image is
    // 16 pieces of 1 KiB (fewer for M0 < 4); wave w moves pieces w, w + 8: lane l's 16 bytes land at piece + 16 l.
    // The DMA instruction is issued through inline asm on purpose: hipcc orders every later ds_read behind an LDS-DMA it
    // can see with s_waitcnt vmcnt(0) (it cannot prove the chunk being filled is not the chunk being read), which would
    // stall each step on the loads it has just requested.  Unseen, the DMA only makes hipcc's own vmcnt bookkeeping for
    // the register loads conservative (more operations in the queue than it counts: it can over-wait, never under-wait).
    // Addresses are a scalar base + one per-lane 32-bit offset (the lane index is made opaque per call: otherwise the
    // offsets of all ten layer >= 1 chunks are hoisted out of the persistent loop — live VGPRs across the pass, i.e. spills)
    auto dma16 = [&](const void* sbase, uint32_t voff, float* dst) {
        const uint32_t lds_addr = (uint32_t)(size_t)(lds_ptr_t)dst;
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma_l0 = [&](int b, float* dst) {
        constexpr int ROW_B = 256 * M0;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t o = (uint32_t)(1024 * wave + 16 * ln);                  // this lane's byte inside piece `wave`
        const int row0 = (FPB > 1 && b >= NBE) ? emb_rows + 16 * (b - NBE) : 16 * b;     // first W0 row of the block
        if (row0 + 16 <= p.in_dim) {                                           // a whole block: 16 KiB (M0 = 4) as they lie
            const char* base = reinterpret_cast<const char*>(p.W[0]) + (size_t)row0 * ROW_B;
#pragma unroll
            for (int pc0 = 0; pc0 < 4 * M0; pc0 += NW)
                if (pc0 + wave < 4 * M0) dma16(base + pc0 * 1024, o, dst + (pc0 + wave) * 256);
        } else {                                                               // K tail: rows past K get a finite stand-in
#pragma unroll
            for (int pc0 = 0; pc0 < 4 * M0; pc0 += NW)
                if (pc0 + wave < 4 * M0) {
                    const uint32_t oo = o + pc0 * 1024;
                    const int krow = min(row0 + (int)(oo / ROW_B), k_last);
                    dma16(p.W[0], (uint32_t)krow * ROW_B + (oo % ROW_B), dst + (pc0 + wave) * 256);
                }
        }
    };
    auto dma_ln = [&](const float* W, int N, int mg, int mg1, float* dst) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t o = (uint32_t)(1024 * wave + 16 * ln);
        const uint32_t voff = (o >> 8) * (uint32_t)(N * 4) + (o & 255u);       // image row o / 256 <-> weight row, 256 B of it
        const char* base = reinterpret_cast<const char*>(W) + ((size_t)(64 * mg) * N + 64 * mg1) * 4;   // scalar
#pragma unroll
        for (int pc0 = 0; pc0 < 16; pc0 += NW) dma16(base + (size_t)(4 * pc0) * N * 4, voff, dst + (pc0 + wave) * 256);
    };
    auto dma_chunk = [&](int ci, float* dst) {
        if (ci >= STEPS) ci -= STEPS;
        if (ci < NB) {
            dma_l0(ci, dst);
        } else {
            int c = ci - NB;
            if (c < S1) {
                dma_ln(p.W[1], 64 * M1, c / M1, c % M1, dst);
            } else if constexpr (M2 > 0) {
                c -= S1;
                dma_ln(p.W[2], 64 * M2, c / M2, c % M2, dst);
            }
        }
    };

    // ---- ring position: chunk of step s (counted over the whole launch) lives in slot s % 3
    int slot = 0;                                  // slot of the CURRENT step's chunk
    auto slot_ptr = [&](int ahead) -> float* {
        int s = slot + ahead;
        s = s >= NSLOT ? s - NSLOT : s;
        return ring + s * SLOT_F;
    };
    auto slot_next = [&]() { slot = slot + 1 == NSLOT ? 0 : slot + 1; };

    // ---- rows.  MFMA layout: lane (g, j) works for rows 16 nt + j of the wave's 32 (launches are cut to < 2^31 rows by
    // the host).  Ids and linear-table entries are handled ROW PER LANE for a PAIR of fields at once: lane l = row (l & 31)
    // of field 2 pr + (l >> 5) — one id load, one range check, one linear-table load per field pair instead of per k-block
    // and N tile; a k-block's ids reach the (g, j) lanes through ds_bpermute (the LDS crossbar, no memory traffic)
    auto row_of = [&](int pass, int nt) -> int { return row_base + pass * PROWS + WROWS * wave + 16 * nt + j; };
    auto brow_of = [&](int pass, int nt) -> int { return min(row_of(pass, nt), row_end - 1); };
    // request the ids of field pair pr for the rows of `pass`
    auto request_pair_ids = [&](int pr, int pass, uint32_t& lo, uint32_t& hi) {
        // element (field fi = min(2 pr + q, n_fields - 1), row r) of the id matrix: the pair's first field and the step to its second
        // one are SCALAR 64-bit products (pr is uniform), the lane adds its half's step and its row — rows are contiguous
        // (ids_stride_b == 1: chain_kernels.hip, eligible), so no per-lane 64-bit multiply (they were 6 quarter-rate instructions
        // per pair, and fp32 MFMAs share the vector lanes)
        const int ln = opaque_lane(), q = ln >> 5;
        const int f0 = min(2 * pr, NSF - 1);
        const int64_t sbase = (int64_t)f0 * p.ids_stride_f;
        const int64_t qstep = 2 * pr + 1 <= NSF - 1 ? p.ids_stride_f : (int64_t)0;
        const int r = min(row_base + pass * PROWS + WROWS * wave + min(ln & 31, WROWS - 1), row_end - 1);
        const int64_t eo = sbase + (q ? qstep : (int64_t)0) + (int64_t)r;
        if constexpr (I64) {
            const u32x2 v = *(gbl_u2_t)(reinterpret_cast<const u32x2*>(p.ids) + eo);
            lo = v[0];
            hi = v[1];
        } else {
            lo = *(gbl_u_t)(reinterpret_cast<const uint32_t*>(p.ids) + eo);
        }
    };
    // landed ids -> rows: range check against the field's vocabulary, out-of-range ids read row 0 and raise the flag
    auto fold_pair_ids = [&](int pr, int pass, uint32_t lo, uint32_t hi) -> uint32_t {
        // the lane's field of the pair: its vocabulary and identity flag come from the descriptors in LDS PER LANE (two reads; as
        // scalars of both fields + per-lane selects they were ~10 more vector instructions per pair)
        const int ln = opaque_lane(), q = ln >> 5;
        const int fi = min(2 * pr + q, NSF - 1);
        const uint2 vv = *reinterpret_cast<const uint2*>(fdesc + 12 * fi + 4);
        const uint32_t lim = vv.y != 0u ? 0xffffffffu : vv.x;
        // identity fields (pre-pooled by dctr_embed_pool: dctr_field_t.identity): the row is the sample's index in the launch
        const bool ident = reinterpret_cast<const uint32_t*>(fdesc)[12 * fi + 10] != 0u;
        const int rr = row_base + pass * PROWS + WROWS * wave + (ln & 31);
        const uint32_t upper = I64 ? hi : (uint32_t)((int32_t)lo >> 31);      // anything but 0: negative or >= 2^32
        const bool ok = ident || (upper == 0u && lo < lim);
        const bool counts = (ln & 31) < WROWS && rr < row_end && 2 * pr + q < NSF;
        if (__any(!ok && counts)) oor = 1;                 // (wave-uniform flag: a scalar register, not a VGPR)
        return ident ? (uint32_t)min(rr, row_end - 1) : (ok ? lo : 0u);
    };
    // linear-table entries of the pair (row per lane); fields without a linear table, or past the last field, give 0
    auto pair_lin_ptr = [&](int pr, uint32_t idc, bool& has) -> gbl_f_t {
        const int q = opaque_lane() >> 5;
        const int fi = min(2 * pr + q, NSF - 1);
        const uint2 l = *reinterpret_cast<const uint2*>(fdesc + 12 * fi + 2);          // (per lane: its field's linear table)
        const uint64_t base = ((uint64_t)l.y << 32) | l.x;
        has = base != 0 && 2 * pr + q < NSF;
        const float* t = has ? reinterpret_cast<const float*>(base) : reinterpret_cast<const float*>(p.fields);
        return (gbl_f_t)(t + (has ? (REC ? idc << 5 : idc) : 0u));            // (REC: linear entries lie 32 floats apart, as the rows)
    };
    // issue the row loads of embedding k-block cb: its ids are half `half` of the folded pair ids `idc`
    auto issue_x = [&](int cb, uint32_t idc, int half, XBlk& X) {
        const int f = cb / EB, h = cb % EB;
        const uint2 tw = *reinterpret_cast<const uint2*>(fdesc + 12 * f);
        const char* table = reinterpret_cast<const char*>(sgpr64(tw.x, tw.y));
        // (lane-derived constants are rebuilt from an opaque copy of the lane index: as loop invariants they would be kept
        // in scratch and reloaded — with a vmcnt wait behind the DMA just issued — in every step)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t gg = (uint32_t)ln >> 4, jj = (uint32_t)ln & 15u;
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            const uint32_t idv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((jj + (uint32_t)(32 * half + 16 * nt)) << 2), (int)idc);
            const uint64_t piece = (uint64_t)idv * (uint32_t)(REC ? E / 2 : E / 4) + (uint64_t)(4u * (uint32_t)h + gg);   // 16-B pieces from the table base
            X.x[nt] = *(gbl_f4_t)(table + (piece << 4));
        }
    };
    // the same for ONE N tile (the spread request phase issues the tiles in different micro-steps)
    // lane constants of the spread row requests, PERSISTENT (the block-form issue_x above rebuilds them from an opaque lane copy: round 2's
    // register budget; with ~25 registers to spare since then, keeping the two costs six registers and saves four vector instructions
    // per request — + 0.5 % in a same-box A/B, scripts/ab_libs.sh; the same treatment of the id / linear-entry requests measured - 1.7 %)
    const uint32_t bp4_ = ((uint32_t)lane & 15u) << 2;
    const uint64_t gg16_ = (uint64_t)(((uint32_t)lane >> 4) << 4);
    constexpr bool KEEP_LC = !CROSS && !POOL;         // (the folded-CrossNet and in-pass pooling instantiations sit at 252 - 256 registers: they rebuild)
    auto issue_x1 = [&](int cb, uint32_t idc, int half, XBlk& X, auto NTc) {
        constexpr int nt = decltype(NTc)::value;
        const int f = cb / EB, h = cb % EB;
        const uint2 tw = *reinterpret_cast<const uint2*>(fdesc + 12 * f);
        const char* table = reinterpret_cast<const char*>(sgpr64(tw.x, tw.y));
        if constexpr (KEEP_LC) {
            const uint32_t idv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(bp4_ + (uint32_t)((32 * half + 16 * nt) << 2)), (int)idc);
            X.x[nt] = *(gbl_f4_t)(table + (((uint64_t)idv << ((E == 16 ? 6 : E == 32 ? 7 : 8) + (REC ? 1 : 0))) + (uint64_t)(64 * h) + gg16_));
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const uint32_t gg = (uint32_t)ln >> 4, jj = (uint32_t)ln & 15u;
            const uint32_t idv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((jj + (uint32_t)(32 * half + 16 * nt)) << 2), (int)idc);
            const uint64_t piece = (uint64_t)idv * (uint32_t)(REC ? E / 2 : E / 4) + (uint64_t)(4u * (uint32_t)h + gg);
            X.x[nt] = *(gbl_f4_t)(table + (piece << 4));
        }
    };
    // FPB > 1: the rows of embedding k-block cb, one N tile: lane group g = field slot fb (and 16-B piece pc for E = 8); the slot's ids
    // are half fb & 1 of pair fb >> 1's register (idA: the block's first pair, idB: its second, E = 4)
    auto issue_xq1 = [&](int cb, uint32_t idA, uint32_t idB, XBlk& X, auto NTc) {
        constexpr int nt = decltype(NTc)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const uint32_t gg = (uint32_t)ln >> 4, jj = (uint32_t)ln & 15u;
        const uint32_t fb = FPB == 4 ? gg : (gg >> 1), pc = FPB == 4 ? 0u : (gg & 1u);
        const int f = min(FPB * cb + (int)fb, p.n_fields - 1);
        const uint2 tw = *reinterpret_cast<const uint2*>(fdesc + 12 * f);                      // (per lane: this slot's table)
        const int src = (int)(((fb & 1u) * 32u + (uint32_t)(16 * nt) + jj) << 2);
        uint32_t idv = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)idA);
        if constexpr (FPB == 4) {
            const uint32_t idb = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)idB);
            idv = (fb >> 1) ? idb : idv;
        }
        const uint64_t base = ((uint64_t)tw.y << 32) | tw.x;
        X.x[nt] = *(gbl_f4_t)(base + (uint64_t)idv * (uint32_t)(E * 4) + 16u * pc);
    };
    auto issue_xq = [&](int cb, uint32_t idA, uint32_t idB, XBlk& X) {
        issue_xq1(cb, idA, idB, X, std::integral_constant<int, 0>{});
        if constexpr (RT > 1) issue_xq1(cb, idA, idB, X, std::integral_constant<int, RT - 1>{});
    };
    // dense features of a pass: requested, then (a step later) written zero-padded to this wave's LDS rows — the dense
    // k-blocks of layer 0 read their B operand from there, so the hot loop has ONE kind of global load.  Lane (g, j) moves
    // columns 4g .. 4g + 3 of dense k-block c for its two rows; the lane's share of dense . dense_lin_w comes out on the way
    auto dense_request = [&](int c, int pass, float (&td)[RT][4]) {
        const int d0 = 16 * c + 4 * g;
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            gbl_f_t src = (gbl_f_t)(p.dense + (int64_t)brow_of(pass, nt) * p.dense_stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) td[nt][e] = src[min(d0 + e, p.n_dense - 1)];
        }
    };
    auto dense_store = [&](int c, const float (&td)[RT][4]) {
        const int d0 = 16 * c + 4 * g;
        float* dlacc = dlacc_ptr();
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            f32x4 v;
            float dl = c == 0 ? 0.f : dlacc[nt * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = d0 + e < p.n_dense ? td[nt][e] : 0.f;
                dl = fmaf(v[e], dlw[min(d0 + e, 255)], dl);
            }
            *reinterpret_cast<f32x4*>(dreg + (16 * nt + j) * (16 * NDB) + d0) = v;
            dlacc[nt * 64] = dl;
        }
    };
    // dense k-blocks 1.. (rare): synchronously
    auto dense_rest = [&](int pass) {
        for (int c = 1; c < NDB; ++c) {
            float td[RT][4];
            dense_request(c, pass, td);
            dense_store(c, td);
        }
    };

    // ---- the barrier of a step: everything this wave requested has landed (its share of the chunk after this one, the
    // next k-block's rows, ids, linear entries), all eight waves have finished reading the previous chunk.  The in-flight
    // registers are named so that hipcc places its own bookkeeping wait here and not in front of their first use
#define CHAIN_TOP_X(X)                                                                                           \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(lvn), "+v"(idr_lo) : : "memory");   \
        if constexpr (I64) asm volatile("" : "+v"(idr_hi));                                                      \
    } while (0)
#define CHAIN_TOP_ID()                                                                                           \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(idr_lo) : : "memory");              \
        if constexpr (I64) asm volatile("" : "+v"(idr_hi));                                                      \
    } while (0)
#define CHAIN_TOP() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // FPB > 1: also the second pair's in-flight registers
#define CHAIN_TOP_Q()                                                                                            \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(lvn), "+v"(idr_lo), "+v"(lvnB), "+v"(idrB_lo) : : "memory");   \
        if constexpr (I64) asm volatile("" : "+v"(idr_hi), "+v"(idrB_hi));                                       \
    } while (0)

    // A-operand lane offsets (floats) inside a chunk image
    const int l0off = (4 * g) * (64 * M0) + 4 * j;     // layer 0: k-step t reads row 4g + t, M-group mg at + 64 mg
    const int lnoff = (16 * g) * 64 + 4 * j;           // layers >= 1: k-step (mt, r) reads row 16g + 4r + mt

    // ---- prologue: chunks 0 and 1, the dense values, ids of field pair 0, rows of k-block 0 of the first pass
    XBlk XA, XB;
    uint32_t idr_lo = 0u, idr_hi = 0u;                 // raw ids of a field pair between their request and the range check
    uint32_t idc = 0u;                                 // checked ids (= table rows) of the current field pair
    float lvn = 0.f;                                   // linear-table entries of the current pair, in flight / landed
    bool lvn_has = false, lvnB_has = false;            // ... and whether the lane's field has a linear table at all (kept from the
                                                       // request: asking the descriptors again a step later was ~15 instructions)
    // FPB > 1: a block's SECOND pair (E = 4) has its own raw / checked ids and linear entries; idc / idr_* / lvn serve the first
    uint32_t idrB_lo = 0u, idrB_hi = 0u, idcB = 0u;
    float lvnB = 0.f;
#pragma unroll
    for (int nt = 0; nt < RT; ++nt) {
        XA.x[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        XB.x[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // in_fm of every field as one scalar bit mask (n_fields <= 64)
    const uint64_t fm_mask = __ballot(lane < p.n_fields && reinterpret_cast<const uint32_t*>(fdesc)[12 * min(lane, p.n_fields - 1) + 8] != 0u);
    {
        const int pass0 = first;
        dma_chunk(0, slot_ptr(0));
        dma_chunk(1, slot_ptr(1));
        request_pair_ids(0, pass0, idr_lo, idr_hi);
        float td[RT][4];
        if (NDB > 0) dense_request(0, pass0, td);
        CHAIN_TOP_ID();
        if (NDB > 0) {
            dense_store(0, td);
            dense_rest(pass0);
        }
        if constexpr (FPB > 1) {
            // block 0's pairs (the second pair's ids are requested and waited for here: once per phase), block 1's ids, block 0's rows
            if constexpr (PPB > 1) {
                request_pair_ids(1, pass0, idrB_lo, idrB_hi);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(idrB_lo) : : "memory");
                if constexpr (I64) asm volatile("" : "+v"(idrB_hi));
                idcB = fold_pair_ids(1, pass0, idrB_lo, idrB_hi);
            }
            idc = fold_pair_ids(0, pass0, idr_lo, idr_hi);
            request_pair_ids(PPB, pass0, idr_lo, idr_hi);
            if constexpr (PPB > 1) request_pair_ids(PPB + 1, pass0, idrB_lo, idrB_hi);
            issue_xq(0, idc, idcB, XA);
        } else {
        idc = fold_pair_ids(0, pass0, idr_lo, idr_hi);
        issue_x(0, idc, 0, XA);
        }
    }

    // Request phase of a step = gather part (rows of the next k-block, ids, linear entries: latency-critical, behind micro-
    // step 0 for every wave) + DMA part (this wave's share of the chunk after next: L2 hits, latency-tolerant).  The DMA part
    // sits behind micro-step 0 for waves 0-3 and behind micro-step CHAIN_DMA_LATE for waves 4-7, so that the CU's vector-
    // memory address path does not get all eight waves' requests in one burst (it has no register results, so placing it
    // twice costs nothing; the gather part placed twice merges in-flight registers of the two placements and spills).
    const bool dma_early = wave < NW / 2;
    constexpr int DMA_LATE0 = CHAIN_DMA_LATE < 4 * M0 ? CHAIN_DMA_LATE : 2 * M0;     // layer 0 has 4 M0 micro-steps per step (even index)
    static_assert(DMA_LATE0 % 2 == 0 && DMA_LATE0 < 4 * M0 && CHAIN_DMA_LATE % 2 == 0 && CHAIN_DMA_LATE < 16, "late DMA slot outside the step");
    for (int it = 0, pass = first; pass < n_pass; ++it, pass += stride) {
        constexpr int PH = 0;
        (void)it;
        const int pass_n = min(pass + stride, n_pass - 1);               // rows the gather prologue at the pass's end is for
        // ================= layer 0: acc0[4 mg + mt][nt] = C tile of output features 64 mg + 4 i + mt
        f32x4 acc0[4 * M0][RT];
#pragma unroll
        for (int mg = 0; mg < M0; ++mg) {
            float bv[16];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const float4 t = *reinterpret_cast<const float4*>(cpar + 64 * mg + 16 * g + 4 * qq);
                bv[4 * qq] = t.x; bv[4 * qq + 1] = t.y; bv[4 * qq + 2] = t.z; bv[4 * qq + 3] = t.w;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt)
                    acc0[4 * mg + mt][nt] = f32x4{bv[mt], bv[4 + mt], bv[8 + mt], bv[12 + mt]};
        }
        f32x4 sum[EB][RT];
        float sq[RT];
        // FPB == 1: the sum of squares as TWO partial sums per N tile (elements 0, 2 / 1, 3 of every 16-B piece): its fmas then pair up
        // on registers that already lie side by side (v_pk_fma_f32 on X.x[nt][0..1], [2..3]); as one chain per tile hipcc packed
        // ACROSS the tiles and paid eight register moves per k-block to line the operands up (same-box A/B: + 0.4 %)
        typedef float sq2_t __attribute__((ext_vector_type(2)));
        sq2_t sq2[RT];
        // CROSS: this lane's share of the row's dot products with the cross vectors (reduced over g after layer 0)
        float cp[CROSS ? CROSS_NV : 1][RT];
#pragma unroll
        for (int v = 0; v < (CROSS ? CROSS_NV : 1); ++v)
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) cp[v][nt] = 0.f;
        auto cross_x = [&](int b, const XBlk& X) {
            if constexpr (CROSS) {
                const float* xv = smem + p.xv_off + 16 * b + 4 * (opaque_lane() >> 4);
#pragma unroll
                for (int v = 0; v < CROSS_NV; ++v) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(xv + v * (16 * NB));
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) cp[v][nt] = fmaf(X.x[nt][e], w[e], cp[v][nt]);
                }
            }
        };
        float linacc = 0.f;                            // row-per-lane: this lane's field of every pair
        uint32_t idcn = 0u;                            // checked ids of the NEXT field pair
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            sq[nt] = 0.f;
            sq2[nt] = sq2_t{0.f, 0.f};
#pragma unroll
            for (int h = 0; h < EB; ++h) sum[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- POOL: VarLenSparseFeat pooled inside the pass (reference inputs.py:120-158 varlen_embedding_lookup + get_varlen_pooling_list,
        // layers/sequence.py:76-106 SequencePoolingLayer: sum / mean over the valid positions).  The rows of a sequence are requested TWO
        // POSITIONS PER LAYER-0 STEP beside the step's MFMAs — a three-stage pipeline across steps, every stage's loads landing behind the
        // next step's barrier: ids of positions (t, t + 1) [one 8-B load per N tile] -> their rows (lane (g, j): 16-B piece g of sample j's
        // row, the layout of the block's B operand) + linear entries -> acc += row * mask in the order of t (dctr_embed_pool's order and
        // arithmetic: the pooled vector is bit-identical to the pre-pass's).  A finished sequence's vector waits, lane-private, in the
        // wave's park area (free during layer 0) until its k-block's step reads it as the dense k-blocks read theirs.  All of this state
        // dies with layer 0: the register peak of the kernel (layer 1: 192 accumulator registers) is untouched.
        int pl_next_f = 0, pl_next_t = 0;              // (wave-uniform) the next (sequence, position) whose ids are requested
        int pl_ids_f = -1, pl_ids_t = 0;               // sequence / first position of the ids in flight (-1: none)
        int pl_rows_f = -1;                            // sequence of the rows in flight (-1: none)
        bool pl_ids_last = false, pl_rows_last = false;    // ... and whether they are that sequence's last
        u32x2 pl_id[RT];
        f32x4 pl_row[2][RT], pl_acc[RT];
        float pl_lv[2][RT], pl_lin[RT], pl_cnt[RT];    // pl_cnt: the mean's denominator so far (valid positions, or the row's length)
        float pl_lf[2] = {0.f, 0.f};                   // finished sequences' pooled first-order terms, row per lane (sequence 2 r + lane half -> [r])
        int pl_len[RT];                                // valid lengths of the sequence whose ids are in flight (length_name given)
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            pl_acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            pl_lin[nt] = 0.f;
            pl_cnt[nt] = 0.f;
            pl_len[nt] = 0;
            pl_id[nt] = u32x2{0u, 0u};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pl_row[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                pl_lv[u][nt] = 0.f;
            }
        }
        // (descriptor i lies in LDS since the launch's start — a load through p.pool here would be a vector load + a wait for EVERYTHING
        // in flight: the DMA, the rows — in every step: measured, + 1.7 us per step)
        auto pool_seq = [&](int i) -> dctr_pool_seq_t {
            const uint4 a = *reinterpret_cast<const uint4*>(cpar + POOLD_OFF + 8 * i), b = *reinterpret_cast<const uint4*>(cpar + POOLD_OFF + 8 * i + 4);
            dctr_pool_seq_t d;
            d.idx = reinterpret_cast<const void*>(sgpr64(a.x, a.y));
            d.length = reinterpret_cast<const int32_t*>(sgpr64(a.z, a.w));
            d.idx_stride = (int64_t)sgpr64(b.x, b.y);
            d.maxlen = __builtin_amdgcn_readfirstlane((int)b.z);
            d.combiner = __builtin_amdgcn_readfirstlane((int)b.w);
            return d;
        };
        auto pool_piece = [&]() {
            if constexpr (POOL) {
                // (A) the rows requested a step ago have landed: accumulate; a finished sequence leaves for the park area
                if (pl_rows_f >= 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int nt = 0; nt < RT; ++nt) {
                            // (a masked or out-of-range position arrives as zeros — stage B — where the pre-pass adds row * 0: acc + 0 = acc)
                            pl_acc[nt] += pl_row[u][nt];
                            pl_lin[nt] += pl_lv[u][nt];
                        }
                    if (pl_rows_last) {
                        const bool mean = ((p.pool_flags >> pl_rows_f) & 1) != 0;
                        f32x4* pv = park_ptr();
                        float lfin[RT];
#pragma unroll
                        for (int nt = 0; nt < RT; ++nt) {
                            f32x4 v = pl_acc[nt];
                            float l = pl_lin[nt];
                            if (mean) {
                                const float denom = pl_cnt[nt] + 1e-8f;                                                 // sequence.py:65,103
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] / denom;
                                l = l / denom;
                            }
                            pv[(pl_rows_f * RT + nt) * 64] = v;
                            lfin[nt] = l;
                            pl_acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            pl_lin[nt] = 0.f;
                            pl_cnt[nt] = 0.f;
                        }
                        // the pooled first-order term joins the linear sum where the pre-pooled (identity) form added it: row per lane,
                        // lanes 0-31 / 32-63 = the even / odd field of a pair, after the SparseFeat pairs in field order
                        const int ln = opaque_lane(), r = ln & 31;
                        float take = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (r & 15), __builtin_bit_cast(int, lfin[0])));
                        if constexpr (RT > 1) {
                            const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (r & 15), __builtin_bit_cast(int, lfin[RT - 1])));
                            take = (r >> 4) ? t1 : take;
                        }
                        // (kept until the SparseFeat pairs' entries are in — below, behind the step loop — so that the sum's order is the
                        // identity form's: sequence i is field NSF + i, the lane half of its parity, pairs in field order)
                        const uint2 lt = *reinterpret_cast<const uint2*>(fdesc + 12 * (NSF + pl_rows_f) + 2);
                        const bool has_lin = (lt.x | lt.y) != 0u;
                        const bool mine = has_lin && (ln >> 5) == ((NSF + pl_rows_f) & 1);
                        const int rnd = pl_rows_f >> 1;        // sequences i and i + 2 share a lane half: [0] the half's first, [1] its second
                        if (rnd == 0) pl_lf[0] = mine ? take : pl_lf[0];
                        else pl_lf[1] = mine ? take : pl_lf[1];
                    }
                    pl_rows_f = -1;
                }
                // (B) the ids requested a step ago have landed: range check, masks, row + linear-entry requests
                if (pl_ids_f >= 0) {
                    const bool by_len = ((p.pool_flags >> (4 + pl_ids_f)) & 1) != 0;
                    const int fi = NSF + pl_ids_f;
                    const uint4 tw = *reinterpret_cast<const uint4*>(fdesc + 12 * fi);           // table, lin_table
                    const uint2 vw = *reinterpret_cast<const uint2*>(fdesc + 12 * fi + 4);       // vocab
                    const char* table = reinterpret_cast<const char*>(sgpr64(tw.x, tw.y));
                    const uint64_t linb = sgpr64(tw.z, tw.w);
                    const uint32_t lim = (uint32_t)__builtin_amdgcn_readfirstlane((int)(vw.y != 0u ? 0xffffffffu : vw.x));
                    const int gg = opaque_lane() >> 4;
                    bool bad = false;
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) {
                        const bool row_ok = row_of(pass, nt) < row_end;
                        if (by_len) pl_cnt[nt] = (float)pl_len[nt];                    // (mean over length_name's lengths)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const uint32_t id = pl_id[nt][u];
                            const int t = pl_ids_t + u;                               // (< maxlen: maxlen is even, pieces are pairs)
                            const bool ok = id < lim && (int32_t)id >= 0;
                            const bool m = by_len ? t < pl_len[nt] : id != 0u;
                            bad = bad || (row_ok && !ok);
                            if (!by_len && m) pl_cnt[nt] += 1.f;                       // (mean over the valid positions, in range or not)
                            pl_row[u][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            pl_lv[u][nt] = 0.f;
                            if (m && ok) {                                             // masked positions request nothing
                                // (scalar base + one 32-bit lane offset: the host admits tables below 4 GiB here — no 64-bit address pair per load)
                                pl_row[u][nt] = *(gbl_f4_t)(table + (uint32_t)((id << 6) + (uint32_t)(16 * gg)));
                                if (linb != 0) pl_lv[u][nt] = *(gbl_f_t)(reinterpret_cast<const char*>(linb) + (uint32_t)(id << 2));
                            }
                        }
                    }
                    if (__any(bad)) oor = 1;
                    pl_rows_f = pl_ids_f;
                    pl_rows_last = pl_ids_last;
                    pl_ids_f = -1;
                }
                // (C) the next two positions' ids (one 8-B load per N tile; at a sequence's first positions also its rows' lengths)
                if (pl_next_f < p.n_pool) {
                    const dctr_pool_seq_t sq = pool_seq(pl_next_f);
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) {
                        // (32-bit lane offsets from the scalar bases: the host admits id matrices below 4 GiB)
                        const uint32_t br = (uint32_t)brow_of(pass, nt) + (uint32_t)p.pool_row0;
                        pl_id[nt] = *(gbl_u2_t)(reinterpret_cast<const char*>(sq.idx) + (uint32_t)((br * (uint32_t)sq.idx_stride + (uint32_t)pl_next_t) << 2));
                        if (pl_next_t == 0 && sq.length != nullptr) pl_len[nt] = *(gbl_i_t)(reinterpret_cast<const char*>(sq.length) + (uint32_t)(br << 2));
                    }
                    pl_ids_f = pl_next_f;
                    pl_ids_t = pl_next_t;
                    pl_next_t += 2;
                    pl_ids_last = pl_next_t >= sq.maxlen;
                    if (pl_ids_last) {
                        ++pl_next_f;
                        pl_next_t = 0;
                    }
                }
            }
        };
        // operand pipeline of every layer: micro-steps of ONE ds_read_b128 (4 A fragments = 4 M-tiles) + 4 x RT MFMAs; the
        // fragment of micro-step u + 1 is requested before the MFMAs of micro-step u (two 4-register buffers, c0 / c1).
        // Layer 0, k-block step: micro-step u = (k-step t = u / M0, M-group mg = u % M0) reads weight row 4g + t
        f32x4 c0, c1;
        // RT = 1 (the tail phase): a micro-step is 4 MFMAs = 128 cycles, about one LDS round trip — a fragment requested one
        // micro-step ahead arrives as its MFMAs want to issue (stamps: ~60 cycles lost per micro-step).  There the fragments are
        // requested TWO micro-steps ahead (c2 / c3 = the next pair)
        constexpr bool DEEP = RT == 1;
        f32x4 c2, c3;
        auto read_l0 = [&](const float* sb, int u) -> f32x4 {
            return *reinterpret_cast<const f32x4*>(sb + l0off + (u / M0) * (64 * M0) + 64 * (u % M0));
        };
        auto mfma_l0 = [&](const f32x4& a, const XBlk& X, int u) {
            const int t = u / M0, mg = u % M0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt)
                    mfma_ip(acc0[4 * mg + mt][nt], a[mt], X.x[nt][t]);
        };
        // FM bookkeeping of the embedding block being multiplied (lane-local)
        auto consume_x = [&](int b, const XBlk& X) {
            if constexpr (FPB > 1) {
                // the lane's field slot: in_fm per lane (slots past the last field hold zeros)
                const int gq = opaque_lane() >> 4;
                const int f = FPB * b + (FPB == 4 ? gq : (gq >> 1));
                const bool in = f < p.n_fields && ((fm_mask >> (f & 63)) & 1ull) != 0ull;
#pragma unroll
                for (int nt = 0; nt < RT; ++nt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = in ? X.x[nt][e] : 0.f;
                        sum[0][nt][e] += v;
                        sq[nt] = fmaf(v, v, sq[nt]);
                    }
                return;
            }
            const int f = b / EB, h = b % EB;
            if ((fm_mask >> f) & 1ull) {
#pragma unroll
                for (int nt = 0; nt < RT; ++nt) {
#pragma unroll
                    for (int hh = 0; hh < EB; ++hh)
                        if (hh == h) sum[hh][nt] += X.x[nt];
                    {
                        const sq2_t lo = sq2_t{X.x[nt][0], X.x[nt][1]}, hi = sq2_t{X.x[nt][2], X.x[nt][3]};
                        sq2[nt] = __builtin_elementwise_fma(lo, lo, sq2[nt]);
                        sq2[nt] = __builtin_elementwise_fma(hi, hi, sq2[nt]);
                    }
                }
            }
        };
        // the request phase of layer-0 step (k-block b_ = step s_ of field pair pr_): DMA share of the chunk after next,
        // rows of the next k-block, and once per pair: the pair's linear entries (s = 0), the next pair's ids (s = PAIR - 2),
        // their range check (s = PAIR - 1).  A dense k-block takes its operand from the staging rows in LDS instead
#define CHAIN_PHASE0(XC, XN)                                                                                     \
        {                                                                                                        \
            if (dma_early) dma_chunk(b_ + 2, slot_ptr(2));                                                       \
            {                                                                                  \
                /* every request is issued unconditionally (past the last field / pair: a clamped, redundant one):   \
                   a conditionally written register keeps its old value alive — through layers 1.. where 192 of the   \
                   256 registers hold accumulators */                                                             \
                const int prn_ = min(pr_ + 1, (NBS - 1) / PAIR);                                                 \
                if (s_ == 1) linacc += (lvn_has && b_ - 1 < NBS) ? lvn : 0.f;                                    \
                if (s_ == PAIR - 1) idcn = fold_pair_ids(prn_, pass, idr_lo, idr_hi);                            \
                /* (past the last embedding block the clamped request would pair the LAST field's table with another field's ids — \
                   beyond the table where that field's vocabulary is the larger one: such a request reads row 0) */                \
                if (s_ + 1 < PAIR) issue_x(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idc, (s_ + 1) / EB, XN);                  \
                else issue_x(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idcn, 0, XN);                                           \
                if (s_ == 0) {                                                                                   \
                    bool has_;                                                                                   \
                    lvn = *pair_lin_ptr(min(pr_, (NBS - 1) / PAIR), idc, has_);                                  \
                    lvn_has = has_;                                                                              \
                }                                                                                                \
                if (s_ == PAIR - 2) request_pair_ids(prn_, pass, idr_lo, idr_hi);                                \
                if (b_ < NBE) consume_x(b_, XC);                                                                 \
                cross_x(b_, XC);                                                                                 \
            }                                                                                                    \
        }
        // The same requests SPREAD over the micro-steps of the step (CHAIN_SPREAD): all eight waves leave the step's barrier
        // together, and as one block the phase puts ~40 vector-memory instructions of the workgroup into the CU's address path at
        // once — the last wave in the queue waited ~3k cycles inside its phase (cycle stamps, profiles/r03_chain_lab_stamps.log)
        // with its MFMA stream stopped behind it.  Slot i (behind the MFMAs of micro-step 2 i) of the 2 M0 slots of a step:
        //   0: DMA share (waves 0-3), linear entries of the previous step's pair, range check of the next pair's ids;
        //   1 / 2: rows of the next k-block, N tile 0 / 1;  3: the pair's linear entries or the next pair's ids;
        //   DMA_LATE0 / 2: DMA share (waves 4-7);  last: FM bookkeeping of the block being multiplied (VALU only).
#define CHAIN_PIECE0(I, XC, XN)                                                                                  \
        {                                                                                                        \
            constexpr int i_ = (I);                                                                              \
            const int prn_ = min(pr_ + 1, (NBS - 1) / PAIR);                                                     \
            if (i_ == 0) {                                                                                       \
                if (dma_early) dma_chunk(b_ + 2, slot_ptr(2));                                                   \
                if (s_ == 1) linacc += (lvn_has && b_ - 1 < NBS) ? lvn : 0.f;                    \
                if (s_ == PAIR - 1) idcn = fold_pair_ids(prn_, pass, idr_lo, idr_hi);            \
            }                                                                                                    \
            if (i_ == 1) {                                                                       \
                if (s_ + 1 < PAIR) issue_x1(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idc, (s_ + 1) / EB, XN, std::integral_constant<int, 0>{});      \
                else issue_x1(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idcn, 0, XN, std::integral_constant<int, 0>{});             \
            }                                                                                                    \
            if (i_ == 2 && RT > 1) {                                                             \
                if (s_ + 1 < PAIR) issue_x1(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idc, (s_ + 1) / EB, XN, std::integral_constant<int, RT - 1>{}); \
                else issue_x1(min(b_ + 1, NBS - 1), b_ + 1 > NBS - 1 ? 0u : idcn, 0, XN, std::integral_constant<int, RT - 1>{});        \
            }                                                                                                    \
            if (i_ == 3) {                                                                       \
                if (s_ == 0) {                                                                                   \
                    bool has_;                                                                                   \
                    lvn = *pair_lin_ptr(min(pr_, (NBS - 1) / PAIR), idc, has_);                                  \
                    lvn_has = has_;                                                                              \
                }                                                                                                \
                if (s_ == PAIR - 2) request_pair_ids(prn_, pass, idr_lo, idr_hi);                                \
            }                                                                                                    \
            if (i_ == 2 * M0 - 1) {                                                              \
                if (b_ < NBE) consume_x(b_, XC);                                                                 \
                cross_x(b_, XC);                                                                                 \
            }                                                                                                    \
        }
#define CHAIN_STEP0(S, XC, XN)                                                                                   \
        {                                                                                                        \
            constexpr int s_ = (S);                                                                              \
            const int b_ = pr_ * PAIR + s_;                                                                      \
            CHAIN_TOP_X(XC);                                                                                     \
            if constexpr (POOL) {   /* (the pooling pipeline's loads have landed behind the same wait: hipcc's bookkeeping goes here) */ \
                _Pragma("unroll") for (int nt_ = 0; nt_ < RT; ++nt_)                                             \
                    asm volatile("" : "+v"(pl_id[nt_]), "+v"(pl_len[nt_]), "+v"(pl_row[0][nt_]), "+v"(pl_row[1][nt_]),  \
                                 "+v"(pl_lv[0][nt_]), "+v"(pl_lv[1][nt_]));                                      \
            }                                                                                                    \
            const float* sb_ = slot_ptr(0);                                                                      \
            if (b_ == 0) c0 = read_l0(sb_, 0);                                                                   \
            if (DEEP && b_ == 0) c1 = read_l0(sb_, 1);                                                           \
            if (b_ >= NBE) {                                                                                     \
                _Pragma("unroll") for (int nt_ = 0; nt_ < RT; ++nt_)                                             \
                    XC.x[nt_] = *reinterpret_cast<const f32x4*>(dreg + (16 * nt_ + j) * (16 * NDB) + 16 * (b_ - NBE) + 4 * g); \
            } else if (POOL && b_ >= NBS) {                                                                      \
                const f32x4* pv_ = park_ptr();                /* the pooled vector of sequence b_ - NBS (pool_piece) */ \
                _Pragma("unroll") for (int nt_ = 0; nt_ < RT; ++nt_) XC.x[nt_] = pv_[((b_ - NBS) * RT + nt_) * 64]; \
            }                                                                                                    \
            _Pragma("unroll") for (int u_ = 0; u_ < 4 * M0; u_ += 2) {                                           \
                if (!DEEP) c1 = read_l0(sb_, u_ + 1);                                                            \
                else if (u_ + 2 < 4 * M0) c2 = read_l0(sb_, u_ + 2);                                             \
                else if (b_ + 1 < NB) c2 = read_l0(slot_ptr(1), 0);                                              \
                DCTR_SB;                                                                                         \
                mfma_l0(c0, XC, u_);                                                                             \
                DCTR_SB;                                                                                         \
                if (POOL && u_ == 0 && b_ < NBS) pool_piece();   /* first in the step: its loads get the whole step to land */ \
                if (CHAIN_SPREAD) {                                                                              \
                    if (u_ == 0) CHAIN_PIECE0(0, XC, XN)                                                         \
                    if (u_ == 2) CHAIN_PIECE0(1, XC, XN)                                                         \
                    if (u_ == 4) CHAIN_PIECE0(2, XC, XN)                                                         \
                    if (u_ == 6) CHAIN_PIECE0(3, XC, XN)                                                         \
                    if (u_ == 4 * M0 - 2 && u_ > 6) CHAIN_PIECE0(2 * M0 - 1, XC, XN)                             \
                } else if (u_ == PH) CHAIN_PHASE0(XC, XN)                                                        \
                if (u_ == DMA_LATE0 && !dma_early) dma_chunk(b_ + 2, slot_ptr(2));                               \
                if (!DEEP) {                                                                                     \
                    if (u_ + 2 < 4 * M0) c0 = read_l0(sb_, u_ + 2);                                              \
                    else if (b_ + 1 < NB) c0 = read_l0(slot_ptr(1), 0);                                          \
                } else {                                                                                         \
                    if (u_ + 3 < 4 * M0) c3 = read_l0(sb_, u_ + 3);                                              \
                    else if (b_ + 1 < NB) c3 = read_l0(slot_ptr(1), 1);                                          \
                }                                                                                                \
                DCTR_SB;                                                                                         \
                mfma_l0(c1, XC, u_ + 1);                                                                         \
                DCTR_SB;                                                                                         \
                if (DEEP && (u_ + 2 < 4 * M0 || b_ + 1 < NB)) {                                                  \
                    c0 = c2;                                                                                     \
                    c1 = c3;                                                                                     \
                }                                                                                                \
            }                                                                                                    \
            slot_next();                                                                                         \
        }
        // FPB > 1: a step = one k-block of FPB fields.  Slot i behind the MFMAs of micro-step 2 i: 0: DMA share (waves 0-3), range check
        // of the NEXT block's ids (requested a step ago), the previous block's linear entries; 1 / 2: rows of the next block, N tile
        // 0 / 1; 3: this block's linear entries, the ids of the block after next; last: FM bookkeeping
        uint32_t idcnB = 0u;                           // (idcn: the next block's first pair)
#define CHAIN_PIECEQ(I, XC, XN)                                                                                  \
        {                                                                                                        \
            constexpr int i_ = (I);                                                                              \
            if (i_ == 0) {                                                                                       \
                if (dma_early) dma_chunk(b_ + 2, slot_ptr(2));                                                   \
                {                                                                              \
                    idcn = fold_pair_ids(PPB * (b_ + 1), pass, idr_lo, idr_hi);                                  \
                    if constexpr (PPB > 1) idcnB = fold_pair_ids(PPB * (b_ + 1) + 1, pass, idrB_lo, idrB_hi);    \
                    linacc += (lvn_has && b_ >= 1 && b_ - 1 < NBE) ? lvn : 0.f;                                  \
                    if constexpr (PPB > 1) linacc += (lvnB_has && b_ >= 1 && b_ - 1 < NBE) ? lvnB : 0.f;         \
                }                                                                                                \
            }                                                                                                    \
            if (i_ == 1) issue_xq1(min(b_ + 1, NBE - 1), b_ + 1 > NBE - 1 ? 0u : idcn, b_ + 1 > NBE - 1 ? 0u : idcnB, XN, std::integral_constant<int, 0>{});         \
            if (i_ == 2 && RT > 1) issue_xq1(min(b_ + 1, NBE - 1), b_ + 1 > NBE - 1 ? 0u : idcn, b_ + 1 > NBE - 1 ? 0u : idcnB, XN, std::integral_constant<int, RT - 1>{}); \
            if (i_ == 3) {                                                                       \
                bool has_;                                                                                       \
                lvn = *pair_lin_ptr(PPB * b_, idc, has_);                                                        \
                lvn_has = has_;                                                                                  \
                if constexpr (PPB > 1) {                                                                         \
                    lvnB = *pair_lin_ptr(PPB * b_ + 1, idcB, has_);                                              \
                    lvnB_has = has_;                                                                             \
                }                                                                                                \
                request_pair_ids(PPB * (b_ + 2), pass, idr_lo, idr_hi);                                          \
                if constexpr (PPB > 1) request_pair_ids(PPB * (b_ + 2) + 1, pass, idrB_lo, idrB_hi);             \
            }                                                                                                    \
            if (i_ == 2 * M0 - 1) {                                                              \
                if (b_ < NBE) consume_x(b_, XC);                                                                 \
            }                                                                                                    \
        }
#define CHAIN_STEPQ(XC, XN)                                                                                      \
        {                                                                                                        \
            CHAIN_TOP_Q();                                                                                       \
            const float* sb_ = slot_ptr(0);                                                                      \
            if (b_ == 0) c0 = read_l0(sb_, 0);                                                                   \
            if (DEEP && b_ == 0) c1 = read_l0(sb_, 1);                                                           \
            if (b_ >= NBE) {                                                                                     \
                _Pragma("unroll") for (int nt_ = 0; nt_ < RT; ++nt_)                                             \
                    XC.x[nt_] = *reinterpret_cast<const f32x4*>(dreg + (16 * nt_ + j) * (16 * NDB) + 16 * (b_ - NBE) + 4 * g); \
            } else if (b_ == NBE - 1 && p.n_fields % FPB != 0) {                                                 \
                /* field slots past the last field: zeros (their loads were clamped to the last field) */        \
                const int gq_ = opaque_lane() >> 4;                                                              \
                const bool ok_ = FPB * b_ + (FPB == 4 ? gq_ : (gq_ >> 1)) < p.n_fields;                          \
                _Pragma("unroll") for (int nt_ = 0; nt_ < RT; ++nt_)                                             \
                    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) XC.x[nt_][e_] = ok_ ? XC.x[nt_][e_] : 0.f;  \
            }                                                                                                    \
            _Pragma("unroll") for (int u_ = 0; u_ < 4 * M0; u_ += 2) {                                           \
                if (!DEEP) c1 = read_l0(sb_, u_ + 1);                                                            \
                else if (u_ + 2 < 4 * M0) c2 = read_l0(sb_, u_ + 2);                                             \
                else if (b_ + 1 < NB) c2 = read_l0(slot_ptr(1), 0);                                              \
                DCTR_SB;                                                                                         \
                mfma_l0(c0, XC, u_);                                                                             \
                DCTR_SB;                                                                                         \
                if (u_ == 0) CHAIN_PIECEQ(0, XC, XN)                                                             \
                if (u_ == 2) CHAIN_PIECEQ(1, XC, XN)                                                             \
                if (u_ == 4) CHAIN_PIECEQ(2, XC, XN)                                                             \
                if (u_ == 6) CHAIN_PIECEQ(3, XC, XN)                                                             \
                if (u_ == 4 * M0 - 2 && u_ > 6) CHAIN_PIECEQ(2 * M0 - 1, XC, XN)                                 \
                if (u_ == DMA_LATE0 && !dma_early) dma_chunk(b_ + 2, slot_ptr(2));                               \
                if (!DEEP) {                                                                                     \
                    if (u_ + 2 < 4 * M0) c0 = read_l0(sb_, u_ + 2);                                              \
                    else if (b_ + 1 < NB) c0 = read_l0(slot_ptr(1), 0);                                          \
                } else {                                                                                         \
                    if (u_ + 3 < 4 * M0) c3 = read_l0(sb_, u_ + 3);                                              \
                    else if (b_ + 1 < NB) c3 = read_l0(slot_ptr(1), 1);                                          \
                }                                                                                                \
                DCTR_SB;                                                                                         \
                mfma_l0(c1, XC, u_ + 1);                                                                         \
                DCTR_SB;                                                                                         \
                if (DEEP && (u_ + 2 < 4 * M0 || b_ + 1 < NB)) {                                                  \
                    c0 = c2;                                                                                     \
                    c1 = c3;                                                                                     \
                }                                                                                                \
            }                                                                                                    \
            slot_next();                                                                                         \
            idc = idcn;                                                                                          \
            idcB = idcnB;                                                                                        \
        }
        if constexpr (FPB > 1) {
            for (int b0_ = 0; b0_ < NB; b0_ += 2) {
                {
                    const int b_ = b0_;
                    CHAIN_STEPQ(XA, XB);
                }
                if (b0_ + 1 < NB) {
                    const int b_ = b0_ + 1;
                    CHAIN_STEPQ(XB, XA);
                }
            }
            // the last block's linear entries (added a block later, which never came)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lvn), "+v"(lvnB) : : "memory");
            linacc += (lvn_has && NB - 1 < NBE) ? lvn : 0.f;
            if constexpr (PPB > 1) linacc += (lvnB_has && NB - 1 < NBE) ? lvnB : 0.f;
        } else {
        for (int pr_ = 0; pr_ * PAIR < NB; ++pr_) {
            CHAIN_STEP0(0, XA, XB);
            if (pr_ * PAIR + 1 < NB) CHAIN_STEP0(1, XB, XA);
            if constexpr (EB >= 2) {
                if (pr_ * PAIR + 2 < NB) CHAIN_STEP0(2, XA, XB);
                if (pr_ * PAIR + 3 < NB) CHAIN_STEP0(3, XB, XA);
            }
            if constexpr (EB == 4) {                   // embedding_dim 64: eight k-blocks per field pair
                if (pr_ * PAIR + 4 < NB) CHAIN_STEP0(4, XA, XB);
                if (pr_ * PAIR + 5 < NB) CHAIN_STEP0(5, XB, XA);
                if (pr_ * PAIR + 6 < NB) CHAIN_STEP0(6, XA, XB);
                if (pr_ * PAIR + 7 < NB) CHAIN_STEP0(7, XB, XA);
            }
            idc = idcn;
        }
        }
#undef CHAIN_STEPQ
#undef CHAIN_PIECEQ
#undef CHAIN_STEP0
#undef CHAIN_PIECE0
#undef CHAIN_PHASE0
        if constexpr (POOL) {      // the pooled first-order terms, in the pairs' order
            linacc += pl_lf[0];
            linacc += pl_lf[1];
        }
        if (FPB == 1 && (NB - 1) % PAIR == 0 && NB - 1 < NBS) {
            // the last step was step 0 of a field pair (odd field count, no dense k-block behind it): the pair's linear
            // entries, added up in a pair's step 1, are still on their way
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(lvn) : : "memory");
            linacc += lvn_has ? lvn : 0.f;
        }
        mfma_drain();
        // ---- gather epilogue of the pass: FM = 0.5 (sum_d (sum_f e)^2 - sum_{f,d} e^2) lane-local, then over g; the linear
        // terms sit row per lane (lane l and l + 32: the two fields of every pair) and go to the (g, j) lanes by bpermute
        float extras[RT];
        {
            const float lin_rows = linacc + __shfl_xor(linacc, 32, 64);
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) {
                float fm = FPB > 1 ? -sq[nt] : -(sq2[nt][0] + sq2[nt][1]);
                if constexpr (FPB > 1) {
                    // the lane groups of one 16-B piece hold different FIELDS' shares of the same dimensions: sum them first, square,
                    // and count each square once (FPB lane groups hold the same sum)
                    float s2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float sv = sum[0][nt][e];
                        if constexpr (FPB == 4) sv += __shfl_xor(sv, 16, 64);
                        sv += __shfl_xor(sv, 32, 64);
                        s2 = fmaf(sv, sv, s2);
                    }
                    fm = fmaf(s2, 1.f / FPB, fm);
                } else {
#pragma unroll
                for (int h = 0; h < EB; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fm = fmaf(sum[h][nt][e], sum[h][nt][e], fm);
                }
                fm += __shfl_xor(fm, 16, 64);
                fm += __shfl_xor(fm, 32, 64);
                fm *= 0.5f;
                float dl = NDB > 0 ? dlacc_ptr()[nt * 64] : 0.f;
                dl += __shfl_xor(dl, 16, 64);
                dl += __shfl_xor(dl, 32, 64);
                const float lin_all = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * (16 * nt + j), __builtin_bit_cast(int, lin_rows))) + dl;
                extras[nt] = (p.fm_used ? fm : 0.f) + (p.lin_used ? lin_all : 0.f);
                if constexpr (CROSS) {
                    const float* cq = smem + p.xv_off + CROSS_NV * 16 * NB;
                    float dots[CROSS_NV], cst[CROSS_NV];
#pragma unroll
                    for (int v = 0; v < CROSS_NV; ++v) {
                        float d = cp[v][nt];
                        d += __shfl_xor(d, 16, 64);
                        d += __shfl_xor(d, 32, 64);
                        dots[v] = d;
                        cst[v] = cq[v];
                    }
                    extras[nt] += cross_logit(dots, cst, p.cross_layers);
                }
                const int r = row_of(pass, nt);
                if (g == 0 && r < row_end) {
                    if (p.fm_logit != nullptr) p.fm_logit[r] = fm;
                    if (p.lin_logit != nullptr) p.lin_logit[r] = lin_all;
                }
            }
        }
        // (BatchNormalization scale / shift,) activation in place: acc0 is now the B operand of layer 1
        if (p.bn_scale[0] != nullptr) bn_block<4 * M0, RT>(cpar + Off::BN_S, cpar + Off::BN_T, g, acc0);
        act_block<4 * M0, RT, EXPACT>(p.activation, acc0);
        if constexpr (M0 > 1) {
            f32x4* park = park_ptr();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < RT; ++nt) park[(mt * RT + nt) * 64] = acc0[4 * (M0 - 1) + mt][nt];
        }

        // ================= layers >= 1: one step per 64 x 64 sub-block (mg, mg1): 16 k-steps (mt, r) of 4 M-tiles x RT MFMAs
        auto read_an = [&](const float* sb, int ks) -> f32x4 {
            const int mt = ks >> 2, r = ks & 3;
            return *reinterpret_cast<const f32x4*>(sb + lnoff + (4 * r + mt) * 64);
        };
        int sidx = 0;                                  // step index behind layer 0 (compile-time after unrolling)
        float td[RT][4];                               // next pass's dense values between its request and its LDS store
        auto init_acc = [&](auto& acc, int boff, auto MG) {
            constexpr int MGc = decltype(MG)::value;
#pragma unroll
            for (int mg = 0; mg < MGc; ++mg) {
                float bv[16];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 t = *reinterpret_cast<const float4*>(cpar + boff + 64 * mg + 16 * g + 4 * qq);
                    bv[4 * qq] = t.x; bv[4 * qq + 1] = t.y; bv[4 * qq + 2] = t.z; bv[4 * qq + 3] = t.w;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < RT; ++nt) acc[4 * mg + mt][nt] = f32x4{bv[mt], bv[4 + mt], bv[8 + mt], bv[12 + mt]};
            }
        };
        // (the generic lambda is instantiated per layer; MI / MO = M-groups of its input / output)
        auto dense_layer = [&](auto& accin, auto& accout, auto MIc, auto MOc, auto PARKc) {
            constexpr int MI = decltype(MIc)::value, MO = decltype(MOc)::value;
            constexpr bool PARKED = decltype(PARKc)::value && MI > 1;      // accin's last M-group waits in LDS
#pragma unroll
            for (int mg = 0; mg < MI; ++mg) {
#pragma unroll
                for (int mg1 = 0; mg1 < MO; ++mg1) {
                    const bool first = sidx == 0;                  // no prefetch across the layer-0 boundary
                    const bool last = sidx == SL - 1;
                    if (last) CHAIN_TOP_ID();                      // the ids requested in the step before
                    else CHAIN_TOP();
                    const float* sb = slot_ptr(0);
                    if (last && NDB > 0) {                         // the next pass's dense values have landed
                        dense_store(0, td);
                        dense_rest(pass_n);
                    }
                    if (first) c0 = read_an(sb, 0);
                    if (DEEP && first) c1 = read_an(sb, 1);
                    if (PARKED && mg == MI - 1 && mg1 == 0) {
                        const f32x4* park = park_ptr();
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int nt = 0; nt < RT; ++nt) accin[4 * mg + mt][nt] = park[(mt * RT + nt) * 64];
                    }
#pragma unroll
                    for (int ks = 0; ks < 16; ks += 2) {
                        if (!DEEP) c1 = read_an(sb, ks + 1);
                        else if (ks + 2 < 16) c2 = read_an(sb, ks + 2);
                        else if (!last) c2 = read_an(slot_ptr(1), 0);
                        DCTR_SB;
#pragma unroll
                        for (int mt1 = 0; mt1 < 4; ++mt1)
#pragma unroll
                            for (int nt = 0; nt < RT; ++nt)
                                mfma_bi(accout[4 * mg1 + mt1][nt], c0[mt1], accin[4 * mg + (ks >> 2)][nt][ks & 3]);
                        DCTR_SB;
                        if (ks == PH) {
                            // the chunk after next, and at the pass's end the next pass's gather prologue
                            if (dma_early) dma_chunk(NB + sidx + 2, slot_ptr(2));
                            if (sidx == SL - 2) {
                                request_pair_ids(0, pass_n, idr_lo, idr_hi);
                                if constexpr (FPB > 1 && PPB > 1) request_pair_ids(1, pass_n, idrB_lo, idrB_hi);
                                if (NDB > 0) dense_request(0, pass_n, td);
                            }
                            if (last) {
                                idc = fold_pair_ids(0, pass_n, idr_lo, idr_hi);
                                if constexpr (FPB > 1) {
                                    if constexpr (PPB > 1) idcB = fold_pair_ids(1, pass_n, idrB_lo, idrB_hi);
                                    request_pair_ids(PPB, pass_n, idr_lo, idr_hi);
                                    if constexpr (PPB > 1) request_pair_ids(PPB + 1, pass_n, idrB_lo, idrB_hi);
                                    issue_xq(0, idc, idcB, XA);
                                } else {
                                    issue_x(0, idc, 0, XA);
                                }
                            }
                        }
                        if (ks == CHAIN_DMA_LATE && !dma_early) dma_chunk(NB + sidx + 2, slot_ptr(2));
                        if (!DEEP) {
                            if (ks + 2 < 16) c0 = read_an(sb, ks + 2);
                            else if (!last) c0 = read_an(slot_ptr(1), 0);
                        } else {
                            if (ks + 3 < 16) c3 = read_an(sb, ks + 3);
                            else if (!last) c3 = read_an(slot_ptr(1), 1);
                        }
                        DCTR_SB;
#pragma unroll
                        for (int mt1 = 0; mt1 < 4; ++mt1)
#pragma unroll
                            for (int nt = 0; nt < RT; ++nt)
                                mfma_bi(accout[4 * mg1 + mt1][nt], c1[mt1], accin[4 * mg + ((ks + 1) >> 2)][nt][(ks + 1) & 3]);
                        DCTR_SB;
                        if (DEEP && (ks + 2 < 16 || !last)) {
                            c0 = c2;
                            c1 = c3;
                        }
                    }
                    slot_next();
                    ++sidx;
                }
            }
        };
        f32x4 acc1[4 * M1][RT];
        init_acc(acc1, B1_OFF, std::integral_constant<int, M1>{});
        dense_layer(acc0, acc1, std::integral_constant<int, M0>{}, std::integral_constant<int, M1>{}, std::true_type{});
        mfma_drain();
        float hs[RT];
        // head of the last layer: act(acc) . head_w over this lane's 16 features per M-group, then over g
        auto head = [&](auto& acc, auto MGc, int layer, int bn_off) {
            constexpr int MG = decltype(MGc)::value;
            if (p.bn_scale[layer] != nullptr) bn_block<4 * MG, RT>(cpar + Off::BN_S + bn_off, cpar + Off::BN_T + bn_off, g, acc);
            act_block<4 * MG, RT, EXPACT>(p.activation, acc);
#pragma unroll
            for (int nt = 0; nt < RT; ++nt) hs[nt] = 0.f;
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) {
                float hw[16];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 t = *reinterpret_cast<const float4*>(cpar + HW_OFF + 64 * mg + 16 * g + 4 * qq);
                    hw[4 * qq] = t.x; hw[4 * qq + 1] = t.y; hw[4 * qq + 2] = t.z; hw[4 * qq + 3] = t.w;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nt = 0; nt < RT; ++nt)
                            hs[nt] = fmaf(acc[4 * mg + mt][nt][r], hw[4 * r + mt], hs[nt]);
            }
        };
        if constexpr (M2 > 0) {
            if (p.bn_scale[1] != nullptr) bn_block<4 * M1, RT>(cpar + Off::BN_S + 64 * M0, cpar + Off::BN_T + 64 * M0, g, acc1);
            act_block<4 * M1, RT, EXPACT>(p.activation, acc1);
            f32x4 acc2[4 * (M2 > 0 ? M2 : 1)][RT];
            init_acc(acc2, B2_OFF, std::integral_constant<int, M2>{});
            dense_layer(acc1, acc2, std::integral_constant<int, M1>{}, std::integral_constant<int, M2>{}, std::false_type{});
            mfma_drain();
            head(acc2, std::integral_constant<int, M2>{}, 2, 64 * (M0 + M1));
        } else {
            head(acc1, std::integral_constant<int, M1>{}, 1, 64 * M0);
        }
        // ---- Dense(1) + linear / FM logits + add[] + global bias, PredictionLayer
#pragma unroll
        for (int nt = 0; nt < RT; ++nt) {
            float v = hs[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            v += extras[nt];
            const int r = row_of(pass, nt);
            if (g == 0 && r < row_end) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p.add[i] != nullptr) v += p.add[i][r];
                v += cpar[GB_OFF];
                if (p.sigmoid_out) v = dctr::sigmoidf_(v);
                p.y[r] = v;
            }
        }
    }
#undef CHAIN_TOP_X
#undef CHAIN_TOP_ID
#undef CHAIN_TOP
#undef CHAIN_TOP_Q
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the wrapped-around DMA of the last pass must not outlive the phase
}

// RT, NW: launch shape of the main phase; TAIL: the kernel also carries the tail phase (64-row units: 4 waves x 16 rows)
template <int RT, int NW, int EB, bool I64, int M0, int M1, int M2, bool TAIL, bool CROSS = false, int FPB = 1, bool EXPACT = false, bool REC = false,
          bool POOL = false>
__global__ __launch_bounds__(64 * NW, 1) void chain_kernel(ChainParams p) {
    constexpr int NT = 64 * NW;
    typedef ChainOff<M0, M1, M2> Off;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cpar = smem + CPAR_OFF;
    float* fdesc = smem + FDESC_OFF;
    float* dlw = smem + DLW_OFF;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (p.probe != nullptr && threadIdx.x == 0) atomicMin(p.probe, (unsigned long long)wall_clock64());
    // ---- once per launch: descriptors, biases, head weights, BatchNormalization scale / shift, dense linear weights -> LDS
    for (int i = threadIdx.x; i < 12 * p.n_fields; i += NT)
        reinterpret_cast<uint32_t*>(fdesc)[i] = reinterpret_cast<const uint32_t*>(p.fields)[i];
    for (int i = threadIdx.x; i < 64 * M0; i += NT) cpar[i] = p.bias[0] != nullptr ? p.bias[0][i] : 0.f;
    for (int i = threadIdx.x; i < 64 * M1; i += NT) cpar[Off::B1 + i] = p.bias[1] != nullptr ? p.bias[1][i] : 0.f;
    if constexpr (M2 > 0)
        for (int i = threadIdx.x; i < 64 * M2; i += NT) cpar[Off::B2 + i] = p.bias[2] != nullptr ? p.bias[2][i] : 0.f;
    for (int i = threadIdx.x; i < 64 * Off::ML; i += NT) cpar[Off::HW + i] = p.head_w[i];
    if (threadIdx.x == 0) cpar[Off::GB] = p.global_bias != nullptr ? p.global_bias[0] : 0.f;
    {
        constexpr int MS[3] = {M0, M1, M2};
        int off = 0;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            if (MS[l] > 0 && p.bn_scale[l] != nullptr)
                for (int i = threadIdx.x; i < 64 * MS[l]; i += NT) {
                    cpar[Off::BN_S + off + i] = p.bn_scale[l][i];
                    cpar[Off::BN_T + off + i] = p.bn_shift[l] != nullptr ? p.bn_shift[l][i] : 0.f;
                }
            off += 64 * MS[l];
        }
    }
    for (int i = threadIdx.x; i < 256; i += NT) dlw[i] = (p.dense_lin_w != nullptr && i < p.n_dense) ? p.dense_lin_w[i] : 0.f;
    if constexpr (POOL) {       // the pooled sequences' descriptors (8 dwords each)
        for (int i = threadIdx.x; i < 8 * p.n_pool; i += NT)
            reinterpret_cast<uint32_t*>(cpar + POOLD_OFF)[i] = reinterpret_cast<const uint32_t*>(p.pool)[i];
    }
    if constexpr (CROSS) {
        // the cross vectors, zero-padded to whole k-blocks, and the row-independent constants of the recurrence
        float* xv = smem + p.xv_off;
        const int XVS = 16 * ((p.in_dim + 15) >> 4), L = p.cross_layers, d = p.in_dim;
        for (int i = threadIdx.x; i < CROSS_NV * XVS; i += NT) {
            const int v = i / XVS, k = i - v * XVS;
            xv[i] = (v <= L && k < d) ? (v < L ? p.cross_w[(size_t)v * d + k] : p.cross_head[k]) : 0.f;
        }
        if (p.cross_const != nullptr) {
            if (threadIdx.x < CROSS_NV) xv[CROSS_NV * XVS + threadIdx.x] = p.cross_const[threadIdx.x];
        } else if (wave == 0) {
            float cst[CROSS_NV];
            cross_constants(p.cross_w, p.cross_b, p.cross_head, L, d, lane, cst);
            if (lane == 0) {
#pragma unroll
                for (int v = 0; v < CROSS_NV; ++v) xv[CROSS_NV * XVS + v] = cst[v];
            }
        }
    }
    __syncthreads();                                   // LDS parameters written
    int oor = 0;
    const int main_end = (int)(TAIL ? p.main_rows : p.batch);
    chain_passes<RT, NW, EB, I64, M0, M1, M2, CROSS, FPB, EXPACT, REC, POOL>(p, smem, wave, lane, 0, main_end, (int)blockIdx.x, (int)gridDim.x, p.n_pass, oor);
    if constexpr (TAIL) {
        if (p.n_tail > 0) {
            // every wave is through with the ring and the staging areas of the main phase; waves 4.. leave (s_barrier waits for
            // the surviving waves of a workgroup only), waves 0-3 take the workgroup's 64-row units
            __syncthreads();
            if (wave < 4)
                chain_passes<1, 4, EB, I64, M0, M1, M2, CROSS, FPB, EXPACT, REC, POOL>(p, smem, wave, lane, main_end, (int)p.batch, (int)blockIdx.x,
                                                                         (int)gridDim.x, p.n_tail, oor);
        }
    }
    if (p.status != nullptr && oor && lane == 0) atomicOr(p.status, (int)DCTR_STATUS_INDEX_OOR);
    if (p.probe != nullptr && lane == 0) atomicMax(p.probe + 1, (unsigned long long)wall_clock64());
}


// launchers, one translation unit per launch shape and (units[0], units[1]) (chain_kernels_r{RT}w{NW}_m{M0}{M1}.hip); M2 = units[2] / 64
int launch_r2w8_m42(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);
int launch_r2w8_m41(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);
int launch_r2w8_m22(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);
int launch_r2w8_m21(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);
int launch_r2w4_m42(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);
int launch_r2w8_m42x(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // CROSS (chain_kernels_r2w8_m42_x.hip)
int launch_r2w8_m42q(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // embedding_dim 8 / 4 (chain_kernels_r2w8_m42_q.hip)
int launch_r2w8_m42t(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // sigmoid / tanh DNNs (chain_kernels_r2w8_m42_t.hip)
int launch_r2w8_m42w(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // embedding_dim 64 (chain_kernels_r2w8_m42_w.hip)
int launch_r2w8_m42r(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // record-form tables, embedding_dim 16 (chain_kernels_r2w8_m42_r.hip)
int launch_r2w8_m42p(const ChainParams& p, int E, int M2, unsigned blocks, hipStream_t stream);     // in-pass sequence pooling, embedding_dim 16 (chain_kernels_r2w8_m42_p.hip)

}  // namespace dctr_chain
