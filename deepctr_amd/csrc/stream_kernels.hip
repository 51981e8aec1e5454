// dctr_embed_mlp_fwd, streaming form — the throughput kernel of the DeepFM-family forward (reference
// deepctr/inputs.py:101-117 embedding_lookup, feature_column.py:171-210 linear logit, layers/interaction.py:588-604 FM,
// layers/core.py:189-208 DNN, :250-259 PredictionLayer) for launches that cover many batches' worth of rows.
//
// Why a second kernel.  mlp_kernel<2> (mlp_device.h) gives a workgroup 32 rows: every weight fragment feeds two row
// tiles, so a CU streams 603 KB of weights from L2 per 32 rows = 16 B/clk/CU at the MFMA rate — exactly the rate a
// CU can pull from L2 (16-21 B/clk measured), and the gather of a tile runs before its MFMAs, covered only by the
// other co-resident workgroup.  Here:
//   * PERSISTENT workgroups (one per CU) walk 64-row tiles: every weight fragment feeds FOUR row tiles -> 8 B/clk/CU;
//   * wave specialisation: 8 MFMA waves (two per SIMD) + 1 LOADER wave.  The loader turns ids into embedding-row
//     addresses and moves the rows HBM/MALL -> LDS with LDS-DMA (global_load_lds_dwordx4: one 1-KiB coalesced request
//     per wave instruction = 16 samples x 64 B of one field, no VGPR round trip), into a RING of three 16-KiB slots
//     (64 input columns x 64 rows each).  Its memory queue holds only gathers and the MFMA waves' queues only weight
//     loads: vmcnt retires in order, so one wave doing both would stall its weight pipeline on every gather;
//   * layer 0 is a K-loop over the ring: the DNN-input tile never exists whole anywhere (110 KB would not fit beside
//     the activations); the DMA image of a column block is [16 samples][16 floats], read conflict-free as the MFMA A
//     operand with ONE ds_read_b128 per row tile and column block (k-slot g takes columns 16*blk + 4*g + t: the K
//     order is permuted, which an fp32 fmaf chain does not care about; the weight rows are fetched in the same order);
//   * while the MFMA waves run layers 1.., the head and the first column blocks of layer 0, the loader is already
//     filling the ring for the NEXT tile: the gather latency is off the critical path;
//   * the loader also owns the FM / linear / dense-passthrough work (FM partial sums from the landed LDS image,
//     4-byte linear rows gathered row-per-lane) — the MFMA waves execute MFMAs, their operand loads and epilogues only.
// Synchronisation is by monotonic counters in LDS (ready / freed per ring slot, an 8-wave barrier counter): s_barrier
// would tie the loader to the MFMA waves' phases.  Every wait is bounded; on a timeout the workgroup sets an abort word
// (all later waits fall through) and ORs DCTR_STATUS_TIMEOUT into the status word: a bug shows up as an error, never
// as a hung GPU.
//
// Eligibility (host, mlp_kernels.hip): uniform embedding_dim E with E % 16 == 0 and out_offset = field * E, no hashing /
// identity fields, dense columns right behind the embeddings (all copied), units[0] <= 256, a head.  Everything else
// takes mlp_kernel.  Same arithmetic: v_mfma_f32_16x16x4_f32 = exact fp32.
#include "mlp_device.h"


namespace dctr_stream {

using dctr::f32x4;
using namespace dctr_mlp;

constexpr int NCONS = 8;                       // MFMA waves
constexpr int NLOAD = 4;                       // loader waves, one per row group of 16 samples
constexpr int NTHREADS = 64 * (NCONS + NLOAD);
constexpr int ROWS = 64;
constexpr int NSLOT = 3;
constexpr int SLOT_F = 4096;                   // floats per ring slot: 4 column blocks x 4 row groups x 256
constexpr int SPIN_CHECK = 1 << 10;            // spins between looks at the wall clock
constexpr unsigned long long WAIT_LIMIT_TICKS = 200000000ull;   // 2 s of the 100 MHz constant clock: a hang, not a memory stall
#ifndef DCTR_STREAM_DMA_AUX
#define DCTR_STREAM_DMA_AUX 0                  // cache policy bits of the gather DMA (lab: 2 = nt, 16 = sc1)
#endif

enum { S_READY = 0, S_FREED = 4, S_BAR = 8, S_XREADY = 9, S_XDONE = 10, S_ABORT = 11, S_WORDS = 16 };

struct StreamParams {
    const dctr_field_t* fields;
    const void* ids;
    int64_t ids_stride_f, ids_stride_b;
    int32_t ids_is_i64, n_fields, dim, n_dense;
    const float* dense;
    int64_t dense_stride;
    const float* dense_lin_w;
    int64_t batch;
    float* fm_logit;
    float* lin_logit;
    int32_t* status;
    int32_t fm_used, lin_used;
    int32_t in_dim, n_layers;
    int32_t units[MAX_LAYERS];
    const float* W[MAX_LAYERS];
    const float* bias[MAX_LAYERS];
    int32_t activation, sigmoid_out;
    const float* head_w;
    const float* add[4];
    const float* global_bias;
    float* y;
    unsigned long long* probe;
    int32_t extras_off, fdesc_off, cpar_off, hpart_off, ring_off, act_off[2];  // LDS layout, float offsets
    int32_t bias_off[MAX_LAYERS];              // biases of layer l at cpar_off + bias_off[l] (-1: none); head_w at + headw_off
    int32_t headw_off;
    int32_t n_tiles;
};

// ---------------------------------------------------------------------------------------------------
// LDS counters
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sync_load(int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// wait until *word >= target (wrap-safe), bounded; wave-uniform
__device__ __forceinline__ void wait_ge(int* sync, int word, int target, int32_t* status) {
    asm volatile("" ::: "memory");
    int spins = 0;
    unsigned long long t0 = 0;
    for (;;) {
        const int v = __builtin_amdgcn_readfirstlane(sync_load(sync + word));
        if (v - target >= 0) break;
        if (__builtin_amdgcn_readfirstlane(sync_load(sync + S_ABORT)) != 0) break;
        // bounded by WALL-CLOCK time (a long but legitimate stall — contended HBM, first-touch page faults, a shared GPU — must
        // not discard valid work): the clock is read every SPIN_CHECK spins only
        bool expired = false;
        if ((++spins & (SPIN_CHECK - 1)) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            expired = now - t0 > WAIT_LIMIT_TICKS;
        }
        if (expired) {
            if ((threadIdx.x & 63) == 0) {
                __hip_atomic_store(sync + S_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (status != nullptr) atomicOr(status, (int)DCTR_STATUS_TIMEOUT);
            }
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

// one arrival; LDS executes a wave's operations in order, so everything this wave did to LDS before is visible to
// whoever observes the count
__device__ __forceinline__ void sig_add(int* sync, int word) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(sync + word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void sig_set(int* sync, int word, int v) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(sync + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// barrier of the 8 MFMA waves
__device__ __forceinline__ void cons_barrier(int* sync, int& epoch, int32_t* status) {
    ++epoch;
    sig_add(sync, S_BAR);
    wait_ge(sync, S_BAR, NCONS * epoch, status);
}

// ---------------------------------------------------------------------------------------------------
// MFMA waves
// ---------------------------------------------------------------------------------------------------
// layer 0 over the ring.  Column block b (16 input columns) = one pipeline stage: 4 A reads (one ds_read_b128 per row
// tile), 4 weight loads (k-steps t = 0..3, slot g <-> weight row 16*b + 4*g + t), 16*TPW MFMAs.  Three register stages
// rotate as in tile_gemm_pipe.  The buffer descriptor is rebuilt per block (scalar ALU) with base = W + 16*b rows and
// num_records = the rows left, so the K tail is cut by the hardware bounds check on the per-lane offset.
template <int TPW>
__device__ __forceinline__ void l0_stream(const StreamParams& p, const float* ring, int* sync, int seq0, int NB,
                                          int n_base, f32x4 (&acc)[4][TPW], int tid) {
    const int lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int N = p.units[0];
    int n0 = n_base + TPW * j;
    if (n0 + TPW > N) n0 = N - TPW;
    int voff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) voff[t] = ((4 * g + t) * N + n0) * 4;
    const int blk_bytes = 16 * N * 4;
    const int w_bytes = p.in_dim * N * 4;
    // DMA image of a (column block, row group): sample s at 64*s bytes, its 16-B piece r at 16 * (r ^ ((s >> 2) & 2)) —
    // the loader swaps pieces 0<->2, 1<->3 for samples 8-15 so that every 16-lane group of this ds_read_b128 covers all
    // 64 banks once (with the plain order lanes j and j+8 of a group would collide: 2-way)
    const float* abase = ring + j * 16 + ((g ^ ((j >> 2) & 2)) << 2);
    const int b_last = NB - 1;
    float a0[4][4], a1[4][4], a2[4][4];
    float b0[4][TPW], b1[4][TPW], b2[4][TPW];
#define L0_LOAD(BI, AR, BR)                                                                              \
    {   /* loads past the last block are clamped, not skipped: a conditional load makes hipcc's vmcnt bookkeeping   \
           assume the shortest queue on every path and wait for the loads it has just issued */                    \
        const int b_ = min((BI), b_last);                                                                \
        const int sq_ = seq0 + (b_ >> 2);                                                                \
        const int sl_ = sq_ % NSLOT;                                                                     \
        if ((b_ & 3) == 0) wait_ge(sync, S_READY + sl_, NLOAD * (sq_ / NSLOT + 1), p.status);            \
        const float* ap_ = abase + sl_ * SLOT_F + (b_ & 3) * 1024;                                       \
        _Pragma("unroll") for (int rt_ = 0; rt_ < 4; ++rt_) {                                            \
            const float4 v_ = *reinterpret_cast<const float4*>(ap_ + rt_ * 256);                         \
            AR[rt_][0] = v_.x; AR[rt_][1] = v_.y; AR[rt_][2] = v_.z; AR[rt_][3] = v_.w;                  \
        }                                                                                                \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                            \
            const_cast<char*>(reinterpret_cast<const char*>(p.W[0])) + (size_t)b_ * blk_bytes, 0,        \
            w_bytes - b_ * blk_bytes, 0x00020000);                                                       \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) buf_load_cols<TPW>(rs_, voff[t_], 0, BR[t_]);   \
    }
#define L0_MFMA(BI, AR, BR)                                                                              \
    do {                                                                                                 \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                                 \
            _Pragma("unroll") for (int rt_ = 0; rt_ < 4; ++rt_)                                          \
                _Pragma("unroll") for (int c_ = 0; c_ < TPW; ++c_)                                       \
                    acc[rt_][c_] = __builtin_amdgcn_mfma_f32_16x16x4f32(AR[rt_][t_], BR[t_][c_], acc[rt_][c_], 0, 0, 0); \
        if ((((BI) & 3) == 3) || (BI) == b_last) sig_add(sync, S_FREED + (seq0 + ((BI) >> 2)) % NSLOT);  \
    } while (0)
    L0_LOAD(0, a0, b0);
    L0_LOAD(1, a1, b1);
    for (int b = 0; b < NB; b += 3) {
        L0_LOAD(b + 2, a2, b2);
        DCTR_SB;
        L0_MFMA(b, a0, b0);
        DCTR_SB;
        L0_LOAD(b + 3, a0, b0);
        DCTR_SB;
        if (b + 1 < NB) L0_MFMA(b + 1, a1, b1);
        DCTR_SB;
        L0_LOAD(b + 4, a1, b1);
        DCTR_SB;
        if (b + 2 < NB) L0_MFMA(b + 2, a2, b2);
        DCTR_SB;
    }
#undef L0_LOAD
#undef L0_MFMA
}

// a wave with no layer-0 column tile still takes part in the ring protocol
__device__ __forceinline__ void l0_idle(const StreamParams& p, int* sync, int seq0, int NCH) {
    for (int c = 0; c < NCH; ++c) {
        const int sq = seq0 + c;
        wait_ge(sync, S_READY + sq % NSLOT, NLOAD * (sq / NSLOT + 1), p.status);
        sig_add(sync, S_FREED + sq % NSLOT);
    }
}

// bias + activation of RTL row tiles x TPW column tiles -> the next layer's LDS tile (column-permuted, stride lda)
template <int TPW, int ACT, int RTL>
__device__ __forceinline__ void epilogue(const float* bias, float* out, int lda, int N, int n_base,
                                         const f32x4 (&acc)[RTL][TPW], int tid) {
    const int lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int KQn = pad64(N) / 4;
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int n = n_base + TPW * j + c;
        if (n < N) {
            const float bv = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RTL; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(rt * 16 + 4 * g + r) * lda + lds_pos(n, KQn)] = act_t<ACT>(acc[rt][c][r] + bv, 0.f, 0.f, 1.f, 0.f);
        }
    }
}

// LAST layer: bias + activation, then straight into the Dense(1) head — partial dot products of this wave's 16*TPW columns
// for its 16*RTL rows go to hpart[column tile][row]; the activations of the last layer are never stored
template <int TPW, int ACT, int RTL>
__device__ __forceinline__ void epilogue_head(const float* bias, const float* headw, float* hpart_rows, int N, int n_base,
                                              const f32x4 (&acc)[RTL][TPW], int tid) {
    const int lane = tid & 63, g = lane >> 4, j = lane & 15;
    float bv[TPW], hw[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int n = n_base + TPW * j + c;
        bv[c] = (n < N && bias != nullptr) ? bias[n] : 0.f;
        hw[c] = n < N ? headw[n] : 0.f;
    }
#pragma unroll
    for (int rt = 0; rt < RTL; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < TPW; ++c) s = fmaf(act_t<ACT>(acc[rt][c][r] + bv[c], 0.f, 0.f, 1.f, 0.f), hw[c], s);
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
            if (j == 0) hpart_rows[rt * 16 + 4 * g + r] = s;
        }
}

template <int TPW, int RTL>
__device__ __forceinline__ void epilogue_head_act(int act, const float* bias, const float* headw, float* hpart_rows, int N,
                                                  int n_base, const f32x4 (&acc)[RTL][TPW], int tid) {
    switch (act) {
        case DCTR_ACT_RELU: epilogue_head<TPW, DCTR_ACT_RELU, RTL>(bias, headw, hpart_rows, N, n_base, acc, tid); break;
        case DCTR_ACT_SIGMOID: epilogue_head<TPW, DCTR_ACT_SIGMOID, RTL>(bias, headw, hpart_rows, N, n_base, acc, tid); break;
        case DCTR_ACT_TANH: epilogue_head<TPW, DCTR_ACT_TANH, RTL>(bias, headw, hpart_rows, N, n_base, acc, tid); break;
        default: epilogue_head<TPW, DCTR_ACT_LINEAR, RTL>(bias, headw, hpart_rows, N, n_base, acc, tid); break;
    }
}

template <int TPW, int RTL>
__device__ __forceinline__ void epilogue_act(int act, const float* bias, float* out, int lda, int N, int n_base,
                                             const f32x4 (&acc)[RTL][TPW], int tid) {
    switch (act) {
        case DCTR_ACT_RELU: epilogue<TPW, DCTR_ACT_RELU, RTL>(bias, out, lda, N, n_base, acc, tid); break;
        case DCTR_ACT_SIGMOID: epilogue<TPW, DCTR_ACT_SIGMOID, RTL>(bias, out, lda, N, n_base, acc, tid); break;
        case DCTR_ACT_TANH: epilogue<TPW, DCTR_ACT_TANH, RTL>(bias, out, lda, N, n_base, acc, tid); break;
        default: epilogue<TPW, DCTR_ACT_LINEAR, RTL>(bias, out, lda, N, n_base, acc, tid); break;
    }
}

// columns [N, pad64(N)) of a layer output are K padding of the next layer: zero them (threads of the MFMA waves)
__device__ __forceinline__ void zero_pad_cols(float* out, int lda, int N, int tid) {
    const int npad = pad64(N) - N;
    const int KQn = pad64(N) / 4;
    if (npad > 0) {
        for (int i = tid; i < ROWS * 64; i += 64 * NCONS) {
            const int r = i >> 6, c = i & 63;
            if (c < npad) out[r * lda + lds_pos(N + c, KQn)] = 0.f;
        }
    }
}

// layers >= 1: activations in LDS (column-permuted, as mlp_kernel keeps them), weights from L2.  Work units are
// (column tile, row part): RTL = 4 -> one unit per column tile covering all 64 rows; RTL = 2 -> two units per column
// tile (rows 0-31 / 32-63), used when there are too few column tiles to occupy the 8 waves.
template <int TPW, int RTL>
__device__ __forceinline__ void layer_units(const StreamParams& p, int l, const float* in, int lda_in, float* out,
                                            int lda_out, int K, int N, int wave, int tid, const float* cpar, float* hpart) {
    constexpr int PARTS = 4 / RTL;
    const float* bias = p.bias_off[l] >= 0 ? cpar + p.bias_off[l] : nullptr;
    const bool last = l + 1 == p.n_layers;
    const int n_ct = (N + 16 * TPW - 1) / (16 * TPW);
    for (int u = wave; u < n_ct * PARTS; u += NCONS) {
        const int ct = u / PARTS, part = u % PARTS;
        const int n_base = ct * 16 * TPW;
        f32x4 acc[RTL][TPW];
#pragma unroll
        for (int rt = 0; rt < RTL; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm_pipe<TPW, RTL, 4>(in + part * 16 * RTL * lda_in, lda_in, pad64(K) / 4, K, p.W[l], N, n_base, acc, tid & 63);
        if (last) epilogue_head_act<TPW, RTL>(p.activation, bias, cpar + p.headw_off, hpart + ct * ROWS + part * 16 * RTL, N, n_base, acc, tid);
        else epilogue_act<TPW, RTL>(p.activation, bias, out + part * 16 * RTL * lda_out, lda_out, N, n_base, acc, tid);
    }
}

__device__ __forceinline__ void consumer(const StreamParams& p, float* smem, int wave) {
    int* sync = reinterpret_cast<int*>(smem);
    const float* ring = smem + p.ring_off;
    const float* extras = smem + p.extras_off;
    const float* cpar = smem + p.cpar_off;        // biases + head weights, copied once per launch
    const int NB = (p.in_dim + 15) >> 4;          // column blocks of the DNN input
    const int NCH = (NB + 3) >> 2;                // ring chunks per tile
    const int N0 = p.units[0];
    const bool wide0 = N0 % 32 == 0 && N0 > 16 * NCONS;
    const int n_base0 = wave * (wide0 ? 32 : 16);
    int seq0 = 0, epoch = 0;
    for (int it = 0, tile = blockIdx.x; tile < p.n_tiles; ++it, tile += gridDim.x) {
        const int64_t b0 = (int64_t)tile * ROWS;
        // the thread index is made opaque once per tile: otherwise every lane-derived LDS address of the epilogues / head
        // is hoisted out of this persistent loop, stays live across the MFMA loops and spills (~60 VGPRs)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // ---- layer 0: stream the ring
        float* out = smem + p.act_off[0];
        int lda_out = pad64(N0) + 4;
        float* hpart = smem + p.hpart_off + (it & 1) * 8 * ROWS;     // [column tile <= 8][row] head partials of this tile
        const float* bias0 = p.bias_off[0] >= 0 ? cpar + p.bias_off[0] : nullptr;
        const bool single = p.n_layers == 1;
        if (n_base0 < N0) {
            if (wide0) {
                f32x4 acc[4][2];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) { acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                l0_stream<2>(p, ring, sync, seq0, NB, n_base0, acc, tid);
                if (single) epilogue_head_act<2, 4>(p.activation, bias0, cpar + p.headw_off, hpart + wave * ROWS, N0, n_base0, acc, tid);
                else epilogue_act<2, 4>(p.activation, bias0, out, lda_out, N0, n_base0, acc, tid);
            } else {
                f32x4 acc[4][1];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                l0_stream<1>(p, ring, sync, seq0, NB, n_base0, acc, tid);
                if (single) epilogue_head_act<1, 4>(p.activation, bias0, cpar + p.headw_off, hpart + wave * ROWS, N0, n_base0, acc, tid);
                else epilogue_act<1, 4>(p.activation, bias0, out, lda_out, N0, n_base0, acc, tid);
            }
        } else {
            l0_idle(p, sync, seq0, NCH);
        }
        if (!single) zero_pad_cols(out, lda_out, N0, tid);
        cons_barrier(sync, epoch, p.status);
        // ---- layers 1..
        const float* in = out;
        int lda_in = lda_out, K = N0;
        int n_ct_last = wide0 ? (N0 + 31) / 32 : (N0 + 15) / 16;
        for (int l = 1; l < p.n_layers; ++l) {
            const int N = p.units[l];
            out = smem + p.act_off[l & 1];
            lda_out = pad64(N) + 4;
            const int n_ct16 = (N + 15) / 16;
            if (N % 32 == 0 && N >= 32 * NCONS) layer_units<2, 4>(p, l, in, lda_in, out, lda_out, K, N, wave, tid, cpar, hpart);
            else if (n_ct16 > NCONS / 2) layer_units<1, 4>(p, l, in, lda_in, out, lda_out, K, N, wave, tid, cpar, hpart);
            else layer_units<1, 2>(p, l, in, lda_in, out, lda_out, K, N, wave, tid, cpar, hpart);
            n_ct_last = (N % 32 == 0 && N >= 32 * NCONS) ? N / 32 : n_ct16;
            if (l + 1 < p.n_layers) zero_pad_cols(out, lda_out, N, tid);
            cons_barrier(sync, epoch, p.status);
            in = out;
            lda_in = lda_out;
            K = N;
        }
        // ---- head: the last layer's epilogue left partial dot products with head_w per column tile; after that layer's
        // barrier wave 0 adds them up with the loader's FM / linear logits, add[], the global bias, and applies the sigmoid.
        // (hpart and extras are double-buffered by tile parity: the other waves are already in the next tile's layer 0)
        if (wave == 0) {
            wait_ge(sync, S_XREADY, NLOAD * (it + 1), p.status);
            const int row = tid & 63;
            float v = extras[(it & 1) * ROWS + row];
            for (int ct = 0; ct < n_ct_last; ++ct) v += hpart[ct * ROWS + row];
            const int64_t b = b0 + row;
            if (b < p.batch) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (p.add[i] != nullptr) v += p.add[i][b];
                if (p.global_bias != nullptr) v += p.global_bias[0];
                if (p.sigmoid_out) v = dctr::sigmoidf_(v);
                p.y[b] = v;
            }
            sig_set(sync, S_XDONE, it + 1);
        }
        seq0 += NCH;
    }
}

// ---------------------------------------------------------------------------------------------------
// loader waves
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// broadcast lane (quad base + K) of every quad: one DPP move, no LDS traffic
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xf, 0xf, false);
}

// Loader wave w owns ROW GROUP w (samples 16*w .. 16*w + 15 of the tile) for every column block: its FM / linear sums
// are complete per sample, so the four loader waves never exchange anything.  Lane (s, q): sample s of the group;
// for the DMA q is the 16-B piece of the row (swapped for samples 8-15, see l0_stream), for the id / linear loads q is
// the column block of the chunk (ids of 4 blocks arrive with ONE load, a quad broadcast hands block k's id to the
// quad).  Per chunk and wave: 1 id load, 4 DMA instructions, <= 4 sixteen-lane linear gathers, 4 ds_read_b128 + ~40 VALU of FM
// sums.  (One loader wave doing all four row groups with row-per-lane ids + ds_bpermute was instruction-bound: ~12k cycles
// per chunk against the 8k cycles the MFMA waves need to consume one.)
// EB = embedding_dim / 16 column blocks per field.  An id outside [0, vocabulary) reads row 0 and raises
// DCTR_STATUS_INDEX_OOR (the host turns that into the reference's IndexError; the launch's outputs are then void).
// field descriptors live in LDS for the whole launch (copied once by stream_kernel): the loaders read them with uniform
// ds_reads.  Scalar loads of the descriptor array (s_load through the constant cache) cost a memory round trip per column
// block in this kernel — 10-14k cycles per chunk, measured — whenever the scalar cache had dropped the lines.
struct FieldLds {
    const float* table;
    const float* lin_table;
    int64_t vocab;
    int in_fm;
};
__device__ __forceinline__ FieldLds field_lds(const float* fdesc, int f) {
    const uint4 a = *reinterpret_cast<const uint4*>(fdesc + 12 * f);          // table, lin_table
    const uint2 b = *reinterpret_cast<const uint2*>(fdesc + 12 * f + 4);      // vocab
    const uint32_t c = *reinterpret_cast<const uint32_t*>(fdesc + 12 * f + 8);  // in_fm
    FieldLds r;
    r.table = reinterpret_cast<const float*>(((uint64_t)__builtin_amdgcn_readfirstlane((int)a.y) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)a.x));
    r.lin_table = reinterpret_cast<const float*>(((uint64_t)__builtin_amdgcn_readfirstlane((int)a.w) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)a.z));
    r.vocab = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((int)b.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b.x));
    r.in_fm = __builtin_amdgcn_readfirstlane((int)c);
    return r;
}

template <int EB>
__device__ __forceinline__ void loader(const StreamParams& p, float* smem, int w) {
    int* sync = reinterpret_cast<int*>(smem);
    float* ring = smem + p.ring_off;
    float* extras = smem + p.extras_off;
    const int lane = threadIdx.x & 63;
    const int s = lane >> 2, q = lane & 3;
    const int piece = q ^ ((s >> 2) & 2);
    const int E = 16 * EB;
    const int NBE = p.n_fields * EB;               // embedding column blocks
    const int NB = (p.in_dim + 15) >> 4;
    const int NCH = (NB + 3) >> 2;
    const float* fdesc = smem + p.fdesc_off;
    int seq0 = 0;
    int oor = 0;
    // id of column block 4*c + q's field for this lane's sample; the NEXT chunk's ids are requested while the current
    // chunk's DMA is in flight (ids are streamed from HBM once: their round trip must not sit in front of every gather)
    auto request_ids = [&](int tile_n, int c_n) -> RawId {
        const int64_t brow_n = min((int64_t)tile_n * ROWS + 16 * w + s, p.batch - 1);
        const int f = min((4 * c_n + q) / EB, p.n_fields - 1);
        return load_id(p.ids, (int64_t)f * p.ids_stride_f + brow_n * p.ids_stride_b, p.ids_is_i64);
    };
    RawId nid = request_ids(min((int)blockIdx.x, p.n_tiles - 1), 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nid.lo), "+v"(nid.hi) : : "memory");
    for (int it = 0, tile = blockIdx.x; tile < p.n_tiles; ++it, tile += gridDim.x) {
        const int64_t b0 = (int64_t)tile * ROWS + 16 * w;
        const int64_t brow = min(b0 + s, p.batch - 1);
        const bool row_live = b0 + s < p.batch;
        float sum[EB][4], sq = 0.f, lin = 0.f, dlin = 0.f;
#pragma unroll
        for (int h = 0; h < EB; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[h][e] = 0.f;
        for (int c = 0; c < NCH; ++c) {
            const int sq_ = seq0 + c;
            const int slot = sq_ % NSLOT;
            wait_ge(sync, S_FREED + slot, NCONS * (sq_ / NSLOT), p.status);
            float* sbase = ring + slot * SLOT_F + w * 256;             // + k * 1024: block k, this wave's row group
            const RawId cid = nid;
            float lv[4];
            int fm_on[4] = {0, 0, 0, 0};
#define DCTR_LD_BLOCK(K)                                                                                          \
            {                                                                                                         \
                const int cb = 4 * c + K;                                                                             \
                lv[K] = 0.f;                                                                                          \
                if (cb < NBE) {                                                                                       \
                    const int f = cb / EB, h = cb % EB;                                                               \
                    const FieldLds fd = field_lds(fdesc, f);                                                          \
                    const float* table = fd.table;                                                                    \
                    const float* lin_table = fd.lin_table;                                                            \
                    const int64_t vocab = fd.vocab;                                                                   \
                    fm_on[K] = fd.in_fm;                                                                              \
                    RawId r;                                                                                          \
                    r.lo = quad_bcast<K>(cid.lo);                                                                     \
                    r.hi = quad_bcast<K>(cid.hi);                                                                     \
                    int64_t idk = id_value(r, p.ids_is_i64);                                                          \
                    const bool ok = (uint64_t)idk < (uint64_t)vocab;                                                  \
                    if (row_live && !ok) oor = 1;                                                                     \
                    idk = ok ? idk : 0;                                                                               \
                    const float* src = table + idk * E + h * 16 + piece * 4;                                          \
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sbase + K * 1024), 16, 0, DCTR_STREAM_DMA_AUX); \
                    if (h == 0 && lin_table != nullptr && q == K) lv[K] = lin_table[idk];             \
                } else if (cb < NB) {                                                                                 \
                    /* dense passthrough block: lane (s, q) supplies floats 4*piece .. 4*piece + 3 of its sample */   \
                    const int d0 = (cb - NBE) * 16 + 4 * piece;                                                       \
                    const float* src = p.dense + brow * p.dense_stride;                                               \
                    float x[4];                                                                                       \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                        const int m = min(d0 + e, p.n_dense - 1);                                                     \
                        const bool real = d0 + e < p.n_dense;                                                         \
                        const float xv = src[m];                                                                      \
                        const float wv = p.dense_lin_w != nullptr ? p.dense_lin_w[m] : 0.f;                           \
                        x[e] = real ? xv : 0.f;                                                                       \
                        dlin = fmaf(x[e], wv, dlin);                                                                  \
                    }                                                                                                 \
                    *reinterpret_cast<float4*>(sbase + K * 1024 + 4 * lane) = make_float4(x[0], x[1], x[2], x[3]);    \
                }                                                                                                     \
            }
            DCTR_LD_BLOCK(0)
            DCTR_LD_BLOCK(1)
            DCTR_LD_BLOCK(2)
            DCTR_LD_BLOCK(3)
#undef DCTR_LD_BLOCK
            {   // ids of the next chunk (of this tile, or the first of this workgroup's next tile)
                const bool last_c = c + 1 == NCH;
                const int tile_n = last_c ? min(tile + (int)gridDim.x, p.n_tiles - 1) : tile;
                nid = request_ids(tile_n, last_c ? 0 : c + 1);
            }
            // the chunk's DMA has landed (and the next ids, the linear rows).  The registers are named as operands so that
            // hipcc knows they are valid from here on: it cannot see an asm wait, and would otherwise put its own
            // vmcnt(0) in front of every later use — i.e. between the DMA instructions of the next chunk
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(nid.lo), "+v"(nid.hi), "+v"(lv[0]), "+v"(lv[1]), "+v"(lv[2]), "+v"(lv[3])
                         :
                         : "memory");
            // FM partial sums from the landed image: this lane's 16 B of block k are floats 4*piece.. of sample s
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cb = 4 * c + k;
                if (cb < NBE) {
                    const int h = cb % EB;
                    lin += lv[k];
                    if (fm_on[k]) {
                        const float4 v = *reinterpret_cast<const float4*>(sbase + k * 1024 + 4 * lane);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
#pragma unroll
                            for (int hh = 0; hh < EB; ++hh)
                                if (hh == h) sum[hh][e] += vv[e];
                            sq = fmaf(vv[e], vv[e], sq);
                        }
                    }
                }
            }
            sig_add(sync, S_READY + slot);
        }
        // ---- per-row logits of the gather epilogue: FM = 0.5 * (sum_d (sum_f e)^2 - sum_{f,d} e^2), linear
        float fm = -sq;
#pragma unroll
        for (int h = 0; h < EB; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) fm = fmaf(sum[h][e], sum[h][e], fm);
        // quad reductions, each term on its own: the piece <-> lane assignment differs between samples 0-7 and 8-15, and only
        // sums of the SAME four partials in a symmetric tree are independent of it (a row's result must not depend on where
        // in a tile the row sits: permuting the rows of a launch permutes its outputs bit for bit)
        fm += __shfl_xor(fm, 1, 64);
        fm += __shfl_xor(fm, 2, 64);
        fm *= 0.5f;
        lin += __shfl_xor(lin, 1, 64);
        lin += __shfl_xor(lin, 2, 64);
        dlin += __shfl_xor(dlin, 1, 64);
        dlin += __shfl_xor(dlin, 2, 64);
        const float lin_all = lin + dlin;
        wait_ge(sync, S_XDONE, it - 1, p.status);                        // the head of tile it-2 has read its extras
        if (q == 0) {
            extras[(it & 1) * ROWS + 16 * w + s] = (p.fm_used ? fm : 0.f) + (p.lin_used ? lin_all : 0.f);
            if (row_live) {
                if (p.fm_logit != nullptr) p.fm_logit[b0 + s] = fm;
                if (p.lin_logit != nullptr) p.lin_logit[b0 + s] = lin_all;
            }
        }
        sig_add(sync, S_XREADY);
        seq0 += NCH;
    }
    if (p.status != nullptr && __any(oor) && lane == 0) atomicOr(p.status, (int)DCTR_STATUS_INDEX_OOR);
}

__global__ __launch_bounds__(NTHREADS) void stream_kernel(StreamParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (p.probe != nullptr && threadIdx.x == 0) atomicMin(p.probe, (unsigned long long)wall_clock64());
    if (threadIdx.x < S_WORDS) reinterpret_cast<int*>(smem)[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 12 * p.n_fields; i += NTHREADS)
        reinterpret_cast<uint32_t*>(smem + p.fdesc_off)[i] = reinterpret_cast<const uint32_t*>(p.fields)[i];
    {
        float* cpar = smem + p.cpar_off;
        for (int l = 0; l < p.n_layers; ++l)
            if (p.bias_off[l] >= 0)
                for (int i = threadIdx.x; i < p.units[l]; i += NTHREADS) cpar[p.bias_off[l] + i] = p.bias[l][i];
        for (int i = threadIdx.x; i < p.units[p.n_layers - 1]; i += NTHREADS) cpar[p.headw_off + i] = p.head_w[i];
    }
    __syncthreads();
    if (wave < NCONS) {
        consumer(p, smem, wave);
    } else {
        if (p.dim == 16) loader<1>(p, smem, wave - NCONS);
        else if (p.dim == 32) loader<2>(p, smem, wave - NCONS);
        else loader<4>(p, smem, wave - NCONS);
    }
    if (p.probe != nullptr && (threadIdx.x & 63) == 0) atomicMax(p.probe + 1, (unsigned long long)wall_clock64());
}

// ---------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------
static int n_cus() { return dctr_n_cus(); }

// 1: the streaming kernel can take this call (the same conditions try_launch applies while it marshals); 0: not eligible
int eligible(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, bool forced) {
    const int E = g->uniform_dim;
    if (E != 16 && E != 32 && E != 64) return 0;
    if (g->any_hash || g->any_identity || g->any_pitch || !a->has_head || a->n_layers < 1 || a->save_acts != nullptr || a->cross_layers > 0) return 0;
    if (a->activation == DCTR_ACT_DICE || a->bn_scale != nullptr) return 0;
    if (a->units[0] > 256 || a->units[0] < 16) return 0;
    if (g->n_dense > 0 && (g->dense_out_offset != g->n_fields * E || g->dense_copy_cols != g->n_dense)) return 0;
    if (a->in_dim != g->n_fields * E + (g->n_dense > 0 ? g->n_dense : 0)) return 0;
    if ((int64_t)a->in_dim * a->units[0] * 4 >= (1LL << 31)) return 0;
    const int64_t n_tiles = dctr_ceil_div(a->batch, (int64_t)ROWS);
    if (!forced && n_tiles < n_cus()) return 0;
    if (n_tiles > 0x7fffffffLL / 64 || g->n_fields > 64) return 0;
    size_t act[2] = {0, 0};
    int off = 0;
    for (int l = 0; l < a->n_layers; ++l) {
        const size_t need = (size_t)ROWS * (((a->units[l] + 63) & ~63) + 4);
        if (need > act[l & 1]) act[l & 1] = need;
        off += a->biases[l] != nullptr ? a->units[l] : 0;
    }
    const int nl = a->units[a->n_layers - 1];
    if (off + nl > 1024) return 0;
    const bool wide = a->n_layers == 1 ? (nl % 32 == 0 && nl > 16 * NCONS) : (nl % 32 == 0 && nl >= 32 * NCONS);
    if ((wide ? (nl + 31) / 32 : (nl + 15) / 16) > 8) return 0;
    if (((size_t)3072 + NSLOT * SLOT_F + act[0] + act[1]) * sizeof(float) > 160 * 1024) return 0;
    return 1;
}

// 0: not eligible (caller falls back to mlp_kernel); 1: launched; < 0 / > 1: error code
int try_launch(const dctr_mlp_args_t* a, const dctr_gather_fm_args_t* g, int fm_used, int lin_used, bool forced,
               hipStream_t stream, int* rc) {
    *rc = DCTR_OK;
    if (!eligible(a, g, forced)) return 0;
    const int E = g->uniform_dim;
    if (E != 16 && E != 32 && E != 64) return 0;
    if (g->any_hash || g->any_identity || g->any_pitch || !a->has_head || a->n_layers < 1 || a->save_acts != nullptr || a->cross_layers > 0) return 0;
    if (a->activation == DCTR_ACT_DICE || a->bn_scale != nullptr) return 0;
    if (a->units[0] > 256 || a->units[0] < 16) return 0;
    if (g->n_dense > 0 && (g->dense_out_offset != g->n_fields * E || g->dense_copy_cols != g->n_dense)) return 0;
    if (a->in_dim != g->n_fields * E + (g->n_dense > 0 ? g->n_dense : 0)) return 0;
    if ((int64_t)a->in_dim * a->units[0] * 4 >= (1LL << 31)) return 0;
    const int64_t n_tiles = dctr_ceil_div(a->batch, (int64_t)ROWS);
    if (!forced && n_tiles < n_cus()) return 0;                       // fewer tiles than CUs: the 32-row kernel fills the chip better
    if (n_tiles > 0x7fffffffLL / 64) return 0;
    StreamParams p{};
    size_t act[2] = {0, 0};
    for (int l = 0; l < a->n_layers; ++l) {
        const size_t need = (size_t)ROWS * (((a->units[l] + 63) & ~63) + 4);
        if (need > act[l & 1]) act[l & 1] = need;
        p.units[l] = a->units[l];
        p.W[l] = a->kernels[l];
        p.bias[l] = a->biases[l];
    }
    if (g->n_fields > 64) return 0;
    p.extras_off = S_WORDS;                                            // [2][64] floats
    p.fdesc_off = 160;                                                 // [n_fields <= 64][12 dwords]
    p.cpar_off = 1024;                                                 // biases of every layer, head weights (< 1024 floats)
    p.hpart_off = 2048;                                                // [2][8][64] head partials
    p.ring_off = 3072;                                                 // floats: 12 KiB in, 1-KiB aligned slots
    {
        int off = 0;
        for (int l = 0; l < a->n_layers; ++l) {
            p.bias_off[l] = a->biases[l] != nullptr ? off : -1;
            off += a->biases[l] != nullptr ? a->units[l] : 0;
        }
        p.headw_off = off;
        off += a->units[a->n_layers - 1];
        if (off > 1024) return 0;
        // column tiles of the last layer (its epilogue parks one head partial per tile and row): at most 8
        const int nl = a->units[a->n_layers - 1];
        const bool wide = a->n_layers == 1 ? (nl % 32 == 0 && nl > 16 * NCONS) : (nl % 32 == 0 && nl >= 32 * NCONS);
        if ((wide ? (nl + 31) / 32 : (nl + 15) / 16) > 8) return 0;
    }
    p.act_off[0] = p.ring_off + NSLOT * SLOT_F;
    p.act_off[1] = p.act_off[0] + (int)act[0];
    const size_t lds = ((size_t)p.act_off[1] + act[1]) * sizeof(float);
    if (lds > 160 * 1024) return 0;
    p.fields = g->fields;
    p.ids = g->ids;
    p.ids_stride_f = g->ids_stride_f;
    p.ids_stride_b = g->ids_stride_b;
    p.ids_is_i64 = g->ids_is_i64;
    p.n_fields = g->n_fields;
    p.dim = E;
    p.n_dense = g->n_dense > 0 ? g->n_dense : 0;
    p.dense = g->dense;
    p.dense_stride = g->dense_stride;
    p.dense_lin_w = g->dense_lin_w;
    p.batch = a->batch;
    p.fm_logit = g->fm_logit;
    p.lin_logit = g->lin_logit;
    p.status = g->status;
    p.fm_used = fm_used;
    p.lin_used = lin_used;
    p.in_dim = a->in_dim;
    p.n_layers = a->n_layers;
    p.activation = a->activation;
    p.sigmoid_out = a->sigmoid_out;
    p.head_w = a->head_w;
    for (int i = 0; i < 4; ++i) p.add[i] = a->add[i];
    p.global_bias = a->global_bias;
    p.y = a->y;
    p.probe = a->probe;
    p.n_tiles = (int)n_tiles;
    static thread_local size_t lds_granted[DCTR_MAX_DEVICES] = {0};   // the attribute call costs ~10 us: once per device and size
    {
        hipError_t e = dctr_grant_lds((const void*)stream_kernel, lds, lds_granted);
        if (e != hipSuccess) {
            dctr_set_error("embed_mlp_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
            *rc = (int)e;
            return 1;
        }
    }
    const unsigned blocks = (unsigned)(n_tiles < n_cus() ? n_tiles : n_cus());
    DCTR_LAUNCH(stream_kernel, dim3(blocks), dim3(NTHREADS), lds, stream, p);
    *rc = dctr_launch_status("dctr_embed_mlp_fwd(stream)");
    return 1;
}

}  // namespace dctr_stream
