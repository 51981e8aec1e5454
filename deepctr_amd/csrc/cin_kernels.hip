// a10 — CIN.call (reference deepctr/layers/interaction.py:277-325) on the f32 matrix cores.
//
// Reference per layer k: z[b,d,i*F_k+j] = x0[b,i,d] * x_k[b,j,d]  (tf.matmul of split tensors + reshape,
// :288-295 — MATERIALISED as [D,B,F0*F_k]: 436 MB at C3 layer 1), then conv1d(k=1) = z @ W_k [F0*F_k, H_k]
// (:299-300), bias, activation, transpose to [B,H,D], split_half (first half -> next hidden, second half
// -> output, :308-317), and finally sum over D of the concatenated direct maps (:322-323).
//
// Here the whole network runs in ONE kernel and z is never materialised: a workgroup owns SB samples,
// i.e. M = SB*D GEMM rows (b,d); the A operand of v_mfma_f32_16x16x4_f32 is formed in registers as
// x0[row,i] * x_k[row,j] from two LDS-resident tiles, the four k-slots of one MFMA being four consecutive
// j of the same i; B streams W_k rows from L2; the layer output y[b,h,d] is written back to LDS where it
// is both the next layer's x_{k+1} and the source of the sum over D.  Exact fp32 (fmaf-chain numerics).
// Cost model (C3, F0=26, D=16, H=128,128): 9.58 MFLOP/sample -> 155 TF f32-MFMA => >= 253 us / 4096.
#include "dctr_common.h"
#include "mfma_tile.h"

namespace {

constexpr int CIN_MAX_LAYERS = 8;

struct CinParams {
    const float* x;
    int64_t batch;
    int64_t x_stride;
    int32_t F0, D, n_layers, split_half, activation;
    int32_t SB;        // samples per workgroup
    int32_t RT;        // row tiles = ceil(SB*D/16)
    int32_t Hmax;      // maps a y buffer in LDS holds
    int32_t reg_reduce;  // 1: the direct maps are summed over D in registers and never stored (D % 4 == 0)
    int32_t out_dim;   // featuremap_num
    int32_t H[CIN_MAX_LAYERS];
    const float* W[CIN_MAX_LAYERS];
    const float* bias[CIN_MAX_LAYERS];
    float* save[CIN_MAX_LAYERS];     // per layer NULL or [B*D, H] row-major: the activations, for the backward pass
    float* out;
    // layer 0 folded over its symmetry (x_k = x_0 there: z[i,j] = z[j,i]) — see cin_fold_kernel; NULL = plain layer 0
    const float* Wsym;               // [4 * sym_ks, H[0]] folded filter rows, pair p = j (j + 1) / 2 + i  (i <= j)
    const unsigned* sym_tab;         // [4 * sym_ks] per pair: i | j << 16
    int32_t sym_ks;                  // k-steps of the folded layer (a multiple of 4)
    int32_t two_y;                   // 1: two y buffers in LDS (a layer reads one while it writes the other)
    // dctr_cin_gather_fwd: x == NULL, the workgroup's x_0 tile is read from the embedding tables (field f of sample b = row
    // ids[f, b] of gfields[f].table: inputs.py:101-117 inside the kernel; D % 4 == 0, plain ids)
    const dctr_field_t* gfields;
    const void* ids;
    int64_t ids_stride_f, ids_stride_b;
    int32_t ids_i64;
    int32_t* status;
    // ... and the Dense(1) over the summed maps (models/xdeepfm.py:64-66) taken on chip: the maps of the workgroup's samples wait in
    // LDS, logit[b] = maps[b, :] . head_w leaves instead of `out`
    const float* head_w;             // [out_dim] or NULL
    float* logit;                    // [B]
};

typedef unsigned int cin_u32x2 __attribute__((ext_vector_type(2)));
typedef float cin_f32x2 __attribute__((ext_vector_type(2)));

template <int TPW>
__device__ __forceinline__ void cin_buf_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, float (&b)[TPW]) {
    if constexpr (TPW == 2) {
        const cin_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        b[0] = __uint_as_float(t.x); b[1] = __uint_as_float(t.y);
    } else {
        b[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
    }
}

#define CIN_SB __builtin_amdgcn_sched_barrier(0)

// LDS layout (round 3): every tile is FIELD-major — x0t[i][m], y[n][m] with m = s*D + d the workgroup's GEMM row and a row stride of
// ROWS_P = 16*RT + 16 floats.  A lane's row inside a 16-row tile is its only per-lane address part, the tile is an immediate
// offset (64 B * rt) and the field a scalar (or, in the folded layer, one table value): no per-tile address registers, no integer
// division by D in the loop set-up, the epilogue's four rows of a lane leave as one ds_write_b128.  The 16 floats of padding
// put consecutive fields 16 banks apart: the folded layer's four k-slots read four different fields in one ds_read.
// Rows >= M (padding of the last tile) hold finite junk whose results are dropped; rows never mix in this GEMM.
//
// One CIN layer for the workgroup's M = SB*D rows.  K = F0*Fk is walked in STAGES of SS k-steps of one i:
// stage (c, i) covers j = 4*(SS*c + tt) + g, tt < SS (slot g of the MFMA takes j = 4*jt + g).  Software pipeline:
//   * B (filter rows i*Fk + j, a wave's 16*TPW-column slice) comes from L2 through raw buffer loads — lane-constant
//     offset, scalar row offset, zero VALU — into NB rotating register stages (NB - 1 stages of MFMAs of cover);
//   * the A operand x0[row,i] * x_k[row,j] is formed in registers: its LDS reads for stage s+1 are issued before the
//     MFMAs of stage s and multiplied after them;
//   * sched_barriers pin "issue loads, then MFMAs" (hipcc otherwise sinks each load next to its use).
// The earlier form (one filter load, four LDS reads, eight MFMAs, wait) ran at 48 % of the nominal f32-MFMA rate; this
// one at 68 % (C3: 364 us per 4096 samples; a pure-MFMA loop sustains 139 of the nominal 157 TFLOP/s on this part).
#ifndef CIN_NB
#define CIN_NB 4
#endif
template <int TPW, int RT, bool SAVE, bool SYM>
__device__ __forceinline__ void cin_layer(const CinParams& p, int k, const float* x0t, const float* xk, int Fk, float* ycur, int Hn,
                                          int d0, int64_t bbase, int out_off, const unsigned* tab, float* outl) {
    using dctr::f32x4;
    constexpr int ROWS_P = RT * 16 + 16;
    constexpr int NB = CIN_NB;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, g = lane >> 4, jl = lane & 15;
    const int D = p.D, F0 = p.F0, H = p.H[k];
    const int M = p.SB * D;
    const float* x0l = x0t + jl;          // this lane's row of tile 0, field 0
    const float* xkl = xk + jl;
    const int n_tiles = (H + 16 * TPW - 1) / (16 * TPW);
    constexpr int SS = RT > 4 ? 2 : 4;    // k-steps per stage: 16*TPW... = SS*RT*TPW MFMAs; fewer with 8 row tiles (VGPRs)
    const int JT = (Fk + 3) / 4;          // k-steps per i
    const int NC = (JT + SS - 1) / SS;    // stages per i
    const float* Wk = SYM ? p.Wsym : p.W[k];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wk), 0, (SYM ? 4 * p.sym_ks : F0 * Fk) * H * 4, 0x00020000);
    for (int wt = wave; wt < n_tiles; wt += 4) {
        const int n_base = wt * 16 * TPW;
        int n0 = n_base + TPW * jl;
        if (n0 + TPW > H) n0 = H - TPW;
        const int voff = (g * H + n0) * 4;            // lane-constant byte offset: slot row g, column slice
        f32x4 acc[RT][TPW];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < TPW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a[SS][RT];
        float bq[NB][SS][TPW];
        auto mfmas = [&](const float (&b)[SS][TPW]) {
#pragma unroll
            for (int tt = 0; tt < SS; ++tt)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int c = 0; c < TPW; ++c)
                        acc[rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt][rt], b[tt][c], acc[rt][c], 0, 0, 0);
        };

        if constexpr (SYM) {
            // Folded layer 0 (x_k = x_0): K walks the F0 (F0 + 1) / 2 pairs i <= j, four consecutive pairs per k-step (slot g takes
            // pair 4 t + g), against filter rows W[i F0 + j] + W[j F0 + i] (W[i F0 + i] on the diagonal) that cin_fold_kernel
            // wrote: 88 k-steps at F0 = 26 where the plain walk takes 26 x 8.  Per k-step a lane reads its pair's two x_0 values
            // per row tile (field offsets from the pair table in LDS, fetched one stage ahead of the reads that use them).
            const int n_st = p.sym_ks / SS;
            int sB = 0, sE = 0;
            auto load_b = [&](float (&b)[SS][TPW]) {
#pragma unroll
                for (int tt = 0; tt < SS; ++tt) cin_buf_load<TPW>(rsrc, voff, 4 * (SS * sB + tt) * H * 4, b[tt]);
                sB = min(sB + 1, n_st - 1);
            };
            unsigned ent[SS];
            auto load_ent = [&]() {
#pragma unroll
                for (int tt = 0; tt < SS; ++tt) ent[tt] = tab[4 * (SS * sE + tt) + g];
                sE = min(sE + 1, n_st - 1);
            };
            float ri[SS][RT], rj[SS][RT];
            auto load_raw = [&]() {
#pragma unroll
                for (int tt = 0; tt < SS; ++tt) {
                    const float* qi = x0l + (ent[tt] & 0xffffu);
                    const float* qj = x0l + (ent[tt] >> 16);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        ri[tt][rt] = qi[rt * 16];
                        rj[tt][rt] = qj[rt * 16];
                    }
                }
            };
            auto make_a = [&]() {
#pragma unroll
                for (int tt = 0; tt < SS; ++tt)
#pragma unroll
                    for (int rt = 0; rt < RT; rt += 2) {
                        // (explicit pairs: each comes out of ONE ds_read2_b32; left to itself hipcc pairs across the reads and pays a
                        //  register move per product)
                        const cin_f32x2 pr = cin_f32x2{ri[tt][rt], ri[tt][rt + 1]} * cin_f32x2{rj[tt][rt], rj[tt][rt + 1]};
                        a[tt][rt] = pr[0];
                        a[tt][rt + 1] = pr[1];
                    }
            };
#pragma unroll
            for (int u = 0; u < NB - 1; ++u) load_b(bq[u]);
            load_ent();
            load_raw();
            load_ent();
            make_a();
            for (int s = 0; s < n_st; s += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (s + u < n_st) {
                        load_b(bq[(u + NB - 1) % NB]);
                        load_raw();
                        load_ent();
                        CIN_SB;
                        mfmas(bq[u]);
                        CIN_SB;
                        make_a();
                        CIN_SB;
                    }
                }
            }
        } else {
            // Stage order: chunk c of the j range OUTER, i INNER — stage (c, i) multiplies x0[:, i] with the chunk's SS x 4 values
            // of x_k, which therefore stay in registers for F0 stages (read from LDS once per chunk, masked there), and only
            // x0[:, i] (RT values) is read per stage.  (i outer / c inner read 3x as much LDS per stage.)  The order only
            // permutes the terms of the K sum.  Stage counters of the two prefetch streams (scalar): B runs NB - 1 stages ahead,
            // the A reads one.
            const int n_stage = F0 * NC;
            int iB = 0, cB = 0, iR = 0, cR = 0;
            auto load_b = [&](float (&b)[SS][TPW]) {      // stage (cB, iB), then advance
#pragma unroll
                for (int tt = 0; tt < SS; ++tt) {
                    const int jt = min(SS * cB + tt, JT - 1);             // steps past JT are masked on the A side
                    cin_buf_load<TPW>(rsrc, voff, (iB * Fk + 4 * jt) * H * 4, b[tt]);
                }
                if (++iB == F0) { iB = 0; cB = min(cB + 1, NC - 1); }
            };
            float rxi[RT], rxk[SS][RT];
            auto load_raw = [&]() {                       // stage (cR, iR), then advance
                if (iR == 0) {                            // new chunk: its x_k values, zero where j >= Fk
#pragma unroll
                    for (int tt = 0; tt < SS; ++tt) {
                        const int j = 4 * (SS * cR + tt) + g;
                        const float* q = xkl + min(j, Fk - 1) * ROWS_P;
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const float v = q[rt * 16];
                            rxk[tt][rt] = j < Fk ? v : 0.f;
                        }
                    }
                }
                const float* q0 = x0l + iR * ROWS_P;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) rxi[rt] = q0[rt * 16];
                if (++iR == F0) { iR = 0; cR = min(cR + 1, NC - 1); }
            };
            auto make_a = [&]() {
#pragma unroll
                for (int tt = 0; tt < SS; ++tt)
#pragma unroll
                    for (int rt = 0; rt < RT; rt += 2) {
                        // two row tiles per v_pk_mul_f32: fp32 MFMAs share the vector lanes, every VALU instruction is matrix time
                        const cin_f32x2 pr = cin_f32x2{rxi[rt], rxi[rt + 1]} * cin_f32x2{rxk[tt][rt], rxk[tt][rt + 1]};
                        a[tt][rt] = pr[0];
                        a[tt][rt + 1] = pr[1];
                    }
            };
#pragma unroll
            for (int u = 0; u < NB - 1; ++u) load_b(bq[u]);
            load_raw();
            make_a();
            // stage s: B of stage s+NB-1 and the A reads of stage s+1 are issued, then the MFMAs of stage s, then the
            // products of stage s+1 overwrite `a` (its last reader has issued).  Interleaving those products with the
            // MFMAs instead (double-buffered a, sched_group_barrier 1:1) measured 6 % SLOWER.
            for (int s = 0; s < n_stage; s += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (s + u < n_stage) {
                        load_b(bq[(u + NB - 1) % NB]);
                        load_raw();
                        CIN_SB;
                        mfmas(bq[u]);
                        CIN_SB;
                        make_a();
                        CIN_SB;
                    }
                }
            }
        }
        // epilogue: bias + activation (C layout: row m = 16rt + 4g + r = (sample s, d), col n = n_base + TPW*jl + c).
        // Maps [0, Hn) feed the next layer -> y[n][m] in LDS (the lane's four rows as one 16-B store; rows >= M are junk
        // nobody reads).  Maps [d0, H) go to the output summed over d
        // (interaction.py:322-323): with reg_reduce that sum is taken here — 4 rows in the lane, lanes 16 / 32 apart
        // (the other k-slots' rows of the same sample), then the D/16 row tiles of a sample — and written straight
        // to `out`; otherwise (D not 4, 8 or a multiple of 16) every map is stored and cin_kernel sums from LDS.
        // (SAVE) descriptor over the whole [B*D, H] buffer of this layer, this lane's byte offset of (row 4g of the workgroup, column n_base + TPW*jl)
        __amdgpu_buffer_rsrc_t save_rsrc = rsrc;
        int save_voff = 0;
        if constexpr (SAVE) {
            if (p.save[k] != nullptr) {
                save_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.save[k], 0, (int)(p.batch * D * H * 4), 0x00020000);
                save_voff = (int)(((bbase * D + 4 * g) * H + n_base + TPW * jl) * 4);
            }
        }
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int n = n_base + TPW * jl + c;
            const bool nok = n < H;
            const float bv = nok ? p.bias[k][n] : 0.f;
            float dsum[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = dctr::apply_act(acc[rt][c][r] + bv, p.activation);
                const bool store = nok && (!p.reg_reduce || n < Hn);
                if (store) *reinterpret_cast<float4*>(ycur + n * ROWS_P + rt * 16 + 4 * g) = float4{v[0], v[1], v[2], v[3]};
                if constexpr (SAVE) {
                    // training: y_k[(b, d), n] (all H maps, row-major) for dctr_cin_bwd.  Buffer stores: lane-constant byte
                    // offset + a scalar row offset (per-row 64-bit addresses cost 64 VGPRs and spilled), rows past the batch
                    // fall outside the descriptor and are dropped by the hardware.
                    if (p.save[k] != nullptr) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = rt * 16 + 4 * g + r;
                            const int vo = (nok && m < M) ? save_voff + c * 4 : 0x7ffffff0;
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), save_rsrc, vo, (rt * 16 + r) * H * 4, 0);
                        }
                    }
                }
                float t = (v[0] + v[1]) + (v[2] + v[3]);
                if (D >= 8) t += __shfl_xor(t, 16, 64);
                if (D >= 16) t += __shfl_xor(t, 32, 64);
                dsum[rt] = t;
            }
            if (p.reg_reduce && nok && n >= d0) {
                const int per = D >= 16 ? D / 16 : 1;                 // row tiles per sample
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    if (rt % per != 0) continue;
                    float t = dsum[rt];
#pragma unroll
                    for (int q = 1; q < RT; ++q)
                        if (q < per && rt + q < RT) t += dsum[rt + q];
                    const int m = rt * 16 + 4 * g;                    // first row of this lane's group
                    const bool lead = D >= 16 ? g == 0 : (4 * g) % D == 0;
                    if (lead && m < M) {
                        const int64_t b = bbase + m / D;
                        if (outl != nullptr) outl[(m / D) * p.out_dim + out_off + (n - d0)] = t;      // (fused head: the maps wait in LDS)
                        else if (b < p.batch) p.out[b * p.out_dim + out_off + (n - d0)] = t;
                    }
                }
            }
        }
    }
}

// SAVE: the training-mode instantiation that also writes every layer's activations (p.save) — a separate one because keeping the
// activated accumulators alive for those stores costs the inference kernel registers (580 B of scratch per lane when it was a
// run-time branch of the one kernel).
template <int RT, bool SAVE>
__global__ __launch_bounds__(256, 2) void cin_kernel(CinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS_P = RT * 16 + 16;
    const int D = p.D, F0 = p.F0, SB = p.SB;
    const int M = SB * D;
    float* x0t = smem;                              // [F0][ROWS_P]
    float* y0 = x0t + F0 * ROWS_P;                  // [Hmax][ROWS_P]
    float* y1 = y0 + p.Hmax * ROWS_P;               // (only when a layer reads one y buffer while writing the other: p.two_y)
    unsigned* tab = reinterpret_cast<unsigned*>(y1 + (p.two_y ? p.Hmax * ROWS_P : 0));      // [4 * sym_ks] pair table of the folded layer 0
    float* outl = p.head_w != nullptr ? reinterpret_cast<float*>(tab + 4 * p.sym_ks) : nullptr;   // [SB][out_dim] summed maps (fused head)
    const int64_t b0 = (int64_t)blockIdx.x * SB;
    if (p.Wsym != nullptr)                          // (i, j) of the fold kernel's table -> float offsets of the two fields' rows in x0t
        for (int t = threadIdx.x; t < 4 * p.sym_ks; t += 256) {
            const unsigned e = p.sym_tab[t];
            tab[t] = (e & 0xffffu) * ROWS_P | ((e >> 16) * ROWS_P) << 16;
        }
    if (p.gfields != nullptr) {
        // x0 tile from the tables: item (sample s, field f, 16-B piece q) — ids of eight items, then their rows, in flight together;
        // an id outside the vocabulary reads row 0 and raises the status flag; samples past the batch: zeros
        const int Q = D >> 2, fq = F0 * Q, total = SB * fq;
        for (int base = 0; base < total; base += 256 * 8) {
            int64_t id[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + u * 256 + (int)threadIdx.x, total - 1);
                const int s = i / fq, f = (i - s * fq) / Q;
                const int64_t b = min(b0 + s, p.batch - 1);
                const int64_t eo = (int64_t)f * p.ids_stride_f + b * p.ids_stride_b;
                id[u] = p.ids_i64 ? reinterpret_cast<const int64_t*>(p.ids)[eo] : (int64_t)reinterpret_cast<const int32_t*>(p.ids)[eo];
            }
            float4 v[8];
            bool bad = false;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + u * 256 + (int)threadIdx.x, total - 1);
                const int s = i / fq, r = i - s * fq;
                const int f = r / Q, q = r - f * Q;
                const bool ok = (uint64_t)id[u] < (uint64_t)p.gfields[f].vocab;
                bad = bad || (!ok && base + u * 256 + (int)threadIdx.x < total && b0 + s < p.batch);
                v[u] = *reinterpret_cast<const float4*>(p.gfields[f].table + (ok ? id[u] : 0) * D + 4 * q);
                if (b0 + s >= p.batch) v[u] = float4{0.f, 0.f, 0.f, 0.f};
            }
            if (bad && p.status != nullptr) atomicOr(p.status, (int)DCTR_STATUS_INDEX_OOR);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < total) {
                    const int s = i / fq, r = i - s * fq;
                    const int f = r / Q, q = r - f * Q;
                    *reinterpret_cast<float4*>(x0t + f * ROWS_P + s * D + 4 * q) = v[u];
                }
            }
        }
        const int padr = RT * 16 - M;               // rows of the last tile past M: zeros
        for (int t = threadIdx.x; t < F0 * padr; t += 256) x0t[(t / padr) * ROWS_P + M + t % padr] = 0.f;
    } else {
        // x0 tile: all global loads of a pass in flight before the LDS stores
        const int total = SB * F0 * D, fd = F0 * D;
        for (int base = 0; base < total; base += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(base + u * 256 + (int)threadIdx.x, total - 1);
                const int s = i / fd;
                const int64_t b = min(b0 + s, p.batch - 1);
                v[u] = p.x[b * p.x_stride + (i - s * fd)];
                if (b0 + s >= p.batch) v[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                if (i < total) {
                    const int s = i / fd, r = i - s * fd;
                    const int f = r / D, d = r - f * D;
                    x0t[f * ROWS_P + s * D + d] = v[u];
                }
            }
        }
        const int padr = RT * 16 - M;               // rows of the last tile past M: zeros
        for (int t = threadIdx.x; t < F0 * padr; t += 256) x0t[(t / padr) * ROWS_P + M + t % padr] = 0.f;
    }
    __syncthreads();

    const float* xk = x0t;
    int Fk = F0;
    float* ycur = y0;
    float* ynext = p.two_y ? y1 : y0;
    int out_off = 0;
    for (int k = 0; k < p.n_layers; ++k) {
        const int H = p.H[k];
        // split (interaction.py:308-317): maps [0, Hn) feed the next layer, maps [d0, H) go to the output
        const bool last = k == p.n_layers - 1;
        int Hn, d0;
        if (p.split_half) {
            Hn = last ? 0 : H / 2;
            d0 = last ? 0 : H / 2;
        } else {
            Hn = last ? 0 : H;
            d0 = 0;
        }
        if (k == 0 && p.Wsym != nullptr) {
            if (H % 32 == 0) cin_layer<2, RT, SAVE, true>(p, k, x0t, xk, Fk, ycur, Hn, d0, b0, out_off, tab, outl);
            else cin_layer<1, RT, SAVE, true>(p, k, x0t, xk, Fk, ycur, Hn, d0, b0, out_off, tab, outl);
        } else if (H % 32 == 0) cin_layer<2, RT, SAVE, false>(p, k, x0t, xk, Fk, ycur, Hn, d0, b0, out_off, tab, outl);
        else cin_layer<1, RT, SAVE, false>(p, k, x0t, xk, Fk, ycur, Hn, d0, b0, out_off, tab, outl);
        __syncthreads();
        const int nd = H - d0;
        if (!p.reg_reduce) {
            // result = reduce_sum(concat(direct), -1): deterministic serial sum over d
            for (int t = threadIdx.x; t < SB * nd; t += 256) {
                const int s = t / nd, n = d0 + t % nd;
                if (b0 + s < p.batch) {
                    const float* yp = ycur + n * ROWS_P + s * D;
                    float acc = 0.f;
                    for (int d = 0; d < D; ++d) acc += yp[d];
                    if (outl != nullptr) outl[s * p.out_dim + out_off + (n - d0)] = acc;      // (fused head: the maps wait in LDS)
                    else p.out[(b0 + s) * p.out_dim + out_off + (n - d0)] = acc;
                }
            }
        }
        out_off += nd;
        xk = ycur;
        Fk = Hn;
        float* t = ycur;
        ycur = ynext;
        ynext = t;
        // no barrier needed here: the next layer writes the OTHER y buffer and only reads this one
    }
    if (outl != nullptr) {
        // fused head: one wave per sample, lanes over the maps in a fixed order (deterministic)
        __syncthreads();
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int s = wave; s < SB; s += 4) {
            float acc = 0.f;
            for (int c = lane; c < p.out_dim; c += 64) acc = fmaf(outl[s * p.out_dim + c], p.head_w[c], acc);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
            if (lane == 0 && b0 + s < p.batch) p.logit[b0 + s] = acc;
        }
    }
}

// Layer 0 of a CIN multiplies x_0 with itself: z[i F0 + j] = z[j F0 + i], so  sum_{i,j} z_ij W[ij, h] = sum_{i <= j} x_i x_j Wf[p(i,j), h]
// with Wf = W[ij] + W[ji] (W[ii] on the diagonal), p = j (j + 1) / 2 + i.  This kernel writes Wf (rows >= the pair count: zeros) and the
// pair table the forward kernel walks; it runs in front of a dctr_cin_fwd that was given a workspace (unless the caller vouches it is current).
__global__ __launch_bounds__(256) void cin_fold_kernel(const float* __restrict__ W, int F0, int H, int rows, float* __restrict__ Wf,
                                                       unsigned* __restrict__ tab) {
    const int P = F0 * (F0 + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)rows * H; e += (int64_t)gridDim.x * 256) {
        const int pr = (int)(e / H), h = (int)(e % H);
        int i = 0, j = 0;
        if (pr < P) {
            j = (int)((sqrtf(8.f * pr + 1.f) - 1.f) * 0.5f);
            while ((j + 1) * (j + 2) / 2 <= pr) ++j;          // (float rounding of the root, either way)
            while (j * (j + 1) / 2 > pr) --j;
            i = pr - j * (j + 1) / 2;
        }
        float v = 0.f;
        if (pr < P) v = i == j ? W[(int64_t)(i * F0 + i) * H + h] : W[(int64_t)(i * F0 + j) * H + h] + W[(int64_t)(j * F0 + i) * H + h];
        Wf[e] = v;
        if (h == 0) tab[pr] = (unsigned)i | ((unsigned)j << 16);
    }
}

// k-steps of the folded layer 0 (multiple of 4: a whole number of stages whatever the stage length), 0 = cannot fold
int cin_sym_ksteps(int F0, int D) {
    (void)D;
    if (F0 * 144 > 0xffff) return 0;                       // (field offsets in the LDS tile as 16-bit halves of a table word)
    const int P = F0 * (F0 + 1) / 2;
    return ((P + 3) / 4 + 3) & ~3;
}

int cin_out_dim(const dctr_cin_args_t* a) {
    int o = 0;
    for (int k = 0; k < a->n_layers; ++k) {
        const int H = a->layer_size[k];
        const bool last = k == a->n_layers - 1;
        o += a->split_half ? (last ? H : H - H / 2) : H;
    }
    return o;
}

// ---- embedding_dim > 128 (interaction.py:277-325 has no limit on D) ----------------------------------------------------------
// CIN never mixes the embedding dimensions: every (sample, d) is an independent GEMM row until the final reduce_sum over d
// (:322-323).  A sample of width D is therefore walked as D / dd pseudo-samples of width dd (the largest divisor of D that the
// kernel's tile holds, <= 128) whose summed maps add up to the sample's: a pre-pass lays x [B, F0, D] out as [B * D/dd, F0, dd] in
// the caller's workspace, the kernel above runs on that, a post-pass adds the D / dd partial map vectors of a sample in the order
// of d.  The two passes move 2 x 4 F0 D bytes per sample beside 2 F0 sum(F_k H_k) D flops: under 2 % of the kernel's time.
constexpr int CIN_WIDE_ROWS = 1024;            // samples per chunk the workspace query provides for (any workspace of >= 64 works)

__global__ __launch_bounds__(256) void cin_reslice_kernel(const float* __restrict__ x, int64_t x_stride, int64_t rows, int F0, int D, int dd,
                                                          float* __restrict__ xs) {
    const int nsl = D / dd;
    const int64_t fd = (int64_t)F0 * D, total = rows * fd;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / fd;
        const int r = (int)(o - b * fd);
        const int f = r / D, d = r - f * D;
        const int sl = d / dd, dl = d - sl * dd;
        xs[((b * nsl + sl) * F0 + f) * dd + dl] = x[b * x_stride + r];
    }
}

__global__ __launch_bounds__(256) void cin_slice_sum_kernel(const float* __restrict__ part, int64_t rows, int nsl, int out_dim,
                                                            float* __restrict__ out) {
    const int64_t total = rows * out_dim;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t b = o / out_dim;
        const int c = (int)(o - b * out_dim);
        const float* pp = part + (b * nsl) * out_dim + c;
        float acc = 0.f;
        for (int sl = 0; sl < nsl; ++sl) acc += pp[(int64_t)sl * out_dim];      // (fixed order: deterministic)
        out[o] = acc;
    }
}

size_t cin_fold_bytes(const dctr_cin_args_t* a) {
    const size_t ks = (size_t)cin_sym_ksteps(a->fields, a->dim);
    return ks * 4 * ((size_t)a->layer_size[0] + 1) * sizeof(float);
}

size_t cin_wide_row_bytes(const dctr_cin_args_t* a, int dd) {
    return ((size_t)a->fields * a->dim + (size_t)(a->dim / dd) * cin_out_dim(a)) * sizeof(float);
}

}  // namespace

// intermediates live in LDS; the workspace holds layer 0's folded filter rows + pair table (optional: without it layer 0 walks all
// F0 x F0 products — same result up to the rounding of W[ij] + W[ji], ~1.2x the time at C3)
static int cin_fwd_impl(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g, const float* head_w, float* logit, void* stream,
                        bool dry = false);
static int cin_gather_checks(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g);
static int cin_fwd_wide(const dctr_cin_args_t* a, int dd, void* stream, bool dry);

// the slice width dctr_cin_fwd walks a sample in: a->dim itself when the kernel takes the sample whole, else the largest divisor of
// a->dim it takes (cin_fwd_impl's own shape decisions, dry), 0 = none
static int cin_pick_slice(const dctr_cin_args_t* a) {
    if (a->dim < 1) return 0;
    dctr_cin_args_t b = *a;
    b.batch = 1;
    b.save_y = nullptr;
    for (int dd = a->dim < 128 ? a->dim : 128; dd >= 1; --dd) {
        if (a->dim % dd != 0) continue;
        b.dim = dd;
        if (cin_fwd_impl(&b, nullptr, nullptr, nullptr, nullptr, true) == DCTR_OK) return dd;
    }
    return 0;
}

// train_kernels.hip: CIN layer by layer on the library's GEMM (z materialised per chunk of samples) — any layer sizes
size_t dctr_cin_layered_sample_floats(const dctr_cin_args_t* a);
int dctr_cin_fwd_layered(const dctr_cin_args_t* a, void* workspace, size_t workspace_bytes, void* stream);
constexpr int CIN_LAYERED_SAMPLES = 256;       // samples per chunk the workspace query provides for (any room for >= 16 works)

// the argument checks every route shares (sizes, enums): what is left to fail in cin_fwd_impl afterwards is the LDS the shape needs
static int cin_shape_checks(const dctr_cin_args_t* a) {
    DCTR_REQUIRE(a != nullptr && a->layer_size != nullptr, DCTR_E_NULL, "cin_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->fields >= 1 && a->dim >= 1 && a->n_layers >= 1 && a->n_layers <= CIN_MAX_LAYERS, DCTR_E_DIM,
                 "cin_fwd: bad sizes (B=%lld F=%d D=%d layers=%d)", (long long)a->batch, a->fields, a->dim, a->n_layers);
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_TANH, DCTR_E_ENUM, "cin_fwd: activation %d", a->activation);
    for (int k = 0; k < a->n_layers; ++k) {
        DCTR_REQUIRE(a->layer_size[k] >= 1, DCTR_E_DIM, "cin_fwd: layer_size[%d]=%d", k, a->layer_size[k]);
        if (a->split_half && k != a->n_layers - 1)
            DCTR_REQUIRE(a->layer_size[k] % 2 == 0, DCTR_E_DIM, "cin_fwd: layer_size must be even except for the last layer when split_half=True");
    }
    return DCTR_OK;
}

// 1: the one-kernel form takes the sample whole; 2: in slices of *dd dimensions; 3: neither — layer by layer (train_kernels.hip)
static int cin_route_of(const dctr_cin_args_t* a, int* dd) {
    *dd = cin_pick_slice(a);
    if (*dd == a->dim) return 1;
    return *dd > 0 ? 2 : 3;
}

// dctr_cin_fwd: the sample whole, in slices of d, or — layer sizes no tile of the kernel holds — layer by layer
static int cin_fwd_route(const dctr_cin_args_t* a, void* stream, bool dry) {
    const int rc = cin_shape_checks(a);
    if (rc != DCTR_OK) return rc;
    int dd = 0;
    const int route = cin_route_of(a, &dd);
    if (route == 1) return cin_fwd_impl(a, nullptr, nullptr, nullptr, stream, dry);
    if (route == 2) return cin_fwd_wide(a, dd, stream, dry);
    if (dry || a->batch == 0) return DCTR_OK;
    DCTR_REQUIRE(a->x_stride >= (int64_t)a->fields * a->dim, DCTR_E_DIM, "cin_fwd: x_stride < fields*dim");
    return dctr_cin_fwd_layered(a, a->workspace, a->workspace_bytes, stream);
}

// ABI 13: samples the kernel does not take whole (embedding_dim > 128, or > 64 with more maps than fit the LDS beside a 128-row
// tile): + room for CIN_WIDE_ROWS samples of the sliced route — REQUIRED there, whatever the batch
extern "C" size_t dctr_cin_workspace_bytes(const dctr_cin_args_t* a) {
    if (a == nullptr || a->fields < 1 || a->dim < 1 || a->n_layers < 1 || a->n_layers > CIN_MAX_LAYERS || a->layer_size == nullptr ||
        a->layer_size[0] < 1)
        return 0;
    const size_t fold = cin_fold_bytes(a);
    for (int k = 0; k < a->n_layers; ++k)
        if (a->layer_size[k] < 1) return fold;
    dctr_cin_args_t b = *a;
    if (b.activation < DCTR_ACT_LINEAR || b.activation > DCTR_ACT_TANH) b.activation = DCTR_ACT_RELU;   // (size queries come without one)
    b.batch = 1;
    int dd = 0;
    const int route = cin_route_of(&b, &dd);
    if (route == 1) return fold;
    if (route == 2) return ((fold + 15) & ~(size_t)15) + (size_t)CIN_WIDE_ROWS * cin_wide_row_bytes(a, dd);
    // (at most 256 MiB unless 16 samples need more: z is F0 * F_k products per (sample, d))
    const size_t per = dctr_cin_layered_sample_floats(a) * sizeof(float), cap = (size_t)256 << 20;
    size_t n = (size_t)CIN_LAYERED_SAMPLES;
    if (n * per > cap) n = cap / per < 16 ? 16 : cap / per;
    return n * per;
}

extern "C" int dctr_cin_fwd(const dctr_cin_args_t* a, void* stream) { return cin_fwd_route(a, stream, false); }

// ABI 13 — would dctr_cin_fwd (gather == NULL) / dctr_cin_gather_fwd (gather != NULL; fused_head: with head_w / logit) take these
// arguments?  Every shape check and kernel-shape decision of the call, no launch; device pointers are not looked at (args->layer_size
// is: a HOST array).  1 = yes, 0 = no (dctr_last_error() says why).
extern "C" int dctr_cin_fwd_supported(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g, int32_t fused_head) {
    if (a == nullptr || a->layer_size == nullptr) return 0;
    if (g == nullptr) return cin_fwd_route(a, nullptr, true) == DCTR_OK ? 1 : 0;
    if (cin_gather_checks(a, g) != DCTR_OK) return 0;
    static const float one_float = 0.f;
    return cin_fwd_impl(a, g, fused_head ? &one_float : nullptr, nullptr, nullptr, true) == DCTR_OK ? 1 : 0;
}

// ABI 8 — CIN.call over the embeddings of a gather (models/xdeepfm.py:52-66): exFM_in = concat of the fields' rows is read from the
// tables inside the kernel (g: plain ids, every field `dim` wide, dim % 4 == 0, no pre-pooled field; g->dnn_in etc. unused), and with
// head_w / logit the Dense(1, use_bias=False) over the summed maps is taken on chip: logit[b] leaves, `out` may be NULL
extern "C" int dctr_cin_gather_fwd(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g, const float* head_w, float* logit, void* stream) {
    DCTR_REQUIRE(a != nullptr && g != nullptr, DCTR_E_NULL, "cin_gather_fwd: null args");
    DCTR_REQUIRE((head_w == nullptr) == (logit == nullptr), DCTR_E_NULL, "cin_gather_fwd: head_w and logit come together");
    DCTR_REQUIRE(g->fields != nullptr && g->ids != nullptr, DCTR_E_NULL, "cin_gather_fwd: null descriptors / ids");
    const int rc = cin_gather_checks(a, g);
    if (rc != DCTR_OK) return rc;
    return cin_fwd_impl(a, g, head_w, logit, stream);
}

static int cin_gather_checks(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g) {
    DCTR_REQUIRE(g->n_fields == a->fields && g->batch == a->batch, DCTR_E_DIM, "cin_gather_fwd: gather of %d fields x %lld rows against CIN over %d x %lld",
                 g->n_fields, (long long)g->batch, a->fields, (long long)a->batch);
    DCTR_REQUIRE(a->dim % 4 == 0 && g->uniform_dim == a->dim && g->all_dim4 && !g->any_hash && !g->any_identity && !g->any_pitch, DCTR_E_UNSUPPORTED,
                 "cin_gather_fwd: every field must be a plain (unhashed, not pre-pooled) lookup of width dim = %d, a multiple of 4", a->dim);
    DCTR_REQUIRE(a->save_y == nullptr, DCTR_E_UNSUPPORTED, "cin_gather_fwd: inference only (save_y: use dctr_embed_gather_fm + dctr_cin_fwd)");
    DCTR_REQUIRE(a->dim <= 128, DCTR_E_UNSUPPORTED, "cin_gather_fwd: embedding_dim %d > 128 (use dctr_embed_gather_fm + dctr_cin_fwd: the sliced route)", a->dim);
    return DCTR_OK;
}

// dctr_cin_fwd over samples the kernel does not take whole, in slices of dd embedding dimensions (see cin_reslice_kernel)
static int cin_fwd_wide(const dctr_cin_args_t* a, int dd, void* stream, bool dry) {
    const int D = a->dim, nsl = D / dd, F0 = a->fields;
    const int out_dim = cin_out_dim(a);
    dctr_cin_args_t b = *a;
    b.dim = dd;
    b.x_stride = (int64_t)F0 * dd;
    b.save_y = nullptr;
    if (dry) {                                  // the kernel's own shape decisions for the slice width
        b.batch = (a->batch > 0 ? a->batch : 1) * nsl;
        b.workspace = nullptr;
        b.workspace_bytes = 0;
        return cin_fwd_impl(&b, nullptr, nullptr, nullptr, nullptr, true);
    }
    if (a->batch == 0) return DCTR_OK;
    const size_t fold = cin_fold_bytes(a), fold_al = (fold + 15) & ~(size_t)15, per_row = cin_wide_row_bytes(a, dd);
    DCTR_REQUIRE(a->workspace != nullptr && dctr_aligned16(a->workspace) && a->workspace_bytes >= fold_al + 64 * per_row, DCTR_E_NULL,
                 "cin_fwd: embedding_dim %d goes out in slices of %d and needs a 16-B aligned workspace (dctr_cin_workspace_bytes: %zu B; at "
                 "least %zu B)", D, dd, dctr_cin_workspace_bytes(a), fold_al + 64 * per_row);
    DCTR_REQUIRE(a->x != nullptr && a->out != nullptr, DCTR_E_NULL, "cin_fwd: null pointer");
    int64_t cap = (int64_t)((a->workspace_bytes - fold_al) / per_row);
    cap = cap > 65536 ? 65536 : cap & ~(int64_t)15;
    if (a->save_y != nullptr) {                 // (a chunk's part of a y_k stays inside the 2 GiB a buffer descriptor addresses)
        int hmax = 1;
        for (int k = 0; k < a->n_layers; ++k) hmax = a->layer_size[k] > hmax ? a->layer_size[k] : hmax;
        const int64_t lim = (0x7fffffffLL / ((int64_t)D * hmax * 4) - 1) & ~(int64_t)15;
        DCTR_REQUIRE(lim >= 16, DCTR_E_DIM, "cin_fwd: save_y rows of %d x %d floats are too long", D, hmax);
        cap = cap > lim ? lim : cap;
    }
    float* xs = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + fold_al);
    float* part = xs + (size_t)cap * F0 * D;    // (cap is a multiple of 16: 16-B aligned)
    hipStream_t st = (hipStream_t)stream;
    for (int64_t r0 = 0; r0 < a->batch; r0 += cap) {
        const int64_t rows = a->batch - r0 < cap ? a->batch - r0 : cap;
        int64_t nb = dctr_ceil_div(rows * F0 * D, (int64_t)256);
        DCTR_LAUNCH(cin_reslice_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, st, a->x + r0 * a->x_stride, a->x_stride,
                    rows, F0, D, dd, xs);
        b.x = xs;
        b.batch = rows * nsl;
        b.out = part;
        b.workspace = fold > 0 ? a->workspace : nullptr;
        b.workspace_bytes = fold;
        b.workspace_ready = r0 == 0 ? a->workspace_ready : 1;
        // save_y: row (b, d) of y_k is row b * D + d = (b * nsl + sl) * dd + dl — the pseudo-samples' rows ARE the sample's rows
        float* sv[CIN_MAX_LAYERS];
        if (a->save_y != nullptr) {
            for (int k = 0; k < a->n_layers; ++k)
                sv[k] = a->save_y[k] != nullptr ? a->save_y[k] + (size_t)r0 * D * a->layer_size[k] : nullptr;
            b.save_y = sv;
        }
        const int rc = cin_fwd_impl(&b, nullptr, nullptr, nullptr, stream);
        if (rc != DCTR_OK) return rc;
        nb = dctr_ceil_div(rows * out_dim, (int64_t)256);
        DCTR_LAUNCH(cin_slice_sum_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, st, part, rows, nsl, out_dim,
                    a->out + r0 * out_dim);
    }
    return dctr_launch_status("dctr_cin_fwd");
}

static int cin_fwd_impl(const dctr_cin_args_t* a, const dctr_gather_fm_args_t* g, const float* head_w, float* logit, void* stream, bool dry) {
    DCTR_REQUIRE(a != nullptr, DCTR_E_NULL, "cin_fwd: null args");
    DCTR_REQUIRE(a->batch >= 0 && a->fields >= 1 && a->dim >= 1 && a->n_layers >= 1 && a->n_layers <= CIN_MAX_LAYERS,
                 DCTR_E_DIM, "cin_fwd: bad sizes (B=%lld F=%d D=%d layers=%d)", (long long)a->batch, a->fields, a->dim,
                 a->n_layers);
    if (a->batch == 0 && !dry) return DCTR_OK;
    DCTR_REQUIRE(dry ? a->layer_size != nullptr : ((a->x || g) && (a->out || logit) && a->layer_size && a->filters && a->bias), DCTR_E_NULL,
                 "cin_fwd: null pointer");
    DCTR_REQUIRE(a->activation >= DCTR_ACT_LINEAR && a->activation <= DCTR_ACT_TANH, DCTR_E_ENUM, "cin_fwd: activation %d",
                 a->activation);
    DCTR_REQUIRE(g != nullptr || dry || a->x_stride >= (int64_t)a->fields * a->dim, DCTR_E_DIM, "cin_fwd: x_stride < fields*dim");
    DCTR_REQUIRE(a->dim <= 128, DCTR_E_UNSUPPORTED, "cin kernel: embedding_dim %d > 128 (a sample's rows are one workgroup's MFMA tiles: dctr_cin_fwd "
                 "walks such samples in slices)", a->dim);
    CinParams p{};
    p.x = g != nullptr ? nullptr : a->x;
    if (g != nullptr) {
        p.gfields = g->fields;
        p.ids = g->ids;
        p.ids_stride_f = g->ids_stride_f;
        p.ids_stride_b = g->ids_stride_b;
        p.ids_i64 = g->ids_is_i64;
        p.status = g->status;
    }
    p.head_w = head_w;
    p.logit = logit;
    p.batch = a->batch;
    p.x_stride = a->x_stride;
    p.F0 = a->fields;
    p.D = a->dim;
    p.n_layers = a->n_layers;
    p.split_half = a->split_half ? 1 : 0;
    p.activation = a->activation;
    // the in-register sum over d pairs lanes 16 / 32 apart: a sample's rows must be 4, 8 or whole 16-row tiles (D = 12, 20, 24 ...
    // straddle them: those sum from LDS)
    p.reg_reduce = (a->dim == 4 || a->dim == 8 || a->dim % 16 == 0) ? 1 : 0;
    int hmax = 1;
    for (int k = 0; k < a->n_layers; ++k) {
        const int H = a->layer_size[k];
        DCTR_REQUIRE(H >= 1, DCTR_E_DIM, "cin_fwd: layer_size[%d]=%d", k, H);
        const bool last = k == a->n_layers - 1;
        if (a->split_half && !last)
            DCTR_REQUIRE(H % 2 == 0, DCTR_E_DIM,
                         "cin_fwd: layer_size must be even except for the last layer when split_half=True");
        p.H[k] = H;
        if (!dry) {
            DCTR_REQUIRE(a->filters[k] && a->bias[k], DCTR_E_NULL, "cin_fwd: filters/bias[%d] null", k);
            DCTR_REQUIRE((((uintptr_t)a->filters[k]) & 7u) == 0, DCTR_E_ALIGN, "cin_fwd: filters[%d] not 8-B aligned", k);
            p.W[k] = a->filters[k];
            p.bias[k] = a->bias[k];
        }
        p.save[k] = (!dry && a->save_y != nullptr) ? a->save_y[k] : nullptr;
        DCTR_REQUIRE(p.save[k] == nullptr || a->batch * (int64_t)a->dim * H * 4 < 0x7fffffffLL, DCTR_E_DIM,
                     "cin_fwd: save_y[%d] of %lld x %d floats exceeds the 2 GiB a buffer descriptor addresses", k,
                     (long long)(a->batch * a->dim), H);
        DCTR_REQUIRE((((uintptr_t)p.save[k]) & 7u) == 0, DCTR_E_ALIGN, "cin_fwd: save_y[%d] not 8-B aligned", k);
        // maps kept in LDS: all of them without reg_reduce, else only those the next layer reads
        const int keep = !p.reg_reduce ? H : (last ? 1 : (a->split_half ? H / 2 : H));
        hmax = keep > hmax ? keep : hmax;
    }
    p.Hmax = hmax;
    p.out_dim = cin_out_dim(a);
    p.out = a->out;
    // rows per workgroup: 128 (eight 16-row tiles, every filter fragment feeds eight MFMAs per column tile: the
    // filter stream from L2, 851 KB per workgroup at C3 layer 2, is what bounds this kernel) when two workgroups
    // still fit a CU's LDS, else 64
    // layer 0 folded over its symmetry when the caller brought the workspace for it
    const size_t sym_need = cin_fold_bytes(a);
    if (dry) p.sym_ks = sym_need > 0 ? cin_sym_ksteps(p.F0, p.D) : 0;     // (the LDS of the folded form: the larger one)
    else if (a->workspace != nullptr && sym_need > 0) {
        DCTR_REQUIRE(a->workspace_bytes >= sym_need && dctr_aligned16(a->workspace), DCTR_E_DIM,
                     "cin_fwd: workspace of %zu B (16-B aligned) needed, got %zu B", sym_need, a->workspace_bytes);
        p.sym_ks = cin_sym_ksteps(p.F0, p.D);
        float* wf = static_cast<float*>(a->workspace);
        unsigned* tb = reinterpret_cast<unsigned*>(wf + (size_t)4 * p.sym_ks * p.H[0]);
        const int64_t n = (int64_t)4 * p.sym_ks * p.H[0];
        if (!a->workspace_ready)
            hipLaunchKernelGGL(cin_fold_kernel, dim3((unsigned)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           p.W[0], p.F0, p.H[0], 4 * p.sym_ks, wf, tb);
        p.Wsym = wf;
        p.sym_tab = tb;
    }
    p.two_y = (p.n_layers >= 3 || (p.n_layers == 2 && !p.reg_reduce)) ? 1 : 0;   // (the last layer of a reg_reduce net stores nothing)
    auto lds_of = [&](int rt_) {
        const size_t rows_p = (size_t)rt_ * 16 + 16;
        const size_t sb = (size_t)(rt_ == 8 ? 128 : 64) / (size_t)a->dim + 1;                   // (>= SB)
        return (((size_t)p.F0 + (size_t)(p.two_y ? 2 : 1) * p.Hmax) * rows_p + (size_t)4 * p.sym_ks + (head_w != nullptr ? sb * p.out_dim : 0)) *
               sizeof(float);
    };
    int rt = 8;
    p.SB = 128 / a->dim;
    if (p.SB < 1) p.SB = 1;
    if ((lds_of(8) > 80 * 1024 || p.SB * a->dim > 128) && a->dim <= 64) {      // (embedding_dim 65 .. 128: one sample = up to eight row tiles)
        rt = 4;
        p.SB = 64 / a->dim;
        if (p.SB < 1) p.SB = 1;
    }
    p.RT = (p.SB * a->dim + 15) / 16;
    DCTR_REQUIRE(p.RT <= rt, DCTR_E_UNSUPPORTED, "cin_fwd: embedding_dim %d needs %d row tiles", a->dim, p.RT);
#ifndef CIN_LDS_PAD
#define CIN_LDS_PAD 0           // lab: extra dynamic LDS, to pin the number of workgroups per CU
#endif
    const size_t lds = lds_of(rt) + CIN_LDS_PAD;
    DCTR_REQUIRE(lds <= 160 * 1024, DCTR_E_UNSUPPORTED, "cin_fwd: needs %zu B of LDS (> 160 KiB)", lds);
    if (dry) return DCTR_OK;
    const bool save = a->save_y != nullptr;
    const void* fn = rt == 8 ? (save ? (const void*)cin_kernel<8, true> : (const void*)cin_kernel<8, false>)
                             : (save ? (const void*)cin_kernel<4, true> : (const void*)cin_kernel<4, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DCTR_REQUIRE(e == hipSuccess, (int)e, "cin_fwd: cannot raise dynamic LDS to %zu B: %s", lds, hipGetErrorString(e));
    }
    const int64_t blocks = dctr_ceil_div(a->batch, p.SB);
    DCTR_REQUIRE(blocks <= 0x7fffffffLL, DCTR_E_DIM, "cin_fwd: batch too large");
    if (rt == 8 && save) DCTR_LAUNCH((cin_kernel<8, true>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    else if (rt == 8) DCTR_LAUNCH((cin_kernel<8, false>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    else if (save) DCTR_LAUNCH((cin_kernel<4, true>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    else DCTR_LAUNCH((cin_kernel<4, false>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, p);
    return dctr_launch_status("dctr_cin_fwd");
}
