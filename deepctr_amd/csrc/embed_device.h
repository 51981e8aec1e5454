// Device helpers shared by the gather / pooling kernels (embed_kernels.hip) and by the fused
// gather -> DNN kernel (mlp_kernels.hip): vector row loads, branch-free id reads, in-register Hash,
// descriptor access through the constant address space.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dctr.h"
#include "farmhash_device.h"

namespace {

typedef dctr_gather_fm_args_t GatherParams;   // passed by value in the kernarg segment (scalar loads)

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

// Branch-free id read: one 32-bit load for the low word plus one for the high word (int64 ids) so that
// no control flow separates the id loads of different fields (they must all be in flight together).
struct RawId {
    uint32_t lo, hi;
};
__device__ __forceinline__ RawId load_id(const void* idx, int64_t pos, int is_i64) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(idx);
    RawId r;
    r.lo = w[is_i64 ? 2 * pos : pos];
    r.hi = w[is_i64 ? 2 * pos + 1 : pos];
    return r;
}
__device__ __forceinline__ int64_t id_value(RawId r, int is_i64) {
    return is_i64 ? (int64_t)(((uint64_t)r.hi << 32) | r.lo) : (int64_t)(int32_t)r.lo;
}
__device__ __forceinline__ int64_t read_id(const void* idx, int64_t pos, int is_i64) {
    return id_value(load_id(idx, pos, is_i64), is_i64);
}

__device__ __forceinline__ int64_t resolve_row(int64_t raw, int hash_mode, int is_i64, int64_t vocab) {
    if (hash_mode == 0) return raw;
    return dctr::hash_bucket_id(raw, !is_i64, (uint64_t)vocab, hash_mode == 2);
}

template <int LPR>
__device__ __forceinline__ float reduce_lpr(float v) {
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

struct GatherAcc {
    float lin;
    int oor;
};

// The descriptor array is read-only for the whole launch.  Viewing it through the AMDGPU constant
// address space (4) lets the compiler fetch it with scalar loads (s_load_*, scalar cache) into SGPRs:
// descriptor-dependent branches become scalar branches and never wait on the vector memory counter.
#define DCTR_CONSTANT __attribute__((address_space(4)))
typedef const dctr_field_t DCTR_CONSTANT* cfield_ptr;

struct FieldRegs {
    const float* table;
    const float* lin_table;
    int64_t vocab;
    int dim, out_offset, in_fm, hash_mode, identity;
    int pitch, lin_pitch;      // floats between consecutive rows / linear entries (dctr_field_t.row_pitch: the record form)
};
__device__ __forceinline__ FieldRegs load_field(cfield_ptr F, int j) {
    FieldRegs r;
    r.table = F[j].table;
    r.lin_table = F[j].lin_table;
    r.vocab = F[j].vocab;
    r.dim = F[j].dim;
    r.out_offset = F[j].out_offset;
    r.in_fm = F[j].in_fm;
    r.hash_mode = F[j].hash_mode;
    r.identity = F[j].identity;
    const int rp = F[j].row_pitch;
    r.pitch = rp != 0 ? rp : r.dim;
    r.lin_pitch = rp != 0 ? rp : 1;
    return r;
}

// One chunk of U fields (j0, j0+step, ...).  Phases with NO control flow inside a phase:
//   1. U id loads — addresses come from kernel arguments only (id matrix base + strides), so they are
//      issued before the descriptor scalar loads have returned;
//   2. rows resolved (optional in-register hash), bounds check;
//   3. U row loads + U one-wide linear loads, all in flight before the first use
//      (out-of-range / tail / inactive lanes read a valid dummy address and are masked afterwards);
//   4. FM / linear accumulation and the concat write.
// `store(out_offset_plus_chunk, v)` receives the column of the first element inside a DNN-input row and the VEC
// values: the stand-alone gather writes them to dnn_in in HBM, the fused gather->DNN kernel to its LDS tile.
// `e0`: first element of the pass (rows wider than LPR * VEC elements are walked in passes of that width; the linear entry counts in pass 0).
template <int VEC, int LPR, int U, bool HASH, typename Store>
__device__ __forceinline__ void gather_chunk(const GatherParams& p, int j0, int step, int f_end, int64_t b, bool valid,
                                             int q, float (&sum)[VEC], float (&sq)[VEC], GatherAcc& acc, Store store, int e0 = 0) {
    const int qe = e0 + q * VEC;                      // this lane's first element of a row
    const int last = f_end - 1;
    const int64_t bb = valid ? b : 0;
    RawId raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = min(j0 + u * step, last);
        raw[u] = load_id(p.ids, (int64_t)j * p.ids_stride_f + bb * p.ids_stride_b, p.ids_is_i64);
    }
    cfield_ptr F = (cfield_ptr)p.fields;
    FieldRegs fr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fr[u] = load_field(F, min(j0 + u * step, last));
    int64_t row[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int jj = j0 + u * step;
        const FieldRegs& f = fr[u];
        int64_t r = f.identity ? b : id_value(raw[u], p.ids_is_i64);
        if constexpr (HASH) {
            if (f.hash_mode != 0) r = resolve_row(r, f.hash_mode, p.ids_is_i64, f.vocab);
        }
        const bool live = valid && jj <= last;
        ok[u] = live && (uint64_t)r < (uint64_t)f.vocab;
        if (live && !ok[u]) acc.oor = 1;
        row[u] = ok[u] ? r : 0;
    }
    float v[U][VEC];
    float lv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const FieldRegs& f = fr[u];
        const int qq = (qe < f.dim) ? qe : 0;
        load_vec<VEC>(f.table + row[u] * f.pitch + qq, v[u]);
        const float* lp = f.lin_table != nullptr ? f.lin_table + row[u] * f.lin_pitch : reinterpret_cast<const float*>(p.fields);
        lv[u] = *lp;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const FieldRegs& f = fr[u];
        const bool act = ok[u] && qe < f.dim;
        if (ok[u] && qe == 0 && f.lin_table != nullptr) acc.lin += lv[u];
#pragma unroll
        for (int c = 0; c < VEC; ++c) v[u][c] = act ? v[u][c] : 0.f;
        if (f.in_fm) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                sum[c] += v[u][c];
                sq[c] = fmaf(v[u][c], v[u][c], sq[c]);
            }
        }
        if (f.out_offset >= 0 && valid && (j0 + u * step) <= last && qe < f.dim) store(f.out_offset + qe, v[u]);
    }
}

// fields j = f_begin, f_begin + f_step, ... < f_end: full chunks of 8, then one clamped tail chunk sized to what is left
template <int VEC, int LPR, bool HASH, typename Store>
__device__ __forceinline__ void gather_fields(const GatherParams& p, int f_begin, int f_step, int f_end, int64_t b,
                                              bool valid, int q, float (&sum)[VEC], float (&sq)[VEC], GatherAcc& acc,
                                              Store store, int e0 = 0) {
    int j = f_begin;
    for (; j + 7 * f_step < f_end; j += 8 * f_step)
        gather_chunk<VEC, LPR, 8, HASH>(p, j, f_step, f_end, b, valid, q, sum, sq, acc, store, e0);
    if (j < f_end) {
        const int left = (f_end - j + f_step - 1) / f_step;   // 1..7, wave-uniform
        if (left > 4) gather_chunk<VEC, LPR, 8, HASH>(p, j, f_step, f_end, b, valid, q, sum, sq, acc, store, e0);
        else if (left > 2) gather_chunk<VEC, LPR, 4, HASH>(p, j, f_step, f_end, b, valid, q, sum, sq, acc, store, e0);
        else gather_chunk<VEC, LPR, 2, HASH>(p, j, f_step, f_end, b, valid, q, sum, sq, acc, store, e0);
    }
}

}  // namespace
