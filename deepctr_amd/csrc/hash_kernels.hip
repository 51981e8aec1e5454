// a2 — Hash.call (reference deepctr/layers/utils.py:89-112) as stand-alone kernels.
// Integer/byte work, HBM-bound: one id per lane, coalesced 4/8-byte reads, 8-byte writes.
#include "dctr_common.h"
#include "farmhash_device.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void hash_int_kernel(const T* __restrict__ x, int64_t n, uint64_t num_buckets,
                                                       int mask_zero, int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        out[i] = dctr::hash_bucket_id((int64_t)x[i], sizeof(T) == 4, num_buckets, mask_zero != 0);
    }
}

__global__ __launch_bounds__(256) void hash_bytes_kernel(const uint8_t* __restrict__ bytes,
                                                         const int64_t* __restrict__ offsets, int64_t n,
                                                         uint64_t num_buckets, int mask_zero,
                                                         int64_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t nb = mask_zero ? num_buckets - 1 : num_buckets;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t b = offsets[i], e = offsets[i + 1];
        const uint8_t* s = bytes + b;
        const uint64_t len = (uint64_t)(e - b);
        int64_t h = (int64_t)(dctr::dctr_fp64_bytes(s, len) % nb);
        if (mask_zero) h = (len == 1 && s[0] == (uint8_t)'0') ? 0 : h + 1;
        out[i] = h;
    }
}

// a2 over a whole id matrix [F, B]: one launch for every field of a gather (the pre-pass of dctr_embed_mlp_fwd's persistent kernels,
// which take plain rows).  blockIdx.y = field; a thread takes IPT ids 256 apart (coalesced); the bucket count's reciprocal is
// formed once per thread: h mod nb = h - mulhi64(h, floor((2^64 - 1) / nb)) * nb, corrected upwards at most twice — the 64-bit
// division of the generic `%` costs as much as the rest of the hash.
constexpr int HF_IPT = 8;
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void hash_fields_kernel(const dctr_field_t* __restrict__ fields, const TI* __restrict__ ids,
                                                          int64_t stride_f, int64_t stride_b, int64_t batch, TO* __restrict__ out,
                                                          int64_t out_stride_f) {
    const int f = blockIdx.y;
    const int mode = fields[f].hash_mode;
    const uint64_t nb = (uint64_t)fields[f].vocab - (mode == 2 ? 1u : 0u);
    const TI* src = ids + (int64_t)f * stride_f;
    TO* dst = out + (int64_t)f * out_stride_f;
    const int64_t b0 = (int64_t)blockIdx.x * (256 * HF_IPT) + threadIdx.x;
    if (mode == 0 || fields[f].identity) {
#pragma unroll
        for (int k = 0; k < HF_IPT; ++k) {
            const int64_t b = b0 + 256 * k;
            if (b < batch) dst[b] = (TO)src[b * stride_b];
        }
        return;
    }
    const uint64_t magic = ~0ull / nb;
    TI x[HF_IPT];
#pragma unroll
    for (int k = 0; k < HF_IPT; ++k) {
        const int64_t b = b0 + 256 * k;
        x[k] = b < batch ? src[b * stride_b] : (TI)0;
    }
#pragma unroll
    for (int k = 0; k < HF_IPT; ++k) {
        const int64_t b = b0 + 256 * k;
        const dctr::Packed24 s = sizeof(TI) == 4 ? dctr::decimal_ascii_i32_fast((int32_t)x[k]) : dctr::decimal_ascii((int64_t)x[k]);
        const uint64_t h = dctr::dctr_fp64_packed(s);
        uint64_t r = h - __umul64hi(h, magic) * nb;
        while (r >= nb) r -= nb;
        if (mode == 2) r = (x[k] != 0) ? r + 1 : 0;
        if (b < batch) dst[b] = (TO)r;
    }
}

int check_common(const void* x, int64_t n, int64_t num_buckets, int mask_zero, const void* out) {
    DCTR_REQUIRE(n >= 0, DCTR_E_DIM, "hash_bucket: n=%lld < 0", (long long)n);
    if (n == 0) return 1;  // nothing to do
    DCTR_REQUIRE(x && out, DCTR_E_NULL, "hash_bucket: null pointer");
    DCTR_REQUIRE(num_buckets - (mask_zero ? 1 : 0) >= 1, DCTR_E_DIM,
                 "hash_bucket: num_buckets=%lld leaves no bucket (mask_zero=%d)", (long long)num_buckets, mask_zero);
    return 0;
}

inline unsigned grid_for(int64_t n) {
    int64_t g = dctr_ceil_div(n, 256);
    if (g > 2048) g = 2048;  // grid-stride the rest (256 CUs x 8 blocks)
    return (unsigned)g;
}

}  // namespace

extern "C" int dctr_hash_bucket_i32(const int32_t* x, int64_t n, int64_t num_buckets, int mask_zero, int64_t* out,
                                    void* stream) {
    int c = check_common(x, n, num_buckets, mask_zero, out);
    if (c < 0) return c;
    if (c == 1) return DCTR_OK;
    DCTR_LAUNCH(hash_int_kernel<int32_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n,
                       (uint64_t)num_buckets, mask_zero, out);
    return dctr_launch_status("dctr_hash_bucket_i32");
}

extern "C" int dctr_hash_bucket_i64(const int64_t* x, int64_t n, int64_t num_buckets, int mask_zero, int64_t* out,
                                    void* stream) {
    int c = check_common(x, n, num_buckets, mask_zero, out);
    if (c < 0) return c;
    if (c == 1) return DCTR_OK;
    DCTR_LAUNCH(hash_int_kernel<int64_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n,
                       (uint64_t)num_buckets, mask_zero, out);
    return dctr_launch_status("dctr_hash_bucket_i64");
}

extern "C" int dctr_hash_bucket_bytes(const uint8_t* bytes, const int64_t* offsets, int64_t n, int64_t num_buckets,
                                      int mask_zero, int64_t* out, void* stream) {
    int c = check_common(bytes, n, num_buckets, mask_zero, out);
    if (c < 0) return c;
    if (c == 1) return DCTR_OK;
    DCTR_REQUIRE(offsets, DCTR_E_NULL, "hash_bucket_bytes: null offsets");
    DCTR_LAUNCH(hash_bytes_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, bytes, offsets, n,
                       (uint64_t)num_buckets, mask_zero, out);
    return dctr_launch_status("dctr_hash_bucket_bytes");
}

extern "C" int dctr_hash_fields(const dctr_field_t* fields, int32_t n_fields, const void* ids, int64_t ids_stride_f,
                                int64_t ids_stride_b, int32_t ids_is_i64, int64_t batch, void* out, int64_t out_stride_f,
                                int32_t out_is_i64, void* stream) {
    DCTR_REQUIRE(n_fields >= 0 && batch >= 0, DCTR_E_DIM, "hash_fields: n_fields=%d batch=%lld", n_fields, (long long)batch);
    if (n_fields == 0 || batch == 0) return DCTR_OK;
    DCTR_REQUIRE(fields && ids && out, DCTR_E_NULL, "hash_fields: null pointer");
    DCTR_REQUIRE(n_fields <= 65535, DCTR_E_DIM, "hash_fields: n_fields=%d > 65535", n_fields);
    const int64_t bx = dctr_ceil_div(batch, (int64_t)256 * HF_IPT);
    DCTR_REQUIRE(bx <= 0x7fffffffLL, DCTR_E_DIM, "hash_fields: batch too large");
    const dim3 grid((unsigned)bx, (unsigned)n_fields), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (ids_is_i64 && out_is_i64)
        DCTR_LAUNCH((hash_fields_kernel<int64_t, int64_t>), grid, block, 0, st, fields, (const int64_t*)ids, ids_stride_f, ids_stride_b, batch, (int64_t*)out, out_stride_f);
    else if (ids_is_i64)
        DCTR_LAUNCH((hash_fields_kernel<int64_t, int32_t>), grid, block, 0, st, fields, (const int64_t*)ids, ids_stride_f, ids_stride_b, batch, (int32_t*)out, out_stride_f);
    else if (out_is_i64)
        DCTR_LAUNCH((hash_fields_kernel<int32_t, int64_t>), grid, block, 0, st, fields, (const int32_t*)ids, ids_stride_f, ids_stride_b, batch, (int64_t*)out, out_stride_f);
    else
        DCTR_LAUNCH((hash_fields_kernel<int32_t, int32_t>), grid, block, 0, st, fields, (const int32_t*)ids, ids_stride_f, ids_stride_b, batch, (int32_t*)out, out_stride_f);
    return dctr_launch_status("dctr_hash_fields");
}
