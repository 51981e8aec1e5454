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
