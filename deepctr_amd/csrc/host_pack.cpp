// Host input pipeline: pack the caller's feature columns into the feature-major staging matrix
// that crosses PCIe.  The reference hands Keras a dict of ndarrays (examples/run_classification_criteo.py:40-50) and lets
// TF copy each one; here the columns of one row range are converted (int32/int64/float32/float64 -> the staging dtype)
// and written into ONE page-locked [n_cols, n_rows] matrix by a few host threads, so the copy engine gets one large
// transfer per chunk while the next chunk is being packed.  Pure host code: no HIP calls.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "dctr.h"
#include "dctr_common.h"

namespace {

constexpr int64_t BLOCK_ROWS = 1 << 16;

template <typename S, typename D>
void convert_block(const char* src, int64_t stride, D* dst, int64_t n) {
    if (stride == (int64_t)sizeof(S)) {
        const S* s = reinterpret_cast<const S*>(src);
        if constexpr (std::is_same<S, D>::value) {
            std::memcpy(dst, s, (size_t)n * sizeof(D));
        } else {
            for (int64_t i = 0; i < n; ++i) dst[i] = static_cast<D>(s[i]);
        }
    } else {
        for (int64_t i = 0; i < n; ++i) {
            S v;
            std::memcpy(&v, src + i * stride, sizeof(S));
            dst[i] = static_cast<D>(v);
        }
    }
}

template <typename D>
void pack_block(const dctr_host_col_t& c, int64_t row, D* dst, int64_t n) {
    const char* src = static_cast<const char*>(c.src) + row * c.stride_bytes;
    switch (c.kind) {
        case DCTR_HOST_I32: convert_block<int32_t, D>(src, c.stride_bytes, dst, n); break;
        case DCTR_HOST_I64: convert_block<int64_t, D>(src, c.stride_bytes, dst, n); break;
        case DCTR_HOST_F32: convert_block<float, D>(src, c.stride_bytes, dst, n); break;
        default: convert_block<double, D>(src, c.stride_bytes, dst, n); break;
    }
}

template <typename D>
void pack_all(const dctr_host_col_t* cols, int n_cols, int64_t row_lo, int64_t n_rows, D* dst, int64_t dst_stride, int n_threads) {
    const int64_t blocks_per_col = dctr_ceil_div(n_rows, BLOCK_ROWS);
    const int64_t jobs = blocks_per_col * n_cols;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t j = next.fetch_add(1, std::memory_order_relaxed);
            if (j >= jobs) return;
            const int c = (int)(j / blocks_per_col);
            const int64_t r0 = (j - (int64_t)c * blocks_per_col) * BLOCK_ROWS;
            const int64_t n = n_rows - r0 < BLOCK_ROWS ? n_rows - r0 : BLOCK_ROWS;
            pack_block<D>(cols[c], row_lo + r0, dst + (int64_t)c * dst_stride + r0, n);
        }
    };
    int T = n_threads < 1 ? 1 : n_threads;
    if ((int64_t)T > jobs) T = (int)jobs;
    std::vector<std::thread> pool;
    pool.reserve(T > 1 ? T - 1 : 0);
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" int dctr_host_pack_columns(const dctr_host_col_t* cols, int32_t n_cols, int64_t row_lo, int64_t n_rows, void* dst,
                                      int64_t dst_col_stride, int32_t dst_kind, int32_t n_threads) {
    DCTR_REQUIRE(n_cols >= 0 && row_lo >= 0 && n_rows >= 0 && dst_col_stride >= n_rows, DCTR_E_DIM, "host_pack_columns: bad sizes");
    DCTR_REQUIRE(dst_kind == DCTR_HOST_I32 || dst_kind == DCTR_HOST_I64 || dst_kind == DCTR_HOST_F32, DCTR_E_ENUM,
                 "host_pack_columns: dst_kind %d (int32, int64 or float32)", dst_kind);
    if (n_cols == 0 || n_rows == 0) return DCTR_OK;
    DCTR_REQUIRE(cols != nullptr && dst != nullptr, DCTR_E_NULL, "host_pack_columns: null pointer");
    for (int c = 0; c < n_cols; ++c) {
        DCTR_REQUIRE(cols[c].src != nullptr, DCTR_E_NULL, "host_pack_columns: column %d has no data", c);
        DCTR_REQUIRE(cols[c].kind >= DCTR_HOST_I32 && cols[c].kind <= DCTR_HOST_F64, DCTR_E_ENUM, "host_pack_columns: column %d kind %d",
                     c, cols[c].kind);
    }
    if (dst_kind == DCTR_HOST_I32)
        pack_all<int32_t>(cols, n_cols, row_lo, n_rows, static_cast<int32_t*>(dst), dst_col_stride, n_threads);
    else if (dst_kind == DCTR_HOST_I64)
        pack_all<int64_t>(cols, n_cols, row_lo, n_rows, static_cast<int64_t*>(dst), dst_col_stride, n_threads);
    else
        pack_all<float>(cols, n_cols, row_lo, n_rows, static_cast<float*>(dst), dst_col_stride, n_threads);
    return DCTR_OK;
}
