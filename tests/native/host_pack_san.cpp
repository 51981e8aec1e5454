// Harness for tests/test_cabi.py::test_host_packer_is_clean_under_asan_ubsan: drives dctr_host_pack_columns (row ranges, strided
// float64 column, every destination dtype, 1 / 3 / 8 threads) in a binary built with -fsanitize=address,undefined.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dctr.h"
int main() {
    const int64_t n = 200003;
    std::vector<int32_t> a(n); std::vector<int64_t> b(n); std::vector<float> c(n); std::vector<double> d(3 * n);
    for (int64_t i = 0; i < n; ++i) { a[i] = (int32_t)i; b[i] = i * 3; c[i] = (float)i * 0.5f; d[3 * i + 1] = (double)i; }
    dctr_host_col_t cols[4] = {{a.data(), 4, DCTR_HOST_I32, 0}, {b.data(), 8, DCTR_HOST_I64, 0}, {c.data(), 4, DCTR_HOST_F32, 0},
                               {&d[1], 24, DCTR_HOST_F64, 0}};
    for (int kind = 0; kind < 3; ++kind) {
        for (int threads : {1, 3, 8}) {
            const int64_t lo = 12345, m = 150001;
            const size_t es = kind == 1 ? 8 : 4;
            std::vector<char> dst((size_t)4 * (m + 7) * es);
            if (dctr_host_pack_columns(cols, 4, lo, m, dst.data(), m + 7, kind, threads) != 0) { puts(dctr_last_error()); return 1; }
        }
    }
    puts("sanitized ok");
    return 0;
}
