// Throughput lab for the stand-alone gather (dctr_embed_gather_fm) on large launches: what does the part give random
// embedding-row reads (64-B / 128-B rows; tables inside / beyond the 256-MiB Infinity Cache), and where does the product
// kernel stand against that?  (bring-up tool, not product code)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I deepctr_amd/csrc scripts/gather_bw_lab.cpp -o scripts/_bin/gather_bw_lab -lrocblas
//   gather_bw_lab [E=16] [V=100000] [B=262144]
#include "../deepctr_amd/csrc/abi.cpp"
#include "../deepctr_amd/csrc/embed_kernels.hip"
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// pure random row reads: every lane reads 16 B of a row (E/4 lanes per row), U rows in flight per lane, sums into one float
template <int U>
__global__ __launch_bounds__(256) void k_read(const int* __restrict__ ids, const float* __restrict__ tables, int F, int64_t V, int E,
                                              int64_t B, float* __restrict__ out) {
    const int lpr = E / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = t / lpr;
    const int q = (int)(t % lpr);
    if (b >= B) return;
    float acc = 0.f;
    for (int f0 = 0; f0 < F; f0 += U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = ids[(int64_t)min(f0 + u, F - 1) * B + b];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(tables + ((int64_t)min(f0 + u, F - 1) * V + id[u]) * E + 4 * q);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// row reads + the linear-table entry of the same id (the product's request pattern: a 4-B read from a separate [V] table per field),
// and the RECORD forms of VERDICT r04 item 4: the linear weight stored behind the row, PITCH floats per record (20: 80-B records, 1.25x
// the memory, half of them straddle a 128-B line; 32: 128-B records, 2x the memory, never straddle) — lane q == 0 reads the fifth piece
template <int U, int PITCH>
__global__ __launch_bounds__(256) void k_read_lin(const int* __restrict__ ids, const float* __restrict__ tables, const float* __restrict__ lin,
                                                  int F, int64_t V, int E, int64_t B, float* __restrict__ out) {
    const int lpr = E / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = t / lpr;
    const int q = (int)(t % lpr);
    if (b >= B) return;
    float acc = 0.f;
    const int pitch = PITCH > 0 ? PITCH : E;
    for (int f0 = 0; f0 < F; f0 += U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = ids[(int64_t)min(f0 + u, F - 1) * B + b];
        float4 v[U];
        float l[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rec = (int64_t)min(f0 + u, F - 1) * V + id[u];
            v[u] = *reinterpret_cast<const float4*>(tables + rec * pitch + 4 * q);
            if (PITCH > 0) l[u] = q == 0 ? tables[rec * pitch + E] : 0.f;
            else l[u] = q == 0 ? lin[rec] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w + l[u];
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the same reads + the concat write (what the stand-alone gather must do at least)
template <int U>
__global__ __launch_bounds__(256) void k_read_write(const int* __restrict__ ids, const float* __restrict__ tables, int F, int64_t V, int E,
                                                    int64_t B, float* __restrict__ dnn_in, int64_t stride) {
    const int lpr = E / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = t / lpr;
    const int q = (int)(t % lpr);
    if (b >= B) return;
    for (int f0 = 0; f0 < F; f0 += U) {
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) id[u] = ids[(int64_t)min(f0 + u, F - 1) * B + b];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(tables + ((int64_t)min(f0 + u, F - 1) * V + id[u]) * E + 4 * q);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (f0 + u < F) *reinterpret_cast<float4*>(dnn_in + b * stride + (int64_t)(f0 + u) * E + 4 * q) = v[u];
    }
}

// streaming write only
__global__ __launch_bounds__(256) void k_write(float* __restrict__ dnn_in, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<float4*>(dnn_in)[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 16;
    const int64_t V = argc > 2 ? atoll(argv[2]) : 100000;
    const int64_t B = argc > 3 ? atoll(argv[3]) : 262144;
    constexpr int F = 26, ND = 13;
    const int64_t stride = ((int64_t)F * E + ND + 3) / 4 * 4;
    float *tables, *lin, *dense, *densew, *dnn_in, *fm, *linl; int* ids; int* status; dctr_field_t* fields;
    const size_t tab_n = (size_t)F * V * E;
    CK(hipMalloc(&tables, tab_n * 4)); CK(hipMemset(tables, 0, tab_n * 4));
    CK(hipMalloc(&lin, (size_t)F * V * 4)); CK(hipMemset(lin, 0, (size_t)F * V * 4));
    std::vector<int> h_ids((size_t)F * B);
    uint32_t st_ = 777u;
    for (auto& x : h_ids) { st_ = st_ * 1664525u + 1013904223u; x = (int)((st_ >> 3) % (uint32_t)V); }
    CK(hipMalloc(&ids, h_ids.size() * 4)); CK(hipMemcpy(ids, h_ids.data(), h_ids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dense, (size_t)B * ND * 4)); CK(hipMemset(dense, 0, (size_t)B * ND * 4));
    CK(hipMalloc(&densew, ND * 4)); CK(hipMemset(densew, 0, ND * 4));
    CK(hipMalloc(&dnn_in, (size_t)B * stride * 4)); CK(hipMalloc(&fm, B * 4)); CK(hipMalloc(&linl, B * 4));
    CK(hipMalloc(&status, 4)); CK(hipMemset(status, 0, 4));
    std::vector<dctr_field_t> fh(F);
    for (int j = 0; j < F; ++j) { fh[j] = dctr_field_t{}; fh[j].table = tables + (size_t)j * V * E; fh[j].lin_table = lin + (size_t)j * V; fh[j].vocab = V; fh[j].dim = E; fh[j].out_offset = j * E; fh[j].in_fm = 1; }
    CK(hipMalloc(&fields, F * sizeof(dctr_field_t))); CK(hipMemcpy(fields, fh.data(), F * sizeof(dctr_field_t), hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* what, double bytes, auto fn) {
        for (int w = 0; w < 2; ++w) fn();
        CK(hipStreamSynchronize(st));
        std::vector<float> t;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 4; ++r) fn();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 4);
        }
        std::sort(t.begin(), t.end());
        const double us = t[2] * 1e3;
        printf("%-58s %9.1f us  %8.1f M rows/s  %7.1f GB/s (%.3f of 8 TB/s)\n", what, us, B / us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
    };
    const double row_b = (double)F * E * 4, id_b = F * 4.0;
    const unsigned blocks = (unsigned)((B * (E / 4) + 255) / 256);
    printf("E=%d V=%lld (tables %.2f GB) B=%lld\n", E, (long long)V, tab_n * 4 / 1e9, (long long)B);
    timeit("random row reads, 2 in flight per lane", B * (row_b + id_b), [&] { hipLaunchKernelGGL(k_read<2>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, fm); });
    timeit("random row reads, 4 in flight per lane", B * (row_b + id_b), [&] { hipLaunchKernelGGL(k_read<4>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, fm); });
    timeit("random row reads, 8 in flight per lane", B * (row_b + id_b), [&] { hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, fm); });
    timeit("random row reads, 13 in flight per lane", B * (row_b + id_b), [&] { hipLaunchKernelGGL(k_read<13>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, fm); });
    {
        timeit("row reads + separate 4-B linear reads, 8 in flight", B * (row_b + 2 * id_b), [&] { hipLaunchKernelGGL((k_read_lin<8, 0>), dim3(blocks), dim3(256), 0, st, ids, tables, lin, F, V, E, B, fm); });
        timeit("row reads + separate 4-B linear reads, 13 in flight", B * (row_b + 2 * id_b), [&] { hipLaunchKernelGGL((k_read_lin<13, 0>), dim3(blocks), dim3(256), 0, st, ids, tables, lin, F, V, E, B, fm); });
        if (E == 16) {
            float* rec;
            CK(hipMalloc(&rec, (size_t)F * V * 32 * 4)); CK(hipMemset(rec, 0, (size_t)F * V * 32 * 4));
            timeit("80-B records [V, 20] (row + linear weight), 8 in flight", B * (row_b + 2 * id_b), [&] { hipLaunchKernelGGL((k_read_lin<8, 20>), dim3(blocks), dim3(256), 0, st, ids, rec, lin, F, V, E, B, fm); });
            timeit("80-B records [V, 20], 13 in flight", B * (row_b + 2 * id_b), [&] { hipLaunchKernelGGL((k_read_lin<13, 20>), dim3(blocks), dim3(256), 0, st, ids, rec, lin, F, V, E, B, fm); });
            timeit("128-B records [V, 32] (2x memory), 8 in flight", B * (row_b + 2 * id_b), [&] { hipLaunchKernelGGL((k_read_lin<8, 32>), dim3(blocks), dim3(256), 0, st, ids, rec, lin, F, V, E, B, fm); });
            CK(hipFree(rec));
        }
    }
    timeit("streaming write of dnn_in", (double)B * stride * 4, [&] { hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, st, dnn_in, B * stride / 4); });
    timeit("row reads (8 in flight) + concat write", B * (2 * row_b + id_b), [&] { hipLaunchKernelGGL(k_read_write<8>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, dnn_in, stride); });
    timeit("row reads (13 in flight) + concat write", B * (2 * row_b + id_b), [&] { hipLaunchKernelGGL(k_read_write<13>, dim3(blocks), dim3(256), 0, st, ids, tables, F, V, E, B, dnn_in, stride); });
    auto prod = [&](float* out_dnn) {
        dctr_gather_fm_args_t g{};
        g.fields = fields; g.ids = ids; g.ids_stride_f = B; g.ids_stride_b = 1; g.ids_is_i64 = 0; g.n_fields = F; g.max_dim = E; g.all_dim4 = 1;
        g.any_hash = 0; g.n_dense = ND; g.dense = dense; g.dense_stride = ND; g.dense_lin_w = densew; g.dense_out_offset = F * E;
        g.dense_copy_cols = ND; g.batch = B; g.status = status; g.dnn_in = out_dnn; g.out_stride = stride; g.fm_logit = fm; g.lin_logit = linl;
        int rc = dctr_embed_gather_fm(&g, st);
        if (rc) { printf("rc=%d %s\n", rc, dctr_last_error()); exit(1); }
    };
    const double alg = B * (row_b + 2 * id_b + ND * 4.0 + 8.0);
    timeit("dctr_embed_gather_fm, logits only (no dnn_in)", alg, [&] { prod(nullptr); });
    timeit("dctr_embed_gather_fm -> dnn_in", alg + (double)B * (F * E + ND) * 4, [&] { prod(dnn_in); });
    return 0;
}
