// Latency lab for the B=4096 gather: which dependent step costs what?  (bring-up tool, not product code)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int F = 26, E = 16, V = 100000, B = 4096, NB = 32;

__global__ void k_empty(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && out == nullptr) out[0] = 1; }

// ids only: lane (s,q) of wave w loads ids of its 7 fields, sums, stores
__global__ __launch_bounds__(256) void k_ids(const int* __restrict__ ids, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane >> 2;
    const int b = blockIdx.x * 16 + s;
    int acc = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { int j = min(wave + 4 * u, F - 1); acc += ids[j * B + b]; }
    if ((lane & 3) == 0 && wave == 0) out[b] = (float)acc;
}

template <bool WRITE, bool LIN, bool RED>
__global__ __launch_bounds__(256) void k_rows(const int* __restrict__ ids, const float* __restrict__ tables,
                                              const float* __restrict__ lin, float* __restrict__ dnn_in,
                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane >> 2, q = lane & 3;
    const int b = blockIdx.x * 16 + s;
    int row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { int j = min(wave + 4 * u, F - 1); row[u] = ids[j * B + b]; }
    float4 v[8]; float lv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int j = min(wave + 4 * u, F - 1);
        v[u] = *reinterpret_cast<const float4*>(tables + ((size_t)j * V + row[u]) * E + q * 4);
        if (LIN) lv[u] = lin[(size_t)j * V + row[u]];
    }
    float sum = 0.f, l = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int jj = wave + 4 * u;
        if (jj < F) {
            sum += v[u].x + v[u].y + v[u].z + v[u].w;
            if (LIN) l += lv[u];
            if (WRITE) *reinterpret_cast<float4*>(dnn_in + (size_t)b * 432 + jj * E + q * 4) = v[u];
        }
    }
    if (RED) {
        __shared__ float red[4][64];
        red[wave][lane] = sum + l;
        __syncthreads();
        if (wave == 0) { sum = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]; sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); if (q == 0) out[b] = sum; }
    } else {
        if (q == 0) atomicAdd(&out[b], sum + l);
    }
}

// one wave per block (1024 blocks x 64 threads), no cross-wave reduction
__global__ __launch_bounds__(64) void k_rows_1w(const int* __restrict__ ids, const float* __restrict__ tables,
                                                const float* __restrict__ lin, float* __restrict__ dnn_in, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x & 3, s = lane >> 2, q = lane & 3;
    const int b = (blockIdx.x >> 2) * 16 + s;
    int row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { int j = min(wave + 4 * u, F - 1); row[u] = ids[j * B + b]; }
    float4 v[8]; float lv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int j = min(wave + 4 * u, F - 1);
        v[u] = *reinterpret_cast<const float4*>(tables + ((size_t)j * V + row[u]) * E + q * 4);
        lv[u] = lin[(size_t)j * V + row[u]];
    }
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int jj = wave + 4 * u;
        if (jj < F) { sum += v[u].x + lv[u]; *reinterpret_cast<float4*>(dnn_in + (size_t)b * 432 + jj * E + q * 4) = v[u]; }
    }
    if (q == 0) atomicAdd(&out[b], sum);
}

template <typename Fn>
float time_kernel(Fn launch, int reps = 200) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { launch(r, e0, e1); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= reps / 4) t.push_back(ms * 1000.f); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    int* ids; float *tables, *lin, *dnn_in, *out;
    CK(hipMalloc(&ids, (size_t)NB * F * B * 4)); CK(hipMalloc(&tables, (size_t)F * V * E * 4)); CK(hipMalloc(&lin, (size_t)F * V * 4));
    CK(hipMalloc(&dnn_in, (size_t)B * 432 * 4)); CK(hipMalloc(&out, B * 4));
    std::vector<int> h((size_t)NB * F * B); srand(1); for (auto& x : h) x = rand() % V;
    CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(tables, 0, (size_t)F * V * E * 4)); CK(hipMemset(lin, 0, (size_t)F * V * 4)); CK(hipMemset(out, 0, B * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto idp = [&](int r) { return ids + (size_t)(r % NB) * F * B; };
    printf("empty        : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a, b, 0, (int*)out); }));
    printf("ids only     : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_ids, dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), out); }));
    printf("rows         : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<false, false, true>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    printf("rows+write   : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<true, false, true>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    printf("rows+lin     : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<false, true, true>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    printf("rows+lin+wr  : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<true, true, true>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    printf("   (atomics) : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<true, true, false>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    printf("1 wave/block : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(k_rows_1w, dim3(1024), dim3(64), 0, st, a, b, 0, (const int*)idp(r), (const float*)tables, (const float*)lin, dnn_in, out); }));
    // same batch every time (ids + rows L2/MALL-hot)
    printf("hot ids: rows+lin+wr : %6.2f us\n", time_kernel([&](int r, hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((k_rows<true, true, true>), dim3(256), dim3(256), 0, st, a, b, 0, (const int*)ids, (const float*)tables, (const float*)lin, dnn_in, out); }));
    return 0;
}
