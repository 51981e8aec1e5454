#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* cyc, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + i, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, unsigned long long* cyc, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + i, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// sustained rate: all CUs, `iters` x 8 independent MFMAs per wave, wall time by HIP events -> TFLOP/s and clock
__global__ __launch_bounds__(256) void ksus(float* out, unsigned long long* cyc, float a, float b, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + i, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(blockIdx.x * 256 + threadIdx.x) & 65535] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    unsigned long long h;
#define RUN(K, N, G) { hipLaunchKernelGGL((K<N>), dim3(G), dim3(256), 0, 0, out, cyc, 1.0f, 2.0f); CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); printf("%s<%d> grid %d: %.1f cycles per MFMA\n", #K, N, G, (double)h / (1000.0 * N)); }
    RUN(k16, 1, 256) RUN(k16, 2, 256) RUN(k16, 4, 256) RUN(k16, 8, 256) RUN(k16, 4, 1)
    RUN(k32, 1, 256) RUN(k32, 2, 256) RUN(k32, 4, 256)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {256, 512, 1024}) for (int iters : {2000, 20000, 100000}) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ksus, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f, 2.0f, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        const double flop = (double)grid * 4 * iters * 8 * 2048.0;
        printf("sustained f32 MFMA: grid %4d iters %6d: %8.1f us  %6.1f TFLOP/s  block0 clock %.2f GHz\n", grid, iters, ms * 1e3,
               flop / (ms * 1e-3) / 1e12, (double)h / (ms * 1e-3) / 1e9);
    }
    return 0;
}
