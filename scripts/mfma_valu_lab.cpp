// Do fp32 MFMAs and vector-ALU instructions overlap on a CDNA4 SIMD?  ("fp32 MFMAs execute on the vector lanes" — the rule
// that shaped the DNN kernels and the reason in-launch sequence pooling does not pay, §4.5.)  One workgroup of 8 waves per CU (two per SIMD,
// the row-chained kernel's occupancy); every iteration issues 8 independent v_mfma_f32_16x16x4_f32 (256 cycles of matrix pipe) and V
// independent v_fma_f32.  mode 0: every wave issues both; mode 1: the SIMD's first wave issues only the MFMAs, its second only the FMAs
// (twice as many, so that the SIMD sees the same totals).  If the two overlapped, time per iteration would stay flat in V until the
// vector pipe itself saturates; if they exclude each other it grows by (cycles per v_fma) x V from V = 0 on.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_lab.cpp -o /tmp/mfma_valu_lab && /tmp/mfma_valu_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

template <int V, int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float a, float b, int iters) {
    extern __shared__ float pad[];                 // (100 KiB: one workgroup per CU)
    f32x4 acc[8];
    float x[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; x[i] = (float)i; }
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || (wave < 4), do_valu = MODE == 0 || (wave >= 4);      // waves w and w + 4 share a SIMD
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int v = (V * i) / 8; v < (V * (i + 1)) / 8; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(a), "v"(b));
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 2 * V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(a), "v"(b));
            if (V == 0) asm volatile("s_nop 0");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i];
    out[(blockIdx.x * 512 + threadIdx.x) & 65535] = s + pad[threadIdx.x & 7];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int V, int MODE>
int run(float* out, unsigned long long* cyc, int iters) {
    CK(hipFuncSetAttribute((const void*)k<V, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V, MODE>), dim3(256), dim3(512), 100 * 1024, 0, out, cyc, 1.0f, 2.0f, iters / 10);      // warm-up (clock)
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<V, MODE>), dim3(256), dim3(512), 100 * 1024, 0, out, cyc, 1.0f, 2.0f, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    const double cyc_it = (double)h / iters;                                        // wave 0 of workgroup 0 (mode 1: an MFMA wave)
    const double tf = 256.0 * 4 * (MODE == 0 ? 2 : 1) * (double)iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
    printf("mode %d  V = %2d v_fma per 8 MFMAs and wave%s: %7.1f cycles per iteration (8 MFMAs = 256 of matrix pipe per wave)  %6.1f TFLOP/s  %.2f GHz\n",
           MODE, V, MODE ? " pair" : "", cyc_it, tf, (double)h / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 65536 * 4));
    CK(hipMalloc(&cyc, 8));
    const int iters = 200000;
    run<0, 0>(out, cyc, iters); run<2, 0>(out, cyc, iters); run<4, 0>(out, cyc, iters); run<8, 0>(out, cyc, iters);
    run<16, 0>(out, cyc, iters); run<32, 0>(out, cyc, iters); run<64, 0>(out, cyc, iters);
    run<0, 1>(out, cyc, iters); run<4, 1>(out, cyc, iters); run<8, 1>(out, cyc, iters); run<16, 1>(out, cyc, iters);
    run<32, 1>(out, cyc, iters); run<64, 1>(out, cyc, iters);
    return 0;
}
