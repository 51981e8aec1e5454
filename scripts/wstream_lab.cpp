// How fast can EVERY CU pull the same 603 KB of DNN weights out of L2 (the bound of a 16-rows-per-CU forward)?  256 workgroups x 8 waves, each
// streams the buffer once: (a) global_load_lds_dwordx4 into a ring of LDS slots with one s_barrier per 16 / 32 KiB chunk (the row-chained
// kernel's mechanism), (b) global_load_dwordx4 into registers (the 32-row kernel's).  No MFMA, no gather: the floor of the weight stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/wstream_lab.cpp -o scripts/_bin/wstream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int CHUNK_KB, int NSLOT>
__global__ __launch_bounds__(512, 1) void k_dma(const float* __restrict__ w, int n_chunks, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CF = CHUNK_KB * 256;                 // floats per chunk
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    auto dma = [&](int c, int slot) {
        const char* base = reinterpret_cast<const char*>(w) + (size_t)c * CHUNK_KB * 1024;
#pragma unroll
        for (int pc = 0; pc < CHUNK_KB; pc += 8) {     // wave moves pieces wave, wave + 8, ... of 1 KiB
            const uint32_t voff = (uint32_t)((pc + wave) * 1024 + 16 * lane);
            const uint32_t lds_addr = (uint32_t)(size_t)(lds_ptr_t)(smem + slot * CF + (pc + wave) * 256);
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
        }
    };
    float acc = 0.f;
    for (int s = 0; s < NSLOT - 1; ++s) dma(min(s, n_chunks - 1), s);
    int slot = 0;
    for (int c = 0; c < n_chunks; ++c) {
        int nx = slot + NSLOT - 1; nx = nx >= NSLOT ? nx - NSLOT : nx;
        // chunk c has landed for everyone once every wave waited for its own share and all passed the barrier
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"((NSLOT - 2) * (CHUNK_KB / 8)) : "memory");
        __syncthreads();
        dma(min(c + NSLOT - 1, n_chunks - 1), nx);
        acc += smem[slot * CF + threadIdx.x];          // touch the chunk
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 1234.5f) out[0] = acc;
}

template <int U>
__global__ __launch_bounds__(512, 1) void k_reg(const float4* __restrict__ w, int n16, float* out) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n16; i += 512 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[min(i + 512 * u, n16 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1234.5f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const size_t bytes = 608 * 1024;                   // 429 x 256 + 256 x 128 + 128 x 64 floats = 603 KB, rounded to whole 32-KiB chunks
    float *w, *out;
    CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0, bytes)); CK(hipMalloc(&out, 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* what, auto fn) {
        for (int i = 0; i < 5; ++i) fn();
        CK(hipStreamSynchronize(st));
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 20; ++r) fn();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 20);
        }
        std::sort(t.begin(), t.end());
        const double us = t[3] * 1e3;
        printf("%-64s %7.2f us per launch  %6.2f TB/s aggregate  %5.1f B/clk/CU at 2.4 GHz\n", what, us, blocks * (double)bytes / us * 1e-6,
               (double)bytes / (us * 2400.0));
    };
    printf("%d workgroups x 512 threads, every one streams the same %zu KB\n", blocks, bytes / 1024);
    CK(hipFuncSetAttribute((const void*)k_dma<16, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * 1024));
    CK(hipFuncSetAttribute((const void*)k_dma<32, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32 * 1024));
    CK(hipFuncSetAttribute((const void*)k_dma<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 1024));
    timeit("LDS-DMA, 16-KiB chunks, ring of 3", [&] { hipLaunchKernelGGL((k_dma<16, 3>), dim3(blocks), dim3(512), 3 * 16 * 1024, st, w, (int)(bytes / 16384), out); });
    timeit("LDS-DMA, 32-KiB chunks, ring of 3", [&] { hipLaunchKernelGGL((k_dma<32, 3>), dim3(blocks), dim3(512), 3 * 32 * 1024, st, w, (int)(bytes / 32768), out); });
    timeit("LDS-DMA, 32-KiB chunks, ring of 4", [&] { hipLaunchKernelGGL((k_dma<32, 4>), dim3(blocks), dim3(512), 4 * 32 * 1024, st, w, (int)(bytes / 32768), out); });
    timeit("global_load_dwordx4 -> registers, 4 in flight per lane", [&] { hipLaunchKernelGGL(k_reg<4>, dim3(blocks), dim3(512), 0, st, (const float4*)w, (int)(bytes / 16), out); });
    timeit("global_load_dwordx4 -> registers, 8 in flight per lane", [&] { hipLaunchKernelGGL(k_reg<8>, dim3(blocks), dim3(512), 0, st, (const float4*)w, (int)(bytes / 16), out); });
    timeit("global_load_dwordx4 -> registers, 16 in flight per lane", [&] { hipLaunchKernelGGL(k_reg<16>, dim3(blocks), dim3(512), 0, st, (const float4*)w, (int)(bytes / 16), out); });
    timeit("empty-ish launch (1 chunk)", [&] { hipLaunchKernelGGL((k_dma<32, 3>), dim3(blocks), dim3(512), 3 * 32 * 1024, st, w, 1, out); });
    return 0;
}
