// NOTE (round 5): the -DDCTR_*_LAB_* / -DDCTR_LAB_TIMING ablation and stamp switches this harness mentions were removed from the product kernels
// (they live in git history up to 5db6128); without them it still builds and times the shipped kernels.
// Lab harness for dctr_mlp_fwd (bring-up tool): per-dispatch time of the C2 DNN at B=4096 under -DDCTR_LAB_* ablations.
#include "../deepctr_amd/csrc/abi.cpp"
#include "../deepctr_amd/csrc/mlp_kernels.hip"
#ifdef DCTR_LAB_TIMING
__device__ unsigned long long dctr_lab_ts[64];
#endif
#include "../deepctr_amd/csrc/mlp_kernels_rt1.hip"
#undef DCTR_MLP_RT
#include "../deepctr_amd/csrc/mlp_kernels_rt2.hip"
#undef DCTR_MLP_RT
#include "../deepctr_amd/csrc/mlp_kernels_rt4.hip"
#include "../deepctr_amd/csrc/mlp_kernels_ring.hip"
namespace dctr_stream { int try_launch(const dctr_mlp_args_t*, const dctr_gather_fm_args_t*, int, int, bool, hipStream_t, int*) { return 0; } int eligible(const dctr_mlp_args_t*, const dctr_gather_fm_args_t*, bool) { return 0; } }
namespace dctr_chain { int eligible(const dctr_mlp_args_t*, const dctr_gather_fm_args_t*, bool) { return 0; } int launch(const dctr_mlp_args_t*, const dctr_gather_fm_args_t*, int, int, int, hipStream_t) { return -5; }
int plan(int64_t, int, int64_t*, int32_t*, int) { return 0; } size_t bf3_workspace_bytes(int) { return 0; } }
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096;
    const int tile_rows = argc > 2 ? atoi(argv[2]) : 0;
    const int dims[4] = {429, 256, 128, 64};
    float *x, *y, *W[3], *bias[3], *head;
    CK(hipMalloc(&x, (size_t)B * 432 * 4)); CK(hipMalloc(&y, B * 4)); CK(hipMemset(x, 0, (size_t)B * 432 * 4));
    for (int l = 0; l < 3; ++l) { CK(hipMalloc(&W[l], (size_t)110080 * 4 * 33)); CK(hipMemset(W[l], 0, (size_t)110080 * 4 * 33)); CK(hipMalloc(&bias[l], dims[l + 1] * 4)); CK(hipMemset(bias[l], 0, dims[l + 1] * 4)); }
    CK(hipMalloc(&head, 64 * 4)); CK(hipMemset(head, 0, 256));
    int32_t units[3] = {256, 128, 64};
    const float* ks[3] = {W[0], W[1], W[2]}; const float* bs[3] = {bias[0], bias[1], bias[2]};
    dctr_mlp_args_t a{};
    a.x = x; a.batch = B; a.x_stride = 432; a.in_dim = 429; a.n_layers = 3; a.units = units; a.kernels = ks; a.biases = bs;
    a.tile_rows = tile_rows;
    a.activation = DCTR_ACT_RELU; a.has_head = 1; a.sigmoid_out = 1; a.head_w = head; a.y = y;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<float> t;
    for (int r = 0; r < 60; ++r) {
        dctr_profile_next_launch();
        int rc = dctr_mlp_fwd(&a, st);
        if (rc) { printf("rc=%d %s\n", rc, dctr_last_error()); return 1; }
        float ms = dctr_profile_last_ms();
        if (r >= 20) t.push_back(ms * 1000.f);
    }
#ifdef DCTR_LAB_TIMING
    {
        unsigned long long ts[64];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(dctr_lab_ts), sizeof(ts)));
        printf("cycles: stage %llu  L0 %llu  L1 %llu  L2 %llu  head %llu  total %llu\n", ts[1]-ts[0], ts[2]-ts[1], ts[3]-ts[2], ts[4]-ts[3], ts[10]-ts[4], ts[10]-ts[0]);
    }
#endif
    std::sort(t.begin(), t.end());
    printf("%-12s tile_rows=%d B=%d  median %.2f us  min %.2f us\n", LABNAME, tile_rows, B, t[t.size() / 2], t[0]);
    return 0;
}
