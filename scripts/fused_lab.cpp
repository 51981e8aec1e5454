// NOTE (round 5): the -DDCTR_*_LAB_* / -DDCTR_LAB_TIMING ablation and stamp switches this harness mentions were removed from the product kernels
// (they live in git history up to 5db6128); without them it still builds and times the shipped kernels.
// Lab harness for dctr_embed_mlp_fwd at the C2 shape (bring-up tool): per-dispatch time + cycle stamps of block 0.
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
    const int tile_rows = argc > 2 ? atoi(argv[2]) : 32;
    constexpr int F = 26, E = 16, V = 100000, ND = 13, NB = 16;
    const int dims[4] = {429, 256, 128, 64};
    float *tables, *lin, *dense, *densew, *y, *W[3], *bias[3], *head, *gb; int* ids; int* status; dctr_field_t* fields;
    CK(hipMalloc(&tables, (size_t)F * V * E * 4)); CK(hipMemset(tables, 0, (size_t)F * V * E * 4));
    CK(hipMalloc(&lin, (size_t)F * V * 4)); CK(hipMemset(lin, 0, (size_t)F * V * 4));
    CK(hipMalloc(&ids, (size_t)NB * F * B * 4));
    std::vector<int> h((size_t)NB * F * B); srand(1); for (auto& x : h) x = rand() % V;
    CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dense, (size_t)B * ND * 4)); CK(hipMemset(dense, 0, (size_t)B * ND * 4));
    CK(hipMalloc(&densew, ND * 4)); CK(hipMemset(densew, 0, ND * 4));
    CK(hipMalloc(&y, B * 4)); CK(hipMalloc(&status, 4)); CK(hipMemset(status, 0, 4));
    std::vector<dctr_field_t> fh(F);
    for (int j = 0; j < F; ++j) { fh[j] = dctr_field_t{}; fh[j].table = tables + (size_t)j * V * E; fh[j].lin_table = lin + (size_t)j * V; fh[j].vocab = V; fh[j].dim = E; fh[j].out_offset = j * E; fh[j].in_fm = 1; }
    CK(hipMalloc(&fields, F * sizeof(dctr_field_t))); CK(hipMemcpy(fields, fh.data(), F * sizeof(dctr_field_t), hipMemcpyHostToDevice));
    for (int l = 0; l < 3; ++l) { CK(hipMalloc(&W[l], (size_t)dims[l] * dims[l + 1] * 4)); CK(hipMemset(W[l], 0, (size_t)dims[l] * dims[l + 1] * 4)); CK(hipMalloc(&bias[l], dims[l + 1] * 4)); CK(hipMemset(bias[l], 0, dims[l + 1] * 4)); }
    CK(hipMalloc(&head, 64 * 4)); CK(hipMemset(head, 0, 256)); CK(hipMalloc(&gb, 4)); CK(hipMemset(gb, 0, 4));
    int32_t units[3] = {256, 128, 64};
    const float* ks[3] = {W[0], W[1], W[2]}; const float* bs[3] = {bias[0], bias[1], bias[2]};
    dctr_mlp_args_t a{};
    a.batch = B; a.in_dim = 429; a.n_layers = 3; a.units = units; a.kernels = ks; a.biases = bs; a.tile_rows = tile_rows;
    a.activation = DCTR_ACT_RELU; a.has_head = 1; a.sigmoid_out = 1; a.head_w = head; a.global_bias = gb; a.y = y;
    dctr_gather_fm_args_t g{};
    g.fields = fields; g.ids_stride_f = B; g.ids_stride_b = 1; g.ids_is_i64 = 0; g.n_fields = F; g.max_dim = E; g.all_dim4 = 1;
    g.any_hash = 0; g.n_dense = ND; g.dense = dense; g.dense_stride = ND; g.dense_lin_w = densew; g.dense_out_offset = F * E;
    g.dense_copy_cols = ND; g.batch = B; g.status = status; g.split_col = 256; g.split_field = 16;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<float> t;
    for (int r = 0; r < 80; ++r) {
        g.ids = ids + (size_t)(r % NB) * F * B;
        dctr_profile_next_launch();
        int rc = dctr_embed_mlp_fwd(&g, &a, 1, 1, st);
        if (rc) { printf("rc=%d %s\n", rc, dctr_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        float ms = dctr_profile_last_ms();
        if (r >= 20) t.push_back(ms * 1000.f);
    }
#ifdef DCTR_LAB_TIMING
    {
        unsigned long long ts[64];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(dctr_lab_ts), sizeof(ts)));
        printf("cycles from kernel start:");
        for (int i = 1; i < 40; ++i) if (ts[i] > ts[0] && ts[i] - ts[0] < (1ull << 30)) printf("  [%d] %llu", i, ts[i] - ts[0]);
        printf("\n");
    }
#endif
    std::sort(t.begin(), t.end());
    printf("fused tile_rows=%d B=%d  median %.2f us  min %.2f us\n", tile_rows, B, t[t.size() / 2], t[0]);
    return 0;
}
