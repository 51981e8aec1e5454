// NOTE (round 5): the -DDCTR_*_LAB_* / -DDCTR_LAB_TIMING ablation and stamp switches this harness mentions were removed from the product kernels
// (they live in git history up to 5db6128); without them it still builds and times the shipped kernels.
// Lab harness for dctr_din_attn_pool_fwd (bring-up tool): per-dispatch time and cycle stamps of the C4 shape.
#include "../deepctr_amd/csrc/abi.cpp"
#include "../deepctr_amd/csrc/din_kernels.hip"
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048, T = argc > 2 ? atoi(argv[2]) : 50, E = argc > 3 ? atoi(argv[3]) : 64;
    const int act = argc > 4 ? atoi(argv[4]) : DCTR_ACT_DICE;
    int32_t units[2] = {80, 40};
    float *q, *k, *out, *ws, *W[2], *bias[2], *da[2], *dm[2], *dv[2], *ok, *ob; uint8_t* mask;
    CK(hipMalloc(&q, (size_t)B * E * 4)); CK(hipMemset(q, 0, (size_t)B * E * 4));
    CK(hipMalloc(&k, (size_t)B * T * E * 4)); CK(hipMemset(k, 0, (size_t)B * T * E * 4));
    CK(hipMalloc(&mask, (size_t)B * T)); CK(hipMemset(mask, 1, (size_t)B * T));
    CK(hipMalloc(&out, (size_t)B * E * 4)); CK(hipMalloc(&ws, (size_t)B * T * 4));
    int K = 4 * E;
    for (int l = 0; l < 2; ++l) {
        CK(hipMalloc(&W[l], (size_t)K * units[l] * 4)); CK(hipMemset(W[l], 0, (size_t)K * units[l] * 4));
        for (float** pp : {&bias[l], &da[l], &dm[l], &dv[l]}) { CK(hipMalloc(pp, units[l] * 4)); CK(hipMemset(*pp, 0, units[l] * 4)); }
        K = units[l];
    }
    CK(hipMalloc(&ok, 40 * 4)); CK(hipMemset(ok, 0, 160)); CK(hipMalloc(&ob, 4)); CK(hipMemset(ob, 0, 4));
    const float* ks[2] = {W[0], W[1]}; const float* bs[2] = {bias[0], bias[1]};
    const float* a1[2] = {da[0], da[1]}; const float* a2[2] = {dm[0], dm[1]}; const float* a3[2] = {dv[0], dv[1]};
    dctr_din_attn_args_t a{};
    a.query = q; a.keys = k; a.key_mask = mask; a.batch = B; a.maxlen = T; a.dim = E; a.n_layers = 2; a.activation = act;
    a.units = units; a.kernels = ks; a.biases = bs; a.dice_alpha = a1; a.dice_mean = a2; a.dice_var = a3; a.dice_eps = 1e-9f;
    a.out_kernel = ok; a.out_bias = ob; a.out = out; a.out_stride = E; a.workspace = ws; a.workspace_bytes = (size_t)B * T * 4;
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<float> t;
    for (int r = 0; r < 30; ++r) {
        dctr_profile_next_launch();
        int rc = dctr_din_attn_pool_fwd(&a, st);
        if (rc) { printf("rc=%d %s\n", rc, dctr_last_error()); return 1; }
        CK(hipStreamSynchronize(st));
        float ms = dctr_profile_last_ms();
        if (r >= 10) t.push_back(ms * 1000.f);
    }
#ifdef DCTR_LAB_TIMING
    unsigned long long ts[64];
    CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(dctr_din_ts), sizeof(ts)));
    printf("cycles: weights->LDS %llu  barrier %llu | last tile: stage %llu  L0 %llu  L1.. %llu | total %llu\n", ts[1] - ts[0],
           ts[2] - ts[1], ts[3] - ts[2] > (1ull << 40) ? 0 : ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4], ts[5] - ts[0]);
#endif
    std::sort(t.begin(), t.end());
    printf("din score kernel B=%d T=%d E=%d act=%d  median %.2f us  min %.2f us\n", B, T, E, act, t[t.size() / 2], t[0]);
    return 0;
}
