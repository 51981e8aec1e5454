// Lab harness for dctr_cin_fwd (bring-up tool): per-launch time of the C3 shape, with and without the folded layer 0, checked
// against a double-precision host restatement of CIN.call (reference deepctr/layers/interaction.py:277-325) on a few samples.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I deepctr_amd/csrc [-DCIN_SRC='"path"'] [-DCIN_NB=4] scripts/cin_lab.cpp -o scripts/_bin/cin_lab
#include "../deepctr_amd/csrc/abi.cpp"
#ifndef CIN_SRC
#define CIN_SRC "../deepctr_amd/csrc/cin_kernels.hip"
#endif
#include CIN_SRC
#include <vector>
#include <algorithm>
#include <cmath>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static void host_cin(const float* x, int F0, int D, const std::vector<int>& Hs, const std::vector<std::vector<float>>& W,
                     const std::vector<std::vector<float>>& bias, bool split_half, std::vector<double>& out) {
    std::vector<double> x0((size_t)F0 * D), xk;
    for (int i = 0; i < F0 * D; ++i) x0[i] = x[i];
    xk = x0;
    int Fk = F0;
    out.clear();
    for (size_t k = 0; k < Hs.size(); ++k) {
        const int H = Hs[k];
        std::vector<double> y((size_t)H * D);
        for (int h = 0; h < H; ++h)
            for (int d = 0; d < D; ++d) {
                double acc = 0;
                for (int i = 0; i < F0; ++i)
                    for (int j = 0; j < Fk; ++j) acc += x0[i * D + d] * xk[j * D + d] * (double)W[k][(size_t)(i * Fk + j) * H + h];
                acc += bias[k][h];
                y[(size_t)h * D + d] = acc > 0 ? acc : 0;       // relu
            }
        const bool last = k + 1 == Hs.size();
        const int Hn = split_half ? (last ? 0 : H / 2) : (last ? 0 : H), d0 = split_half ? (last ? 0 : H / 2) : 0;
        for (int h = d0; h < H; ++h) {
            double s = 0;
            for (int d = 0; d < D; ++d) s += y[(size_t)h * D + d];
            out.push_back(s);
        }
        xk.assign(y.begin(), y.begin() + (size_t)Hn * D);
        Fk = Hn;
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, F0 = argc > 2 ? atoi(argv[2]) : 26, D = argc > 3 ? atoi(argv[3]) : 16;
    const int H0 = argc > 4 ? atoi(argv[4]) : 128, H1 = argc > 5 ? atoi(argv[5]) : 128, H2 = argc > 6 ? atoi(argv[6]) : 0;
    const int reps = 40;
    std::vector<int> Hs = {H0, H1};
    if (H2 > 0) Hs.push_back(H2);
    const int L = (int)Hs.size();
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx((size_t)B * F0 * D);
    for (auto& v : hx) v = 0.5f * nd(rng);
    std::vector<std::vector<float>> hW(L), hb(L);
    int Fk = F0;
    for (int k = 0; k < L; ++k) {
        hW[k].resize((size_t)F0 * Fk * Hs[k]);
        const float sc = 1.f / std::sqrt((float)(F0 * Fk));
        for (auto& v : hW[k]) v = sc * nd(rng);
        hb[k].resize(Hs[k]);
        for (auto& v : hb[k]) v = 0.1f * nd(rng);
        Fk = k + 1 == L ? 0 : Hs[k] / 2;
    }
    float *x, *out, *ws = nullptr;
    std::vector<float*> dW(L), db(L);
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    for (int k = 0; k < L; ++k) {
        CK(hipMalloc(&dW[k], hW[k].size() * 4)); CK(hipMemcpy(dW[k], hW[k].data(), hW[k].size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&db[k], hb[k].size() * 4)); CK(hipMemcpy(db[k], hb[k].data(), hb[k].size() * 4, hipMemcpyHostToDevice));
    }
    dctr_cin_args_t a{};
    std::vector<int32_t> ls(Hs.begin(), Hs.end());
    std::vector<const float*> fp(dW.begin(), dW.end()), bp(db.begin(), db.end());
    a.x = x; a.batch = B; a.x_stride = (int64_t)F0 * D; a.fields = F0; a.dim = D; a.n_layers = L; a.split_half = 1;
    a.activation = DCTR_ACT_RELU; a.layer_size = ls.data(); a.filters = fp.data(); a.bias = bp.data();
    int odim = 0;
    for (int k = 0; k < L; ++k) odim += k + 1 == L ? Hs[k] : Hs[k] - Hs[k] / 2;
    CK(hipMalloc(&out, (size_t)B * odim * 4));
    a.out = out;
    const size_t need = dctr_cin_workspace_bytes(&a);
    if (need) CK(hipMalloc(&ws, need));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int samples[6] = {0, 1, 7, B / 2 + 3, B - 2, B - 1};
    std::vector<std::vector<double>> ref(6);
    for (int q = 0; q < 6; ++q) host_cin(hx.data() + (size_t)samples[q] * F0 * D, F0, D, Hs, hW, hb, true, ref[q]);
    const double flop = 2.0 * B * D * ((double)F0 * F0 * Hs[0] + (L > 1 ? (double)F0 * (Hs[0] / 2) * Hs[1] : 0) +
                                       (L > 2 ? (double)F0 * (Hs[1] / 2) * Hs[2] : 0));
    for (int mode = 0; mode < (need ? 2 : 1); ++mode) {
        a.workspace = mode ? ws : nullptr;
        a.workspace_bytes = mode ? need : 0;
        CK(hipMemsetAsync(out, 0xff, (size_t)B * odim * 4, st));
        std::vector<float> t;
        for (int r = 0; r < reps; ++r) {
            dctr_profile_next_launch();
            int rc = dctr_cin_fwd(&a, st);
            if (rc) { printf("rc=%d %s\n", rc, dctr_last_error()); return 1; }
            CK(hipStreamSynchronize(st));
            if (r >= 10) t.push_back(dctr_profile_last_ms() * 1000.f);
        }
        // back to back: 20 launches between two events
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 20; ++r) dctr_cin_fwd(&a, st);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> ho((size_t)B * odim);
        CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (int q = 0; q < 6; ++q)
            for (int o = 0; o < odim; ++o) {
                const double r = ref[q][o], g = ho[(size_t)samples[q] * odim + o];
                worst = std::max(worst, std::fabs(g - r) / (std::fabs(r) + 1e-3));
                big = std::max(big, std::fabs(r));
            }
        bool finite = true;
        for (float v : ho) finite = finite && std::isfinite(v);
        std::sort(t.begin(), t.end());
        const float med = t[t.size() / 2];
        printf("cin B=%d F0=%d D=%d H=%d,%d,%d  %s  median %.1f us  min %.1f us  back-to-back %.1f us/call (incl. fold)  %.1f TFLOP/s (ref count, median) = %.3f of 157.3   max rel err %.2e (|ref| <= %.2f)  finite=%d\n",
               B, F0, D, H0, H1, H2, mode ? "FOLDED layer 0" : "plain layer 0 ", med, t[0], ms * 1000.f / 20, flop / med * 1e-6,
               flop / med * 1e-6 / 157.3, worst, big, (int)finite);
    }
    return 0;
}
