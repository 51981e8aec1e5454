// NOTE (round 5): the -DDCTR_*_LAB_* / -DDCTR_LAB_TIMING ablation and stamp switches this harness mentions were removed from the product kernels
// (they live in git history up to 5db6128); without them it still builds and times the shipped kernels.
// Lab harness for the row-chained dctr_embed_mlp_fwd kernel (chain_kernels.hip) at the C2 / C5 shapes: correctness against
// mlp_kernel<2>, the streaming kernel and a float64 host reference on a row sample (ragged tail, out-of-range id flag),
// then launch times over a range of rows per launch for tile_rows 0 (auto: chained + remainder), 256 (chained only), 64.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDCTR_CHAIN_LAB_WGTS -I include -I deepctr_amd/csrc scripts/chain_lab.cpp -o scripts/_bin/chain_lab
//   chain_lab [E=16] [V=100000]
#include "../deepctr_amd/csrc/abi.cpp"
#include "../deepctr_amd/csrc/mlp_kernels.hip"
#include "../deepctr_amd/csrc/mlp_kernels_rt1.hip"
#undef DCTR_MLP_RT
#include "../deepctr_amd/csrc/mlp_kernels_rt2.hip"
#undef DCTR_MLP_RT
#include "../deepctr_amd/csrc/mlp_kernels_rt4.hip"
#include "../deepctr_amd/csrc/mlp_kernels_ring.hip"
#include "../deepctr_amd/csrc/stream_kernels.hip"
#include "../deepctr_amd/csrc/chain_kernels.hip"
#include "../deepctr_amd/csrc/chain_kernels_r2w8_m42.hip"
#undef DCTR_CHAIN_RT
#undef DCTR_CHAIN_NW
#undef DCTR_CHAIN_M0
#undef DCTR_CHAIN_M1
#undef DCTR_CHAIN_M2SET
#include "../deepctr_amd/csrc/chain_kernels_r2w4_m42.hip"
#undef DCTR_CHAIN_RT
#undef DCTR_CHAIN_NW
#undef DCTR_CHAIN_M0
#undef DCTR_CHAIN_M1
#undef DCTR_CHAIN_M2SET
#include "../deepctr_amd/csrc/chain_kernels_r2w8_m42_q.hip"
namespace dctr_chain {      // (the lab links the 256-128-64 instantiations only)
int launch_r2w8_m41(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m22(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m21(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m42x(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m42t(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m42w(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
int launch_r2w8_m42r(const ChainParams&, int, int, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
size_t bf3_workspace_bytes(int) { return 0; }
int launch_r2w8_m42_bf3(const ChainParams&, int, void*, bool, unsigned, hipStream_t) { return DCTR_E_UNSUPPORTED; }
}
#include <algorithm>
#include <cmath>
#include <time.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static uint32_t lcg_state = 12345u;
static inline float frand() { lcg_state = lcg_state * 1664525u + 1013904223u; return ((lcg_state >> 8) * (1.0f / 16777216.0f)) - 0.5f; }

int main(int argc, char** argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 16;
    const int V = argc > 2 ? atoi(argv[2]) : 100000;
    constexpr int F = 26, ND = 13;
    const int in_dim = F * E + ND;
    const int dims[4] = {in_dim, 256, 128, 64};
    const int64_t BMAX = 262144;
    float *tables, *lin, *dense, *densew, *y0, *y1, *W[3], *bias[3], *head, *gb; int* ids; int* status; dctr_field_t* fields;
    std::vector<float> h_tab((size_t)F * V * E), h_lin((size_t)F * V), h_dense((size_t)BMAX * ND), h_dw(ND);
    const bool zero = getenv("DCTR_LAB_ZERO") != nullptr;      // DVFS probe: all-zero operands draw less power -> higher clock
    for (auto& x : h_tab) x = zero ? 0.f : 0.2f * frand();
    for (auto& x : h_lin) x = 0.2f * frand();
    for (auto& x : h_dense) x = frand() + 0.5f;
    for (auto& x : h_dw) x = frand();
    std::vector<int> h_ids((size_t)F * BMAX);
    for (auto& x : h_ids) { lcg_state = lcg_state * 1664525u + 1013904223u; x = (int)((lcg_state >> 4) % (uint32_t)V); }
    CK(hipMalloc(&tables, h_tab.size() * 4)); CK(hipMemcpy(tables, h_tab.data(), h_tab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&lin, h_lin.size() * 4)); CK(hipMemcpy(lin, h_lin.data(), h_lin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&ids, h_ids.size() * 4)); CK(hipMemcpy(ids, h_ids.data(), h_ids.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dense, h_dense.size() * 4)); CK(hipMemcpy(dense, h_dense.data(), h_dense.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&densew, ND * 4)); CK(hipMemcpy(densew, h_dw.data(), ND * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&y0, BMAX * 4)); CK(hipMalloc(&y1, BMAX * 4)); CK(hipMalloc(&status, 4)); CK(hipMemset(status, 0, 4));
    std::vector<dctr_field_t> fh(F);
    for (int j = 0; j < F; ++j) { fh[j] = dctr_field_t{}; fh[j].table = tables + (size_t)j * V * E; fh[j].lin_table = lin + (size_t)j * V; fh[j].vocab = V; fh[j].dim = E; fh[j].out_offset = j * E; fh[j].in_fm = 1; }
    CK(hipMalloc(&fields, F * sizeof(dctr_field_t))); CK(hipMemcpy(fields, fh.data(), F * sizeof(dctr_field_t), hipMemcpyHostToDevice));
    std::vector<float> h_W[3], h_b[3], h_head(64);
    for (int l = 0; l < 3; ++l) {
        h_W[l].resize((size_t)dims[l] * dims[l + 1]); h_b[l].resize(dims[l + 1]);
        const float sc = 2.0f * sqrtf(2.0f / (dims[l] + dims[l + 1]));
        for (auto& x : h_W[l]) x = zero ? 0.f : sc * frand();
        for (auto& x : h_b[l]) x = 0.1f * frand();
        CK(hipMalloc(&W[l], h_W[l].size() * 4)); CK(hipMemcpy(W[l], h_W[l].data(), h_W[l].size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&bias[l], h_b[l].size() * 4)); CK(hipMemcpy(bias[l], h_b[l].data(), h_b[l].size() * 4, hipMemcpyHostToDevice));
    }
    for (auto& x : h_head) x = frand();
    const float h_gb = 0.123f;
    CK(hipMalloc(&head, 64 * 4)); CK(hipMemcpy(head, h_head.data(), 256, hipMemcpyHostToDevice));
    CK(hipMalloc(&gb, 4)); CK(hipMemcpy(gb, &h_gb, 4, hipMemcpyHostToDevice));
    int32_t units[3] = {256, 128, 64};
    const float* ks[3] = {W[0], W[1], W[2]}; const float* bs[3] = {bias[0], bias[1], bias[2]};
    hipStream_t st; CK(hipStreamCreate(&st));

    auto run = [&](int64_t B, int tile_rows, float* y, int sigmoid) -> int {
        dctr_mlp_args_t a{};
        a.batch = B; a.in_dim = in_dim; a.n_layers = 3; a.units = units; a.kernels = ks; a.biases = bs; a.tile_rows = tile_rows;
        a.activation = DCTR_ACT_RELU; a.has_head = 1; a.sigmoid_out = sigmoid; a.head_w = head; a.global_bias = gb; a.y = y;
        dctr_gather_fm_args_t g{};
        g.fields = fields; g.ids = ids; g.ids_stride_f = BMAX; g.ids_stride_b = 1; g.ids_is_i64 = 0; g.n_fields = F; g.max_dim = E; g.all_dim4 = 1;
        g.any_hash = 0; g.n_dense = ND; g.dense = dense; g.dense_stride = ND; g.dense_lin_w = densew; g.dense_out_offset = F * E;
        g.dense_copy_cols = ND; g.batch = B; g.status = status; g.split_col = E == 16 ? 256 : 0; g.split_field = E == 16 ? 16 : 0;
        g.uniform_dim = E;
        int rc = dctr_embed_mlp_fwd(&g, &a, 1, 1, st);
        if (rc) printf("rc=%d %s\n", rc, dctr_last_error());
        return rc;
    };

    // ---- correctness: B with a ragged tail, raw logits
    const int64_t Bc = 4096 * 5 + 37;
    CK(hipMemset(y0, 0xff, BMAX * 4)); CK(hipMemset(y1, 0xff, BMAX * 4));
    if (run(Bc, 32, y0, 0) || run(Bc, 256, y1, 0)) return 1;
    CK(hipStreamSynchronize(st));
    std::vector<float> r0(Bc + 8), r1(Bc + 8);
    CK(hipMemcpy(r0.data(), y0, (Bc + 8) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), y1, (Bc + 8) * 4, hipMemcpyDeviceToHost));
    int st_h = 0; CK(hipMemcpy(&st_h, status, 4, hipMemcpyDeviceToHost));
    double md = 0; int64_t worst = -1; int nan1 = 0;
    for (int64_t b = 0; b < Bc; ++b) { if (!(r1[b] == r1[b])) ++nan1; const double d = fabs((double)r0[b] - r1[b]); if (d > md) { md = d; worst = b; } }
    printf("status=%d  chain vs mlp_kernel<2>: max |diff| %.3e at row %lld (%.6f vs %.6f), NaN rows %d, tail untouched: %s\n", st_h, md, (long long)worst,
           worst >= 0 ? r0[worst] : 0.f, worst >= 0 ? r1[worst] : 0.f, nan1, (r1[Bc] != r1[Bc]) ? "yes" : "NO");
    // float64 reference on a row sample
    double mref0 = 0, mref1 = 0;
    for (int64_t b : {int64_t(0), int64_t(1), int64_t(17), int64_t(63), int64_t(64), int64_t(4095), int64_t(4096), int64_t(12345), Bc - 38, Bc - 2, Bc - 1}) {
        std::vector<double> x(in_dim), S(E, 0.0); double sq = 0, linv = 0;
        for (int f = 0; f < F; ++f) {
            const int id = h_ids[(size_t)f * BMAX + b];
            linv += h_lin[(size_t)f * V + id];
            for (int e = 0; e < E; ++e) { const double v = h_tab[((size_t)f * V + id) * E + e]; x[f * E + e] = v; S[e] += v; sq += v * v; }
        }
        double fm = -sq; for (int e = 0; e < E; ++e) fm += S[e] * S[e]; fm *= 0.5;
        for (int m = 0; m < ND; ++m) { x[F * E + m] = h_dense[(size_t)b * ND + m]; linv += (double)h_dense[(size_t)b * ND + m] * h_dw[m]; }
        std::vector<double> cur = x;
        for (int l = 0; l < 3; ++l) {
            std::vector<double> nx(dims[l + 1]);
            for (int n = 0; n < dims[l + 1]; ++n) { double acc = h_b[l][n]; for (int k = 0; k < dims[l]; ++k) acc += cur[k] * h_W[l][(size_t)k * dims[l + 1] + n]; nx[n] = acc > 0 ? acc : 0; }
            cur = nx;
        }
        double logit = h_gb + fm + linv; for (int n = 0; n < 64; ++n) logit += cur[n] * h_head[n];
        mref0 = std::max(mref0, fabs(logit - r0[b]) / (fabs(logit) + 1e-2)); mref1 = std::max(mref1, fabs(logit - r1[b]) / (fabs(logit) + 1e-2));
        if (b < 2 || b == Bc - 1) printf("  row %lld: ref %.6f  mlp_kernel %.6f  chain %.6f\n", (long long)b, logit, r0[b], r1[b]);
    }
    printf("vs float64 reference (11 rows): rel err mlp_kernel<2> %.2e, chain %.2e\n", mref0, mref1);

#ifdef DCTR_CHAIN_LAB_TS
    {
        if (run(DCTR_CHAIN_TS_RT == 2 ? 262144 : 81920, 0, y1, 1)) return 1;       // (RT 1: the tail phase of the 20-batch call)
        CK(hipStreamSynchronize(st));
        unsigned long long ts[2][64];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(dctr_chain_ts), sizeof(ts)));
        const unsigned long long t0 = ts[0][0];
        const char* nm[2] = {"wave0", DCTR_CHAIN_TS_RT == 2 ? "wave7" : "wave3"};
        for (int w = 0; w < 2; ++w) {
            printf("stamps %s (cycles after wave 0's pass start):", nm[w]);
            for (int i = 0; i < 64; ++i) if (ts[w][i]) printf(" [%d]%lld", i, (long long)(ts[w][i] - t0));
            printf("\n");
        }
    }
#endif
    {   // auto split (chained kernel for whole multiples of 256 rows x CUs + the rest) against the 32-row kernel
        const int64_t Ba = 65536 + 32768 + 16384 + 77;
        CK(hipMemset(y0, 0xff, BMAX * 4)); CK(hipMemset(y1, 0xff, BMAX * 4));
        if (run(Ba, 32, y0, 0) || run(Ba, 0, y1, 0)) return 1;
        CK(hipStreamSynchronize(st));
        std::vector<float> a0(Ba + 8), a1(Ba + 8);
        CK(hipMemcpy(a0.data(), y0, (Ba + 8) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a1.data(), y1, (Ba + 8) * 4, hipMemcpyDeviceToHost));
        double m = 0; int nn = 0;
        for (int64_t b = 0; b < Ba; ++b) { if (!(a1[b] == a1[b])) ++nn; m = std::max(m, fabs((double)a0[b] - a1[b])); }
        printf("auto split, %lld rows: max |diff| vs mlp_kernel<2> %.3e, NaN rows %d, tail untouched: %s\n", (long long)Ba, m, nn, (a1[Ba] != a1[Ba]) ? "yes" : "NO");
    }
    {   // the tail phase against the forced 256-row shape: the same bits (every phase walks k in the same order)
        for (int64_t Bt : {int64_t(81920), int64_t(65536 + 16384 - 77), int64_t(65536 + 100), int64_t(65536 + 32768 - 5), int64_t(131072 + 16384)}) {
            CK(hipMemset(y0, 0xff, BMAX * 4)); CK(hipMemset(y1, 0xff, BMAX * 4));
            if (run(Bt, 256, y0, 0) || run(Bt, 0, y1, 0)) return 1;
            CK(hipStreamSynchronize(st));
            std::vector<uint32_t> b0(Bt + 8), b1(Bt + 8);
            CK(hipMemcpy(b0.data(), y0, (Bt + 8) * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b1.data(), y1, (Bt + 8) * 4, hipMemcpyDeviceToHost));
            int64_t nd = 0, firstd = -1;
            for (int64_t b = 0; b < Bt; ++b) if (b0[b] != b1[b]) { if (firstd < 0) firstd = b; ++nd; }
            int64_t pr[4]; int32_t pw[4];
            const int np = dctr_chain::plan(Bt, 0, pr, pw, 4);
            printf("tail vs forced 256-row passes, %7lld rows (plan:", (long long)Bt);
            for (int i = 0; i < np; ++i) printf(" %lld x %d", (long long)pr[i], pw[i]);
            printf("): %lld rows differ (first %lld), tail untouched: %s\n", (long long)nd, (long long)firstd, (b1[Bt] == 0xffffffffu) ? "yes" : "NO");
        }
    }
#ifdef DCTR_CHAIN_LAB_WGTS
    // ---- where an isolated launch spends its time: wall-clock stamps (100 MHz) and shader-cycle stamps of every workgroup — kernel
    // entry, end of the main phase, end of the tail phase — against the host-side event pair around the launch; `second`: the same
    // launch issued right behind an identical one (no idle GPU in front of it); stagger: workgroup b waits (b % 16) x S ticks at entry
    {
        const int stagger = 0;
        for (int64_t Bw : {int64_t(65536), int64_t(81920), int64_t(262144)}) {
            hipEvent_t w0, w1; CK(hipEventCreate(&w0)); CK(hipEventCreate(&w1));
            for (int second = 0; second < 2; ++second) {
                float ms = 0;
                static unsigned long long wg[1024][8];
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipDeviceSynchronize());
                    if (second && run(Bw, 0, y1, 1)) return 1;
                    CK(hipEventRecord(w0, st));
                    if (run(Bw, 0, y1, 1)) return 1;
                    CK(hipEventRecord(w1, st)); CK(hipEventSynchronize(w1));
                    CK(hipEventElapsedTime(&ms, w0, w1));
                    CK(hipMemcpyFromSymbol(wg, HIP_SYMBOL(dctr_chain_wgts), sizeof(wg)));
                }
                unsigned long long t0 = ~0ull;
                for (int b = 0; b < 256; ++b) t0 = std::min(t0, wg[b][0]);
                std::vector<double> s0, s1, s2, mhz;
                for (int b = 0; b < 256; ++b) {
                    s0.push_back((wg[b][0] - t0) * 0.01); s1.push_back((wg[b][1] - t0) * 0.01); s2.push_back((wg[b][2] - t0) * 0.01);
                    mhz.push_back((double)(wg[b][5] - wg[b][4]) / ((double)(wg[b][1] - wg[b][0]) * 0.01));     // shader cycles per us, main phase
                }
                std::sort(s0.begin(), s0.end()); std::sort(s1.begin(), s1.end()); std::sort(s2.begin(), s2.end()); std::sort(mhz.begin(), mhz.end());
                printf("stamps %6lld rows stagger %3d %s (us after the first entry; min / median / max over 256 workgroups): entry %.1f / %.1f / %.1f, "
                       "main done %.1f / %.1f / %.1f, tail done %.1f / %.1f / %.1f; event pair %.1f us; shader clock in the main phase %.0f / %.0f / %.0f MHz\n",
                       (long long)Bw, stagger, second ? "2nd of two" : "isolated  ", s0[0], s0[128], s0[255], s1[0], s1[128], s1[255], s2[0], s2[128], s2[255],
                       ms * 1e3, mhz[0], mhz[128], mhz[255]);
            }
        }
    }
#endif
    if (getenv("CHAIN_LAB_STAMPS_ONLY") != nullptr) return 0;
    // ---- one-shot region of 20 batches (81,920 rows = 65,536 on <2,8> + 16,384 on <1,4>), as bench.py --steps 20 times it:
    // host clock around issue + synchronise, idle GPU before.  (a) one call (two launches on one stream), (b) the two parts as
    // two calls on one stream, (c) the two parts on TWO streams (fork / join with events): does the remainder fill the main
    // kernel's tail?
    if (getenv("CHAIN_LAB_ONESHOT") != nullptr) {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        auto run_on = [&](hipStream_t s, int64_t row0, int64_t B, float* y) -> int {
            dctr_mlp_args_t a{};
            a.batch = B; a.in_dim = in_dim; a.n_layers = 3; a.units = units; a.kernels = ks; a.biases = bs; a.tile_rows = 0;
            a.activation = DCTR_ACT_RELU; a.has_head = 1; a.sigmoid_out = 1; a.head_w = head; a.global_bias = gb; a.y = y + row0;
            dctr_gather_fm_args_t g{};
            g.fields = fields; g.ids = ids + row0; g.ids_stride_f = BMAX; g.ids_stride_b = 1; g.ids_is_i64 = 0; g.n_fields = F; g.max_dim = E; g.all_dim4 = 1;
            g.any_hash = 0; g.n_dense = ND; g.dense = dense + row0 * ND; g.dense_stride = ND; g.dense_lin_w = densew; g.dense_out_offset = F * E;
            g.dense_copy_cols = ND; g.batch = B; g.status = status; g.split_col = E == 16 ? 256 : 0; g.split_field = E == 16 ? 16 : 0;
            g.uniform_dim = E;
            return dctr_embed_mlp_fwd(&g, &a, 1, 1, s);
        };
        auto now_us = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<double> ts;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now_us();
                if (mode == 0) run_on(st, 0, 81920, y1);
                else if (mode == 1) { run_on(st, 0, 65536, y1); run_on(st, 65536, 16384, y1); }
                else {
                    CK(hipEventRecord(ef, st)); CK(hipStreamWaitEvent(st2, ef, 0));
                    run_on(st2, 65536, 16384, y1);
                    run_on(st, 0, 65536, y1);
                    CK(hipEventRecord(ej, st2)); CK(hipStreamWaitEvent(st, ej, 0));
                }
                CK(hipStreamSynchronize(st));
                ts.push_back(now_us() - t0);
            }
            std::sort(ts.begin(), ts.end());
            printf("one-shot 81,920 rows, mode %d (%s): median %.1f us, min %.1f us -> %.1f M samples/s\n", mode,
                   mode == 0 ? "one call" : mode == 1 ? "two calls, one stream" : "two streams", ts[ts.size() / 2], ts[0], 81920.0 / ts[ts.size() / 2]);
        }
    }
    // ---- timing
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop_row = 2.0 * ((double)in_dim * 256 + 256 * 128 + 128 * 64 + 64);
    const bool quick = getenv("CHAIN_LAB_QUICK") != nullptr;
    for (int64_t B : {int64_t(16384), int64_t(32768), int64_t(65536), int64_t(81920), int64_t(131072), int64_t(262144)}) {
        if (quick && B != 262144) continue;
        for (int tr : {256, 128, 0, 64}) {
            if (quick && tr != 256 && !(tr == 128 && getenv("CHAIN_LAB_128") != nullptr)) continue;
            if (tr == 128 && B > 32768 && getenv("CHAIN_LAB_128") == nullptr) continue;
            for (int w = 0; w < 3; ++w) if (run(B, tr, y1, 1)) return 1;
            CK(hipStreamSynchronize(st));
            const int R = 10;
            std::vector<float> t;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < R; ++r) run(B, tr, y1, 1);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / R);
            }
            std::sort(t.begin(), t.end());
            const double us = t[t.size() / 2] * 1e3;
            printf("rows/launch %7lld tile_rows %3d: %9.2f us/launch  %7.1f M samples/s  %6.1f TFLOP/s (%.3f of 157.3)  [%.2f us per 4096 rows]\n",
                   (long long)B, tr, us, B / us, flop_row * B / us * 1e-6, flop_row * B / us * 1e-6 / 157.3, us * 4096 / B);
        }
    }
    CK(hipMemcpy(&st_h, status, 4, hipMemcpyDeviceToHost));
    printf("final status=%d\n", st_h);
    return 0;
}
