// Stand-alone check + timing of the stream-K fused-MLP kernel (csrc/mlp_streamk.h) against a float64 host reference and against the
// tile-per-workgroup GEMMs of csrc/gemm_mfma.h, on uniform random [-1, 1) operands (zero-filled operands clock ~15-20 % higher: never bench on them).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../me-trpo_amd/csrc -I../../include mlp_sk_bench.hip -o mlp_sk_bench
// run:   ./mlp_sk_bench            (correctness on small shapes, then the C2 / C3 / C4 shapes)
#define SK_DEBUG 1
#include "gemm_mfma.h"
#include "mlp_streamk.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed, float scale = 1.0f) {
    std::mt19937 g(seed); std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<float> v(n); for (auto& x : v) x = d(g) * scale; return v;
}
template <class T> static T* up(const std::vector<T>& v) { T* p; CK(hipMalloc(&p, v.size() * sizeof(T))); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

struct Net {   // one 2-hidden-layer ensemble in the resident layout: per head [W0 (nin x H1), b0, W1 (H1 x H2), b1, W2 (H2 x no), b2]
    int heads, nin, H1, H2, no; long long np, oW0, ob0, oW1, ob1, oW2, ob2; std::vector<float> w;
};
static Net make_net(int heads, int nin, int H1, int H2, int no, unsigned seed) {
    Net n; n.heads = heads; n.nin = nin; n.H1 = H1; n.H2 = H2; n.no = no;
    auto a4 = [](long long x) { return (x + 3) & ~3LL; };
    n.oW0 = 0; n.ob0 = a4((long long)nin * H1); n.oW1 = a4(n.ob0 + H1); n.ob1 = a4(n.oW1 + (long long)H1 * H2); n.oW2 = a4(n.ob1 + H2); n.ob2 = a4(n.oW2 + (long long)H2 * no); n.np = a4(n.ob2 + no);
    n.w = rnd((size_t)heads * n.np, seed);
    for (int h = 0; h < heads; ++h) {
        float* p = n.w.data() + (size_t)h * n.np;
        for (long long i = 0; i < (long long)nin * H1; ++i) p[n.oW0 + i] *= 1.0f / sqrtf((float)nin);
        for (long long i = 0; i < (long long)H1 * H2; ++i) p[n.oW1 + i] *= 1.0f / sqrtf((float)H1);
        for (long long i = 0; i < (long long)H2 * no; ++i) p[n.oW2 + i] *= 1.0f / sqrtf((float)H2);
    }
    return n;
}

static double timeit(const std::function<void()>& f, int reps = 10) {      // >= 0.3 s of back-to-back launches: the sustained clock, not the burst clock
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    { float ms1; CK(hipEventElapsedTime(&ms1, e0, e1)); reps = std::max(reps, (int)(300.0f / std::max(ms1, 0.01f))); }
    for (int i = 0; i < reps / 2; ++i) f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e3;
}

// ---- fused 2-hidden-layer MLP: x -> relu(x W0 + b0) -> relu(. W1 + b1) -> . W2 (partials per 256-column block; b2 left to the consumer) ----
template <int S0, int OT, int NW = 8>
static void run_fused(const char* name, int M, int heads, int nin, int Hd, int no, bool check, int grid_override, int n_sm) {
    Net net = make_net(heads, nin, Hd, Hd, no, 7);
    const int lda = 4 * S0;
    std::vector<float> x((size_t)M * lda, 0.0f);
    { auto r = rnd((size_t)M * nin, 11); for (int m = 0; m < M; ++m) { for (int i = 0; i < nin; ++i) x[(size_t)m * lda + i] = r[(size_t)m * nin + i]; x[(size_t)m * lda + nin] = 1.0f; } }
    float* dW = up(net.w); float* dX = up(x);
    SkArgs a = {};
    a.M = M; a.heads = heads; a.K1 = Hd; a.N = Hd;
    a.A = dX; a.strideA = 0; a.lda = lda;
    a.W0 = dW + net.oW0; a.strideW0 = net.np; a.W1 = dW + net.oW1; a.strideW1 = net.np;
    SkPlan p = sk_plan<SK_A_PRODUCER, SK_EPI_OUT, S0, OT, NW>(a, n_sm, grid_override);
    const int CB = a.CB, E = SkEpi<OT>::E;
    float* dImg; CK(hipMalloc(&dImg, (size_t)heads * CB * E * SkEpi<OT>::FLOATS * 4));
    const long long tot = (long long)heads * CB * E * SkEpi<OT>::FLOATS;
    hipLaunchKernelGGL((k_sk_epi_image<OT>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, dW + net.ob1, net.np, dW + net.oW2, net.np, no, Hd, heads, dImg);
    a.epi = dImg; a.ldp = 16 * OT; a.stridePart = (long long)M * a.ldp;
    float* dPart; CK(hipMalloc(&dPart, (size_t)CB * heads * a.stridePart * 4)); CK(hipMemset(dPart, 0xff, (size_t)CB * heads * a.stridePart * 4));
    a.part = dPart;
    CK(hipMalloc(&a.xacc, p.xacc_floats * 4)); CK(hipMalloc(&a.xflag, p.nflags * 4)); CK(hipMemset(a.xflag, 0, p.nflags * 4));
    CK(hipMalloc(&a.err, 8)); CK(hipMemset(a.err, 0, 8));
    void* smem; CK(hipMalloc(&smem, p.sched_bytes)); CK((sk_build_sched<SK_A_PRODUCER, SK_EPI_OUT, S0, OT>(a, p, smem, 0)));
    unsigned epoch = 0;
    auto go = [&]() { a.epoch = ++epoch; CK((sk_launch<SK_A_PRODUCER, SK_EPI_OUT, S0, OT, NW>(a, p, 0))); };
    go(); CK(hipDeviceSynchronize());
    double err; CK(hipMemcpy(&err, a.err, 8, hipMemcpyDeviceToHost));
    if (err) printf("  !! hand-over time-out reported\n");
    if (check) {
        std::vector<float> part((size_t)CB * heads * a.stridePart);
        CK(hipMemcpy(part.data(), dPart, part.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; int nbad = 0;
        std::vector<double> h1(Hd), h2(Hd);
        for (int hd = 0; hd < heads; ++hd) {
            const float* w = net.w.data() + (size_t)hd * net.np;
            for (int m = 0; m < M; m += (M > 600 ? 37 : 1)) {
                for (int k = 0; k < Hd; ++k) { double s = w[net.ob0 + k]; for (int i = 0; i < nin; ++i) s += (double)x[(size_t)m * lda + i] * w[net.oW0 + (size_t)i * Hd + k]; h1[k] = s > 0 ? s : 0; }
                for (int n = 0; n < Hd; ++n) { double s = w[net.ob1 + n]; for (int k = 0; k < Hd; ++k) s += h1[k] * w[net.oW1 + (size_t)k * Hd + n]; h2[n] = s > 0 ? s : 0; }
                for (int o = 0; o < no; ++o) {
                    double ref = 0; for (int n = 0; n < Hd; ++n) ref += h2[n] * w[net.oW2 + (size_t)n * no + o];
                    double got = 0; for (int cb = 0; cb < CB; ++cb) got += part[((size_t)cb * heads + hd) * a.stridePart + (size_t)m * a.ldp + o];
                    maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
                    if (fabs(got - ref) > 1e-4 && nbad < 12 && o == 0) {
                        ++nbad; printf("    bad head %d m %d (rb %d wave %d) got %.5f ref %.5f :", hd, m, m / 128, (m % 128) / 16, got, ref);
                        for (int cb = 0; cb < CB; ++cb) { double rc = 0; for (int n = cb * 256; n < cb * 256 + 256; ++n) rc += h2[n] * w[net.oW2 + (size_t)n * no + o];
                            printf("  cb%d got %.5f ref %.5f", cb, part[((size_t)cb * heads + hd) * a.stridePart + (size_t)m * a.ldp + o], rc); }
                        printf("\n");
                    }
                }
            }
        }
        printf("  check %-28s M=%5d heads=%2d nin=%2d H=%4d no=%2d grid=%3d : max|err| %.3e (max|ref| %.3f) %s\n", name, M, heads, nin, Hd, no, p.grid, maxerr, maxref,
               maxerr <= 2e-5 * fmax(1.0, maxref) ? "OK" : "FAIL");
    } else {
        const double us = timeit(go);
        const double fl = 2.0 * M * heads * ((double)nin * Hd + (double)Hd * Hd + (double)Hd * no);
        printf("  %-34s M=%5d heads=%2d nin=%2d H=%4d no=%2d grid=%3d lds=%zu : %8.1f us  %6.1f TFLOP/s (algorithmic: x W0, . W1, . W2)\n", name, M, heads, nin, Hd, no, p.grid,
               p.lds_bytes, us, fl / us / 1e6);
    }
    CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dImg)); CK(hipFree(dPart)); CK(hipFree(a.xacc)); CK(hipFree(a.xflag)); CK(hipFree(a.err));
}

// ---- one wide layer from activations in HBM: C = relu(A W1 + b1) ----
template <int NW = 8>
static void run_layer(const char* name, int M, int heads, int Kd, int N, bool check, int grid_override, int n_sm, bool old_too) {
    auto A = rnd((size_t)heads * M * Kd, 3), W = rnd((size_t)heads * Kd * N, 5, 1.0f / sqrtf((float)Kd)), b = rnd((size_t)heads * N, 9);
    float *dA = up(A), *dW = up(W), *db = up(b), *dC, *dC2;
    CK(hipMalloc(&dC, (size_t)heads * M * N * 4)); CK(hipMalloc(&dC2, (size_t)heads * M * N * 4));
    SkArgs a = {};
    a.M = M; a.heads = heads; a.K1 = Kd; a.N = N; a.A = dA; a.strideA = (long long)M * Kd; a.lda = Kd; a.W1 = dW; a.strideW1 = (long long)Kd * N;
    a.b1 = db; a.strideB1 = N; a.C = dC; a.strideC = (long long)M * N; a.ldc = N;
    a.skip = (!check && getenv("SK_SKIP")) ? atoi(getenv("SK_SKIP")) : 0;
    SkPlan p = sk_plan<SK_A_GLOBAL, SK_EPI_STORE, 1, 1, NW>(a, n_sm, grid_override);
    CK(hipMalloc(&a.xacc, p.xacc_floats * 4)); CK(hipMalloc(&a.xflag, p.nflags * 4)); CK(hipMemset(a.xflag, 0, p.nflags * 4));
    CK(hipMalloc(&a.err, 8)); CK(hipMemset(a.err, 0, 8));
    void* smem; CK(hipMalloc(&smem, p.sched_bytes)); CK((sk_build_sched<SK_A_GLOBAL, SK_EPI_STORE, 1, 1>(a, p, smem, 0)));
    unsigned epoch = 0;
    auto go = [&]() { a.epoch = ++epoch; CK((sk_launch<SK_A_GLOBAL, SK_EPI_STORE, 1, 1, NW>(a, p, 0))); };
    GemmEpi ep = {}; ep.bias = db; ep.strideBias = N;
    auto go_old = [&]() { gemm_auto<EPI_BIAS_RELU, false, false>(dA, (long long)M * Kd, Kd, dW, (long long)Kd * N, N, dC2, (long long)M * N, N, M, N, Kd, heads, ep, 0); };
    go(); CK(hipDeviceSynchronize());
    double err; CK(hipMemcpy(&err, a.err, 8, hipMemcpyDeviceToHost));
    if (err) printf("  !! hand-over time-out reported\n");
    if (check) {
        std::vector<float> C((size_t)heads * M * N);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int hd = 0; hd < heads; ++hd)
            for (int m = 0; m < M; m += (M > 600 ? 41 : 1))
                for (int n = 0; n < N; n += 3) {
                    double s = b[(size_t)hd * N + n];
                    for (int k = 0; k < Kd; ++k) s += (double)A[((size_t)hd * M + m) * Kd + k] * W[((size_t)hd * Kd + k) * N + n];
                    s = s > 0 ? s : 0;
                    maxerr = fmax(maxerr, fabs(s - C[((size_t)hd * M + m) * N + n]));
                }
        printf("  check %-28s M=%5d heads=%2d K=%4d N=%4d grid=%3d : max|err| %.3e %s\n", name, M, heads, Kd, N, p.grid, maxerr, maxerr <= 2e-5 ? "OK" : "FAIL");
        // bitwise: another grid must give the same sums
        std::vector<float> C1 = C;
        SkPlan p2 = sk_plan<SK_A_GLOBAL, SK_EPI_STORE, 1, 1, NW>(a, n_sm, p.grid > 3 ? p.grid - 3 : p.grid + 1);
        float* xa2; unsigned* xf2; CK(hipMalloc(&xa2, p2.xacc_floats * 4)); CK(hipMalloc(&xf2, p2.nflags * 4)); CK(hipMemset(xf2, 0, p2.nflags * 4));
        float* xa1 = a.xacc; unsigned* xf1 = a.xflag; a.xacc = xa2; a.xflag = xf2; a.epoch = 1;
        void* smem2; CK(hipMalloc(&smem2, p2.sched_bytes)); CK((sk_build_sched<SK_A_GLOBAL, SK_EPI_STORE, 1, 1>(a, p2, smem2, 0)));
        CK(hipMemset(dC, 0, C.size() * 4));
        CK((sk_launch<SK_A_GLOBAL, SK_EPI_STORE, 1, 1, NW>(a, p2, 0))); CK(hipDeviceSynchronize());
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t q = 0; q < C.size(); ++q) diff += (memcmp(&C[q], &C1[q], 4) != 0);
        printf("        grid %d vs grid %d: %zu of %zu outputs differ bitwise %s\n", p.grid, p2.grid, diff, C.size(), diff == 0 ? "OK" : "FAIL");
        a.xacc = xa1; a.xflag = xf1; CK(hipFree(xa2)); CK(hipFree(xf2));
        sk_plan<SK_A_GLOBAL, SK_EPI_STORE, 1, 1, NW>(a, n_sm, grid_override);
        CK(hipFree(smem2));
    } else {
        if (getenv("SK_STAMPS")) { unsigned long long* d2; CK(hipMalloc(&d2, 48 * 8)); CK(hipMemset(d2, 0, 48 * 8)); a.dbg2 = d2; go(); go(); CK(hipDeviceSynchronize());
          unsigned long long h2[48]; CK(hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost)); a.dbg2 = nullptr; CK(hipFree(d2));
          for (int w = 0; w < 2; ++w) for (int c = 0; c < 4; ++c) { const unsigned long long* t = h2 + (w * 4 + c) * 6;
              printf("    wave %d chunk %d: top +%5lld front-bk +%5lld steps +%5lld epi +%5lld back-bk/wait +%5lld barrier | since wave0 chunk8 top: %lld\n", w * 4, 8 + c,
                     (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]), (long long)(t[0] - h2[0])); } }
        unsigned long long* dbg; CK(hipMalloc(&dbg, 8 * 8 * p.grid)); CK(hipMemset(dbg, 0, 8 * 8 * p.grid)); a.dbg = dbg;          // total cycles / ticks of the LAST launch of the timed loop (clocks settled)
        const double us = timeit(go), fl = 2.0 * M * N * Kd * heads;
        CK(hipMemset(dbg, 0, 8 * 8 * p.grid)); go(); CK(hipDeviceSynchronize());
        { std::vector<unsigned long long> hd(8 * p.grid); CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
          double cyc = 0, tk = 0, nqs = 0, cmax = 0; unsigned long long t_in0 = ~0ull, t_in1 = 0, t_l0 = ~0ull, t_l1 = 0, t_e0 = ~0ull, t_e1 = 0;
          for (int b = 0; b < p.grid; ++b) { const unsigned long long* d = &hd[8 * b]; cyc += d[0]; tk += d[1]; nqs += d[2]; cmax = fmax(cmax, (double)d[0]);
              t_in0 = std::min(t_in0, d[3]); t_in1 = std::max(t_in1, d[3]); t_l0 = std::min(t_l0, d[4]); t_l1 = std::max(t_l1, d[4]); t_e0 = std::min(t_e0, d[5]); t_e1 = std::max(t_e1, d[5]); }
          printf("    main loop (wave 0 of each workgroup): %.0f cycles (max %.0f) / %.1f us = %.2f GHz; %.0f entries -> %.0f cycles per entry (MFMA issue floor 8192)\n", cyc / p.grid, cmax, tk / p.grid / 100.0,
                 cyc / tk / 10.0, nqs / p.grid, cyc / nqs);
          { double c0 = 0, c1 = 0; for (int b = 0; b < p.grid; ++b) (b < p.grid / 2 ? c0 : c1) += (double)hd[8 * b]; 
            printf("    workgroups [0, grid/2): %.0f cycles, [grid/2, grid): %.0f cycles (mean loop time; dispatch order: first / second workgroup of a CU when grid = 2 x CUs)\n", c0 / (p.grid / 2), c1 / (p.grid - p.grid / 2)); }
          printf("    timeline (us from the first workgroup's entry): entries ..%.1f | loop starts %.1f..%.1f | last-wave exits %.1f..%.1f\n", (t_in1 - t_in0) / 100.0, (t_l0 - t_in0) / 100.0,
                 (t_l1 - t_in0) / 100.0, (t_e0 - t_in0) / 100.0, (t_e1 - t_in0) / 100.0); }
        a.dbg = nullptr; CK(hipFree(dbg));
        printf("  %-34s M=%5d heads=%2d K=%4d N=%4d grid=%3d : %8.1f us  %6.1f TFLOP/s", name, M, heads, Kd, N, p.grid, us, fl / us / 1e6);
        if (old_too) { const double uo = timeit(go_old); printf("   | gemm_mfma.h: %8.1f us %6.1f TFLOP/s", uo, fl / uo / 1e6); }
        printf("\n");
    }
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dC)); CK(hipFree(dC2)); CK(hipFree(a.xacc)); CK(hipFree(a.xflag)); CK(hipFree(a.err));
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_sm = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.name, n_sm);
    const bool perf_only = argc > 1 && argv[1][0] == 'p';
    if (argc > 1 && argv[1][0] == 'd') { run_fused<9, 2>("fused ant split", 300, 2, 35, 512, 29, true, 5, n_sm); return 0; }
    if (!perf_only) {
        run_layer<4>("4w layer small", 300, 3, 64, 256, true, 0, n_sm, false);
        run_layer<4>("4w layer split 7 tiles / 4 wg", 300, 1, 128, 512, true, 4, n_sm, false);
        run_layer<4>("4w layer inside-one-tile", 100, 1, 256, 256, true, 3, n_sm, false);
        run_layer<4>("4w layer C3-like", 2500, 10, 512, 512, true, 0, n_sm, false);
        run_fused<9, 2, 4>("4w fused ant small", 300, 2, 35, 256, 29, true, 0, n_sm);
        run_fused<9, 2, 4>("4w fused ant split", 300, 2, 35, 512, 29, true, 5, n_sm);
        run_fused<10, 2, 4>("4w fused ant 37 in", 300, 2, 37, 512, 29, true, 5, n_sm);
        run_fused<6, 2, 4>("4w fused half-cheetah", 200, 3, 23, 512, 18, true, 7, n_sm);
        run_fused<3, 1, 4>("4w fused swimmer", 100, 5, 10, 512, 10, true, 0, n_sm);
        run_fused<9, 2, 4>("4w fused ant C3 share", 2500, 10, 35, 512, 29, true, 0, n_sm);
        run_layer("layer small", 300, 3, 64, 256, true, 0, n_sm, false);
        run_layer("layer split 7 tiles / 4 wg", 300, 1, 128, 512, true, 4, n_sm, false);       // 3 row blocks x 2 col blocks = 6 tiles on 4 workgroups
        run_layer("layer inside-one-tile", 100, 1, 256, 256, true, 3, n_sm, false);             // 1 tile on 3 workgroups: import AND export in one piece
        run_layer("layer C3-like", 2500, 10, 512, 512, true, 0, n_sm, false);
        run_fused<9, 2>("fused ant small", 300, 2, 35, 256, 29, true, 0, n_sm);
        run_fused<9, 2>("fused ant split", 300, 2, 35, 512, 29, true, 5, n_sm);
        run_fused<6, 2>("fused half-cheetah", 200, 3, 23, 512, 18, true, 7, n_sm);
        run_fused<3, 1>("fused swimmer", 100, 5, 10, 512, 10, true, 0, n_sm);
        run_fused<9, 2>("fused ant C3 share", 2500, 10, 35, 512, 29, true, 0, n_sm);
    }
    printf("-- timing (random operands) --\n");
    run_layer<4>("4w C4 hidden layer", 6250, 20, 1024, 1024, false, 0, n_sm, false);
    run_layer<4>("4w C2 hidden layer", 2500, 5, 1024, 1024, false, 0, n_sm, false);
    run_layer<4>("4w C3 hidden layer", 2500, 10, 512, 512, false, 0, n_sm, false);
    run_layer<4>("4w 8192 x 4096 x 4096", 8192, 1, 4096, 4096, false, 0, n_sm, false);
    run_fused<9, 2, 4>("4w C3 fused MLP (ant 2x512)", 2500, 10, 35, 512, 29, false, 0, n_sm);
    run_fused<6, 2, 4>("4w C2 fused MLP (half-cheetah 2x1024)", 2500, 5, 23, 1024, 18, false, 0, n_sm);
    run_layer("C4 hidden layer", 6250, 20, 1024, 1024, false, 0, n_sm, true);
    run_layer("C2 hidden layer", 2500, 5, 1024, 1024, false, 0, n_sm, true);
    run_layer("C3 hidden layer", 2500, 10, 512, 512, false, 0, n_sm, true);
    run_layer("8192 x 4096 x 4096", 8192, 1, 4096, 4096, false, 0, n_sm, true);
    run_fused<9, 2>("C3 fused MLP (ant 2x512)", 2500, 10, 35, 512, 29, false, 0, n_sm);
    run_fused<6, 2>("C2 fused MLP (half-cheetah 2x1024)", 2500, 5, 23, 1024, 18, false, 0, n_sm);
    return 0;
}
