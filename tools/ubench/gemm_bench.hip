// Stand-alone timing of the batched f32 MFMA GEMM (csrc/gemm_mfma.h) on the shapes the rollout / training paths launch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../me-trpo_amd/csrc -I../../include gemm_bench.hip -o gemm_bench
#include "gemm_mfma.h"
#include <cstdio>
#include <vector>

template <int TM, int TN, int EPI, bool TA, bool TB>
static void bench(const char* name, int M, int N, int Kd, int heads) {
    const size_t nA = (size_t)heads * M * Kd, nW = (size_t)heads * Kd * N, nC = (size_t)heads * M * N;
    float *A, *W, *C, *bias;
    (void)hipMalloc(&A, nA * 4); (void)hipMalloc(&W, nW * 4); (void)hipMalloc(&C, nC * 4); (void)hipMalloc(&bias, (size_t)heads * N * 4);
    (void)hipMemset(A, 0, nA * 4); (void)hipMemset(W, 0, nW * 4); (void)hipMemset(bias, 0, (size_t)heads * N * 4);
    GemmEpi ep = {}; ep.bias = bias; ep.strideBias = N; ep.mask = C; ep.strideMask = (long long)M * N; ep.ldm = N;
    const int lda = TA ? M : Kd, ldw = TB ? Kd : N;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) gemm_mfma_launch<TM, TN, EPI, TA, TB>(A, (long long)M * Kd, lda, W, (long long)Kd * N, ldw, C, (long long)M * N, N, M, N, Kd, heads, ep, 0);
    (void)hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) gemm_mfma_launch<TM, TN, EPI, TA, TB>(A, (long long)M * Kd, lda, W, (long long)Kd * N, ldw, C, (long long)M * N, N, M, N, Kd, heads, ep, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-34s M=%5d N=%5d K=%5d heads=%2d : %8.1f us  %6.1f TFLOP/s\n", name, M, N, Kd, heads, ms * 1e3, 2.0 * M * N * Kd * heads / ms / 1e9);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(bias);
}

int main() {
    for (int heads : {5, 10}) {
        const int M = 2500, N = heads == 5 ? 1024 : 512, K = N;
        bench<2, 2, EPI_BIAS_RELU, false, false>("128x128", M, N, K, heads);
        bench<1, 2, EPI_BIAS_RELU, false, false>(" 64x128", M, N, K, heads);
        bench<2, 1, EPI_BIAS_RELU, false, false>("128x 64", M, N, K, heads);
        bench<1, 1, EPI_BIAS_RELU, false, false>(" 64x 64", M, N, K, heads);
    }
    bench<2, 2, EPI_BIAS_RELU, false, false>("128x128", 1000, 1024, 1024, 5);
    bench<1, 2, EPI_BIAS_RELU, false, false>(" 64x128", 1000, 1024, 1024, 5);
    bench<1, 1, EPI_BIAS_RELU, false, false>(" 64x 64", 1000, 1024, 1024, 5);
    bench<2, 2, EPI_BIAS_RELU, false, false>("128x128", 6250, 1024, 1024, 20);
    bench<2, 2, EPI_BIAS_RELU, false, false>("128x128", 8192, 4096, 4096, 1);
    return 0;
}
