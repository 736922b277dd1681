// Does a hipGraph shorten a chain of small DEPENDENT kernels?  The policy update is ~40 such launches per iteration (k_finalize 8 us, k_try_theta 5 us, ...):
// the same chain of 100 read-modify-write kernels (64 blocks x 256 threads; and 1024-thread 47-block "reduction-sized" ones) launched on a stream and
// as one captured graph.  build: hipcc --offload-arch=gfx950 -O3 graph_chain.hip -o graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_touch(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
static float* d_buf;
static void chain(hipStream_t s, int blocks, int threads, int n) { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(threads), 0, s, d_buf, n); }
int main() {
    (void)hipMalloc(&d_buf, 1 << 22); (void)hipMemset(d_buf, 0, 1 << 22);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape) {
        const int blocks = shape ? 47 : 64, threads = shape ? 1024 : 256, n = blocks * threads;
        chain(st, blocks, threads, n); (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 20; ++r) chain(st, blocks, threads, n);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%2d blocks x %4d threads, stream launches : %6.2f us per kernel\n", blocks, threads, ms * 1e3 / 2000);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        chain(st, blocks, threads, n);
        (void)hipStreamEndCapture(st, &g);
        hipError_t rc = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 20; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%2d blocks x %4d threads, one hipGraph     : %6.2f us per kernel (instantiate rc=%d)\n", blocks, threads, ms * 1e3 / 2000, (int)rc);
    }
    return 0;
}
