// Platform constants behind DESIGN.md's fuse-or-launch decisions: GPU-side cost of a back-to-back kernel launch, of a dependent
// tiny-kernel chain, and of a cooperative grid barrier (512 workgroups x 256 threads, the policy-update grid).
// build: hipcc --offload-arch=gfx950 -O3 launch_sync.hip -o launch_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_empty(float* p) { if (p && threadIdx.x == 1234567) p[0] = 1.f; }
__global__ void k_touch(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void k_gridsync(float* p, int iters) {
    cg::grid_group g = cg::this_grid();
    float v = 0.f;
    for (int i = 0; i < iters; ++i) { v += 1.f; g.sync(); }
    if (blockIdx.x == 0 && threadIdx.x == 0) p[0] = v;
}

static float timeit(void (*fn)(hipStream_t), int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fn(0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) fn(0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
static float* d_buf;
int main() {
    (void)hipMalloc(&d_buf, 1 << 22);
    (void)hipMemset(d_buf, 0, 1 << 22);
    const float t_empty = timeit([](hipStream_t s) { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d_buf); }, 20) / 100;
    const float t_empty_big = timeit([](hipStream_t s) { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, s, d_buf); }, 20) / 100;
    const float t_touch = timeit([](hipStream_t s) { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_touch, dim3(64), dim3(256), 0, s, d_buf, 16384); }, 20) / 100;
    printf("back-to-back launch, 1 block x 64 threads, empty      : %6.2f us per kernel\n", t_empty * 1e3);
    printf("back-to-back launch, 512 blocks x 256 threads, empty  : %6.2f us per kernel\n", t_empty_big * 1e3);
    printf("dependent chain of 64-block read-modify-write kernels : %6.2f us per kernel\n", t_touch * 1e3);
    int iters = 1000;
    void* args[] = {&d_buf, &iters};
    for (int blocks : {256, 512}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipError_t rc = hipLaunchCooperativeKernel((const void*)k_gridsync, dim3(blocks), dim3(256), args, 0, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        rc = hipLaunchCooperativeKernel((const void*)k_gridsync, dim3(blocks), dim3(256), args, 0, 0);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("cooperative grid.sync(), %d blocks x 256 threads       : %6.2f us per barrier (rc=%d)\n", blocks, ms * 1e3 / iters, (int)rc);
    }
    return 0;
}
