// HBM write rate by pattern: 512 MB written by 256 workgroups of 512 threads (16-byte stores), (a) each workgroup one contiguous 2 MB stream, (b) each workgroup a
// 1 KB wide column panel of a [rows][4 KB] matrix (1 KB pieces at a stride of 4 KB: what a 256-column tile of a 1024-wide activation matrix writes), (c) 512 B
// wide panels (128-column tiles).  build: hipcc --offload-arch=gfx950 -O3 write_pattern.hip -o write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k_write(float4* p, long long pieces_per_wg, int piece_f4, long long stride_f4, long long wg_stride_f4, int wgs_per_row, float v) {
    const int wg = blockIdx.x;
    float4* base = p + (long long)(wg / wgs_per_row) * wg_stride_f4 + (long long)(wg % wgs_per_row) * piece_f4;
    const int per_iter = 512 / piece_f4;                       // pieces a workgroup writes per instruction round
    const int t = threadIdx.x, pi = t / piece_f4, off = t % piece_f4;
    const float4 val = {v, v, v, v};
    for (long long q = pi; q < pieces_per_wg; q += per_iter) base[q * stride_f4 + off] = val;
}
int main() {
    const long long bytes = 512ll << 20;
    float4* d; (void)hipMalloc(&d, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* name; int piece_bytes; int row_bytes; } pat[] = {{"contiguous 2 MB per workgroup", 0, 0}, {"1 KB panels of 4 KB rows", 1024, 4096}, {"512 B panels of 4 KB rows", 512, 4096}, {"2 KB panels of 4 KB rows", 2048, 4096}};
    for (auto& pt : pat) {
        const int wgs = 256;
        long long pieces, stride, wgstride; int piece_f4, per_row;
        if (pt.piece_bytes == 0) { piece_f4 = 512; pieces = bytes / wgs / 8192; stride = 512; wgstride = bytes / wgs / 16; per_row = 1; }
        else { piece_f4 = pt.piece_bytes / 16; per_row = pt.row_bytes / pt.piece_bytes; const long long rows = bytes / pt.row_bytes, groups = wgs / per_row;
               pieces = rows / groups; stride = pt.row_bytes / 16; wgstride = pieces * stride; }
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_write, dim3(wgs), dim3(512), 0, 0, d, pieces, piece_f4, stride, wgstride, per_row, 1.0f);
        (void)hipEventRecord(e0, 0);
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_write, dim3(wgs), dim3(512), 0, 0, d, pieces, piece_f4, stride, wgstride, per_row, 2.0f);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-32s %7.1f us per 512 MB = %.2f TB/s\n", pt.name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
