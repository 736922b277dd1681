// Micro-benchmark: can one wave overlap its own VALU work with its own MFMAs on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define FENCE __builtin_amdgcn_sched_barrier(0)

template <int NCHAIN, int NVALU, int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters, float a, float b) {
    f32x4 acc[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned int v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) {
                acc[c] = MFMA16(a, b, acc[c]);
                FENCE;
#pragma unroll
                for (int j = 0; j < NVALU; ++j) {
                    if (OP == 0) v[j & 7] = v[j & 7] ^ v[(j + 1) & 7];          // one full-rate VALU instruction
                    else if (OP == 1) v[j & 7] = v[j & 7] * 0x9E3779B9u;          // v_mul_lo_u32
                    else v[j & 7] = __umulhi(v[j & 7], 0x9E3779B9u);                // v_mul_hi_u32
                    FENCE;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    unsigned int x = 0;
    for (int i = 0; i < 8; ++i) x ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCHAIN, int NVALU, int OP = 0>
void run(int waves_per_simd, float* d_out, unsigned long long* d_cyc) {
    const int iters = 200;
    const int threads = 256 * waves_per_simd;       // 4 SIMDs x waves_per_simd waves in ONE workgroup on one CU
    hipLaunchKernelGGL((k<NCHAIN, NVALU, OP>), dim3(1), dim3(threads), 0, 0, d_out, d_cyc, iters, 1.0f, 0.5f);
    hipLaunchKernelGGL((k<NCHAIN, NVALU, OP>), dim3(1), dim3(threads), 0, 0, d_out, d_cyc, iters, 1.0f, 0.5f);
    unsigned long long c = 0;
    hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * 8 * NCHAIN;
    printf("chains=%d valu_per_mfma=%d op=%d waves/simd=%d : %7.1f cycles per MFMA (per wave), %6.1f cycles per MFMA per SIMD\n", NCHAIN, NVALU, OP, waves_per_simd,
           c / n_mfma, c / n_mfma / waves_per_simd);
}

int main() {
    float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, sizeof(float) * 4096); hipMalloc(&d_cyc, 8);
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>(w, d_out, d_cyc); run<1, 2>(w, d_out, d_cyc); run<1, 6>(w, d_out, d_cyc); run<1, 10>(w, d_out, d_cyc);
        run<2, 0>(w, d_out, d_cyc); run<2, 2>(w, d_out, d_cyc); run<2, 6>(w, d_out, d_cyc);
        run<5, 0>(w, d_out, d_cyc); run<5, 1>(w, d_out, d_cyc); run<5, 2>(w, d_out, d_cyc); run<5, 4>(w, d_out, d_cyc); run<5, 6>(w, d_out, d_cyc); run<5, 8>(w, d_out, d_cyc);
        run<5, 12>(w, d_out, d_cyc);
        run<5, 2, 1>(w, d_out, d_cyc); run<5, 4, 1>(w, d_out, d_cyc); run<5, 2, 2>(w, d_out, d_cyc);
    }
    return 0;
}
