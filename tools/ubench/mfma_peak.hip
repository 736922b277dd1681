// What the f32 matrix pipe sustains on THIS box under its power management, on random operands (zero operands clock higher):
// a register-resident issue loop of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, one and two waves per SIMD, 256 CUs.
// build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(512) k_peak(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xFFFFF]; b[i] = in[(tid * 8 + 4 + i) & 0xFFFFF]; }
    if constexpr (SHAPE == 16) {
        f32x4 acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t & 3], b[(t >> 2) & 3], acc[t], 0, 0, 0);
        }
        float s = 0; for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        out[tid] = s;
    } else {
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 3], b[(t >> 1) & 3], acc[t & 3], 0, 0, 0);
        }
        float s = 0; for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[tid] = s;
    }
}

// the same 16x16x4 issue loop with its srcA operands streamed from LDS: one ds_read_b128 per four MFMAs, four reads in flight (the operand
// pattern of csrc/mlp_streamk.h), 8 waves per CU
__global__ void __launch_bounds__(512) k_peak_lds(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = in[(i + 17 * blockIdx.x) & 0xFFFFF];
    __syncthreads();
    const float b = in[tid & 0xFFFFF];
    f32x4 acc[16], w[8];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (unsigned)(((lane >> 4) * 4 * 256 + 4 * (lane & 15)) * 4);
    asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(addr));
    for (int it = 0; it < iters; it += 2) {
#define GRP(k) asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %0, %2 offset:%3" : "=&v"(w[((k) + 4) & 7]), "+v"(w[(k) & 7]) : "v"(addr), "i"((((k) + 4) & 31) * 1024 + ((k) & 3) * 256)); \
        __builtin_amdgcn_sched_barrier(0); \
        for (int v = 0; v < 4; ++v) acc[4 * ((k) & 3) + v] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[(k) & 7][v], b, acc[4 * ((k) & 3) + v], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);
        GRP(0) GRP(1) GRP(2) GRP(3) GRP(4) GRP(5) GRP(6) GRP(7)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    float s = 0; for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[tid] = s;
}
static void run_lds(const float* din, float* dout, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak_lds, dim3(256), dim3(512), 32768, 0, din, dout, iters / 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_peak_lds, dim3(256), dim3(512), 32768, 0, din, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)iters * 16 * 2048.0 * 8 * 256;
    printf("%-28s 2 waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", "16x16x4 random, srcA from LDS", ms, flops / ms / 1e9);
}

template <int SHAPE> static void run(const char* name, int threads, const float* din, float* dout, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_peak<SHAPE>, dim3(256), dim3(threads), 0, 0, din, dout, iters / 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_peak<SHAPE>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)iters * 16 * 2048.0 * (threads / 64) * 256;     // 16 x (16x16x4) = 8 x (32x32x2) = 32768 flop per wave per iteration
    printf("%-28s %d waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", name, threads / 256, ms, flops / ms / 1e9);
}

int main() {
    std::vector<float> h(1 << 20);
    std::mt19937 g(1); std::uniform_real_distribution<float> d(-1.f, 1.f);
    float *din, *dz, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dz, h.size() * 4); hipMalloc(&dout, 256 * 512 * 4);
    for (auto& x : h) x = d(g);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(dz, 0, h.size() * 4);
    const int iters = 200000;
    for (int rep = 0; rep < 2; ++rep) {
        run<16>("16x16x4 random", 256, din, dout, iters); run<16>("16x16x4 random", 512, din, dout, iters);
        run<32>("32x32x2 random", 256, din, dout, iters); run<32>("32x32x2 random", 512, din, dout, iters);
        run<16>("16x16x4 zeros", 512, dz, dout, iters); run<32>("32x32x2 zeros", 512, dz, dout, iters);
        run_lds(din, dout, iters);
    }
    return 0;
}
