// Measured peaks for the roofline denominators (SURVEY.md 8d "measured peak"): a register-resident f32 MFMA issue loop and a
// streaming copy, on the device the ctx lives on.  Diagnostics hook (not part of include/metrpo.h); bench.py prints the two numbers
// beside the nominal ones of MI355X_MICROARCH.md -- `frac` keeps the nominal peak, the stricter denominator.
#include <algorithm>
#include "metrpo_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_probe_mfma(float* out, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.678f) out[0] = s;                 // keeps the chain alive, never true
}

// Streaming copy, U float4 per thread in flight before the first store (a grid-stride loop of single loads measured 4.8 TB/s: one 16-byte load per
// thread and iteration does not cover the HBM latency at 8 waves per SIMD; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  NT: non-temporal
// loads and stores (streamed data, nothing is re-read).  A block owns a contiguous span of U * 256 float4 per iteration.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_probe_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v* __restrict__ s = (const f4v*)src;
    f4v* __restrict__ d = (f4v*)dst;
    const size_t span = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * span; base < n; base += (size_t)gridDim.x * span) {
        f4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (i < n) v[u] = NT ? __builtin_nontemporal_load(&s[i]) : s[i]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], &d[i]); else d[i] = v[u]; } }
    }
}

// out[0] = dense f32 MFMA TFLOP/s (v_mfma_f32_32x32x2_f32, 8 waves per CU, 4 independent accumulators per wave)
// out[1] = HBM GB/s of a 1 GiB streaming copy (read + write bytes; best of 12 forms of the copy loop)
extern "C" int32_t metrpo_probe_peaks(metrpo_ctx* c, double* out, void* stream) {
    if (!c || !out) return METRPO_ENULL;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
    float* d = nullptr;
    const size_t bytes = (size_t)1 << 30;
    HIP_TRY(c, hipMalloc(&d, 2 * bytes));
    HIP_TRY(c, hipMemsetAsync(d, 0, 2 * bytes, st));
    const int iters = 4096, blocks = c->n_sm * 2;
    float ms = 0.0f;
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(256), 0, st, d, 64);       // warm-up
    HIP_TRY(c, hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(256), 0, st, d, iters);
    HIP_TRY(c, hipEventRecord(e1, st));
    HIP_TRY(c, hipEventSynchronize(e1));
    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    out[0] = (double)blocks * 4 /*waves*/ * iters * 16 /*MFMAs*/ * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
    const size_t n4 = bytes / sizeof(float4);
    // best of a few forms of the same copy (loads in flight per thread x temporal hint x grid size): the figure is the device's, not one loop's
    typedef void (*copy_t)(const float4*, float4*, size_t);
    const copy_t forms[4] = {k_probe_copy<4, false>, k_probe_copy<8, false>, k_probe_copy<4, true>, k_probe_copy<8, true>};
    const int grids[3] = {c->n_sm * 8, c->n_sm * 16, c->n_sm * 32};
    out[1] = 0.0;
    for (int f = 0; f < 4; ++f)
        for (int gi = 0; gi < 3; ++gi) {
            hipLaunchKernelGGL(forms[f], dim3(grids[gi]), dim3(256), 0, st, (const float4*)d, (float4*)((char*)d + bytes), n4);
            HIP_TRY(c, hipEventRecord(e0, st));
            for (int r = 0; r < 4; ++r)
                hipLaunchKernelGGL(forms[f], dim3(grids[gi]), dim3(256), 0, st, (const float4*)d, (float4*)((char*)d + bytes), n4);
            HIP_TRY(c, hipEventRecord(e1, st));
            HIP_TRY(c, hipEventSynchronize(e1));
            HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
            out[1] = std::max(out[1], 4.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        }
    (void)hipFree(d); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}


// ---- how many CUs actually run this process's waves (CU masks, partitioned devices, reserved CUs: multiProcessorCount does not say) ----
// Census: 16 single-wave workgroups per nominal CU, each holding 64 KB of LDS (at most two per CU: they must spread) for ~3 us, marks the
// (XCC, SE, SH, CU) it ran on.  The kernels that exchange data between the workgroups of ONE launch (rollout_resident.hip) need their whole
// grid on the chip at once; they size their grids by this count instead of trusting the property.
__global__ void __launch_bounds__(64) k_cu_census(unsigned* __restrict__ seen) {
    extern __shared__ char census_pad[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: [11:8] CU, [12] SH, [15:13] SE
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID: [3:0]
        const unsigned key = ((xcc & 0xFu) << 8) | ((hw >> 8) & 0xFFu);
        atomicOr(&seen[key >> 5], 1u << (key & 31));
        census_pad[0] = (char)key;                                             // keeps the allocation
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 300ull) __builtin_amdgcn_s_sleep(16);      // 3 us at 100 MHz
    }
}
// schedulable CUs of this context's device: measured in metrpo_create and again by metrpo_set_exclusive (never inside a launch path: the census allocates and
// synchronises).  Two censuses, the larger count: a tenant or an earlier launch holding CUs during the ~3 us window would otherwise shrink every resident
// and cooperative grid for the context's whole life.
int sched_cus(metrpo_ctx* c, hipStream_t st) {
    if (c->n_cu_sched > 0) return c->n_cu_sched;
    unsigned* d = nullptr;
    const int words = 128;                                                      // 12-bit keys
    if (hipMalloc(&d, words * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return c->n_sm; }
    unsigned h[words];
    int best = 0;
    bool ok = true;
    for (int rep = 0; rep < 2 && ok; ++rep) {
        ok = hipMemsetAsync(d, 0, words * sizeof(unsigned), st) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_cu_census, dim3(16 * c->n_sm), dim3(64), 65536, st, d);
            ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        }
        if (ok) { int n = 0; for (int i = 0; i < words; ++i) n += __builtin_popcount(h[i]); best = std::max(best, n); }
    }
    (void)hipFree(d);
    if (!ok) { (void)hipGetLastError(); return c->n_sm; }
    c->n_cu_sched = (best > 0 && best <= c->n_sm) ? best : c->n_sm;
    return c->n_cu_sched;
}
// grid <= (workgroups of this kernel the runtime says fit a CU) x (CUs that really schedule our waves); exclusive use of the device as told by the host
bool grid_is_coresident(metrpo_ctx* c, const void* kernel, int threads, size_t lds, long long grid, hipStream_t st) {
    if (!ctx_exclusive(c)) return false;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (per_cu > 1) per_cu -= 1;                                                // the API answers one too many at some SGPR counts (MI355X_MICROARCH.md, residency)
    return per_cu >= 1 && grid <= (long long)per_cu * sched_cus(c, st);
}
extern "C" int32_t metrpo_schedulable_cus(metrpo_ctx* c, void* stream) {
    if (!c) return METRPO_ENULL;
    return sched_cus(c, (hipStream_t)stream);
}
// exclusive = 0: other compute processes share this GPU -- no kernel that waits on other workgroups of its own launch is selected
extern "C" int32_t metrpo_set_exclusive(metrpo_ctx* c, int32_t exclusive) {
    if (!c) return METRPO_ENULL;
    c->exclusive = exclusive ? 1 : 0;
    if (c->exclusive) { c->n_cu_sched = 0; (void)sched_cus(c, nullptr); }      // the tenants may have changed: count again, here and not inside the next launch
    return METRPO_OK;
}
