// BaseSampler.process_samples on device (samplers/base.py:48-104,163-167):
//   k_gae      -- baseline predict + GAE + discounted returns, one thread per env column, reverse in time
//   k_center   -- [rllab] util.center_advantages
//   k_gram     -- normal equations of [rllab] LinearFeatureBaseline.fit
// All three are HBM-bound streaming kernels over the time-major trajectory tensors: a wave reads
// 64 consecutive envs of one time step, so every access is coalesced without any transposition.
#include "device_common.h"

// features [o, o^2, al, al^2, al^3, 1], o = clip(obs,-10,10), al = t/100   ([rllab] LinearFeatureBaseline._features)
__device__ __forceinline__ double baseline_value(const float* __restrict__ obs_row, int ns, int tpath,
                                                 const double* __restrict__ coeffs) {
    double v = 0.0;
    for (int i = 0; i < ns; ++i) {
        const double o = fmin(fmax((double)obs_row[i], -10.0), 10.0);
        v += coeffs[i] * o + coeffs[ns + i] * o * o;
    }
    const double al = (double)tpath / 100.0;
    v += coeffs[2 * ns] * al + coeffs[2 * ns + 1] * al * al + coeffs[2 * ns + 2] * al * al * al + coeffs[2 * ns + 3];
    return v;
}

__global__ void k_gae(const float* __restrict__ obs, const float* __restrict__ rew, const uint8_t* __restrict__ done,
                      const int32_t* __restrict__ tpath, int T, int B, int ns, const double* __restrict__ coeffs,
                      double gamma, double lam, float* __restrict__ adv, float* __restrict__ ret,
                      uint8_t* __restrict__ valid, double* __restrict__ stats) {
    __shared__ double red[16];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    if (b < B) {
        double a_next = 0.0, v_next = 0.0, r_next = 0.0;
        bool complete = false;            // becomes true at the last done of the column: later (earlier-in-time) samples are whole paths
        for (int t = T - 1; t >= 0; --t) {
            const size_t tb = (size_t)t * B + b;
            if (done[tb]) { a_next = 0.0; v_next = 0.0; r_next = 0.0; complete = true; }   // path_baselines = append(V, 0), base.py:58
            const double r = (double)rew[tb];
            const double v = (coeffs != nullptr) ? baseline_value(obs + tb * ns, ns, tpath[tb], coeffs) : 0.0;
            const double delta = r + gamma * v_next - v;                                   // base.py:59-61
            const double a = delta + gamma * lam * a_next;                                 // discount_cumsum(deltas, g*lam), :62-63
            const double g = r + gamma * r_next;                                           // discount_cumsum(rewards, g), :64
            adv[tb] = (float)a; ret[tb] = (float)g; valid[tb] = complete ? 1 : 0;
            if (complete) { s1 += a; s2 += a * a; cnt += 1.0; }
            a_next = a; v_next = v; r_next = g;
        }
    }
    const double t1 = block_sum(s1, red), t2 = block_sum(s2, red), t3 = block_sum(cnt, red);
    if (threadIdx.x == 0) { atomicAdd(&stats[0], t1); atomicAdd(&stats[1], t2); atomicAdd(&stats[2], t3); }
}

__global__ void k_center(float* __restrict__ adv, const uint8_t* __restrict__ valid, int64_t N,
                         const double* __restrict__ stats) {
    const double n = stats[2];
    const double mean = stats[0] / n;
    const double var = fmax(stats[1] / n - mean * mean, 0.0);
    const double inv = 1.0 / (sqrt(var) + 1e-8);                  // (a - mean) / (std + 1e-8)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = (valid == nullptr || valid[i]) ? (float)(((double)adv[i] - mean) * inv) : 0.0f;
}

// Gram matrix: block stages a tile of GS samples' feature rows (double) in LDS, then every thread owns
// a strided subset of the F*F (+F) outputs and walks the tile.  Per-block results -> double atomics.
#define GRAM_TILE 64
__global__ void k_gram(const float* __restrict__ obs, const float* __restrict__ ret, const int32_t* __restrict__ tpath,
                       const uint8_t* __restrict__ valid, int64_t N, int ns, double* __restrict__ AtA,
                       double* __restrict__ Aty) {
    extern __shared__ __attribute__((aligned(16))) double feat[];     // [GRAM_TILE][F+1]  (last col = return)
    const int F = 2 * ns + 4, LDF = F + 1;
    const int nout = F * F + F;
    // each thread accumulates its outputs across all tiles of this block in registers (<= 8 per thread per pass)
    for (int p0 = 0; p0 < nout; p0 += blockDim.x * 8) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t base = (int64_t)blockIdx.x * GRAM_TILE; base < N; base += (int64_t)gridDim.x * GRAM_TILE) {
            __syncthreads();
            for (int e = threadIdx.x; e < GRAM_TILE * LDF; e += blockDim.x) {
                const int sidx = e / LDF, f = e % LDF;
                const int64_t n = base + sidx;
                double v = 0.0;
                if (n < N && (valid == nullptr || valid[n])) {
                    if (f < ns) v = fmin(fmax((double)obs[n * ns + f], -10.0), 10.0);
                    else if (f < 2 * ns) { const double o = fmin(fmax((double)obs[n * ns + f - ns], -10.0), 10.0); v = o * o; }
                    else if (f == F) v = (double)ret[n];
                    else {
                        const double al = (double)tpath[n] / 100.0;
                        const int q = f - 2 * ns;
                        v = (q == 0) ? al : (q == 1) ? al * al : (q == 2) ? al * al * al : 1.0;
                    }
                }
                feat[e] = v;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + u * blockDim.x + threadIdx.x;
                if (p < nout) {
                    const int i = (p < F * F) ? p / F : p - F * F;
                    const int j = (p < F * F) ? p % F : F;            // Aty uses the return column
                    double a = 0.0;
                    for (int sidx = 0; sidx < GRAM_TILE; ++sidx) a += feat[sidx * LDF + i] * feat[sidx * LDF + j];
                    acc[u] += a;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * blockDim.x + threadIdx.x;
            if (p < nout) {
                if (p < F * F) atomicAdd(&AtA[p], acc[u]);
                else atomicAdd(&Aty[p - F * F], acc[u]);
            }
        }
    }
}

int launch_gae(metrpo_ctx* c, const float* obs, const float* rew, const uint8_t* done, const int32_t* tpath, int T,
               int B, const double* coeffs, double gamma, double lam, float* adv, float* ret, uint8_t* valid,
               double* stats, hipStream_t st) {
    const int bs = 64;
    hipLaunchKernelGGL(k_gae, dim3((B + bs - 1) / bs), dim3(bs), 0, st, obs, rew, done, tpath, T, B, c->pd.ns, coeffs,
                       gamma, lam, adv, ret, valid, stats);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_center(metrpo_ctx* c, float* adv, const uint8_t* valid, int64_t N, const double* stats, hipStream_t st) {
    const int bs = 256;
    const int grid = (int)std::min<int64_t>((N + bs - 1) / bs, (int64_t)c->n_sm * 8);
    hipLaunchKernelGGL(k_center, dim3(grid), dim3(bs), 0, st, adv, valid, N, stats);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_gram(metrpo_ctx* c, const float* obs, const float* ret, const int32_t* tpath, const uint8_t* valid,
                int64_t N, double* AtA, double* Aty, hipStream_t st) {
    const int F = 2 * c->pd.ns + 4;
    const int bs = 256;
    const size_t sh = sizeof(double) * GRAM_TILE * (F + 1);
    const int grid = (int)std::min<int64_t>((N + GRAM_TILE - 1) / GRAM_TILE, (int64_t)c->n_sm * 4);
    hipLaunchKernelGGL(k_gram, dim3(grid), dim3(bs), sh, st, obs, ret, tpath, valid, N, c->pd.ns, AtA, Aty);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
