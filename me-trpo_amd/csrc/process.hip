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
    for (int i0 = 0; i0 < ns; i0 += 8) {                 // 8 observation loads in flight, then the (ordered) float64 sum: the one-load-per-iteration
        float ob[8];                                     // form of this loop was a chain of ns dependent HBM round trips per sample
#pragma unroll
        for (int u = 0; u < 8; ++u) ob[u] = (i0 + u < ns) ? obs_row[i0 + u] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < ns) {
                const double o = fmin(fmax((double)ob[u], -10.0), 10.0);
                v += coeffs[i0 + u] * o + coeffs[ns + i0 + u] * o * o;
            }
        }
    }
    const double al = (double)tpath / 100.0;
    v += coeffs[2 * ns] * al + coeffs[2 * ns + 1] * al * al + coeffs[2 * ns + 2] * al * al * al + coeffs[2 * ns + 3];
    return v;
}

#ifndef GAE_CHUNK
#define GAE_CHUNK 7
#endif
#ifndef GAE_NW
#define GAE_NW 8          // wavefronts per 64-env tile = time chunks scanned concurrently (4 x 10-step batches: 15.1 us at C1, 8 x 7: 12.9, 16 x 7: 15.6)
#endif

// One reverse pass over the steps [t_lo, t_hi) of env column b, starting from the carry (a_next, v_next, r_next, complete) of step t_hi:
// the arithmetic of samplers/base.py:57-64 in float64.  WRITE = false only returns the carry at t_lo (chunk aggregate).
template <bool WRITE>
__device__ __forceinline__ void gae_chunk_pass(const double* __restrict__ V, const float* __restrict__ rew, const uint8_t* __restrict__ done,
                                               int t_lo, int t_hi, int B, int b, double gamma, double lam, double& a_next, double& v_next,
                                               double& r_next, bool& complete, float* __restrict__ adv, float* __restrict__ ret,
                                               uint8_t* __restrict__ valid, double& s1, double& s2, double& cnt) {
    for (int t1 = t_hi; t1 > t_lo; t1 -= GAE_CHUNK) {
        // issue the loads of a whole chunk before entering the (dependent) recurrence
        float rr[GAE_CHUNK]; uint8_t dd[GAE_CHUNK]; double vv[GAE_CHUNK];
#pragma unroll
        for (int u = 0; u < GAE_CHUNK; ++u) {
            const int t = t1 - 1 - u;
            const size_t tb = (size_t)(t < t_lo ? t_lo : t) * B + b;
            rr[u] = rew[tb]; dd[u] = done[tb]; vv[u] = (V != nullptr) ? V[tb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < GAE_CHUNK; ++u) {
            const int t = t1 - 1 - u;
            if (t >= t_lo) {
                if (dd[u]) { a_next = 0.0; v_next = 0.0; r_next = 0.0; complete = true; }   // path_baselines = append(V, 0), base.py:58
                const double r = (double)rr[u], v = vv[u];
                const double delta = r + gamma * v_next - v;                               // base.py:59-61
                const double a = delta + gamma * lam * a_next;                             // discount_cumsum(deltas, g*lam), :62-63
                const double g = r + gamma * r_next;                                       // discount_cumsum(rewards, g), :64
                if (WRITE) {
                    const size_t tb = (size_t)t * B + b;
                    adv[tb] = (float)a; ret[tb] = (float)g; valid[tb] = complete ? 1 : 0;
                    if (complete) { s1 += a; s2 += a * a; cnt += 1.0; }
                }
                a_next = a; v_next = v; r_next = g;
            }
        }
    }
}

// GAE / returns as a prefix sum ACROSS wavefronts (reduce-then-scan over time): a 64-env tile is handled by GAE_NW waves, wave w owning
// the time chunk [w*Tc, (w+1)*Tc) with one lane per env (time-major rows -> coalesced).  The reverse recurrences
//     a_t = delta_t + g*lam*a_{t+1},  ret_t = r_t + g*ret_{t+1}     (reset at every done)
// are affine in the carry that enters a chunk from later time, so
//   pass 1  every wave scans its chunk with a ZERO carry and publishes the chunk aggregate (carry it hands to the earlier chunk for a
//           zero carry-in, the coefficients by which a non-zero carry-in would change it, whether a done cut the dependence);
//   combine after one barrier every wave composes the aggregates of the later chunks (<= GAE_NW-1 fused multiply-adds per env);
//   pass 2  every wave rescans its chunk from its true carry with exactly the float64 arithmetic of the sequential scan and writes
//           adv / ret / valid (equal to a single sequential pass up to float64 rounding of the composed carry).
// The dependent chain per env is 2*T/GAE_NW steps instead of T, and GAE_NW times as many waves are in flight.
// Phase 0 (fused baseline.predict, samplers/base.py:55): the value V[t][b] = features(obs[t][b], tpath) . coeffs of every sample of the
// tile is formed by the same block before the scan -- wave w takes the steps t = w, w + GAE_NW, ...; the 64 x ns observation block of a
// step is ONE contiguous run of the time-major tensor, read with coalesced loads (the block of the wave's NEXT step is already in flight),
// turned through a wave-private LDS buffer so that every lane ends up with its own env's row, evaluated in float64 and kept in the ctx's
// V buffer for the two scan passes (L2-resident: written and read by the same workgroup).  As a separate thread-per-sample kernel this
// read rows of ns floats at a stride of ns floats per lane: 0.26 TB/s on Ant (560 us at the C3 share, 10x the scan itself).
// NS = compile-time ns of the six envs (register-resident prefetch), NS = 0: any ns (no prefetch).
// baseline.predict for ALL (step, 64-env block) pairs at once: a wave per pair.  k_gae's own first phase deals a block's steps over the 8 waves of ONE workgroup
// per 64 env columns -- at the params-file batches (100-500 envs: 2-8 workgroups on 256 CUs) 12-60 dependent steps per wave, each a serial 2 ns + 4 term
// float64 dot product per lane: 150 us (Ant) / 270 us (Humanoid) of a 50 000-sample batch.  Same staging, same baseline_value: the same V bit for bit.
__global__ void __launch_bounds__(256) k_baseline_predict(const float* __restrict__ obs, const int32_t* __restrict__ tpath, const double* __restrict__ coeffs,
                                                          int ns, double* __restrict__ V, int T, int B) {
    extern __shared__ __attribute__((aligned(16))) float stage[];      // [4][64 * ns]
    const int lane0 = threadIdx.x & 63, w0 = threadIdx.x >> 6;
    const int t = blockIdx.y * 4 + w0;
    if (t >= T) return;
    const int b0 = blockIdx.x * 64, nv = min(64, B - b0);
    float* S = stage + (size_t)w0 * 64 * ns;
    const float* __restrict__ src = obs + ((size_t)t * B + b0) * ns;
    for (int e = lane0; e < nv * ns; e += 64) S[e] = src[e];
    const int tp = (lane0 < nv) ? tpath[(size_t)t * B + b0 + lane0] : 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (lane0 < nv) V[(size_t)t * B + b0 + lane0] = baseline_value(S + lane0 * ns, ns, tp, coeffs);
}

template <int NS, int NW>
__global__ void __launch_bounds__(64 * NW) k_gae(const float* __restrict__ obs, const int32_t* __restrict__ tpath, const double* __restrict__ coeffs,
                                                      int ns_rt, int pre_v, double* __restrict__ V, const float* __restrict__ rew, const uint8_t* __restrict__ done,
                                                      int T, int B, double gamma, double lam, float* __restrict__ adv, float* __restrict__ ret,
                                                      uint8_t* __restrict__ valid, double* __restrict__ stats, double* gpart) {
    extern __shared__ __attribute__((aligned(16))) float stage[];      // [NW][64 * ns]
    if (coeffs != nullptr && pre_v) { /* V written by k_baseline_predict (few env columns: the predictions of all steps at once, chip-wide) */ }
    else if (coeffs != nullptr) {
        const int ns = NS ? NS : ns_rt;
        const int lane0 = threadIdx.x & 63, w0 = threadIdx.x >> 6;
        const int b0 = blockIdx.x * 64, nv = min(64, B - b0);
        float* S = stage + (size_t)w0 * 64 * ns;
        constexpr int NR = NS ? NS : 1;
        float cur[NR], nxt[NR];
        auto load = [&](int t, float (&r)[NR]) {
            const float* __restrict__ src = obs + ((size_t)t * B + b0) * NS;
#pragma unroll
            for (int i = 0; i < NR; ++i) { const int e = i * 64 + lane0; r[i] = (e < nv * NS) ? src[e] : 0.0f; }
        };
        // the step's path-time index travels with its observation block (consumed one iteration after it was requested, like the block)
        int tp_cur = 0, tp_nxt = 0;
        auto load_tp = [&](int t) { return (lane0 < nv) ? tpath[(size_t)t * B + b0 + lane0] : 0; };
        if (w0 < T) { if (NS) load(w0, nxt); tp_nxt = load_tp(w0); }
        for (int t = w0; t < T; t += NW) {
            tp_cur = tp_nxt;
            if (t + NW < T) tp_nxt = load_tp(t + NW);
            if (NS) {
#pragma unroll
                for (int i = 0; i < NR; ++i) cur[i] = nxt[i];
                if (t + NW < T) load(t + NW, nxt);
#pragma unroll
                for (int i = 0; i < NR; ++i) S[i * 64 + lane0] = cur[i];
            } else {
                const float* __restrict__ src = obs + ((size_t)t * B + b0) * ns;
                for (int e = lane0; e < nv * ns; e += 64) S[e] = src[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (lane0 < nv) V[(size_t)t * B + b0 + lane0] = baseline_value(S + lane0 * ns, ns, tp_cur, coeffs);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                             // every wave's V rows are visible to the whole workgroup from here on
    } else V = nullptr;
    __shared__ double red[16];
    __shared__ double agg[NW][5][64];           // per chunk and env: A0, G0, v_first, CA, CG   (carry-out = A0 + CA*E_in, G0 + CG*r_in)
    __shared__ uint8_t cut[NW][64];             // a done inside the chunk: the carry-in does not reach the chunk's first step
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int Tc = (T + NW - 1) / NW;
    const int t_lo = min(T, w * Tc), t_hi = min(T, (w + 1) * Tc);
    const bool act = b < B;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    {   // pass 1: zero carry-in
        double a_n = 0.0, v_n = 0.0, r_n = 0.0; bool comp = false;
        if (act) gae_chunk_pass<false>(V, rew, done, t_lo, t_hi, B, b, gamma, lam, a_n, v_n, r_n, comp, adv, ret, valid, s1, s2, cnt);
        const int len = t_hi - t_lo;
        agg[w][0][lane] = a_n; agg[w][1][lane] = r_n; agg[w][2][lane] = v_n;
        // the carry enters the last step of the chunk as E = g*v_in + g*lam*a_in and r_in; without a done it reaches the first step scaled
        // by (g*lam)^(len-1) and g^len
        agg[w][3][lane] = (comp || len == 0) ? 0.0 : pow(gamma * lam, (double)(len - 1));
        agg[w][4][lane] = (comp || len == 0) ? 0.0 : pow(gamma, (double)len);
        cut[w][lane] = comp ? 1 : 0;
    }
    __syncthreads();
    // combine: carry entering chunk w = carry-out of chunk w+1 (which in turn depends on the carry entering it, ...), from the last chunk down
    double a_in = 0.0, v_in = 0.0, r_in = 0.0; bool c_in = false;
    for (int j = NW - 1; j > w; --j) {
        const int lo = min(T, j * Tc), hi = min(T, (j + 1) * Tc);
        if (hi <= lo) continue;                          // empty chunk: the carry passes through
        const double E = gamma * v_in + gamma * lam * a_in;
        a_in = agg[j][0][lane] + agg[j][3][lane] * E;
        r_in = agg[j][1][lane] + agg[j][4][lane] * r_in;
        v_in = agg[j][2][lane];
        c_in = c_in || cut[j][lane];
    }
    if (act) gae_chunk_pass<true>(V, rew, done, t_lo, t_hi, B, b, gamma, lam, a_in, v_in, r_in, c_in, adv, ret, valid, s1, s2, cnt);
    const double t1s = block_sum(s1, red), t2s = block_sum(s2, red), t3s = block_sum(cnt, red);
    // The statistics of the batch = the workgroups' sums added IN WORKGROUP ORDER by whichever workgroup finishes last (float64 atomics would add
    // them in arrival order: the centred advantages, and with them the whole update, would differ in the last bit from run to run).
    // gpart = arrival ticket (first 8 bytes; zero at allocation, reset here), then [gridDim.x][3] sums.
    __shared__ unsigned int s_last;
    unsigned int* ticket = (unsigned int*)gpart;
    gpart += 1;
    if (threadIdx.x == 0) {
        gpart[3 * blockIdx.x + 0] = t1s; gpart[3 * blockIdx.x + 1] = t2s; gpart[3 * blockIdx.x + 2] = t3s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == gridDim.x - 1) ? 1u : 0u;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;                  // lane l adds workgroups l, l + 64, ...; then a butterfly of fixed shape
    for (int j = threadIdx.x; j < (int)gridDim.x; j += 64) {
        a0 += __hip_atomic_load(gpart + 3 * j + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a1 += __hip_atomic_load(gpart + 3 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a2 += __hip_atomic_load(gpart + 3 * j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if (threadIdx.x == 0) {
        stats[0] += a0; stats[1] += a1; stats[2] += a2;       // ACCUMULATED (metrpo.h): one add per launch, in stream order
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
// a strided subset of the F*F (+F) outputs and walks the tile.  Per-block results -> one partial matrix per block (part [gridDim.x][F*F+F]),
// added in block order by k_gram_final like the MFMA kernels' partials (float64 atomics would add them in arrival order).
#define GRAM_TILE 64
__global__ void k_gram(const float* __restrict__ obs, const float* __restrict__ ret, const int32_t* __restrict__ tpath,
                       const uint8_t* __restrict__ valid, int64_t N, int ns, double* __restrict__ part) {
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
            if (p < nout) part[(size_t)blockIdx.x * nout + p] = acc[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Gram matrix on the f32 matrix core.  G = F^T F (+ F^T y as an extra feature column) contracts over
// samples: A[i = feature][k = sample] and B[k = sample][j = feature] are the SAME per-lane value
// (lane l: feature 16cb + (l&15), sample 4s + (l>>4)), so one feature evaluation feeds both operands.
// Products are exact f32; each 16-sample tile is accumulated in the MFMA accumulator, then folded
// into float64 lane accumulators (the reference forms these sums in float64), block-reduced in LDS in
// fixed order and written as one partial matrix per block; k_gram_final adds the blocks in order.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NFB>
__global__ void __launch_bounds__(256) k_gram_mfma(const float* __restrict__ obs, const float* __restrict__ ret,
                                                   const int32_t* __restrict__ tpath, const uint8_t* __restrict__ valid,
                                                   long long N, int ns, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) double red[];      // [4][M*M]
    constexpr int M = NFB * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, q = lane >> 4;
    const int F = 2 * ns + 4;
    double acc[NFB][NFB][4];
#pragma unroll
    for (int a = 0; a < NFB; ++a)
#pragma unroll
        for (int b = 0; b < NFB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;
    const long long ntiles = (N + 15) / 16;
    // Raw inputs of a tile (4 k-steps x {valid, step index, return, NFB observation columns}) are fetched ONE TILE AHEAD: 20 small
    // gathers per lane whose HBM latency otherwise sits between every two tiles of a wave (the kernel ran at 0.5 TB/s).
    struct TileIn { float o[4][NFB]; float rv[4]; int tp[4]; int ok[4]; };
    auto fetch = [&](long long tile, TileIn& in) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long long n = tile * 16 + 4 * s + q;
            const bool inr = tile < ntiles && n < N;
            const long long nc = inr ? n : 0;
            in.ok[s] = inr ? ((valid == nullptr) ? 1 : (int)valid[nc]) : 0;
            in.tp[s] = tpath[nc];
            in.rv[s] = ret[nc];
#pragma unroll
            for (int cb = 0; cb < NFB; ++cb) {
                const int f = 16 * cb + c;
                const int fo = (f < ns) ? f : ((f < 2 * ns) ? f - ns : 0);
                in.o[s][cb] = obs[nc * ns + fo];
            }
        }
    };
    TileIn nxt;
    fetch((long long)blockIdx.x * 4 + wave, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) here, not at the top of every iteration (see policy_mfma.hip)
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const TileIn in = nxt;
        fetch(tile + (long long)gridDim.x * 4, nxt);
        asm volatile("" ::: "memory");                      // the loads are issued here, before this tile's arithmetic
        f32x4 g[NFB][NFB];
#pragma unroll
        for (int a = 0; a < NFB; ++a)
#pragma unroll
            for (int b = 0; b < NFB; ++b) g[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        float val[4][NFB];                                  // branch-free feature evaluation (selection in registers)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float al = (float)in.tp[s] / 100.0f;
            const float rv = in.rv[s];
#pragma unroll
            for (int cb = 0; cb < NFB; ++cb) {
                const int f = 16 * cb + c;
                const float o = fminf(fmaxf(in.o[s][cb], -10.f), 10.f);
                const int kq = f - 2 * ns;
                float x = (f < ns) ? o : o * o;
                x = (kq == 0) ? al : x;
                x = (kq == 1) ? al * al : x;
                x = (kq == 2) ? al * al * al : x;
                x = (kq == 3) ? 1.0f : x;
                x = (f == F) ? rv : x;
                x = (f > F) ? 0.0f : x;
                val[s][cb] = in.ok[s] ? x : 0.0f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NFB; ++a)
#pragma unroll
                for (int b = 0; b < NFB; ++b) g[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(val[s][a], val[s][b], g[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < NFB; ++a)
#pragma unroll
            for (int b = 0; b < NFB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] += (double)g[a][b][r];
    }
    double* mine = red + (size_t)wave * M * M;
#pragma unroll
    for (int a = 0; a < NFB; ++a)
#pragma unroll
        for (int b = 0; b < NFB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 * a + 4 * q + r) * M + 16 * b + c] = acc[a][b][r];   // D layout: row 4q+r, col c
    __syncthreads();
    // compact partial row: [F*F] Gram entries then [F] entries of F^T y (column F of the padded tile matrix)
    const int nout = F * F + F;
    double* out = part + (size_t)blockIdx.x * nout;
    for (int p = threadIdx.x; p < nout; p += 256) {
        const int i = (p < F * F) ? p / F : p - F * F, j = (p < F * F) ? p % F : F;
        const int e = i * M + j;
        out[p] = (red[e] + red[M * M + e]) + (red[2 * M * M + e] + red[3 * M * M + e]);
    }
}

__global__ void __launch_bounds__(1024) k_gram_final(const double* __restrict__ part, int nblocks, int M, int F,
                                                     double* __restrict__ AtA, double* __restrict__ Aty) {
    // block = 16 outputs x 64 row-slices (many small blocks: the sum is latency-bound); fixed order -> deterministic
    __shared__ double sh[64][17];
    const int lc = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int p = blockIdx.x * 16 + lc;
    const int nout = F * F + F;
    double a = 0.0;
    if (p < nout)
        for (int b = sl; b < nblocks; b += 64) a += part[(size_t)b * nout + p];
    sh[sl][lc] = a;
    __syncthreads();
    if (sl == 0 && p < nout) {
        double s = 0.0;
        for (int w = 0; w < 64; ++w) s += sh[w][lc];
        if (p < F * F) AtA[p] += s; else Aty[p - F * F] += s;
    }
}

// Same sums for WIDE feature sets (5 to 8 column blocks: Humanoid's 114 features + the return column): the NFB (NFB + 1) / 2 upper-triangle
// 16 x 16 blocks do not fit one wave's registers in float64, so a BLOCK works on one 16-sample tile at a time: its 256 threads evaluate the
// tile's 16 x M feature values once into LDS ([k-step][q][feature], double-buffered, the next tile's raw inputs loaded a tile ahead), and every
// wave runs the MFMAs of its quarter of the block pairs on LDS operands and keeps those blocks' float64 accumulators.  The generic k_gram
// (thread per sample pair in LDS) needed 77 ms per iteration at C4.
template <int NFB>
__global__ void __launch_bounds__(256) k_gram_mfma_wide(const float* __restrict__ obs, const float* __restrict__ ret,
                                                        const int32_t* __restrict__ tpath, const uint8_t* __restrict__ valid,
                                                        long long N, int ns, double* __restrict__ part) {
    constexpr int M = NFB * 16, NPAIR = NFB * (NFB + 1) / 2, NPW = (NPAIR + 3) / 4, NV = (16 * M + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) double red[];      // epilogue: [M * M]; main loop: float FT[2][4][4][M] in the same memory
    float* FT = (float*)red;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int F = 2 * ns + 4;
    // this wave's block pairs (a <= b), enumerated row-major over the upper triangle: p = wave, wave + 4, ...
    int pa[NPW], pb[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        int p = wave + 4 * j, a = 0;
        if (p >= NPAIR) { pa[j] = -1; pb[j] = -1; continue; }
        while (p >= NFB - a) { p -= NFB - a; ++a; }
        pa[j] = a; pb[j] = a + p;
    }
    double acc[NPW][4];
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.0;
    const long long ntiles = (N + 15) / 16;
    // element v of the tile handled by this thread: sample sl = e / M (0..15), feature f = e % M, e = tid + 256 v
    struct Raw { float o[NV]; float rv[NV]; int tp[NV]; int ok[NV]; };
    auto fetch = [&](long long tile, Raw& in) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + 256 * v, sl = e / M, f = e % M;
            const long long n = tile * 16 + sl;
            const bool inr = e < 16 * M && tile < ntiles && n < N;
            const long long nc = inr ? n : 0;
            const int fo = (f < ns) ? f : ((f < 2 * ns) ? f - ns : 0);
            in.ok[v] = inr ? ((valid == nullptr) ? 1 : (int)valid[nc]) : 0;
            in.tp[v] = tpath[nc]; in.rv[v] = ret[nc]; in.o[v] = obs[nc * ns + fo];
        }
    };
    auto publish = [&](const Raw& in, float* dst) {                  // dst[k-step s][q][feature]: sample sl = 4 s + q
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int e = tid + 256 * v, sl = e / M, f = e % M;
            if (e >= 16 * M) continue;
            const float al = (float)in.tp[v] / 100.0f, o = fminf(fmaxf(in.o[v], -10.f), 10.f);
            const int kq = f - 2 * ns;
            float x = (f < ns) ? o : o * o;
            x = (kq == 0) ? al : x;
            x = (kq == 1) ? al * al : x;
            x = (kq == 2) ? al * al * al : x;
            x = (kq == 3) ? 1.0f : x;
            x = (f == F) ? in.rv[v] : x;
            x = (f > F) ? 0.0f : x;
            dst[((sl >> 2) * 4 + (sl & 3)) * M + f] = in.ok[v] ? x : 0.0f;
        }
    };
    Raw nxt;
    long long tile = blockIdx.x;
    fetch(tile, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int buf = 0;
    if (tile < ntiles) publish(nxt, FT);
    fetch(tile + gridDim.x, nxt);
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const float* cur = FT + buf * 16 * M;
        f32x4 g[NPW];
#pragma unroll
        for (int j = 0; j < NPW; ++j) g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float* row = cur + (s4 * 4 + q) * M + c;
#pragma unroll
            for (int j = 0; j < NPW; ++j)
                if (pa[j] >= 0) g[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(row[16 * pa[j]], row[16 * pb[j]], g[j], 0, 0, 0);
        }
        if (tile + gridDim.x < ntiles) publish(nxt, FT + (buf ^ 1) * 16 * M);      // the other buffer: its readers passed the barrier below one tile ago
        fetch(tile + 2LL * gridDim.x, nxt);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] += (double)g[j][r];
        __syncthreads();
        buf ^= 1;
    }
    __syncthreads();
    // epilogue: every block pair is owned by exactly one wave -> write it (and its mirror image) into the M x M matrix, then the compact row
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        if (pa[j] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                   // D layout: row 4 q + r of block a, col c of block b
            const int i = 16 * pa[j] + 4 * q + r, jj = 16 * pb[j] + c;
            red[i * M + jj] = acc[j][r];
            if (pa[j] != pb[j]) red[jj * M + i] = acc[j][r];
        }
    }
    __syncthreads();
    const int nout = F * F + F;
    double* out = part + (size_t)blockIdx.x * nout;
    for (int p = tid; p < nout; p += 256) {
        const int i = (p < F * F) ? p / F : p - F * F, j = (p < F * F) ? p % F : F;
        out[p] = red[i * M + j];
    }
}

template <int NFB>
static int launch_gram_mfma_wide(metrpo_ctx* c, const float* obs, const float* ret, const int32_t* tpath, const uint8_t* valid,
                                 int64_t N, double* AtA, double* Aty, hipStream_t st) {
    constexpr int M = NFB * 16;
    const int F = 2 * c->pd.ns + 4;
    const long long tiles = (N + 15) / 16;
    const int g = (int)std::max<long long>(1, std::min<long long>(tiles, (long long)c->n_sm * 2));
    const size_t need = (size_t)g * (F * F + F);
    if (need > c->gram_cap) {
        ws_retire(c, c->d_gram_part);
        c->d_gram_part = nullptr; c->gram_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_gram_part, need * sizeof(double)));
        c->gram_cap = need;
    }
    const size_t sh = sizeof(double) * M * M;                         // >= the 2 x 16 x M floats of the main loop
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_gram_mfma_wide<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_gram_mfma_wide<NFB>, dim3(g), dim3(256), sh, st, obs, ret, tpath, valid, (long long)N, c->pd.ns, c->d_gram_part);
    const int nout = F * F + F;
    hipLaunchKernelGGL(k_gram_final, dim3((nout + 15) / 16), dim3(1024), 0, st, c->d_gram_part, g, M, F, AtA, Aty);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

template <int NFB>
static int launch_gram_mfma(metrpo_ctx* c, const float* obs, const float* ret, const int32_t* tpath, const uint8_t* valid,
                            int64_t N, double* AtA, double* Aty, hipStream_t st) {
    constexpr int M = NFB * 16;
    const int F = 2 * c->pd.ns + 4;
    const long long tiles = (N + 15) / 16;
    const int g = (int)std::max<long long>(1, std::min<long long>((tiles + 3) / 4, (long long)c->n_sm * 2));   // inputs are prefetched one tile ahead: few blocks, few partial matrices
    const size_t need = (size_t)g * (F * F + F);
    if (need > c->gram_cap) {
        ws_retire(c, c->d_gram_part);
        c->d_gram_part = nullptr; c->gram_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_gram_part, need * sizeof(double)));
        c->gram_cap = need;
    }
    const size_t sh = sizeof(double) * 4 * M * M;
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_gram_mfma<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_gram_mfma<NFB>, dim3(g), dim3(256), sh, st, obs, ret, tpath, valid, (long long)N, c->pd.ns, c->d_gram_part);
    const int nout = F * F + F;
    hipLaunchKernelGGL(k_gram_final, dim3((nout + 15) / 16), dim3(1024), 0, st, c->d_gram_part, g, M, F, AtA, Aty);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_gae(metrpo_ctx* c, const float* obs, const float* rew, const uint8_t* done, const int32_t* tpath, int T,
               int B, const double* coeffs, double gamma, double lam, float* adv, float* ret, uint8_t* valid,
               double* stats, hipStream_t st) {
    const long long N = (long long)T * B;
    double* V = nullptr;
    if (coeffs != nullptr) {
        if ((size_t)N > c->vbuf_cap) {
            ws_retire(c, c->d_vbuf);
            c->d_vbuf = nullptr; c->vbuf_cap = 0;
            HIP_TRY(c, ws_alloc(c, (void**)&c->d_vbuf, sizeof(double) * (size_t)N));
            c->vbuf_cap = (size_t)N;
        }
        V = c->d_vbuf;
    }
    const int nblk = (B + 63) / 64;
    if ((size_t)nblk * 3 + 2 > c->gae_part_cap) {
        ws_retire(c, c->d_gae_part);
        c->d_gae_part = nullptr; c->gae_part_cap = 0;
        const size_t cap = std::max<size_t>((size_t)nblk * 3 + 2, 1024);
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_gae_part, sizeof(double) * cap));
        HIP_TRY(c, hipMemsetAsync(c->d_gae_part, 0, sizeof(double) * cap, st));       // the ticket (first word) starts at zero; every launch leaves it there
        c->gae_part_cap = cap;
    }
    const int ns = c->pd.ns;
    // few env columns and a long horizon (the params-file batches: B = 100, T = 600): the grid is 2 workgroups and the kernel is the
    // dependent chain of one wave's steps -- 16 time chunks instead of 8 shorten it (84 -> 67 us at C0-params-file); at C1 (79 workgroups) 8 is faster
    const bool wide = (B <= 256 && T >= 128 && ns <= 18);
    const int nw = wide ? 16 : GAE_NW;
    // few env columns: the predictions of all steps by a launch of their own (k_baseline_predict), the scan kernel then starts from V
    const int pre_v = (coeffs != nullptr && nblk <= 32 && T >= 16) ? 1 : 0;
    if (pre_v) {
        const size_t shp = sizeof(float) * 4 * 64 * (size_t)ns;
        if (shp > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_baseline_predict, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp));
        // (V = c->d_vbuf holds T * B doubles: ensured by the caller, metrpo_gae / launch_process_fused, before launch_gae is entered)
        hipLaunchKernelGGL(k_baseline_predict, dim3(nblk, (T + 3) / 4), dim3(256), shp, st, obs, tpath, coeffs, ns, V, T, B);
        HIP_TRY(c, hipGetLastError());                       // a failed launch is reported HERE, not as k_gae's
    }
    const size_t sh = (coeffs != nullptr && !pre_v) ? sizeof(float) * nw * 64 * (size_t)ns : 0;
#define GAE_LAUNCH_NW(NSV, NWV) do { \
        if (sh > 48 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_gae<NSV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        hipLaunchKernelGGL((k_gae<NSV, NWV>), dim3((B + 63) / 64), dim3(64 * NWV), sh, st, obs, tpath, coeffs, ns, pre_v, V, rew, done, T, B, gamma, lam, adv, ret, valid, stats, c->d_gae_part); } while (0)
#define GAE_LAUNCH(NSV) do { if (wide) GAE_LAUNCH_NW(NSV, 16); else GAE_LAUNCH_NW(NSV, GAE_NW); } while (0)
#define GAE_LAUNCH8(NSV) GAE_LAUNCH_NW(NSV, GAE_NW)
    if (sh + 48 * 1024 > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "gae: observation too wide for the staging buffer");
    switch (ns) {                                    // the six envs' widths get the register-prefetching instantiation
    case 10: GAE_LAUNCH(10); break; case 11: GAE_LAUNCH(11); break; case 14: GAE_LAUNCH(14); break;
    case 18: GAE_LAUNCH(18); break; case 29: GAE_LAUNCH8(29); break; case 55: GAE_LAUNCH8(55); break;
    default: GAE_LAUNCH8(0);
    }
#undef GAE_LAUNCH8
#undef GAE_LAUNCH_NW
#undef GAE_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

// Opening launch of process_samples (metrpo_process_begin): the clamped log_std of the policy that generated the batch (agent_infos['log_std'],
// [rllab] GaussianMLPPolicy: max(log_std, log min_std)) and the zero fill of the iteration's float64 accumulators (advantage statistics | AtA | Aty)
// in ONE launch -- as separate operations (a copy of theta, a clamp, a fill) they were three dependent launches between the rollout and k_gae.
__global__ void k_process_begin(const float* __restrict__ theta, int P, int na, float* __restrict__ ls_out, double* __restrict__ acc, int64_t n_acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (acc != nullptr && i < n_acc) acc[i] = 0.0;
    if (ls_out != nullptr && i < na) { const float v = theta[P - na + i]; ls_out[i] = (v < LOG_MIN_STD) ? LOG_MIN_STD : v; }     // NaN stays NaN, as torch.clamp leaves it
}
int launch_process_begin(metrpo_ctx* c, float* ls_out, double* acc, int64_t n_acc, hipStream_t st) {
    const int64_t n = std::max<int64_t>(n_acc, (int64_t)c->pd.na);
    hipLaunchKernelGGL(k_process_begin, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)c->d_theta, c->pd.P, c->pd.na, ls_out, acc, n_acc);
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

// [rllab] LinearFeatureBaseline.fit's solve (linear_feature_baseline.py: lstsq(F^T F + reg I, F^T returns), reg x 10 while the solution has
// a NaN, at most 5 attempts) on the device, so that the coefficients the next process_samples needs never leave it: one workgroup, the
// augmented system [A + reg I | b] in LDS (float64), Gaussian elimination in the natural order (the system is square and, with reg > 0,
// symmetric positive definite: the least-squares solution IS its solution), back substitution by the first wave.  F <= 114 (Humanoid): 0.1 MB of LDS.
__global__ void __launch_bounds__(1024) k_baseline_solve(int F, const double* __restrict__ AtA, const double* __restrict__ Aty, double reg0, double* __restrict__ coeffs) {
    extern __shared__ __attribute__((aligned(16))) double Ms[];      // [F][F + 1] | x [F]
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, W = F + 1;
    double* x = Ms + (size_t)F * W;
    double reg = reg0;
    for (int attempt = 0; attempt < 5; ++attempt) {
        for (int i = tid; i < F * W; i += (int)blockDim.x) { const int r = i / W, c = i % W; Ms[i] = (c < F) ? AtA[r * F + c] + (r == c ? reg : 0.0) : Aty[r]; }
        if (tid == 0) s_bad = 0;
        __syncthreads();
        for (int k = 0; k < F; ++k) {
            // No row exchanges (third pass of round 5; until then: a pivot search by the first wave + the exchange = two more barriers per pivot, 302 us at F = 114):
            // A + reg I is symmetric positive definite, for which elimination in the natural order is as stable as with partial pivoting -- what the one-wave
            // solver of the small envs below has always done; a non-finite result escalates reg like a NaN.
            // A pivot that is not positive is numerical breakdown of a system that is positive definite on paper (columns collinear to within rounding): the attempt
            // counts as failed and reg escalates, like a non-finite solution -- the reference's SVD-based lstsq returns a bounded solution there, large finite
            // coefficients out of a sign-flipped pivot would not be one.
            if (tid == 0 && !(Ms[k * W + k] > 0.0)) s_bad = 1;
            const double inv = 1.0 / Ms[k * W + k];
            // rows below k x columns right of k (rhs included): a thread owns ONE column (W <= 128) and every RS-th row (RS = blockDim / 128 = 8: 16 waves, a
            // pivot's update is ~7 rows per thread) -- no integer division per element (as `i / nc, i % nc` over a flat index the update was 386 us at F = 114, most
            // of it address arithmetic); the same expression per element whatever the block size
            const int RS = (int)blockDim.x >> 7;
            for (int c = k + 1 + (tid & 127); c < W; c += 128) {
                const double mkc = Ms[k * W + c];
                int r = k + 1 + (tid >> 7);
                for (; r + 3 * RS < F; r += 4 * RS) {        // four rows in flight (independent read / multiply / write chains)
                    const double f0 = Ms[r * W + k] * inv, f1 = Ms[(r + RS) * W + k] * inv, f2 = Ms[(r + 2 * RS) * W + k] * inv, f3 = Ms[(r + 3 * RS) * W + k] * inv;
                    const double m0 = Ms[r * W + c], m1 = Ms[(r + RS) * W + c], m2 = Ms[(r + 2 * RS) * W + c], m3 = Ms[(r + 3 * RS) * W + c];
                    Ms[r * W + c] = m0 - f0 * mkc; Ms[(r + RS) * W + c] = m1 - f1 * mkc; Ms[(r + 2 * RS) * W + c] = m2 - f2 * mkc; Ms[(r + 3 * RS) * W + c] = m3 - f3 * mkc;
                }
                for (; r < F; r += RS) Ms[r * W + c] -= (Ms[r * W + k] * inv) * mkc;
            }
            __syncthreads();
        }
        if (tid < 64) {                                      // back substitution: x[k] = (b[k] - sum_{c > k} M[k][c] x[c]) / M[k][k]
            for (int k = F - 1; k >= 0; --k) {
                double a = 0.0;
                for (int c = k + 1 + lane; c < F; c += 64) a += Ms[k * W + c] * x[c];
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                if (lane == 0) { const double v = (Ms[k * W + F] - a) / Ms[k * W + k]; x[k] = v; if (!(fabs(v) <= 1.7e308)) s_bad = 1; }      // NaN or Inf
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (!s_bad) break;
        reg *= 10.0;
        __syncthreads();
    }
    for (int i = tid; i < F; i += (int)blockDim.x) coeffs[i] = x[i];
}

// The same solve for the feature counts of the five small envs (F = 24 ... 62), in ONE wave's registers: lane r holds row r of [A + reg I | b],
// the elimination is unrolled over the pivots, a pivot row is read out of its lane straight into scalar registers (v_readlane with a constant
// lane), no LDS, no barrier: 2 us where the workgroup version takes 37 (its 3 barriers per pivot).  No row exchanges: A + reg I is symmetric
// positive definite, for which elimination in the natural order is as stable as with pivoting; a non-finite result escalates reg like a NaN.
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int FT>
__global__ void __launch_bounds__(64) k_baseline_solve_wave(const double* __restrict__ AtA, const double* __restrict__ Aty, double reg0, double* __restrict__ coeffs) {
    const int lane = threadIdx.x;
    const bool live = lane < FT;
    double reg = reg0;
    double x[FT];
    for (int attempt = 0; attempt < 5; ++attempt) {
        double row[FT + 1];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < FT; ++c) row[c] = live ? AtA[lane * FT + c] + (lane == c ? reg : 0.0) : 0.0;
        row[FT] = live ? Aty[lane] : 0.0;
#pragma unroll
        for (int k = 0; k < FT; ++k) {
            double pv[FT + 1];
#pragma unroll
            for (int c = k; c <= FT; ++c) pv[c] = readlane_f64(row[c], k);
            bad = bad || !(pv[k] > 0.0);                     // (uniform) a pivot that is not positive: numerical breakdown, the attempt fails and reg escalates (k_baseline_solve: why)
            const double f = row[k] * (1.0 / pv[k]);
            if (lane > k) {
#pragma unroll
                for (int c = k + 1; c <= FT; ++c) row[c] = fma(-f, pv[c], row[c]);
            }
        }
#pragma unroll
        for (int k = FT - 1; k >= 0; --k) {
            double a = row[FT];
#pragma unroll
            for (int c = k + 1; c < FT; ++c) a = fma(-row[c], x[c], a);
            x[k] = readlane_f64(a / row[k], k);
            bad = bad || !isfinite(x[k]);
        }
        if (!bad) break;
        reg *= 10.0;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < FT; ++k) coeffs[k] = x[k];
    }
}

int launch_baseline_solve(metrpo_ctx* c, const double* AtA, const double* Aty, double reg, double* coeffs, hipStream_t st) {
    const int F = 2 * c->pd.ns + 4;
#define SOLVE_WAVE(FT_) case FT_: hipLaunchKernelGGL(k_baseline_solve_wave<FT_>, dim3(1), dim3(64), 0, st, AtA, Aty, reg, coeffs); HIP_TRY(c, hipGetLastError()); return METRPO_OK;
    switch (F) { SOLVE_WAVE(24) SOLVE_WAVE(26) SOLVE_WAVE(32) SOLVE_WAVE(40) SOLVE_WAVE(62) default: break; }
#undef SOLVE_WAVE
    const size_t sh = sizeof(double) * ((size_t)F * (F + 1) + F);
    if (sh > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "baseline_solve: feature count too large for one workgroup's LDS");
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_baseline_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_baseline_solve, dim3(1), dim3(1024), sh, st, F, AtA, Aty, reg, coeffs);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_gram(metrpo_ctx* c, const float* obs, const float* ret, const int32_t* tpath, const uint8_t* valid,
                int64_t N, double* AtA, double* Aty, hipStream_t st) {
    const int F = 2 * c->pd.ns + 4;
    const int nfb = (F + 1 + 15) / 16;                        // feature columns incl. the return column, in 16-blocks
    if (nfb <= 2) return launch_gram_mfma<2>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 3) return launch_gram_mfma<3>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 4) return launch_gram_mfma<4>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 5) return launch_gram_mfma_wide<5>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 6) return launch_gram_mfma_wide<6>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 7) return launch_gram_mfma_wide<7>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    if (nfb == 8) return launch_gram_mfma_wide<8>(c, obs, ret, tpath, valid, N, AtA, Aty, st);
    const int bs = 256;
    const size_t sh = sizeof(double) * GRAM_TILE * (F + 1);
    const int grid = (int)std::min<int64_t>((N + GRAM_TILE - 1) / GRAM_TILE, (int64_t)c->n_sm * 4);
    const int nout = F * F + F;
    const size_t need = (size_t)grid * nout;
    if (need > c->gram_cap) {
        ws_retire(c, c->d_gram_part);
        c->d_gram_part = nullptr; c->gram_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_gram_part, sizeof(double) * need));
        c->gram_cap = need;
    }
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_gram, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_gram, dim3(grid), dim3(bs), sh, st, obs, ret, tpath, valid, N, c->pd.ns, c->d_gram_part);
    hipLaunchKernelGGL(k_gram_final, dim3((nout + 15) / 16), dim3(1024), 0, st, c->d_gram_part, grid, 0, F, AtA, Aty);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Loop condition of VectorizedSampler.obtain_samples (samplers/vectorized_sampler.py:60,104) for chunked rollouts:
// per-step completed-path sample counts, then the sequential "first step at which the total reaches batch_size" scan.
// Counts are small integers held in float64 (exact), like the other per-iteration statistics of this file.
__global__ void __launch_bounds__(256) k_path_counts(const uint8_t* __restrict__ done, const int32_t* __restrict__ tpath, int B,
                                                     double* __restrict__ counts, const int32_t* __restrict__ stop) {
    __shared__ double sh[16];
    if (*stop != 0) return;
    const size_t row = (size_t)blockIdx.x * B;
    long long acc = 0;
    for (int b = threadIdx.x; b < B; b += 256)
        if (done[row + b]) acc += (long long)tpath[row + b] + 1;              // path length = 0-based index of its last step + 1
    const double tot = block_sum((double)acc, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

__global__ void k_stop_scan(const double* __restrict__ counts, int T, int t0, double batch_size, double* __restrict__ state,
                            int32_t* __restrict__ stop) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || *stop != 0) return;
    double cum = state[0];
    for (int t = 0; t < T; ++t) {
        cum += counts[t];
        if (cum >= batch_size) { state[0] = cum; state[1] = (double)(t0 + t); *stop = 1; __threadfence(); return; }
    }
    state[0] = cum;
}

int launch_sampler_progress(metrpo_ctx* c, const uint8_t* done, const int32_t* tpath, int T, int B, int t0, long long batch_size,
                            double* counts, double* state, int32_t* stop, hipStream_t st) {
    hipLaunchKernelGGL(k_path_counts, dim3(T), dim3(256), 0, st, done, tpath, B, counts, stop);
    hipLaunchKernelGGL(k_stop_scan, dim3(1), dim3(64), 0, st, counts, T, t0, (double)batch_size, state, stop);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
