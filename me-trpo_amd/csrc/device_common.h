// Device-side helpers shared by the generic (VALU) kernels.  gfx950 only.
#pragma once
#include "metrpo_internal.h"

#define LOG_MIN_STD (-13.815510557964274f)   // log(1e-6): [rllab] GaussianMLPPolicy(min_std=1e-6)
#define KL_EPS 1e-8f                         // [rllab] DiagonalGaussian.kl_sym denominator constant

// tanh(x) = 1 - 2 / (1 + exp(2x)) in 5 VALU ops (v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma): absolute error
// <= ~2e-7 everywhere (exact saturation to +-1); libm's tanhf costs ~40 ops and sat on every kernel's critical path.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);          // exp(2x) = 2^(2x log2 e)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

__device__ __forceinline__ float act_apply(int act, float x) {
    if (act == METRPO_ACT_RELU) return fmaxf(x, 0.0f);
    if (act == METRPO_ACT_TANH) return tanh_fast(x);
    return x;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (production draws; parity tests supply the draws explicitly).
// counter = (env index lo, env index hi, t, purpose<<16 | chunk), key = seed.
// ---------------------------------------------------------------------------------------------
// RNG_STEP, chunk c of (env, t): .x,.y -> Box-Muller pair = policy noise of action dims 2c, 2c+1.  Chunk 0 also carries
//   .z -> step_rand model index of step t (high bits) and cur_model_idx of the reset following step t (low 16 bits),
//   .w -> pool row of that reset.   => the common case (na <= 2) costs ONE Philox block per env-step.
// RNG_SELNOISE, chunk c: 4 normals = model_mean_std noise of state dims 4c..4c+3.  RNG_RESET: initial reset (.x row, .y model).
enum { RNG_STEP = 1, RNG_SELNOISE = 2, RNG_RESET = 3 };

__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

__device__ __forceinline__ uint4 rng_draw(uint64_t seed, uint64_t env, uint32_t t, uint32_t purpose, uint32_t chunk) {
    return philox4x32(make_uint4((uint32_t)env, (uint32_t)(env >> 32), t, (purpose << 16) | chunk),
                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

// 4 standard normals from one Philox block (Box-Muller)
__device__ __forceinline__ void normal4(uint4 r, float out[4]) {
    const float S = 2.3283064365386963e-10f;   // 2^-32
    float u0 = ((float)r.x + 0.5f) * S, u1 = ((float)r.y + 0.5f) * S;
    float u2 = ((float)r.z + 0.5f) * S, u3 = ((float)r.w + 0.5f) * S;
    float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

__device__ __forceinline__ void normal2(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float S = 2.3283064365386963e-10f;
    const float u0 = ((float)a + 0.5f) * S, u1 = ((float)b + 0.5f) * S;
    const float rad = sqrtf(-2.0f * __logf(u0));
    float sn, cs;
    __sincosf(6.283185307179586f * u1, &sn, &cs);
    n0 = rad * cs; n1 = rad * sn;
}

__device__ __forceinline__ int rng_index(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); }
__device__ __forceinline__ int rng_index16(uint32_t r, int n) { return (int)(((r & 0xFFFFu) * (uint32_t)n) >> 16); }   // n <= 65535

// ---------------------------------------------------------------------------------------------
// "Column" layout: one thread = one env/sample; activation u of thread tid lives at buf[u*LD + tid]
// (lanes hit consecutive LDS banks).  Weights are wave-uniform reads of a const __restrict__ kernel
// argument, which hipcc turns into scalar (s_load) traffic: no LDS bandwidth for weights.
// ---------------------------------------------------------------------------------------------
#define DENSE_JB 8
__device__ __forceinline__ void dense_col(const float* __restrict__ W, const float* __restrict__ b, int n_in,
                                          int n_out, int act, const float* in, float* out, int LD, int tid) {
    for (int j0 = 0; j0 < n_out; j0 += DENSE_JB) {
        float acc[DENSE_JB];
        const int nj = min(DENSE_JB, n_out - j0);
        if (nj == DENSE_JB) {
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj) acc[jj] = b[j0 + jj];
            for (int i = 0; i < n_in; ++i) {
                const float a = in[i * LD + tid];
                const float* __restrict__ w = W + (size_t)i * n_out + j0;
#pragma unroll
                for (int jj = 0; jj < DENSE_JB; ++jj) acc[jj] = fmaf(a, w[jj], acc[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj) out[(j0 + jj) * LD + tid] = act_apply(act, acc[jj]);
        } else {
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj) acc[jj] = (jj < nj) ? b[j0 + jj] : 0.0f;
            for (int i = 0; i < n_in; ++i) {
                const float a = in[i * LD + tid];
                const float* __restrict__ w = W + (size_t)i * n_out + j0;
#pragma unroll
                for (int jj = 0; jj < DENSE_JB; ++jj)
                    if (jj < nj) acc[jj] = fmaf(a, w[jj], acc[jj]);
            }
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj)
                if (jj < nj) out[(j0 + jj) * LD + tid] = act_apply(act, acc[jj]);
        }
    }
}

// dense_col restricted to the output columns [j_lo, j_hi) -- lets G thread groups of a block share one layer of one env tile
__device__ __forceinline__ void dense_col_range(const float* __restrict__ W, const float* __restrict__ b, int n_in, int n_out, int j_lo,
                                                int j_hi, int act, const float* in, float* out, int LD, int tid) {
    for (int j0 = j_lo; j0 < j_hi; j0 += DENSE_JB) {
        float acc[DENSE_JB];
        const int nj = min(DENSE_JB, j_hi - j0);
#pragma unroll
        for (int jj = 0; jj < DENSE_JB; ++jj) acc[jj] = (jj < nj) ? b[j0 + jj] : 0.0f;
        for (int i = 0; i < n_in; ++i) {
            const float a = in[i * LD + tid];
            const float* __restrict__ w = W + (size_t)i * n_out + j0;
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj)
                if (jj < nj) acc[jj] = fmaf(a, w[jj], acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < DENSE_JB; ++jj)
            if (jj < nj) out[(j0 + jj) * LD + tid] = act_apply(act, acc[jj]);
    }
}

// mlp_col with the outputs of every layer split over the G thread groups of the block (group g = threadIdx.x / LD); every thread of the
// block must call it (block barriers between layers).  Returns the buffer holding the outputs.
__device__ __forceinline__ float* mlp_col_groups(const NetDesc& net, const float* __restrict__ params, const float* in, float* bufA, float* bufB,
                                                 int LD, int lane, int g, int G) {
    const float* cur = in;
    float* dst = bufA;
    for (int l = 0; l < net.n_layers; ++l) {
        const int n_out = net.dims[l + 1];
        const int per = ((n_out + G * DENSE_JB - 1) / (G * DENSE_JB)) * DENSE_JB;       // columns per group, a multiple of the register block
        const int j_lo = min(n_out, g * per), j_hi = min(n_out, j_lo + per);
        dense_col_range(params + net.w_off[l], params + net.b_off[l], net.dims[l], n_out, j_lo, j_hi, net.act[l], cur, dst, LD, lane);
        __syncthreads();
        cur = dst;
        dst = (dst == bufA) ? bufB : bufA;
    }
    return const_cast<float*>(cur);
}

// Full MLP in column layout.  `in` is read-only; hidden activations ping-pong between bufA/bufB;
// returns the pointer (bufA or bufB) holding the n_out outputs.
__device__ __forceinline__ float* mlp_col(const NetDesc& net, const float* __restrict__ params, const float* in,
                                          float* bufA, float* bufB, int LD, int tid) {
    const float* cur = in;
    float* dst = bufA;
    for (int l = 0; l < net.n_layers; ++l) {
        dense_col(params + net.w_off[l], params + net.b_off[l], net.dims[l], net.dims[l + 1], net.act[l], cur, dst,
                  LD, tid);
        cur = dst;
        dst = (dst == bufA) ? bufB : bufA;
    }
    return const_cast<float*>(cur);
}

// ---------------------------------------------------------------------------------------------
// analytic cost (reward = -cost) and termination; s_next/u accessed through column pointers
//   envs/com_*_env.py cost_np_vec / is_done -- see include/metrpo.h metrpo_env for file:line
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float env_cost(int env, int ns, int na, const float* xn, const float* u, int LD, int tid) {
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) { const float a = u[d * LD + tid]; su2 = fmaf(a, a, su2); }
    switch (env) {
    case METRPO_ENV_SWIMMER: return -(xn[5 * LD + tid] - 1e-2f * (su2 / (float)na));
    case METRPO_ENV_HALF_CHEETAH: return -fminf(fmaxf(xn[9 * LD + tid] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
    case METRPO_ENV_ANT: return -(xn[15 * LD + tid] - 1e-2f * 0.5f * su2 + 0.05f);
    case METRPO_ENV_HUMANOID: { const float h = xn[(ns - 1) * LD + tid] - 1.5f; return h * h + 1e-2f * 1e-3f * su2; }
    case METRPO_ENV_HOPPER: {
        float pen = 0.0f;
        for (int i = 2; i < ns; ++i) pen += fmaxf(fabsf(xn[i * LD + tid]) - 100.0f, 0.0f);
        return -(xn[5 * LD + tid] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0 * LD + tid], 0.0f) -
                 10.0f * fmaxf(fabsf(xn[1 * LD + tid]) - 0.2f, 0.0f) - pen);
    }
    case METRPO_ENV_SNAKE: return -(xn[7 * LD + tid] - 1e-2f * 0.5f * su2);
    }
    return 0.0f;
}

__device__ __forceinline__ bool env_is_done(int env, int ns, const float* xn, int LD, int tid) {
    if (env != METRPO_ENV_ANT) return false;
    bool finite = true;
    for (int i = 0; i < ns; ++i) finite = finite && isfinite(xn[i * LD + tid]);
    const float z = xn[2 * LD + tid];
    return !((z >= 0.2f) && (z <= 1.0f) && finite);
}

// block-wide sum of a double over a 1-D block (<= 1024 threads); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 16 doubles */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sh[i];
    }
    return r;
}
