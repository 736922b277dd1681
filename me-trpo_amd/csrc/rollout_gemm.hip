// Step-wise rollout for LARGE dynamics networks (hidden >= 128: the params-file shapes 2x512 / 2x1024 / 3x1024 of
// BASELINE's C0-params-file, C2, C3, C4).  Same reference path as the fused kernels (samplers/vectorized_sampler.py:45-116,
// env_helpers.py:597-635, training.py:218-269), different mapping: at these widths ONE time step is already tens of
// GFLOP, so the time loop stays on the host (stream-ordered launches, no synchronisation) and every dynamics layer is
// a batched-over-heads GEMM on the f32 matrix core:
//     k_big_pre    thread per env: policy forward (tiny), action, clip, normalise + drop -> X[B][n_in]; writes obs/act/mean
//     k_gemm_bias_act   C[k] = act(A[k] . W[k] + b[k])   128x128(64) tiles, v_mfma_f32_32x32x2_f32, LDS double buffer
//     k_big_post   lane per (env, dim): de-normalise + residual, sam_mode selection over the K heads, reward, done, reset
// The weights are streamed from L2/HBM every step (K x 1-9 MB): the tile shape gives >= 128-fold reuse per fetched
// weight, which keeps the kernel MFMA-bound (AI ~ 60 flop/B at B = 2500).
#include <thread>
#include "gemm_mfma.h"
#include "mfma_common.h"
#include "mlp_streamk.h"
#include "big_prepost.h"
#include "mlp_persist.h"
#include "policy_chain3.h"

// policy.get_actions + clip + normalise/drop for policies without an MFMA pre-kernel (Humanoid's 100-50-25): a block = 64 envs x G thread
// groups; the outputs of every policy layer are split over the groups (activations in LDS columns), group 0 owns the env's bookkeeping
__global__ void k_big_pre(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta, const float* __restrict__ norm,
                          BigState st) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = 64, tid = threadIdx.x & 63, grp = threadIdx.x >> 6, G = blockDim.x >> 6;
    const int b = blockIdx.x * 64 + tid;
    const bool active = b < r.B;
    const int ns = pd.ns, na = pd.na;
    float* Sc = lds; float* A = Sc + ns * LD; float* Bq = A + pd.pol.max_width * LD;
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, b);
    const int tt = t + RK_TOFF(r, b);                            // row of the trajectory tensors / draw counter of this env's step
    if (r.stop != nullptr && *r.stop != 0) return;               // the sampling loop already ended (metrpo_sampler_progress)
    if (grp == 0) {
    if (t == 0 && active && r.init_obs != nullptr) {             // continuation of a chunked rollout
        st.cur_model[b] = r.init_model[b]; st.ts[b] = r.init_ts[b];
        for (int i = 0; i < ns; ++i) st.S[(size_t)b * ns + i] = r.init_obs[(size_t)b * ns + i];
    } else if (t == 0 && active) {                               // vec_env.reset() (env_helpers.py:585-595)
        const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
        st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K);
        st.ts[b] = 0;
        for (int i = 0; i < ns; ++i) st.S[(size_t)b * ns + i] = r.pool[(size_t)row * ns + i];
    }
    for (int i = 0; i < ns; ++i) Sc[i * LD + tid] = active ? st.S[(size_t)b * ns + i] : 0.0f;
    }
    __syncthreads();
    float* m = mlp_col_groups(pd.pol, theta, Sc, A, Bq, LD, tid, grp, G);
    if (!active || grp != 0) return;
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, b);
    const float* __restrict__ log_std = theta + pd.pol.n_params;
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    const uint4 dstep = rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, 0);
    for (int i = 0; i < ns; ++i) {
        const float s = Sc[i * LD + tid];
        r.obs[tb * ns + i] = s;
        if (i >= pd.n_drop) st.X[(size_t)b * st.ldx + i - pd.n_drop] = (s - in_mean[i]) / in_std[i];       // training.py:228,146-151
    }
    for (int j = pd.nin; j < st.ldx; ++j) st.X[(size_t)b * st.ldx + j] = (st.xone && j == pd.nin) ? 1.0f : 0.0f;   // pad columns of the 16-byte aligned rows (layer 0 contracts over ldx)
    for (int d0 = 0; d0 < na; d0 += 2) {
        float z[2] = {0.f, 0.f};
        if (!r.determ && r.eps == nullptr) {
            const uint4 blk = (d0 == 0) ? dstep : rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, d0 >> 1);
            normal2(blk.x, blk.y, z[0], z[1]);
        }
        for (int d = d0; d < min(d0 + 2, na); ++d) {
            const float mu = m[d * LD + tid];
            float a = mu;
            if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * na + d] : z[d - d0], __expf(fmaxf(log_std[d], LOG_MIN_STD)), mu);
            r.act[tb * na + d] = a; r.mean[tb * na + d] = mu;
            const float ac = fminf(fmaxf(a, -1.0f), 1.0f);          // env_helpers.py:599
            st.U[(size_t)b * na + d] = ac;
            st.X[(size_t)b * st.ldx + (ns - pd.n_drop) + d] = (ac - in_mean[ns + d]) / in_std[ns + d];
        }
    }
}

// MFMA variant of k_big_pre for the 2x32 tanh policies (every shipped params file except Humanoid): a wave evaluates the policy of a 16-env
// tile as the transposed MFMA chain of the fused rollout kernels (30 MFMAs) instead of 64 threads walking three dense layers each;
// identical draws (same Philox blocks) and identical outputs layout.  grid = ceil(B/64) blocks of 4 waves.
// POST = true (t >= 1): the same launch first CLOSES step t - 1 for its 16 envs -- k_big_post's work (de-normalise + residual, selection over the
// heads, reward, done, reset; env_helpers.py:597-635) in this kernel's layout: lane (env c, quarter q) owns the state dims 4 q + r and
// 16 + 4 q + r (one 16-byte read per partial and head), the new state goes straight into the wave's LDS state tile, which is what the policy
// chain below reads.  One launch per step instead of two between the ensemble kernels: k_big_post(t - 1) + this kernel's own launch and
// its reload of the state cost ~17 us of a 150 us step at the C3 share.  Same arithmetic in the same order as k_big_post: trajectories are
// bit for bit those of the two-launch sequence (tests/test_gpu_streamk.py).
#ifdef PP_TIMING        // developer instrumentation (SRC=rollout_gemm.hip tools/build_variant.sh pptiming -DPP_TIMING): shader-clock phases of workgroup 0, wave 0 of the merged launch
__device__ unsigned long long g_pp_phase[8];
#define PP_MARK(i) { if (POST && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); g_pp_phase[i] += n_ - pp_t; pp_t = n_; } }
extern "C" int32_t metrpo_debug_pp_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_phase), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1; }
#else
#define PP_MARK(i)
#endif
template <int ENV, bool POST>
__global__ void __launch_bounds__(256) k_big_pre_mfma(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta,
                                                      const float* __restrict__ norm, BigState st) {
    using C = Cfg<ENV, 64, 32>;
    using IM = PreImg<ENV>;
    constexpr int NS = C::NS, IMG = IM::IMG;
    __shared__ __attribute__((aligned(16))) float lds[IMG + 4 * 16 * NS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b0 = (blockIdx.x * 4 + wave) * 16, b = b0 + c;
    const bool active = b < r.B;
    float* ST = lds + IMG + wave * 16 * NS;
    if (r.stop != nullptr && *r.stop != 0) return;               // the sampling loop already ended (metrpo_sampler_progress)
#ifdef PP_TIMING
    unsigned long long pp_t = __builtin_readcyclecounter();
#endif
    // weight image: all gathers of a thread are issued before the first LDS store (one L2 round trip, not one per element) -- and the stores wait until
    // step t - 1 has been closed below: the gathers' round trip passes behind the post part's own loads instead of in front of them
    constexpr int NIT = (IMG + 255) / 256;
    float wv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) wv[it] = IM::entry(theta, tid + 256 * it);
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, b);
    if (t == 0 && active && q == 0 && r.init_obs != nullptr) {   // continuation of a chunked rollout
        st.cur_model[b] = r.init_model[b]; st.ts[b] = r.init_ts[b];
        for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.init_obs[(size_t)b * NS + i];
    } else if (t == 0 && active && q == 0) {                     // vec_env.reset() (env_helpers.py:585-595)
        const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
        st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K);
        st.ts[b] = 0;
        for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.pool[(size_t)row * NS + i];
    }
    if constexpr (!POST) { __builtin_amdgcn_s_waitcnt(0x0F70); wave_lds_sync(); }      // t == 0: the reset rows of this tile were written by lanes of its own wave and are read back below
    PreLane<ENV> pl;
    big_pre_head<ENV, POST>(pd, r, t, theta, norm, st, ST, b0, lane, pl);      // (POST: closes step t - 1 first) state tile, obs[t]
    PP_MARK(0)
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const int i = tid + 256 * it; if (i < IMG) lds[i] = wv[it]; }
    __syncthreads();                                             // image complete
    PP_MARK(1)
    big_pre_tail<ENV, false>(r, t, st, lds, ST, b0, lane, pl);   // policy chain, action, act / mean / U / X
#ifdef PP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PP_MARK(3)
}

// MFMA pre-kernel for tanh policies with THREE hidden layers (Humanoid's 100-50-25, params-humanoid.json): the same transposed chain as
// k_big_pre_mfma, every width padded to whole 16-unit tiles (zero weights, tanh(0) = 0 meets zero rows of the next layer).  One wave per 16-env
// tile: NS_KS C1 + 4 C1 C2 + 4 C2 C3 + 4 C3 CO matrix instructions (258 for Humanoid) instead of the gather -> four small GEMMs -> action
// chain of six launches (41 us per step at 500 rows) or the thread-per-env k_big_pre (302 us).  Same draws, same outputs layout.
template <int NS, int NA, int NDROP, int W1, int W2, int W3, bool POST = false>
__global__ void __launch_bounds__(256) k_big_pre_mfma3(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta,
                                                       const float* __restrict__ norm, BigState st) {
    using PC = P3<NS, NA, W1, W2, W3>;
    constexpr int CO = PC::CO, IMG = PC::IMG, pLS = PC::pLS, NIN = NS - NDROP + NA;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // image, then [4 waves][16 envs][NS] state tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b0 = (blockIdx.x * 4 + wave) * 16, b = b0 + c;
    const bool active = b < r.B;
    float* ST = lds + IMG + wave * 16 * NS;
    if (r.stop != nullptr && *r.stop != 0) return;
    PC::load_image(lds, st.PIMG, tid);                           // built once per launch chain (k_pre_mfma3_image)
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, b);
    const int tt = t + RK_TOFF(r, b);
    if (t == 0 && active && q == 0 && r.init_obs != nullptr) {   // continuation of a chunked rollout / merged rounds
        st.cur_model[b] = r.init_model[b]; st.ts[b] = r.init_ts[b];
        for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.init_obs[(size_t)b * NS + i];
    } else if (t == 0 && active && q == 0) {                     // vec_env.reset() (env_helpers.py:585-595)
        const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
        st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K);
        st.ts[b] = 0;
        for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.pool[(size_t)row * NS + i];
    }
    __syncthreads();                                             // image complete; the reset rows of this tile are written by its own wave
    const int lim = min(16, max(0, r.B - b0)) * NS;
    if constexpr (POST) big_close_step<METRPO_ENV_HUMANOID, NS, NA>(pd, r, t, norm, st, ST, c, q, active, b, genv);      // step t - 1 closed here (k_big_post), as in k_big_pre_mfma<ENV, true>
    else for (int i = lane; i < 16 * NS; i += 64) ST[i] = (i < lim) ? st.S[(size_t)b0 * NS + i] : 0.0f;
    wave_lds_sync();
    for (int i = lane; i < lim; i += 64) {                       // obs[t] (merged rounds: a tile's envs may belong to two rounds)
        const int bi = b0 + i / NS;
        r.obs[((size_t)(t + RK_TOFF(r, bi)) * RK_STRIDE(r) + RK_ENV(r, bi)) * NS + i % NS] = ST[i];
    }
    f32x4 mu[CO];
    PC::forward(lds, ST, lane, c, q, mu);
    if (!active) return;
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, b);
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    const float* __restrict__ log_std = theta + pLS;
    for (int i = q; i < NS; i += 4) if (i >= NDROP) st.X[(size_t)b * st.ldx + i - NDROP] = (ST[c * NS + i] - in_mean[i]) / in_std[i];     // training.py:228,146-151
    if (q == 0) for (int j = NIN; j < st.ldx; ++j) st.X[(size_t)b * st.ldx + j] = (st.xone && j == NIN) ? 1.0f : 0.0f;
    // action dims 16 cb + 4 q .. + 3 of this lane = Philox chunks (dim >> 1) of the step (chunk 0 = the step block), exactly as k_big_pre
#pragma unroll
    for (int cb = 0; cb < CO; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d0 = 16 * cb + 4 * q + 2 * h;
            if (d0 >= NA) continue;
            float z[2] = {0.f, 0.f};
            if (!r.determ && r.eps == nullptr) {
                const uint4 blk = rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, d0 >> 1);
                normal2(blk.x, blk.y, z[0], z[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = d0 + j;
                if (d >= NA) continue;
                const float m = mu[cb][2 * h + j];
                float a = m;
                if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * NA + d] : z[j], __expf(fmaxf(log_std[d], LOG_MIN_STD)), m);
                r.act[tb * NA + d] = a; r.mean[tb * NA + d] = m;
                const float ac = fminf(fmaxf(a, -1.0f), 1.0f);          // env_helpers.py:599
                st.U[(size_t)b * NA + d] = ac;
                st.X[(size_t)b * st.ldx + (NS - NDROP) + d] = (ac - in_mean[NS + d]) / in_std[NS + d];
            }
        }
}
// The same pre-step for SMALL batches (the params-file shape: 500 rows = 32 tiles on 256 CUs), where k_big_pre_mfma3's one wave per tile is a chain of 258
// dependent-ish matrix instructions behind a 67 KB image copy into every workgroup's LDS (23 us per step, a fifth of params-humanoid's rollout).  Here a
// tile is a whole workgroup: wave w owns column blocks w, w + 4 (layer 0: 2 w, 2 w + 1) of every layer -- 28 + 28 + 16 + 8 matrix instructions on its path
// instead of 258 --, loads ITS fragments of the image straight into registers (one round trip, no LDS image), and the layers' D fragments change hands
// through 13 KB of LDS.  Same image, same k order per output unit, same draws: bit for bit k_big_pre_mfma3<.., false>.
template <int NS, int NA, int NDROP, int W1, int W2, int W3>
__global__ void __launch_bounds__(256) k_big_pre_mfma3_split(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta,
                                                             const float* __restrict__ norm, BigState st) {
    using PC = P3<NS, NA, W1, W2, W3>;
    constexpr int NS_KS = PC::NS_KS, C1 = PC::C1, C2 = PC::C2, C3 = PC::C3, CO = PC::CO, pLS = PC::pLS, NIN = NS - NDROP + NA;
    constexpr int NB0 = cdiv(C1, 4), NB1 = cdiv(C2, 4), NB2 = cdiv(C3, 4), NB3 = cdiv(CO, 4);
    static_assert(NB3 == 1, "one output block per wave (na <= 64)");
    __shared__ __attribute__((aligned(16))) float ST[16 * NS];
    __shared__ f32x4 H1[C1 * 64], H2[C2 * 64], H3[C3 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * 16, b = b0 + c;
    const bool active = b < r.B;
    if (r.stop != nullptr && *r.stop != 0) return;
    const float* __restrict__ img = st.PIMG;
    // this wave's fragments and biases: issued first, they fly under the reset / state-tile work
    float f0[NS_KS][NB0], f1[4 * C1][NB1], f2[4 * C2][NB2], f3[4 * C3];
    f32x4 bi0[NB0], bi1[NB1], bi2[NB2], bi3;
#pragma unroll
    for (int j = 0; j < NB0; ++j) {
        const int cb = wave * NB0 + j, cbc = (cb < C1) ? cb : C1 - 1;
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) f0[s_][j] = img[PC::O_F0 + (s_ * C1 + cbc) * 64 + lane];
        bi0[j] = *(const f32x4*)&img[PC::O_B0 + 16 * cbc + 4 * q];
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int cb = wave * NB1 + j, cbc = (cb < C2) ? cb : C2 - 1;
#pragma unroll
        for (int kk = 0; kk < 4 * C1; ++kk) f1[kk][j] = img[PC::O_F1 + (kk * C2 + cbc) * 64 + lane];
        bi1[j] = *(const f32x4*)&img[PC::O_B1 + 16 * cbc + 4 * q];
    }
#pragma unroll
    for (int j = 0; j < NB2; ++j) {
        const int cb = wave * NB2 + j, cbc = (cb < C3) ? cb : C3 - 1;
#pragma unroll
        for (int kk = 0; kk < 4 * C2; ++kk) f2[kk][j] = img[PC::O_F2 + (kk * C3 + cbc) * 64 + lane];
        bi2[j] = *(const f32x4*)&img[PC::O_B2 + 16 * cbc + 4 * q];
    }
    {
        const int cbc = (wave < CO) ? wave : CO - 1;
#pragma unroll
        for (int kk = 0; kk < 4 * C3; ++kk) f3[kk] = img[PC::O_F3 + (kk * CO + cbc) * 64 + lane];
        bi3 = *(const f32x4*)&img[PC::O_B3 + 16 * cbc + 4 * q];
    }
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, b);
    const int tt = t + RK_TOFF(r, b);
    const int lim = min(16, max(0, r.B - b0)) * NS;
    if (t == 0) {                                                // wave 0 resets its tile's envs and reads the rows back itself (as k_big_pre_mfma3's wave does)
        if (wave == 0) {
            if (active && q == 0 && r.init_obs != nullptr) {     // continuation of a chunked rollout / merged rounds
                st.cur_model[b] = r.init_model[b]; st.ts[b] = r.init_ts[b];
                for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.init_obs[(size_t)b * NS + i];
            } else if (active && q == 0) {                       // vec_env.reset() (env_helpers.py:585-595)
                const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
                const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
                st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K);
                st.ts[b] = 0;
                for (int i = 0; i < NS; ++i) st.S[(size_t)b * NS + i] = r.pool[(size_t)row * NS + i];
            }
            wave_lds_sync();
            for (int i = lane; i < 16 * NS; i += 64) ST[i] = (i < lim) ? st.S[(size_t)b0 * NS + i] : 0.0f;
        }
    } else {
        for (int i = tid; i < 16 * NS; i += 256) ST[i] = (i < lim) ? st.S[(size_t)b0 * NS + i] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < lim; i += 256) {                       // obs[t] (merged rounds: a tile's envs may belong to two rounds)
        const int bi = b0 + i / NS;
        r.obs[((size_t)(t + RK_TOFF(r, bi)) * RK_STRIDE(r) + RK_ENV(r, bi)) * NS + i % NS] = ST[i];
    }
    // ---- layer 0
    {
        f32x4 p0[NB0];
#pragma unroll
        for (int j = 0; j < NB0; ++j) p0[j] = bi0[j];
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) {
            const int f = 4 * s_ + q;
            const float x = (f < NS) ? ST[c * NS + f] : 0.0f;
#pragma unroll
            for (int j = 0; j < NB0; ++j) p0[j] = MFMA16(f0[s_][j], x, p0[j]);
        }
#pragma unroll
        for (int j = 0; j < NB0; ++j) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p0[j][rr] = tanh_fast(p0[j][rr]);
            if (wave * NB0 + j < C1) H1[(wave * NB0 + j) * 64 + lane] = p0[j];
        }
    }
    __syncthreads();
    // ---- layer 1
    {
        f32x4 h1[C1], p1[NB1];
#pragma unroll
        for (int cb = 0; cb < C1; ++cb) h1[cb] = H1[cb * 64 + lane];
#pragma unroll
        for (int j = 0; j < NB1; ++j) p1[j] = bi1[j];
#pragma unroll
        for (int kk = 0; kk < 4 * C1; ++kk)
#pragma unroll
            for (int j = 0; j < NB1; ++j) p1[j] = MFMA16(f1[kk][j], h1[kk >> 2][kk & 3], p1[j]);
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p1[j][rr] = tanh_fast(p1[j][rr]);
            if (wave * NB1 + j < C2) H2[(wave * NB1 + j) * 64 + lane] = p1[j];
        }
    }
    __syncthreads();
    // ---- layer 2
    {
        f32x4 h2[C2], p2[NB2];
#pragma unroll
        for (int cb = 0; cb < C2; ++cb) h2[cb] = H2[cb * 64 + lane];
#pragma unroll
        for (int j = 0; j < NB2; ++j) p2[j] = bi2[j];
#pragma unroll
        for (int kk = 0; kk < 4 * C2; ++kk)
#pragma unroll
            for (int j = 0; j < NB2; ++j) p2[j] = MFMA16(f2[kk][j], h2[kk >> 2][kk & 3], p2[j]);
#pragma unroll
        for (int j = 0; j < NB2; ++j) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p2[j][rr] = tanh_fast(p2[j][rr]);
            if (wave * NB2 + j < C3) H3[(wave * NB2 + j) * 64 + lane] = p2[j];
        }
    }
    __syncthreads();
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    if (wave == 3 && active) {                                   // state columns of the dynamics input rows (the last wave has no output block)
        for (int i = q; i < NS; i += 4) if (i >= NDROP) st.X[(size_t)b * st.ldx + i - NDROP] = (ST[c * NS + i] - in_mean[i]) / in_std[i];     // training.py:228,146-151
        if (q == 0) for (int j = NIN; j < st.ldx; ++j) st.X[(size_t)b * st.ldx + j] = (st.xone && j == NIN) ? 1.0f : 0.0f;
    }
    if (wave >= CO) return;
    // ---- output layer: wave w holds action dims 16 w + 4 q .. + 3 of env c
    f32x4 mu = bi3;
    {
        f32x4 h3[C3];
#pragma unroll
        for (int cb = 0; cb < C3; ++cb) h3[cb] = H3[cb * 64 + lane];
#pragma unroll
        for (int kk = 0; kk < 4 * C3; ++kk) mu = MFMA16(f3[kk], h3[kk >> 2][kk & 3], mu);
    }
    if (!active) return;
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, b);
    const float* __restrict__ log_std = theta + pLS;
    const int cb = wave;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                // Philox chunks (dim >> 1) of the step (chunk 0 = the step block), exactly as k_big_pre
        const int d0 = 16 * cb + 4 * q + 2 * h;
        if (d0 >= NA) continue;
        float z[2] = {0.f, 0.f};
        if (!r.determ && r.eps == nullptr) {
            const uint4 blk = rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, d0 >> 1);
            normal2(blk.x, blk.y, z[0], z[1]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = d0 + j;
            if (d >= NA) continue;
            const float m = mu[2 * h + j];
            float a = m;
            if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * NA + d] : z[j], __expf(fmaxf(log_std[d], LOG_MIN_STD)), m);
            r.act[tb * NA + d] = a; r.mean[tb * NA + d] = m;
            const float ac = fminf(fmaxf(a, -1.0f), 1.0f);          // env_helpers.py:599
            st.U[(size_t)b * NA + d] = ac;
            st.X[(size_t)b * st.ldx + (NS - NDROP) + d] = (ac - in_mean[NS + d]) / in_std[NS + d];
        }
    }
}
template <int NS, int NA, int NDROP, int W1, int W2, int W3> constexpr size_t big_pre_mfma3_lds() {
    return sizeof(float) * (size_t)(P3<NS, NA, W1, W2, W3>::IMG + 4 * 16 * NS);
}

// ---- pre-step for policies without an MFMA pre-kernel at LARGE batch (Humanoid's 100-50-25 at B = 6250): the policy layers run as GEMMs over
// the batch (k_gemm_mfma, tanh epilogue) between a gather kernel and an action kernel.  k_big_pre walks the three dense layers on 64-env
// blocks with weights through scalar loads: 0.31 ms per step at C4 for 140 MFLOP.
// gather: vec_env.reset() at t = 0, obs[t] = S, X state columns; one lane per (env, state dim)
__global__ void __launch_bounds__(256) k_big_pre_gather(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ norm, BigState st) {
    const int ns = pd.ns, na = pd.na;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)r.B * ns || (r.stop != nullptr && *r.stop != 0)) return;
    const int b = (int)(idx / ns), i = (int)(idx % ns);
    float s;
    if (t == 0) {
        if (r.init_obs != nullptr) {                              // continuation of a chunked rollout
            s = r.init_obs[(size_t)b * ns + i];
            if (i == 0) { st.cur_model[b] = r.init_model[b]; st.ts[b] = r.init_ts[b]; }
        } else {                                                  // env_helpers.py:585-595
            const uint4 d0 = rng_draw(r.seed, r.stream_offset + (uint64_t)RK_ENV(r, b), 0, RNG_RESET, 0);
            const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
            s = r.pool[(size_t)row * ns + i];
            if (i == 0) { st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K); st.ts[b] = 0; }
        }
        st.S[(size_t)b * ns + i] = s;
    } else s = st.S[(size_t)b * ns + i];
    r.obs[((size_t)(t + RK_TOFF(r, b)) * RK_STRIDE(r) + RK_ENV(r, b)) * ns + i] = s;
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    if (i >= pd.n_drop) st.X[(size_t)b * st.ldx + i - pd.n_drop] = (s - in_mean[i]) / in_std[i];       // training.py:228,146-151
    if (i == 0) for (int j = pd.nin; j < st.ldx; ++j) st.X[(size_t)b * st.ldx + j] = (st.xone && j == pd.nin) ? 1.0f : 0.0f;   // pad columns of the 16-byte aligned rows
}
// action: a = mu + sigma z (policy noise from the step's Philox blocks), clip, act / mean / U / X action columns; one lane per (env, dim pair)
__global__ void __launch_bounds__(256) k_big_pre_action(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta, const float* __restrict__ norm,
                                                        const float* __restrict__ MU, BigState st) {
    const int ns = pd.ns, na = pd.na, npair = (na + 1) / 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)r.B * npair || (r.stop != nullptr && *r.stop != 0)) return;
    const int b = (int)(idx / npair), d0 = 2 * (int)(idx % npair);
    const int tt = t + RK_TOFF(r, b);
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, b);
    const float* __restrict__ log_std = theta + pd.pol.n_params;
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    float z[2] = {0.f, 0.f};
    if (!r.determ && r.eps == nullptr) {
        const uint4 blk = rng_draw(r.seed, r.stream_offset + (uint64_t)RK_ENV(r, b), r.t0 + tt, RNG_STEP, d0 >> 1);
        normal2(blk.x, blk.y, z[0], z[1]);
    }
    for (int d = d0; d < min(d0 + 2, na); ++d) {
        const float mu = MU[(size_t)b * na + d];
        float a = mu;
        if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * na + d] : z[d - d0], __expf(fmaxf(log_std[d], LOG_MIN_STD)), mu);
        r.act[tb * na + d] = a; r.mean[tb * na + d] = mu;
        const float ac = fminf(fmaxf(a, -1.0f), 1.0f);              // env_helpers.py:599
        st.U[(size_t)b * na + d] = ac;
        st.X[(size_t)b * st.ldx + (ns - pd.n_drop) + d] = (ac - in_mean[ns + d]) / in_std[ns + d];
    }
}

typedef void (*big_pre_mfma_t)(ProblemDesc, RolloutK, int, const float*, const float*, BigState);
static big_pre_mfma_t big_pre_mfma_select(const metrpo_ctx* c, size_t* dyn_lds, bool post = false) {
    const ProblemDesc& pd = c->pd;
    *dyn_lds = 0;
    if (pd.env == METRPO_ENV_HUMANOID && pd.ns == 55 && pd.na == 21 && pd.n_drop == 0 && pd.pol.n_layers == 4 && pd.pol.dims[1] == 100 && pd.pol.dims[2] == 50 &&
        pd.pol.dims[3] == 25 && pd.pol.act[0] == METRPO_ACT_TANH && pd.pol.act[1] == METRPO_ACT_TANH && pd.pol.act[2] == METRPO_ACT_TANH) {
        *dyn_lds = big_pre_mfma3_lds<55, 21, 0, 100, 50, 25>();
        return post ? k_big_pre_mfma3<55, 21, 0, 100, 50, 25, true> : k_big_pre_mfma3<55, 21, 0, 100, 50, 25, false>;
    }
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH) return nullptr;
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: return (pd.ns == 10 && pd.na == 2 && pd.n_drop == 2) ? (post ? k_big_pre_mfma<METRPO_ENV_SWIMMER, true> : k_big_pre_mfma<METRPO_ENV_SWIMMER, false>) : nullptr;
    case METRPO_ENV_HALF_CHEETAH: return (pd.ns == 18 && pd.na == 6 && pd.n_drop == 1) ? (post ? k_big_pre_mfma<METRPO_ENV_HALF_CHEETAH, true> : k_big_pre_mfma<METRPO_ENV_HALF_CHEETAH, false>) : nullptr;
    case METRPO_ENV_ANT: return (pd.ns == 29 && pd.na == 8 && pd.n_drop == 2) ? (post ? k_big_pre_mfma<METRPO_ENV_ANT, true> : k_big_pre_mfma<METRPO_ENV_ANT, false>) : nullptr;
    case METRPO_ENV_HOPPER: return (pd.ns == 11 && pd.na == 3 && pd.n_drop == 0) ? (post ? k_big_pre_mfma<METRPO_ENV_HOPPER, true> : k_big_pre_mfma<METRPO_ENV_HOPPER, false>) : nullptr;
    case METRPO_ENV_SNAKE: return (pd.ns == 14 && pd.na == 4 && pd.n_drop == 2) ? (post ? k_big_pre_mfma<METRPO_ENV_SNAKE, true> : k_big_pre_mfma<METRPO_ENV_SNAKE, false>) : nullptr;
    }
    return nullptr;
}

// de-normalise + residual (training.py:257), selection (env_helpers.py:617-634), reward (:601), done (:603-604), reset (:585-595).
// One LANE per (env, state dim): a group of DL = 32 or 64 consecutive lanes owns an env (ns <= DL), so the K-head reads of a dim
// are one coalesced row segment per head and the selection runs ns-fold parallel; the few per-env quantities (reward terms,
// finiteness, the Hopper penalty sum in the reference's dim order) are gathered inside the group with lane shuffles.  The
// thread-per-env version walked the ns dims serially on B threads: 21 us per step at B = 2500, the third largest kernel of C3.
template <int DL>
__global__ void __launch_bounds__(256) k_big_post(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ norm, BigState st) {
    const int i = threadIdx.x % DL, b = blockIdx.x * (256 / DL) + threadIdx.x / DL;
    if (r.stop != nullptr && *r.stop != 0) return;
    const int ns = pd.ns, na = pd.na, K = pd.K;
    const bool env_ok = b < r.B, on = env_ok && i < ns;
    const int bc = env_ok ? b : 0, ic = (i < ns) ? i : 0;
    const int tt = t + RK_TOFF(r, bc);                           // row of the trajectory tensors / draw counter of this env's step
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, bc);
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, bc);
    const float* diff_mean = norm + 2 * (ns + na); const float* diff_std = diff_mean + ns;
    const uint4 dstep = rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, 0);
    int sel = st.cur_model[bc];
    if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (r.model_idx != nullptr) ? r.model_idx[tb] : rng_index(dstep.z, K);
    if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
    const bool simple = (r.sam_mode == METRPO_SAM_STEP_RAND || r.sam_mode == METRPO_SAM_EPS_RAND || r.sam_mode == METRPO_SAM_ONE_MODEL);
    auto grp = [&](float x, int src) { return __shfl(x, src, DL); };           // value of lane `src` of this env's group
    // sum of squared clipped actions in action order (identical association to the per-env loop it replaces)
    const float ua = (env_ok && i < na) ? st.U[(size_t)bc * na + i] : 0.0f;
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) { const float a = grp(ua, d); su2 = fmaf(a, a, su2); }
    float* S = st.S + (size_t)bc * ns;
    const float s_old = S[ic], dm = diff_mean[ic], ds = diff_std[ic];
    auto outv = [&](int k) {                                     // output layer of head k, dim i (partials in split order, like k_splitk_bias_reduce)
        if (st.out_splits == 0) return st.OUT[((size_t)k * r.B + bc) * ns + ic];
        float o = st.out_bias[(size_t)k * st.out_bias_stride + ic];
        for (int sp = 0; sp < st.out_splits; ++sp) o += st.PART[((size_t)sp * K + k) * st.out_stride + (size_t)bc * st.out_ld + ic];
        return o;
    };
    auto head = [&](int k) { return fmaf(ds, outv(k), dm) + s_old; };
    float v;
    if (simple) v = head(sel);
    else {
        float m = 0.0f;
        for (int k = 0; k < K; ++k) m += head(k);
        m /= (float)K;
        v = m;
        if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
            float var = 0.0f;
            for (int k = 0; k < K; ++k) { const float d = head(k) - m; var = fmaf(d, d, var); }
            float z4[4];
            float nz;
            if (r.sel_noise != nullptr) nz = r.sel_noise[tb * ns + ic];
            else { normal4(rng_draw(r.seed, genv, r.t0 + tt, RNG_SELNOISE, ic >> 2), z4); nz = z4[ic & 3]; }
            v = fmaf(nz, sqrtf(var / (float)K), m);
        } else if (r.sam_mode == METRPO_SAM_MODEL_MED) {
            const int r_lo = (K - 1) / 2, r_hi = K / 2;
            float lo = 0.0f, hi = 0.0f;
            for (int k = 0; k < K; ++k) {
                const float xk = head(k);
                int rank = 0;
                for (int j = 0; j < K; ++j) { const float xj = head(j); rank += (xj < xk) || (xj == xk && j < k); }
                if (rank == r_lo) lo = xk;
                if (rank == r_hi) hi = xk;
            }
            v = 0.5f * (lo + hi);
        }
    }
    if (i >= ns) v = 0.0f;
    // per-env quantities
    const int ki = (pd.env == METRPO_ENV_SWIMMER || pd.env == METRPO_ENV_HOPPER) ? 5 : (pd.env == METRPO_ENV_HALF_CHEETAH) ? 9
                   : (pd.env == METRPO_ENV_ANT) ? 15 : (pd.env == METRPO_ENV_SNAKE) ? 7 : 0;
    const float key = grp(v, ki), h0v = grp(v, 0), h1v = grp(v, 1), zc = grp(v, 2), last = grp(v, ns - 1);
    float pen = 0.0f;
    if (pd.env == METRPO_ENV_HOPPER) for (int j = 2; j < ns; ++j) pen += fmaxf(fabsf(grp(v, j)) - 100.0f, 0.0f);
    int fin = isfinite(v) ? 1 : 0;                               // all-finite over the env's dims (lanes beyond ns hold 0)
#pragma unroll
    for (int o = 1; o < DL; o <<= 1) fin &= __shfl_xor(fin, o, DL);
    float cost = 0.0f;
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: cost = -(key - 1e-2f * (su2 / (float)na)); break;
    case METRPO_ENV_HALF_CHEETAH: cost = -fminf(fmaxf(key - 1e-1f * 0.5f * su2, -10.0f), 10.0f); break;
    case METRPO_ENV_ANT: cost = -(key - 1e-2f * 0.5f * su2 + 0.05f); break;
    case METRPO_ENV_HUMANOID: cost = (last - 1.5f) * (last - 1.5f) + 1e-2f * 1e-3f * su2; break;
    case METRPO_ENV_HOPPER: cost = -(key - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - h0v, 0.0f) - 10.0f * fmaxf(fabsf(h1v) - 0.2f, 0.0f) - pen); break;
    case METRPO_ENV_SNAKE: cost = -(key - 1e-2f * 0.5f * su2); break;
    }
    int ts = st.ts[bc] + 1;
    bool dn = (pd.env == METRPO_ENV_ANT) ? !((zc >= 0.2f) && (zc <= 1.0f) && (fin != 0)) : false;
    dn = dn || (ts >= r.H);
    int cur = st.cur_model[bc];
    if (env_ok && i == 0) { r.rew[tb] = -cost; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts - 1; }
    float s_new = v;
    if (dn) {                                                    // uniform over the env's group
        const size_t rb = (size_t)(t + 1) * r.B + bc;
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
        cur = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
        s_new = r.pool[(size_t)row * ns + ic];
        ts = 0;
    }
    if (on) S[i] = s_new;
    if (env_ok && i == 0) { st.ts[b] = ts; if (dn) st.cur_model[b] = cur; }
    if (t == r.T - 1 && RK_LAST_ROUND(r, bc)) {
        const int be = RK_ENV(r, bc);
        if (on && r.last_obs != nullptr) r.last_obs[(size_t)be * ns + i] = s_new;
        if (env_ok && i == 0) { if (r.last_ts != nullptr) r.last_ts[be] = ts; if (r.last_model != nullptr) r.last_model[be] = cur; }
    }
}

// ------------------------------------------------------------------------------------------------
static void gemm_launch(int act, const float* A, long long sA, int lda, const float* W, long long sW, int ldw, const float* bias,
                        long long sB, float* C, long long sC, int ldc, int M, int N, int Kd, int heads, hipStream_t st) {
    GemmEpi ep = {};
    ep.bias = bias; ep.strideBias = sB;
    if (act == METRPO_ACT_RELU) gemm_auto<EPI_BIAS_RELU, false, false>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
    else if (act == METRPO_ACT_TANH) gemm_auto<EPI_BIAS_TANH, false, false>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
    else gemm_auto<EPI_BIAS_ID, false, false>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
}

// ---- stream-K path (mlp_streamk.h) for the shapes with enough tiles to give every CU an equal share: BASELINE's C2 / C3 / C4 ----
// two hidden layers: ONE launch per step (layer 0 as the main layer's operand producer, output layer in its epilogue);
// three hidden layers (Humanoid): layer 0 on the tile GEMM, then two launches (layer 1 stored, layer 2 + output layer).
struct SkVt {
    SkPlan (*plan)(SkArgs&, int);
    hipError_t (*sched)(SkArgs&, const SkPlan&, void*, hipStream_t);
    hipError_t (*launch)(const SkArgs&, const SkPlan&, hipStream_t);
};
template <int AMODE, int EPI, int S0, int OT> static SkVt sk_vt() {
    SkVt v;
    v.plan = [](SkArgs& a, int n_sm) { return sk_plan<AMODE, EPI, S0, OT>(a, n_sm); };
    v.sched = [](SkArgs& a, const SkPlan& p, void* mem, hipStream_t st) { return sk_build_sched<AMODE, EPI, S0, OT>(a, p, mem, st); };
    v.launch = [](const SkArgs& a, const SkPlan& p, hipStream_t st) { return sk_launch<AMODE, EPI, S0, OT>(a, p, st); };
    return v;
}
static bool sk_fused_vt(int S0, int OT, SkVt* v) {          // the (input steps, output tiles) pairs of the six envs: swimmer, hopper, snake, half-cheetah, ant
    if (S0 == 3 && OT == 1) { *v = sk_vt<SK_A_PRODUCER, SK_EPI_OUT, 3, 1>(); return true; }
    if (S0 == 4 && OT == 1) { *v = sk_vt<SK_A_PRODUCER, SK_EPI_OUT, 4, 1>(); return true; }
    if (S0 == 5 && OT == 1) { *v = sk_vt<SK_A_PRODUCER, SK_EPI_OUT, 5, 1>(); return true; }
    if (S0 == 6 && OT == 2) { *v = sk_vt<SK_A_PRODUCER, SK_EPI_OUT, 6, 2>(); return true; }
    if (S0 == 9 && OT == 2) { *v = sk_vt<SK_A_PRODUCER, SK_EPI_OUT, 9, 2>(); return true; }
    return false;
}
static bool sk_out_vt(int OT, SkVt* v) {
    if (OT == 1) { *v = sk_vt<SK_A_GLOBAL, SK_EPI_OUT, 1, 1>(); return true; }
    if (OT == 2) { *v = sk_vt<SK_A_GLOBAL, SK_EPI_OUT, 1, 2>(); return true; }
    if (OT == 4) { *v = sk_vt<SK_A_GLOBAL, SK_EPI_OUT, 1, 4>(); return true; }
    return false;
}
// layer 0 of the stored-layer-0 forms (modes 2, 3) on k_l0_rows: the input widths of the six envs
static bool l0_rows(int S0, const L0Args& a, int n_cu, hipStream_t st, hipError_t* e) {
    switch (S0) {
    case 3: *e = l0_rows_launch<3>(a, n_cu, st); return true;
    case 4: *e = l0_rows_launch<4>(a, n_cu, st); return true;
    case 5: *e = l0_rows_launch<5>(a, n_cu, st); return true;
    case 6: *e = l0_rows_launch<6>(a, n_cu, st); return true;
    case 10: *e = l0_rows_launch<10>(a, n_cu, st); return true;
    case 20: *e = l0_rows_launch<20>(a, n_cu, st); return true;
    default: return false;
    }
}
static bool l0_rows_ok(int S0) { return S0 == 3 || S0 == 4 || S0 == 5 || S0 == 6 || S0 == 10 || S0 == 20; }
static void sk_epi_image(int OT, const float* b1, long long sB1, const float* W2, long long sW2, int no, int N, int heads, float* img, hipStream_t st) {
    const long long per = (OT == 1) ? SkEpi<1>::FLOATS * SkEpi<1>::E : (OT == 2) ? SkEpi<2>::FLOATS * SkEpi<2>::E : SkEpi<4>::FLOATS * SkEpi<4>::E;
    const long long tot = (long long)heads * (N / 256) * per;
    const dim3 grid((unsigned)((tot + 255) / 256));
    if (OT == 1) hipLaunchKernelGGL(k_sk_epi_image<1>, grid, dim3(256), 0, st, b1, sB1, W2, sW2, no, N, heads, img);
    else if (OT == 2) hipLaunchKernelGGL(k_sk_epi_image<2>, grid, dim3(256), 0, st, b1, sB1, W2, sW2, no, N, heads, img);
    else hipLaunchKernelGGL(k_sk_epi_image<4>, grid, dim3(256), 0, st, b1, sB1, W2, sW2, no, N, heads, img);
}
static size_t sk_epi_floats(int OT, int N, int heads) {
    const size_t per = (OT == 1) ? SkEpi<1>::FLOATS * SkEpi<1>::E : (OT == 2) ? SkEpi<2>::FLOATS * SkEpi<2>::E : SkEpi<4>::FLOATS * SkEpi<4>::E;
    return (size_t)heads * (N / 256) * per;
}
struct SkPath {            // mode 0: off; 1: x -> [layer 0 | layer 1 | output layer] in one launch; 2: layer L-3 stored, [layer L-2 | output layer];
                           // 3: two hidden layers with a wide input: layer 0 stored by the tile GEMM, [layer 1 | output layer] in one launch
    int mode, S0, OT;
    int forced;                                                             // option STREAMK=1: the caller wants this family whatever the tile count (parity tests at oracle sizes)
    SkVt v1, v2; SkArgs a1, a2; SkPlan p1, p2;
};
// which path (and its plans) for this context and batch; a1 / a2 carry shapes only (pointers are filled in by rollout_gemm_chunk)
static SkPath sk_select(const metrpo_ctx* c, int B) {
    SkPath sp = {};
    const ProblemDesc& pd = c->pd;
    const int L = pd.dyn.n_layers, K = pd.K;
    const char* fe = ctx_opt(c, OPT_STREAMK);                               // "1": also below one tile per CU (tests at oracle-sized batches)
    const bool force = fe != nullptr && fe[0] == '1';
    sp.forced = force ? 1 : 0;
    if (ctx_opt(c, OPT_NO_STREAMK) != nullptr || L < 3 || L > 4 || pd.ns > 64) return sp;
    for (int l = 0; l < L - 1; ++l) if (pd.dyn.act[l] != METRPO_ACT_RELU) return sp;
    if (pd.dyn.act[L - 1] != METRPO_ACT_IDENTITY) return sp;
    const int K1 = pd.dyn.dims[L - 2], N = pd.dyn.dims[L - 1];            // the layer in front of the output layer: [K1 x N]
    if (K1 % 32 != 0 || N % 256 != 0) return sp;
    sp.OT = (pd.ns <= 16) ? 1 : (pd.ns <= 32) ? 2 : 4;
    // Fewer tiles than CUs: the pieces of a tile side by side, added at their ends (mlp_streamk.h: SkArgs::late).  Built and measured at the Humanoid
    // params file's shape (K = 5, M = 500, 2 x 1024: 80 tiles, 10.6 chunks per workgroup): 71.7 us against the tile GEMM's 69.9 us while the hand-overs at the END
    // of a tile's 3-4 pieces formed a chain of 2-3 links (store + drain + flag + four load passes, ~9 us each); 66.0 us since every piece but the last exports
    // its own sums and the last adds them all (one link).  The launch is then faster than the tile GEMM (69.3), the STEP is not: this path's layer-0 rows
    // (k_l0_rows 21 us) and unsplit pre-step (29 us) cost 20 us more than the GEMM path's (11 + 9.6 + 8.9): 116.6 vs 98.8 us per step.  So it was not selected by itself (METRPO_STREAMK_LATE=1 selected it from 8 chunks per workgroup up); forced launches
    // (METRPO_STREAMK=1: the parity tests) use it from 2 up, METRPO_STREAMK_LATE=0 never (forced launches then run one unsplit tile per workgroup).
    // Third pass of round 5: with ONE hand-over per tile the launch beats the 64 x 64-tile GEMM at that shape (65.3 vs 69.3 us), and together with the tile GEMM
    // for layer 0 and the split pre-step + separate post launch of small batches (rollout_gemm_chunk) the step is 93 us against 98.8: selected by itself for the
    // wide-input two-hidden-layer form (mode 3: Humanoid's 76 inputs) -- and for the narrow-input forms (mode 1: one launch per
    // step), which below one tile per CU are the resident kernel's shapes and get here when that kernel is off (a shared GPU: NO_RESIDENT / not exclusive):
    // params-half-cheetah 9.17 -> 7.56 ms per rollout against the tile GEMMs (resident: 7.29).  STREAMK_LATE=0 keeps the tile GEMMs.
    const char* le = ctx_opt(c, OPT_STREAMK_LATE);
    const bool late_ok = (le != nullptr) ? le[0] == '1' : true;
    auto late_for = [&](int K1_, int N_, int epi_units) -> int {
        const long long tiles = (long long)K * ((B + 127) / 128) * (N_ / 256), units = tiles * (K1_ / 32 + epi_units);
        // from 5 units per workgroup up (tools/late_sweep.py, profiles/r05_e_late_sweep.txt: -3 ... -22 % per rollout against the tile GEMMs at 5.2 - 10.6 units; at 2.6 - 3.3 units
        // it loses -- +14 ... +83 % -- except for the narrow-input 2 x 1024 form)
        return (late_ok && tiles < c->n_sm && units / c->n_sm >= (force ? 2 : 5)) ? 1 : 0;
    };
    const int epi_units = (sp.OT == 4) ? 2 : 1;
    // test hook STREAMK_PLACE: "flat" = consecutive ranges over the workgroups (no XCD-aware deal), "xcd" = XCD-aware ranges without the team walk; default: teams
    const char* place = ctx_opt(c, OPT_STREAMK_PLACE);
    const bool place_flat = place != nullptr && place[0] == 'f', place_xcd = place != nullptr && place[0] == 'x';
    if (!force && (long long)K * ((B + 127) / 128) * (N / 256) < c->n_sm && !late_for(K1, N, epi_units)) return sp;
    auto shape = [&](SkArgs& a, int K1_, int N_, int eu) { a = SkArgs{}; a.M = B; a.heads = K; a.K1 = K1_; a.N = N_; a.late = late_for(K1_, N_, eu);
                                                             a.xcd = place_flat ? 0 : 8;
                                                             a.team = place_xcd ? 0 : 1; };      // (sk_plan keeps it for SK_A_GLOBAL launches with enough tiles only)      // MI355X: 8 XCDs, workgroups dealt round-robin
    if (L == 3) {
        sp.S0 = (pd.nin + 1 + 3) / 4;
        const bool fused = sk_fused_vt(sp.S0, sp.OT, &sp.v1);                   // layer 0 as producer: inputs of up to 36 values (Humanoid's 76 + 1: mode 3)
        if (!fused && !sk_out_vt(sp.OT, &sp.v1)) return sp;
        shape(sp.a1, K1, N, epi_units);
        sp.a1.lda = fused ? 4 * sp.S0 : K1; sp.a1.ldp = 16 * sp.OT; sp.a1.stridePart = (long long)B * sp.a1.ldp;
        sp.p1 = sp.v1.plan(sp.a1, c->n_sm);
        if (sp.p1.lds_bytes > 160 * 1024) return sp;
        sp.mode = fused ? 1 : 3;
    } else {
        const int K0 = pd.dyn.dims[1], N0 = pd.dyn.dims[2];                  // layer 1: [K0 x N0], N0 == K1
        if (K0 % 32 != 0 || N0 % 256 != 0 || N0 != K1) return sp;
        if (!sk_out_vt(sp.OT, &sp.v2)) return sp;
        sp.v1 = sk_vt<SK_A_GLOBAL, SK_EPI_STORE, 1, 1>();
        shape(sp.a1, K0, N0, 1); shape(sp.a2, K1, N, epi_units);
        sp.a2.ldp = 16 * sp.OT; sp.a2.stridePart = (long long)B * sp.a2.ldp;
        sp.p1 = sp.v1.plan(sp.a1, c->n_sm); sp.p2 = sp.v2.plan(sp.a2, c->n_sm);
        if (sp.p1.lds_bytes > 160 * 1024 || sp.p2.lds_bytes > 160 * 1024) return sp;
        sp.mode = 2;
    }
    // element offsets of the schedule records are 32 bits
    if ((long long)K * pd.dyn.n_params >= (1LL << 31) || (long long)K * B * std::max(K1, N) >= (1LL << 31)) { sp.mode = 0; return sp; }
    // ... and the kernel forms BYTE offsets of a row inside one head's activations / outputs in 32 bits (a_voff, the STORE and partial-output rows)
    if ((long long)B * std::max(std::max(K1, N), 64) * 4 >= (1LL << 32)) { sp.mode = 0; return sp; }
    return sp;
}

// ---- persistent stream-K rollout (mlp_persist.h): the (env, input steps, output tiles) instantiations = the five envs of the 2 x 32 MFMA pre-step ----
struct SkpVt {
    const void* kern; size_t lds; bool wide;
    void (*launch)(const SkpArgs&, int grid, size_t lds, hipStream_t);
    void (*tab)(const SkArgs&, bool wide, std::vector<SkRec>&, int (&)[8], int&, int&, int&);
};
template <int ENV, int S0, int OT, bool WIDE> static SkpVt skp_vt() {
    SkpVt v;
    v.kern = (const void*)k_sk_persist<ENV, S0, OT, WIDE>; v.wide = WIDE;
    v.lds = sizeof(float) * (size_t)(4 * SkGeom<SK_A_PRODUCER, SK_EPI_OUT, S0, OT>::STAGE + 4);      // the ring + the workgroup's halt word
    v.launch = [](const SkpArgs& p, int grid, size_t lds, hipStream_t st) { hipLaunchKernelGGL((k_sk_persist<ENV, S0, OT, WIDE>), dim3(grid), dim3(512), lds, st, p); };
    v.tab = [](const SkArgs& a, bool wide, std::vector<SkRec>& t, int (&Jx)[8], int& Jmax, int& L, int& NSL) { skp_build_tab<OT>(a, wide, t, Jx, Jmax, L, NSL); };
    return v;
}
// wide: an even number of 256-column blocks -> tiles of two adjacent blocks sharing the layer-0 producer (option PERSIST_WIDE=0 keeps one block per tile, =1 forces two: tests)
static bool skp_select(const metrpo_ctx* c, int S0, int OT, int CB, int B, SkpVt* v) {
    const ProblemDesc& pd = c->pd;
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH) return false;
    // A row block's steps form a chain: (one tile + its closing) per step.  Two-block tiles double the tile time, so they only pay where the chip has more than
    // ~1.2 of them per compute workgroup and step -- below that the chain, not the matrix pipe, sets the step time (C3 share: 170 vs 128 us per step).
    const long long wide_tiles = (long long)pd.K * (CB / 2) * ((B + 127) / 128);
    const char* wo = ctx_opt(c, OPT_PERSIST_WIDE);
    const bool wide = (CB % 2 == 0) && !(wo && wo[0] == '0') && ((wo && wo[0] == '1') || 10 * wide_tiles >= 12 * (long long)(c->n_sm - 8));
#define SKP_PICK(ENV_, S0_, OT_) { *v = wide ? skp_vt<ENV_, S0_, OT_, true>() : skp_vt<ENV_, S0_, OT_, false>(); return true; }
    switch (pd.env) {
    case METRPO_ENV_SWIMMER:      if (pd.ns == 10 && pd.na == 2 && pd.n_drop == 2 && S0 == 3 && OT == 1) SKP_PICK(METRPO_ENV_SWIMMER, 3, 1) break;
    case METRPO_ENV_HOPPER:       if (pd.ns == 11 && pd.na == 3 && pd.n_drop == 0 && S0 == 4 && OT == 1) SKP_PICK(METRPO_ENV_HOPPER, 4, 1) break;
    case METRPO_ENV_SNAKE:        if (pd.ns == 14 && pd.na == 4 && pd.n_drop == 2 && S0 == 5 && OT == 1) SKP_PICK(METRPO_ENV_SNAKE, 5, 1) break;
    case METRPO_ENV_HALF_CHEETAH: if (pd.ns == 18 && pd.na == 6 && pd.n_drop == 1 && S0 == 6 && OT == 2) SKP_PICK(METRPO_ENV_HALF_CHEETAH, 6, 2) break;
    case METRPO_ENV_ANT:          if (pd.ns == 29 && pd.na == 8 && pd.n_drop == 2 && S0 == 9 && OT == 2) SKP_PICK(METRPO_ENV_ANT, 9, 2) break;
    }
#undef SKP_PICK
    return false;
}
// chunk-record table of this launch shape in device memory (cached in the context: the shape repeats every iteration)
static int skp_table(metrpo_ctx* c, const SkpVt& v, const SkArgs& a, SkpArgs* p, hipStream_t st) {
    const long long key[8] = {a.M, a.heads, a.K1, a.N, (long long)a.strideW1, (long long)a.strideW0, (long long)a.stridePart, (long long)c->pd.env * 2 + (v.wide ? 1 : 0)};
    bool same = c->d_skp_tab != nullptr;
    for (int i = 0; i < 8; ++i) same = same && c->skp_key[i] == key[i];
    if (!same) {
        // first use of a shape only.  The records live in the context (the copy below is ordered on the caller's stream behind the launches that still read the old
        // table; a pageable source is staged before the call returns, and the vector outlives it anyway); no synchronisation of the stream
        std::vector<SkRec> tab;
        v.tab(a, v.wide, tab, c->skp_Jx, c->skp_Jmax, c->skp_L, c->skp_NSL);
        const size_t bytes = tab.size() * sizeof(SkRec);
        c->skp_tab_host.assign((const int*)tab.data(), (const int*)tab.data() + bytes / sizeof(int));
        if (bytes > c->skp_tab_cap) {
            ws_retire(c, c->d_skp_tab);          // (launches already enqueued may still read the old table: retired, not freed)
            c->d_skp_tab = nullptr; c->skp_tab_cap = 0;
            HIP_TRY(c, ws_alloc(c, (void**)&c->d_skp_tab, bytes));
            c->skp_tab_cap = bytes;
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_skp_tab, c->skp_tab_host.data(), bytes, hipMemcpyHostToDevice, st));
        for (int i = 0; i < 8; ++i) c->skp_key[i] = key[i];
    }
    p->tab = (const SkRec*)c->d_skp_tab;
    for (int x = 0; x < 8; ++x) p->Jx[x] = c->skp_Jx[x];
    p->Jmax = c->skp_Jmax; p->L = c->skp_L; p->NSL = c->skp_NSL;
    return METRPO_OK;
}

bool gemm_path_applicable(const metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    if (pd.ns > 64) return false;
    int minw = 1 << 30;
    for (int l = 1; l < pd.dyn.n_layers; ++l) minw = std::min(minw, pd.dyn.dims[l]);
    // (>= 128 until round 5: the widths 65 .. 127 -- and any width the fused 2 x 64 kernels do not hold -- then fell to the thread-per-env kernel, ~80x the time
    // per env step; the tile GEMMs take any width, every tile predicated)
    return pd.dyn.n_layers >= 2 && minw >= 16;
}

// One chunk of the step loop on stream `st`.  ws == nullptr: only report the workspace size (bytes, 256-aligned) through *need_out.
static int rollout_gemm_chunk(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st, char* ws, size_t* need_out, int vB = 0, int vR = 0) {
    const ProblemDesc& pd = c->pd;
    const int B = a->B, K = pd.K, L = pd.dyn.n_layers;
    int maxh = 0;
    for (int l = 1; l < L; ++l) maxh = std::max(maxh, pd.dyn.dims[l]);
    // workspace (floats): S, X, U, HA, HB, OUT + ints ts, cur_model
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };       // keep every sub-buffer 16-byte aligned
    SkPath sk = sk_select(c, B);
    // stored layer 0 (modes 2, 3) by k_l0_rows: the bias rides as input row n_in, as for the producer of mode 1 (METRPO_NO_L0_ROWS: the tile GEMM)
    const int S0all = (pd.nin + 1 + 3) / 4;
    // (below one tile per CU -- the late split -- the 64 x 64-tile GEMM is the faster layer 0: 11.9 vs 21.2 us at 500 rows per head)
    const bool sk_small = sk.mode != 0 && ((sk.mode == 2) ? sk.a2.late : sk.a1.late) != 0;
    const bool l0r = sk.mode >= 2 && !sk_small && l0_rows_ok(S0all) && pd.dyn.dims[1] % 256 == 0 && pd.dyn.act[0] == METRPO_ACT_RELU &&
                     pd.dyn.b_off[0] == pd.dyn.w_off[0] + pd.nin * pd.dyn.dims[1] && ctx_opt(c, OPT_NO_L0_ROWS) == nullptr;
    const int ldx = (sk.mode == 1) ? 4 * sk.S0 : (l0r ? 4 * S0all : ((pd.nin + 3) & ~3));
    const size_t nS = up4((size_t)B * pd.ns), nX = up4((size_t)B * ldx), nU = up4((size_t)B * pd.na), nH = (sk.mode == 1) ? 0 : up4((size_t)K * B * maxh), nO = up4((size_t)K * B * pd.ns);
    size_t nP = 0;
    for (int l = 0; l < L; ++l) nP = std::max(nP, up4(skinny_part_floats(B, pd.dyn.dims[l + 1], pd.dyn.dims[l], K)));
    // stream-K path: output-layer partials per 256-column block, epilogue images, schedules, accumulator hand-over slots + flags
    const SkArgs& ska = (sk.mode == 2) ? sk.a2 : sk.a1;                      // the launch that carries the output layer
    size_t nSkImg = 0, nSkSched = 0, nSkX = 0, nSkFlag = 0;
    if (sk.mode) {
        nP = std::max(nP, up4((size_t)(ska.N / 256) * K * ska.stridePart));
        nSkImg = up4(sk_epi_floats(sk.OT, ska.N, K));
        nSkSched = ((sk.p1.sched_bytes + 255) & ~(size_t)255) / 4 + ((sk.mode == 2 ? sk.p2.sched_bytes + 255 : 0) & ~(size_t)255) / 4 + 64;
        nSkX = up4(std::max(sk.p1.xacc_floats, sk.mode == 2 ? sk.p2.xacc_floats : (size_t)0));
        nSkFlag = up4((size_t)std::max(sk.p1.nflags, sk.mode == 2 ? sk.p2.nflags : 0));
    }
    // last hidden layer + output layer as ONE launch when the hidden layer runs on 64x64 tiles anyway (C0-params-file, C2, C3 shapes): its
    // activations (K x B x width floats: 51 MB at C3) are then neither written nor read back; k_big_post adds the width/64 partials
    const int fuse_tile = (!sk.mode && L >= 2 && pd.dyn.act[L - 2] == METRPO_ACT_RELU && pd.dyn.act[L - 1] == METRPO_ACT_IDENTITY && ctx_opt(c, OPT_NO_FUSED_OUT) == nullptr)
                              ? gemm_fused_out_tile(B, pd.dyn.dims[L - 1], K, pd.ns) : 0;
    const bool fuse_out = fuse_tile > 0;
    if (fuse_out) nP = std::max(nP, up4(gemm_fused_out_part_floats(B, pd.dyn.dims[L - 1], K, pd.ns, fuse_tile)));
    size_t pre_lds = 0;
    const big_pre_mfma_t pre_mfma = big_pre_mfma_select(c, &pre_lds);
    // steps t >= 1: k_big_post(t - 1) rides in the pre-step's launch (the MFMA pre-kernels: 2 x 32 policies and Humanoid's 100-50-25; STEP_MERGE=0 keeps the two launches: A/B runs, tests)
    size_t pre_lds_post = 0;
    // (Humanoid's 55 dims are 16 per lane in that layout: behind the tile GEMMs' 16 output partials per head -- small batches -- the closing part is slower
    //  than k_big_post's lane per dim, 12.9 vs 11.1 ms per params-file rollout; behind stream-K's 4 partials it is merged.  METRPO_STEP_MERGE=1 forces it: tests)
    //  Small batches of the 100-50-25 pre-step (a tile per workgroup: k_big_pre_mfma3_split, below) keep two launches also behind stream-K: 9.6 + 6.3 us against the merged 29.4.)
    const char* sm_opt = ctx_opt(c, OPT_STEP_MERGE);          // "1" forces the merged launch, "0" keeps two launches (A/B runs, tests)
    const bool step_merge_forced = sm_opt != nullptr && sm_opt[0] != '0', step_merge_off = sm_opt != nullptr && sm_opt[0] == '0';
    const bool split_pre_small = pd.ns > 32 && B <= 16 * c->n_sm && ctx_opt(c, OPT_NO_PRE_SPLIT) == nullptr && !step_merge_forced;
    const bool merge_ok = !step_merge_off && !split_pre_small && (pd.ns <= 32 || sk.mode != 0 || step_merge_forced);
    const big_pre_mfma_t pre_post = (pre_mfma && merge_ok) ? big_pre_mfma_select(c, &pre_lds_post, true) : nullptr;
    // policies without an MFMA pre-kernel: GEMM chain over the batch from B = 1024 up -- and at ANY batch when the policy is large
    // (k_big_pre walks the weights through scalar loads, one block's time whatever B: 302 us per step for Humanoid's 100-50-25 at B = 100,
    // the params-humanoid.json shape, against ~40 us for the six small launches of the chain: iteration 77 -> 28 ms)
    // (METRPO_PRE_GEMM=1 forces it, =0 forbids it: tests)
    const char* pg_env = ctx_opt(c, OPT_PRE_GEMM);
    const bool pre_gemm = !pre_mfma && ((pg_env && pg_env[0] == '1') || (!(pg_env && pg_env[0] == '0') && (B >= 1024 || pd.pol.n_params >= 4096)));
    const size_t nPol = pre_gemm ? up4((size_t)B * pd.pol.max_width) : 0;
    const size_t nPimg = pre_lds ? up4((size_t)pre_mfma3_image_floats<55, 21, 100, 50, 25>()) : 0;      // the only three-hidden-layer instantiation (big_pre_mfma_select)
    // persistent form of mode 1 (mlp_persist.h): every step of this chunk in ONE launch.  Needs the MFMA pre-step of the 2 x 32 policies (its wave functions close a
    // step and prepare the next inside the launch), whole-batch rows (no merged rounds), at least two steps, and its grid on the chip at once.
    SkpVt skp = {};
    int skp_grid = 0;
    bool persist = sk.mode == 1 && vB == 0 && a->T >= 2 && ctx_opt(c, OPT_NO_PERSIST) == nullptr && !c->persist_failed && pre_lds == 0 && pre_mfma != nullptr &&
                   skp_select(c, sk.S0, sk.OT, sk.a1.N / 256, B, &skp);
    if (persist) {
        if (skp.lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute(skp.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)skp.lds));
        skp_grid = (std::min(sched_cus(c, st), c->n_sm) / 8) * 8;
        persist = skp_grid >= 16 && grid_is_coresident(c, skp.kern, 512, skp.lds, skp_grid, st);
        // The persistent launch works in WHOLE 128 x 256 tiles: a step cannot be shorter than one tile's k loop on one CU (~150 us at hidden 1024, ~80 at 512), whatever
        // B is.  With fewer tiles per step than workgroups the per-step stream-K launches win -- they split a tile's k range over the idle CUs -- by 2.5 x at the strong-
        // scaling shares of C2 / C3 (B = 312: 33.5 -> 13.2 ms per rollout; B = 625 Ant: 52 -> 30 ms); the crossover sits at one tile per workgroup (240 tiles: per-step
        // launches 7-11 % ahead, 280 tiles: persistent 0-6 % ahead, 400: 5-14 %; profiles/r06_share_sweep.txt).  STREAMK=1 (parity tests at oracle sizes) keeps it.
        const long long tiles_step = (long long)K * ((B + 127) / 128) * (sk.a1.N / 256);
        if (persist && !sk.forced && tiles_step < skp_grid + skp_grid / 16) persist = false;
    }
    int skp_nclose = SKP_NCLOSE;
    if (persist) {
        // per step: RB closings of ~17 us on nclose workgroups against (tiles x chunks x ~4.6 us) / compute workgroups of matrix work
        const int RBp = (B + 127) / 128, CBp = sk.a1.N / 256;
        const double step_us = (double)K * CBp * RBp * (sk.a1.K1 / 32 + 1) * 4.6 / std::max(1, skp_grid - 8);
        skp_nclose = std::max(2, std::min((int)SKP_NCLOSE, (int)std::ceil(RBp * 17.0 / (0.6 * step_us))));
    }
    const size_t nSkpFlag = persist ? up4(2 * (size_t)((B + 127) / 128)) : 0, nSkpPost = persist ? up4((sizeof(SkpPost) + 3) / 4) : 0;
    const bool skp_stop = persist && a->stop_batch > 0 && a->d_stop_cum != nullptr;
    const size_t nSkpStop = skp_stop ? up4(3 * (size_t)a->T + 8) : 0;       // floats: cnt[T] u64 | closed[T] u32 | misc[4] u32 | cum u64
    const size_t need = ((nS + nX + nU + 2 * nH + nO + nP + 2 * nPol + nPimg + nSkImg + nSkSched + nSkX + nSkFlag + nSkpFlag + nSkpPost + nSkpStop) * sizeof(float) + 2 * (size_t)B * sizeof(int) + 1023) & ~(size_t)255;
    if (need_out) *need_out = need;
    if (ws == nullptr) return METRPO_OK;
    BigState bs = {};
    float* p = (float*)ws;
    bs.S = p; p += nS; bs.X = p; p += nX; bs.U = p; p += nU; bs.HA = p; p += nH; bs.HB = p; p += nH; bs.OUT = p; p += nO; bs.PART = nP ? p : nullptr; p += nP; bs.PA = p; p += nPol; bs.PB = p; p += nPol; bs.PIMG = nPimg ? p : nullptr; p += nPimg;
    float* sk_img = p; p += nSkImg;
    char* sk_sched = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255); p += nSkSched;     // 256-byte aligned inside its (64 floats larger) slot
    float* sk_xacc = p; p += nSkX;
    unsigned* sk_flag = (unsigned*)p; p += nSkFlag;
    int* skp_flags = (int*)p; p += nSkpFlag;
    SkpPost* skp_post = (SkpPost*)p; p += nSkpPost;
    float* skp_stopmem = p; p += nSkpStop;
    bs.ts = (int*)p; bs.cur_model = bs.ts + B;
    bs.out_ld = pd.ns; bs.xone = (sk.mode == 1 || l0r) ? 1 : 0;
    if (sk.mode) c->last_rollout_kernel = 5;
    unsigned sk_epoch = 0;                                   // flags are zeroed below; every launch of this chain takes the next epoch
    if (sk.mode) {
        const int Lo = L - 1;                                // output layer
        HIP_TRY(c, hipMemsetAsync(sk_flag, 0, nSkFlag * sizeof(float), st));
        sk_epi_image(sk.OT, c->d_dyn + pd.dyn.b_off[Lo - 1], pd.dyn.n_params, c->d_dyn + pd.dyn.w_off[Lo], pd.dyn.n_params, pd.ns, ska.N, K, sk_img, st);
        SkArgs& o = (sk.mode == 2) ? sk.a2 : sk.a1;
        o.W1 = c->d_dyn + pd.dyn.w_off[Lo - 1]; o.strideW1 = pd.dyn.n_params;
        o.epi = sk_img; o.part = bs.PART;
        if (sk.mode == 1) {
            o.A = bs.X; o.strideA = 0;
            o.W0 = c->d_dyn + pd.dyn.w_off[0]; o.strideW0 = pd.dyn.n_params;      // row nin of the resident layout = b0 (X[nin] = 1), the rows behind it meet X's zero pad
        } else if (sk.mode == 3) {
            o.A = bs.HA; o.strideA = (long long)B * o.K1; o.lda = o.K1;
        } else {
            SkArgs& h = sk.a1;                               // layer 1: relu(HA W1 + b1) -> HB
            h.A = bs.HA; h.strideA = (long long)B * h.K1; h.lda = h.K1;
            h.W1 = c->d_dyn + pd.dyn.w_off[1]; h.strideW1 = pd.dyn.n_params; h.b1 = c->d_dyn + pd.dyn.b_off[1]; h.strideB1 = pd.dyn.n_params;
            h.C = bs.HB; h.strideC = (long long)B * h.N; h.ldc = h.N;
            o.A = bs.HB; o.strideA = (long long)B * o.K1; o.lda = o.K1;
        }
        sk.a1.xacc = sk_xacc; sk.a1.xflag = sk_flag; sk.a1.err = comm_err_cell(c) + 1;      // scal[S_ROLLERR]
        if (!persist) HIP_TRY(c, sk.v1.sched(sk.a1, sk.p1, sk_sched, st));
        if (sk.mode == 2) {
            sk.a2.xacc = sk_xacc; sk.a2.xflag = sk_flag; sk.a2.err = comm_err_cell(c) + 1;
            HIP_TRY(c, sk.v2.sched(sk.a2, sk.p2, sk_sched + ((sk.p1.sched_bytes + 255) & ~(size_t)255), st));
        }
        bs.out_splits = ska.N / 256; bs.out_stride = ska.stridePart; bs.out_ld = ska.ldp;
        bs.out_bias = c->d_dyn + pd.dyn.b_off[Lo]; bs.out_bias_stride = pd.dyn.n_params;
    }
    if (nPimg) hipLaunchKernelGGL((k_pre_mfma3_image<55, 21, 100, 50, 25>), dim3((unsigned)((nPimg + 255) / 256)), dim3(256), 0, st, c->d_theta, bs.PIMG);
    bs.ldx = ldx;
    RolloutK r = make_rollout_k(a);
    r.vB = vB; r.vR = vR;                                    // merged rounds: a->B = vR * vB rows, a->T = H steps (launch_rollout_gemm)
    const int pbs = 256;                                     // 64 envs x 4 output groups
    const size_t psh = (size_t)(pd.ns + 2 * pd.pol.max_width) * 64 * sizeof(float);
    if (psh > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "policy too wide for k_big_pre");
    if (psh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_big_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psh));
    if (pre_mfma && pre_lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pre_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre_lds));
    if (pre_post && pre_lds_post > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pre_post, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre_lds_post));
    if (persist) {
        // step 0's pre-step as its own launch (vec_env.reset() / the continuation state), then ONE launch for the tiles of all T steps: it closes steps 0 .. T - 2 and
        // prepares steps 1 .. T - 1 itself; the last step is closed by k_big_post below, which also hands out the continuation state
        SkpArgs pa = {};
        pa.a = sk.a1; pa.a.epoch = 0;
        { const int rc = skp_table(c, skp, sk.a1, &pa, st); if (rc) return rc; }
        pa.G8 = skp_grid / 8; pa.T = a->T;
        // closing workgroups: each closes ~17 us per (row block, step); enough of them to stay under ~60 % busy at this launch's step time, at most one per XCD
        { const char* nc = ctx_opt(c, OPT_PERSIST_NCLOSE); pa.nclose = nc ? std::max(1, std::min((int)SKP_NCLOSE, atoi(nc))) : skp_nclose; }
        pa.xflag = skp_flags; pa.arrive = (unsigned*)(skp_flags + (B + 127) / 128); pa.stop = r.stop; pa.post = skp_post;
        HIP_TRY(c, hipMemsetAsync(skp_flags, 0, nSkpFlag * sizeof(float), st));
        if (ctx_opt(c, OPT_PERSIST_STATS) != nullptr) {           // developer statistics of this launch (metrpo_debug_persist_stats)
            if (c->skp_stats_n < skp_grid) { ws_retire(c, c->d_skp_stats); c->d_skp_stats = nullptr; HIP_TRY(c, ws_alloc(c, (void**)&c->d_skp_stats, sizeof(unsigned long long) * 8 * skp_grid)); }
            c->skp_stats_n = skp_grid;
            HIP_TRY(c, hipMemsetAsync(c->d_skp_stats, 0, sizeof(unsigned long long) * 8 * skp_grid, st));
            pa.stats = c->d_skp_stats;
            pa.nowait = ctx_opt(c, OPT_PERSIST_STATS)[0] == '2' ? 1 : ctx_opt(c, OPT_PERSIST_STATS)[0] == '3' ? 3 : 0;      // 2: no flag waits; 3: no flag waits and no chunk barrier (timing only)
        }
        if (skp_stop) {                                          // the sampler's stop rule inside the launch (mlp_persist.h)
            HIP_TRY(c, hipMemsetAsync(skp_stopmem, 0, nSkpStop * sizeof(float), st));
            pa.cnt = (unsigned long long*)skp_stopmem; pa.closed = (unsigned*)(pa.cnt + a->T); pa.misc = pa.closed + a->T + (a->T & 1);      // (misc 8-byte aligned: cum follows it)
            pa.cum = (unsigned long long*)(pa.misc + 4);
            HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)(pa.misc + 2), (int)SKP_NOHALT, 1, st));
            pa.stop_batch = a->stop_batch; pa.stop_cum0 = a->d_stop_cum;
        }
        SkpPost po; po.pd = pd; po.r = r; po.st = bs; po.theta = c->d_theta; po.norm = c->d_norm;
        hipLaunchKernelGGL(k_skp_post_args, dim3(1), dim3(64), 0, st, po, skp_post);
        hipLaunchKernelGGL(pre_mfma, dim3((B + 63) / 64), dim3(256), pre_lds, st, pd, r, 0, c->d_theta, c->d_norm, bs);
        skp.launch(pa, skp_grid, skp.lds, st);
        const int tl = a->T - 1;
        if (pd.ns <= 32) hipLaunchKernelGGL(k_big_post<32>, dim3((B + 7) / 8), dim3(256), 0, st, pd, r, tl, c->d_norm, bs);
        else hipLaunchKernelGGL(k_big_post<64>, dim3((B + 3) / 4), dim3(256), 0, st, pd, r, tl, c->d_norm, bs);
        HIP_TRY(c, hipGetLastError());
        c->last_rollout_kernel = 6;
        return METRPO_OK;
    }
    // Humanoid's 100-50-25 pre-step at small batches: a tile per workgroup, column blocks over its waves (k_big_pre_mfma3_split)
    const bool pre_split = pre_mfma != nullptr && pre_lds != 0 && B <= 16 * c->n_sm && ctx_opt(c, OPT_NO_PRE_SPLIT) == nullptr;
    for (int t = 0; t < a->T; ++t) {
        if (pre_split && !(pre_post && t > 0)) hipLaunchKernelGGL((k_big_pre_mfma3_split<55, 21, 0, 100, 50, 25>), dim3((B + 15) / 16), dim3(256), 0, st, pd, r, t, c->d_theta, c->d_norm, bs);
        else if (pre_post && t > 0) hipLaunchKernelGGL(pre_post, dim3((B + 63) / 64), dim3(256), pre_lds_post, st, pd, r, t, c->d_theta, c->d_norm, bs);
        else if (pre_mfma) hipLaunchKernelGGL(pre_mfma, dim3((B + 63) / 64), dim3(256), pre_lds, st, pd, r, t, c->d_theta, c->d_norm, bs);
        else if (pre_gemm) {
            hipLaunchKernelGGL(k_big_pre_gather, dim3((unsigned)(((long long)B * pd.ns + 255) / 256)), dim3(256), 0, st, pd, r, t, c->d_norm, bs);
            const float* pin = bs.S; int ldp = pd.ns;
            float* pbuf[2] = {bs.PA, bs.PB};
            for (int l = 0; l < pd.pol.n_layers; ++l) {
                const int Kp = pd.pol.dims[l], Np = pd.pol.dims[l + 1];
                float* pout = pbuf[l & 1];
                gemm_launch(pd.pol.act[l], pin, 0, ldp, c->d_theta + pd.pol.w_off[l], 0, Np, c->d_theta + pd.pol.b_off[l], 0, pout, 0, Np, B, Np, Kp, 1, st);
                pin = pout; ldp = Np;
            }
            const int npair = (pd.na + 1) / 2;
            hipLaunchKernelGGL(k_big_pre_action, dim3((unsigned)(((long long)B * npair + 255) / 256)), dim3(256), 0, st, pd, r, t, c->d_theta, c->d_norm, pin, bs);
        }
        else hipLaunchKernelGGL(k_big_pre, dim3((B + 63) / 64), dim3(pbs), psh, st, pd, r, t, c->d_theta, c->d_norm, bs);
        const float* in = bs.X; long long sIn = 0; int ldin = bs.ldx;
        float* bufs[2] = {bs.HA, bs.HB};
        if (sk.mode == 1) {                                  // the whole dynamics ensemble of this step in one launch
            sk.a1.epoch = ++sk_epoch;
            HIP_TRY(c, sk.v1.launch(sk.a1, sk.p1, st));
        } else if (sk.mode >= 2) {
            if (l0r) {
                L0Args la = {};
                la.M = B; la.heads = K; la.N = pd.dyn.dims[1]; la.ldx = bs.ldx; la.x = bs.X; la.W0 = c->d_dyn + pd.dyn.w_off[0]; la.strideW0 = pd.dyn.n_params;
                la.C = bs.HA; la.strideC = (long long)B * pd.dyn.dims[1];
                hipError_t e = hipSuccess;
                (void)l0_rows(S0all, la, c->n_sm, st, &e);
                HIP_TRY(c, e);
            } else
            gemm_launch(METRPO_ACT_RELU, bs.X, 0, bs.ldx, c->d_dyn + pd.dyn.w_off[0], pd.dyn.n_params, pd.dyn.dims[1], c->d_dyn + pd.dyn.b_off[0], pd.dyn.n_params,
                        bs.HA, (long long)B * pd.dyn.dims[1], pd.dyn.dims[1], B, pd.dyn.dims[1], bs.ldx, K, st);
            sk.a1.epoch = ++sk_epoch;
            HIP_TRY(c, sk.v1.launch(sk.a1, sk.p1, st));
            if (sk.mode == 2) {
                sk.a2.epoch = ++sk_epoch;
                HIP_TRY(c, sk.v2.launch(sk.a2, sk.p2, st));
            }
        }
        for (int l = 0; l < L && !sk.mode; ++l) {
            // layer 0 contracts over the PADDED input row (X's pad columns are 0, the weight rows they meet are the first bias entries that follow
            // W0 in the resident layout: finite x 0): a contraction length that is a multiple of 4 takes the GEMM's aligned load path
            const int Kd = (l == 0 && L > 1) ? bs.ldx : pd.dyn.dims[l], N = pd.dyn.dims[l + 1];
            const bool lastl = (l == L - 1);
            float* out = lastl ? bs.OUT : bufs[l & 1];
            const long long sOut = (long long)B * N;
            const float* Wl = c->d_dyn + pd.dyn.w_off[l];
            const float* bl = c->d_dyn + pd.dyn.b_off[l];
            if (fuse_out && l == L - 2) {
                const float* W2 = c->d_dyn + pd.dyn.w_off[L - 1];
                gemm_relu_fused_out(fuse_tile, in, sIn, ldin, Wl, pd.dyn.n_params, N, bl, pd.dyn.n_params, W2, pd.dyn.n_params, pd.ns, B, N, Kd, K, bs.PART, st,
                                    &bs.out_splits, &bs.out_stride);
                bs.out_bias = c->d_dyn + pd.dyn.b_off[L - 1]; bs.out_bias_stride = pd.dyn.n_params;
                break;
            }
            SkinnyDefer df = {0, 0};
            gemm_skinny_bias(in, sIn, ldin, Wl, pd.dyn.n_params, N, bl, pd.dyn.n_params, out, sOut, B, N, Kd, K, bs.PART, st, pd.dyn.act[l], lastl ? &df : nullptr);
            if (lastl) { bs.out_splits = df.splits; bs.out_stride = df.stridePart; bs.out_bias = bl; bs.out_bias_stride = pd.dyn.n_params; }
            in = out; sIn = sOut; ldin = N;
        }
        if (pre_post && t + 1 < a->T) continue;             // closed by the next step's launch
        if (pd.ns <= 32) hipLaunchKernelGGL(k_big_post<32>, dim3((B + 7) / 8), dim3(256), 0, st, pd, r, t, c->d_norm, bs);
        else hipLaunchKernelGGL(k_big_post<64>, dim3((B + 3) / 4), dim3(256), 0, st, pd, r, t, c->d_norm, bs);
    }
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}


// Reset states of rounds 1 .. R-1 of a rollout whose envs only end at the horizon: what k_big_post's reset branch at step round * H - 1 produces
// (pool row and next model from that step's draw, env_helpers.py:585-595), without running the steps before it.
__global__ void k_round_init(ProblemDesc pd, RolloutK r, int R, float* __restrict__ init_obs, int32_t* __restrict__ init_ts, int32_t* __restrict__ init_model) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = pd.ns;
    if (idx >= (long long)(R - 1) * r.B * ns) return;
    const int i = (int)(idx % ns), b = (int)((idx / ns) % r.B), round = 1 + (int)(idx / ((long long)ns * r.B));
    const uint4 dstep = rng_draw(r.seed, r.stream_offset + (uint64_t)b, r.t0 + round * r.H - 1, RNG_STEP, 0);
    const int row = rng_index(dstep.w, r.n_pool);
    const size_t o = (size_t)(round - 1) * r.B + b;
    init_obs[o * ns + i] = r.pool[(size_t)row * ns + i];
    if (i == 0) { init_ts[o] = 0; init_model[o] = rng_index16(dstep.z, pd.K); }
}

// Merged rounds: the start states of ALL R rounds, [R][B] -- round 0 from the reset draw of vec_env.reset() (env_helpers.py:585-595, as the
// pre-kernels' own t = 0 branch), rounds 1 .. R-1 as k_round_init.
__global__ void k_round_init_all(ProblemDesc pd, RolloutK r, int R, int Bv, float* __restrict__ init_obs, int32_t* __restrict__ init_ts, int32_t* __restrict__ init_model) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = pd.ns;
    if (idx >= (long long)R * Bv * ns) return;
    const int i = (int)(idx % ns), b = (int)((idx / ns) % Bv), round = (int)(idx / ((long long)ns * Bv));
    int row, model;
    if (round == 0) {
        const uint4 d0 = rng_draw(r.seed, r.stream_offset + (uint64_t)b, 0, RNG_RESET, 0);
        row = rng_index(d0.x, r.n_pool); model = rng_index(d0.y, pd.K);
    } else {
        const uint4 dstep = rng_draw(r.seed, r.stream_offset + (uint64_t)b, r.t0 + round * r.H - 1, RNG_STEP, 0);
        row = rng_index(dstep.w, r.n_pool); model = rng_index16(dstep.z, pd.K);
    }
    const size_t o = (size_t)round * Bv + b;
    init_obs[o * ns + i] = r.pool[(size_t)row * ns + i];
    if (i == 0) { init_ts[o] = 0; init_model[o] = model; }
}

// Step-wise rollout of a large dynamics ensemble.  The R = T / H rounds of a horizon-terminated rollout that starts from a reset are
// independent given the counter-based draws (every env is reset at the same steps, and its reset state is a function of that step's
// draw alone), and at the reference's own batch size (B = 100: params-*.json) a round is a chain of ~3 us launches that leaves most of the
// chip idle: the rounds then run CONCURRENTLY, one per stream, each as a continuation chunk (t0 = round * H) on its own workspace, and
// produce bit for bit what the sequential loop does.
int launch_rollout_gemm(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int B = a->B, H = a->H;
    const int R = (H > 0 && a->T % H == 0) ? a->T / H : 1;
    const bool par = R >= 2 && R <= METRPO_MAX_PAR_ROUNDS && pd.env != METRPO_ENV_ANT && (long long)pd.K * B <= 8192 &&
                     a->t0 == 0 && a->d_init_obs == nullptr && a->d_stop == nullptr && a->d_eps == nullptr && a->d_model_idx == nullptr &&
                     a->d_sel_noise == nullptr && a->d_reset_idx == nullptr && a->d_reset_model == nullptr && ctx_opt(c, OPT_SEQ_ROUNDS) == nullptr;
    // Small batches (R B <= 1024 rows): the rounds as ONE batch of R B envs stepping H times -- one launch chain instead of R concurrent ones, and
    // GEMMs over R B rows instead of R GEMMs over B (B = 100 fills 100 of 128 tile rows, 500 fill 500 of 512; params-humanoid.json: 27.7 -> see DESIGN).
    // Row b of the merged batch = env b % B of round b / B (RolloutK::vB / vR); same draws, same trajectory rows as the round-by-round loop.
    const bool merged = par && (long long)R * B <= 1024 && ctx_opt(c, OPT_NO_MERGED_ROUNDS) == nullptr;
    size_t need1 = 0;
    {
        metrpo_rollout_args probe = *a;
        if (par) probe.T = H;
        if (merged) probe.B = R * B;
        const int rc = rollout_gemm_chunk(c, &probe, st, nullptr, &need1);
        if (rc != METRPO_OK) return rc;
    }
    const size_t init_rows = merged ? (size_t)R * B : (par ? (size_t)(R - 1) * B : 0);
    const size_t init_bytes = (init_rows * (pd.ns * sizeof(float) + 2 * sizeof(int32_t)) + 255) & ~(size_t)255;
    const size_t need = ((par && !merged) ? (size_t)R * need1 : need1) + init_bytes;
    if (need > c->big_cap) {
        ws_retire(c, c->d_big);
        c->d_big = nullptr; c->big_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_big, need));
        c->big_cap = need;
    }
    if (merged) {
        char* base = (char*)c->d_big;
        float* init_obs = (float*)(base + need1);
        int32_t* init_ts = (int32_t*)(init_obs + (size_t)R * B * pd.ns);
        int32_t* init_model = init_ts + (size_t)R * B;
        const RolloutK rk = make_rollout_k(a);
        const long long n_init = (long long)R * B * pd.ns;
        hipLaunchKernelGGL(k_round_init_all, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, st, pd, rk, R, B, init_obs, init_ts, init_model);
        metrpo_rollout_args am = *a;
        am.B = R * B; am.T = H;
        am.d_init_obs = init_obs; am.d_init_ts = init_ts; am.d_init_model = init_model;
        return rollout_gemm_chunk(c, &am, st, base, nullptr, B, R);
    }
    if (!par) return rollout_gemm_chunk(c, a, st, (char*)c->d_big, nullptr);
    if (!c->side_ready) {
        for (int i = 0; i < METRPO_MAX_PAR_ROUNDS - 1; ++i) {
            HIP_TRY(c, hipStreamCreateWithFlags(&c->side_stream[i], hipStreamNonBlocking));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
        }
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        c->side_ready = 1;
    }
    char* base = (char*)c->d_big;
    float* init_obs = (float*)(base + (size_t)R * need1);
    int32_t* init_ts = (int32_t*)(init_obs + (size_t)(R - 1) * B * pd.ns);
    int32_t* init_model = init_ts + (size_t)(R - 1) * B;
    const RolloutK rk = make_rollout_k(a);
    const long long n_init = (long long)(R - 1) * B * pd.ns;
    hipLaunchKernelGGL(k_round_init, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, st, pd, rk, R, init_obs, init_ts, init_model);
    HIP_TRY(c, hipEventRecord(c->ev_fork, st));
    // At these sizes the step loop is bound by the HOST's launch rate (~3 us per launch, 4 launches per step: the GPU is waiting), so every
    // round is enqueued by its own host thread; round 0 by the caller's, on the caller's stream.
    auto run_round = [&](int round) -> int {
        metrpo_rollout_args ar = *a;
        const size_t rows = (size_t)round * H * B;
        ar.T = H; ar.t0 = round * H;
        ar.d_obs = a->d_obs + rows * pd.ns; ar.d_act = a->d_act + rows * pd.na; ar.d_mean = a->d_mean + rows * pd.na;
        ar.d_rew = a->d_rew + rows; ar.d_done = a->d_done + rows; ar.d_tpath = a->d_tpath + rows;
        if (round != R - 1) { ar.d_last_obs = nullptr; ar.d_last_ts = nullptr; ar.d_last_model = nullptr; }
        if (round > 0) {
            ar.d_init_obs = init_obs + (size_t)(round - 1) * B * pd.ns; ar.d_init_ts = init_ts + (size_t)(round - 1) * B; ar.d_init_model = init_model + (size_t)(round - 1) * B;
        }
        hipStream_t rs = (round == 0) ? st : c->side_stream[round - 1];
        if (round > 0) {
            if (hipSetDevice(c->device) != hipSuccess) return METRPO_EHIP;                 // a fresh host thread has no current device
            if (hipStreamWaitEvent(rs, c->ev_fork, 0) != hipSuccess) return METRPO_EHIP;
        }
        const int rc = rollout_gemm_chunk(c, &ar, rs, base + (size_t)round * need1, nullptr);
        if (rc != METRPO_OK) return rc;
        if (round > 0 && hipEventRecord(c->ev_join[round - 1], rs) != hipSuccess) return METRPO_EHIP;
        return METRPO_OK;
    };
    int rcs[METRPO_MAX_PAR_ROUNDS] = {0};
    bool threaded[METRPO_MAX_PAR_ROUNDS] = {false};
    std::thread workers[METRPO_MAX_PAR_ROUNDS - 1];
    for (int round = 1; round < R; ++round) {
        try { workers[round - 1] = std::thread([&, round] { rcs[round] = run_round(round); }); threaded[round] = true; }
        catch (...) { threaded[round] = false; }                  // no thread to be had (std::system_error must not cross the C ABI): enqueue it here
    }
    rcs[0] = run_round(0);
    for (int round = 1; round < R; ++round) { if (threaded[round]) workers[round - 1].join(); else rcs[round] = run_round(round); }
    // The caller's stream joins EVERY round that was started, failed ones included: their side streams may still be writing the
    // trajectory tensors and the shared workspace, and the next call must not race with them.  A round that failed before it
    // recorded its join event is drained on the host instead.
    int first_bad = METRPO_OK;
    for (int round = 0; round < R; ++round) if (rcs[round] != METRPO_OK && first_bad == METRPO_OK) first_bad = rcs[round];
    for (int round = 1; round < R; ++round) {
        if (rcs[round] == METRPO_OK) { if (hipStreamWaitEvent(st, c->ev_join[round - 1], 0) != hipSuccess && first_bad == METRPO_OK) first_bad = METRPO_EHIP; }
        else (void)hipStreamSynchronize(c->side_stream[round - 1]);
    }
    if (first_bad != METRPO_OK) { (void)hipGetLastError(); return first_bad; }       // c->err holds the failing round's message (set_err is serialised)
    return METRPO_OK;
}

