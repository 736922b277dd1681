// Pre-step (policy.get_actions + clip + normalise / drop -> X) and closing of a step (k_big_post's work) of the step-wise rollout paths, as
// WAVE-level device functions: one wave = one 16-env tile.  Shared by the launch-per-step kernels of rollout_gemm.hip (k_big_pre_mfma) and the
// persistent stream-K rollout (mlp_persist.h), which runs them inside its own launch for the row block whose last tile it just finished -- same
// arithmetic in the same order, so the trajectories of both paths are bit for bit the same (tests/test_gpu_persist.py).
// Reference: samplers/vectorized_sampler.py:45-116 (policy.get_actions, vec_env.step), env_helpers.py:597-635 (step: clip, selection over the heads,
// reward, done, reset), training.py:228,146-151 (input normalisation, dropped columns), training.py:257 (de-normalise + residual).
#pragma once
#include "mfma_common.h"

struct BigState { float* S; int* ts; int* cur_model; float* X; float* U; float* HA; float* HB; float* OUT; float* PART;
                  int ldx;
                  // output layer left as split-K partials (gemm_skinny_bias with defer): k_big_post adds them and the bias
                  int out_splits; long long out_stride; const float* out_bias; long long out_bias_stride;
                  int out_ld;      // row stride of a partial (ns; 16 OT on the stream-K path, mlp_streamk.h)
                  int xone;        // 1: X[nin] = 1 -- the stream-K path's layer-0 producer takes the bias as one more input row
                  float* PA; float* PB;
                  float* PIMG; };   // k_big_pre_mfma3: the policy's fragment image, built once per launch chain (k_pre_mfma3_image)      // policy activations of the GEMM pre-path [B][max policy width]   // ldx: row stride of X = n_in rounded up to 4 floats, so every row is 16-byte aligned for the layer-0 GEMM's loads


// Step t - 1 closed for a wave's 16 envs in the lane layout of the MFMA pre-kernels (env c, quarter q: dims 16 hh + 4 q .. + 3 of every 16-dim block):
// what k_big_post computes, same arithmetic in the same order; the new state lands in the wave's LDS tile ST [16][NS] (and in S), ready for the policy chain.
// SC1P (the persistent rollout's closing workgroups): the output partials were written by workgroups of the SAME launch behind other L2s -- read them with
// agent-scope (sc1) loads, which a line this XCD's L2 still holds from the previous step's read cannot serve.
template <int ENV, int NS, int NA, bool SC1P = false>
__device__ __forceinline__ void big_close_step(const ProblemDesc& pd, const RolloutK& r, int t, const float* __restrict__ norm, const BigState& st, float* ST,
                                               int c, int q, bool active, int b, uint64_t genv, int* len_acc = nullptr) {
    constexpr int NH = (NS + 15) / 16;
    // ---- close step t - 1 (k_big_post) for this wave's 16 envs ----
    const int bc = active ? b : max(r.B - 1, 0);
    const int K = pd.K, tp = t - 1;
    const int ttp = tp + RK_TOFF(r, bc);
    const size_t tbp = (size_t)ttp * RK_STRIDE(r) + RK_ENV(r, bc);
    const float* diff_mean = norm + 2 * (NS + NA); const float* diff_std = diff_mean + NS;
    const uint4 dstep = rng_draw(r.seed, genv, r.t0 + ttp, RNG_STEP, 0);
    int sel = st.cur_model[bc];
    if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (r.model_idx != nullptr) ? r.model_idx[tbp] : rng_index(dstep.z, K);
    if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
    const bool simple = (r.sam_mode == METRPO_SAM_STEP_RAND || r.sam_mode == METRPO_SAM_EPS_RAND || r.sam_mode == METRPO_SAM_ONE_MODEL);
    float ua_[NA];                                           // clipped actions of step t - 1 (summed below, behind the loads of the output partials: one round trip for both)
#pragma unroll
    for (int d = 0; d < NA; ++d) ua_[d] = st.U[(size_t)bc * NA + d];
        float vnew[NH][4];
    bool finl = true;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
        const int i0 = 16 * hh + 4 * q;                      // dims i0 .. i0 + 3 (the partial rows are 16 OT floats wide, zero beyond ns)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) vnew[hh][rr] = 0.0f;
        if (i0 >= NS) continue;
        auto outv4 = [&](int k, float (&o)[4]) {             // output layer of head k, dims i0 .. i0 + 3: bias, then the partials in split order (k_big_post: outv)
            if (st.out_splits == 0) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) o[rr] = (i0 + rr < NS) ? st.OUT[((size_t)k * r.B + bc) * NS + i0 + rr] : 0.0f;
                return;
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) o[rr] = (i0 + rr < NS) ? st.out_bias[(size_t)k * st.out_bias_stride + i0 + rr] : 0.0f;
            for (int sp = 0; sp < st.out_splits; ++sp) {
                const float* pr_ = st.PART + ((size_t)sp * K + k) * st.out_stride + (size_t)bc * st.out_ld + i0;
                if constexpr (SC1P) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) if ((st.out_ld & 3) == 0 || i0 + rr < NS) o[rr] += __hip_atomic_load(const_cast<float*>(pr_) + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                else if ((st.out_ld & 3) == 0) { const f32x4 p4 = *(const f32x4*)pr_;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) o[rr] += p4[rr]; }
                else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) if (i0 + rr < NS) o[rr] += pr_[rr];
                }
            }
        };
        float so[4], dm_[4], ds_[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const int i = min(i0 + rr, NS - 1); so[rr] = st.S[(size_t)bc * NS + i]; dm_[rr] = diff_mean[i]; ds_[rr] = diff_std[i]; }
        auto head4 = [&](int k, float (&hv)[4]) { float o[4]; outv4(k, o);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) hv[rr] = fmaf(ds_[rr], o[rr], dm_[rr]) + so[rr]; };
        float v4[4];
        if (simple) head4(sel, v4);
        else {
            float m4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < K; ++k) { float h4[4]; head4(k, h4);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) m4[rr] += h4[rr]; }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { m4[rr] /= (float)K; v4[rr] = m4[rr]; }
            if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
                float var4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < K; ++k) { float h4[4]; head4(k, h4);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) { const float d_ = h4[rr] - m4[rr]; var4[rr] = fmaf(d_, d_, var4[rr]); } }
                float z4[4];
                if (r.sel_noise == nullptr) normal4(rng_draw(r.seed, genv, r.t0 + ttp, RNG_SELNOISE, i0 >> 2), z4);     // dims i0 .. i0 + 3 = chunk i0 / 4
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float nz = (r.sel_noise != nullptr) ? r.sel_noise[tbp * NS + min(i0 + rr, NS - 1)] : z4[rr];
                    v4[rr] = fmaf(nz, sqrtf(var4[rr] / (float)K), m4[rr]);
                }
            } else if (r.sam_mode == METRPO_SAM_MODEL_MED) {
                const int r_lo = (K - 1) / 2, r_hi = K / 2;
                float lo4[4] = {0.f, 0.f, 0.f, 0.f}, hi4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < K; ++k) {
                    float xk[4]; head4(k, xk);
                    int rank[4] = {0, 0, 0, 0};
                    for (int j = 0; j < K; ++j) { float xj[4]; head4(j, xj);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) rank[rr] += (xj[rr] < xk[rr]) || (xj[rr] == xk[rr] && j < k); }
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) { if (rank[rr] == r_lo) lo4[rr] = xk[rr]; if (rank[rr] == r_hi) hi4[rr] = xk[rr]; }
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) v4[rr] = 0.5f * (lo4[rr] + hi4[rr]);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) if (i0 + rr < NS) { vnew[hh][rr] = v4[rr]; finl = finl && isfinite(v4[rr]); ST[c * NS + i0 + rr] = v4[rr]; }
    }
    float su2 = 0.0f;                                        // sum of squared clipped actions in action order (every lane of the env, redundantly)
#pragma unroll
    for (int d = 0; d < NA; ++d) su2 = fmaf(ua_[d], ua_[d], su2);
    int fin = finl ? 1 : 0;                                  // all-finite over the env's dims: the env's four lanes are c, c + 16, c + 32, c + 48
    fin &= __shfl_xor(fin, 16, 64); fin &= __shfl_xor(fin, 32, 64);
    wave_lds_sync();
    const float* Sv = ST + c * NS;                           // the env's next state, every dim
    constexpr int ki = (ENV == METRPO_ENV_SWIMMER || ENV == METRPO_ENV_HOPPER) ? 5 : (ENV == METRPO_ENV_HALF_CHEETAH) ? 9 : (ENV == METRPO_ENV_ANT) ? 15 : (ENV == METRPO_ENV_SNAKE) ? 7 : (ENV == METRPO_ENV_HUMANOID) ? NS - 1 : 0;
    const float key = Sv[ki], h0v = Sv[0], h1v = Sv[1], zc = Sv[2];
    float pen = 0.0f;
    if (ENV == METRPO_ENV_HOPPER) for (int j = 2; j < NS; ++j) pen += fmaxf(fabsf(Sv[j]) - 100.0f, 0.0f);
    float cost = 0.0f;
    switch (ENV) {
    case METRPO_ENV_SWIMMER: cost = -(key - 1e-2f * (su2 / (float)NA)); break;
    case METRPO_ENV_HALF_CHEETAH: cost = -fminf(fmaxf(key - 1e-1f * 0.5f * su2, -10.0f), 10.0f); break;
    case METRPO_ENV_ANT: cost = -(key - 1e-2f * 0.5f * su2 + 0.05f); break;
    case METRPO_ENV_HOPPER: cost = -(key - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - h0v, 0.0f) - 10.0f * fmaxf(fabsf(h1v) - 0.2f, 0.0f) - pen); break;
    case METRPO_ENV_SNAKE: cost = -(key - 1e-2f * 0.5f * su2); break;
    case METRPO_ENV_HUMANOID: cost = (key - 1.5f) * (key - 1.5f) + 1e-2f * 1e-3f * su2; break;      // key = the last state dim (k_big_post: last)
    }
    int ts = st.ts[bc] + 1;
    bool dn = (ENV == METRPO_ENV_ANT) ? !((zc >= 0.2f) && (zc <= 1.0f) && (fin != 0)) : false;
    dn = dn || (ts >= r.H);
    int cur = st.cur_model[bc];
    if (active && q == 0) { r.rew[tbp] = -cost; r.done[tbp] = dn ? 1 : 0; r.tpath[tbp] = ts - 1; }
    if (len_acc != nullptr && active && q == 0 && dn) *len_acc += ts;      // samples of the path that just completed (the sampler's n_samples: vectorized_sampler.py:104)
    wave_lds_sync();                                         // every lane has read its env's scalars: the reset rows may overwrite the tile
    if (dn) {                                                // uniform over the env's four lanes
        const size_t rb = (size_t)(tp + 1) * r.B + bc;
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
        cur = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { const int i = 16 * hh + 4 * q + rr; if (i < NS) { vnew[hh][rr] = r.pool[(size_t)row * NS + i]; ST[c * NS + i] = vnew[hh][rr]; } }
        ts = 0;
    }
    if (active) {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { const int i = 16 * hh + 4 * q + rr; if (i < NS) st.S[(size_t)b * NS + i] = vnew[hh][rr]; }
        if (q == 0) { st.ts[b] = ts; if (dn) st.cur_model[b] = cur; }
    } else {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { const int i = 16 * hh + 4 * q + rr; if (i < NS) ST[c * NS + i] = 0.0f; }       // rows beyond the batch: zeros, as the reload below gives
    }
}


// ---- the pre-step of the 2 x 32 tanh policies (every shipped params file except Humanoid) as a transposed MFMA chain, one wave per 16-env tile ----
// LDS image of the policy (floats): W0 fragments [NS_KS][2][64] | W1 fragments [8][2][64] | Wout fragments [8][64] | b0 [32] | b1 [32] | bout [16]
template <int ENV> struct PreImg {
    using C = Cfg<ENV, 64, 32>;
    static constexpr int NS = C::NS, NA = C::NA, PH = 32, NS_KS = C::NS_KS;
    static constexpr int O_PF1 = NS_KS * 2 * 64, O_PF2 = O_PF1 + 16 * 64, O_B0 = O_PF2 + 8 * 64, O_B1 = O_B0 + 32, O_B2 = O_B1 + 32, IMG = O_B2 + 16;
    // value of image entry i (a gather from the flat policy vector, rllab order)
    static __device__ __forceinline__ float entry(const float* __restrict__ theta, int i) {
        float w = 0.0f;
        const int ln = i & 63, cc = ln & 15, qq = ln >> 4;
        if (i < O_PF1) { const int f = i >> 6, s_ = f >> 1, cb = f & 1, in = 4 * s_ + qq; if (in < NS) w = theta[C::pW0 + in * PH + 16 * cb + cc]; }
        else if (i < O_PF2) { const int f = (i - O_PF1) >> 6, kk = f >> 1, cb = f & 1; w = theta[C::pW1 + (16 * (kk >> 2) + 4 * qq + (kk & 3)) * PH + 16 * cb + cc]; }
        else if (i < O_B0) { const int kk = (i - O_PF2) >> 6; if (cc < NA) w = theta[C::pW2 + (16 * (kk >> 2) + 4 * qq + (kk & 3)) * NA + cc]; }
        else if (i < O_B1) w = theta[C::pb0 + (i - O_B0)];
        else if (i < O_B2) w = theta[C::pb1 + (i - O_B1)];
        else if (i < IMG) { const int d = i - O_B2; if (d < NA) w = theta[C::pb2 + d]; }
        return w;
    }
};

// what the END of a pre-step needs that depends on nothing computed in it: loaded / drawn up front, while the loads of the closing part are under way
template <int ENV> struct PreLane {
    static constexpr int NSQ = (Cfg<ENV, 64, 32>::NS + 3) / 4;
    float smn[NSQ], ssd[NSQ], amn[4], asd[4], lsd[4], zn[4];
};

// First half of a wave's pre-step for its 16 envs b0 .. b0 + 15 at step t: normaliser rows, log_std and the action noise of step t; then (POST) step t - 1
// closed (big_close_step: the new state lands in ST and S) or (!POST) the state tile loaded from S; obs[t] written.  ST: this wave's [16][NS] LDS tile.
template <int ENV, bool POST, bool SC1P = false>
__device__ __forceinline__ void big_pre_head(const ProblemDesc& pd, const RolloutK& r, int t, const float* __restrict__ theta, const float* __restrict__ norm,
                                             const BigState& st, float* ST, int b0, int lane, PreLane<ENV>& pl, int* len_acc = nullptr) {
    using C = Cfg<ENV, 64, 32>;
    constexpr int NS = C::NS, NA = C::NA, NSQ = PreLane<ENV>::NSQ;
    const int c = lane & 15, q = lane >> 4, b = b0 + c;
    const bool active = b < r.B;
    const uint64_t genv = r.stream_offset + (uint64_t)RK_ENV(r, b);
    const int tt = t + RK_TOFF(r, b);                            // row of the trajectory tensors / draw counter of this env's step
    const int lim = min(16, max(0, r.B - b0)) * NS;
    // Everything the END of this launch needs that depends on nothing computed here goes out NOW, with the loads of the post part: the normaliser
    // rows and log_std of this lane's dims (L2 hits, but a round trip of their own when first touched behind the policy chain) and the action
    // noise of step t (Philox + Box-Muller: ~150 vector instructions that run while the loads are under way).  Same values, same arithmetic.
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    const float* __restrict__ log_std = theta + C::pLS;
#pragma unroll
    for (int j = 0; j < NSQ; ++j) { const int i = min(q + 4 * j, NS - 1); pl.smn[j] = in_mean[i]; pl.ssd[j] = in_std[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int d = min(4 * q + j, NA - 1); pl.amn[j] = in_mean[NS + d]; pl.asd[j] = in_std[NS + d]; pl.lsd[j] = log_std[d]; pl.zn[j] = 0.0f; }
    if (!r.determ && r.eps == nullptr) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d0 = 4 * q + 2 * h;
            if (d0 >= NA) continue;
            const uint4 blk = rng_draw(r.seed, genv, r.t0 + tt, RNG_STEP, d0 >> 1);
            normal2(blk.x, blk.y, pl.zn[2 * h], pl.zn[2 * h + 1]);
        }
    }
    if constexpr (POST) {
        big_close_step<ENV, NS, NA, SC1P>(pd, r, t, norm, st, ST, c, q, active, b, genv, len_acc);
    } else {
        for (int i = lane; i < 16 * NS; i += 64) ST[i] = (i < lim) ? st.S[(size_t)b0 * NS + i] : 0.0f;
    }
    wave_lds_sync();
    if (lim > 0) {
        if (r.vB == 0) { const size_t base = ((size_t)t * r.B + b0) * NS; for (int i = lane; i < lim; i += 64) r.obs[base + i] = ST[i]; }
        else for (int i = lane; i < lim; i += 64) {              // merged rounds: a tile's envs may belong to two rounds
            const int bi = b0 + i / NS;
            r.obs[((size_t)(t + RK_TOFF(r, bi)) * r.vB + RK_ENV(r, bi)) * NS + i % NS] = ST[i];
        }
    }
}

// store of one element of the normalised input rows X.  SC1 (the persistent rollout): an agent-scope (write-through) store -- X is read by workgroups
// of the same launch on other XCDs with agent-scope loads (MI355X_MICROARCH.md, inter-workgroup visibility); otherwise a plain store.
template <bool SC1> __device__ __forceinline__ void store_x(float* p, float v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// Second half: the policy chain on the state tile ST (30 matrix instructions; img = the PreImg<ENV> image in LDS), a = mean + sigma z, clip, act / mean / U and
// the normalised, dropped input row X of step t.
template <int ENV, bool SC1X>
__device__ __forceinline__ void big_pre_tail(const RolloutK& r, int t, const BigState& st, const float* img, const float* ST, int b0, int lane, const PreLane<ENV>& pl) {
    using C = Cfg<ENV, 64, 32>;
    using IM = PreImg<ENV>;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NS_KS = C::NS_KS, NSQ = PreLane<ENV>::NSQ;
    constexpr int O_PF1 = IM::O_PF1, O_PF2 = IM::O_PF2, O_B0 = IM::O_B0, O_B1 = IM::O_B1, O_B2 = IM::O_B2;
    const int c = lane & 15, q = lane >> 4, b = b0 + c;
    const bool active = b < r.B;
    const int tt = t + RK_TOFF(r, b);
    f32x4 p0[2], p1[2];
    p0[0] = *(const f32x4*)&img[O_B0 + 4 * q]; p0[1] = *(const f32x4*)&img[O_B0 + 16 + 4 * q];
#pragma unroll
    for (int s_ = 0; s_ < NS_KS; ++s_) {
        const int f = 4 * s_ + q;
        const float x = (f < NS) ? ST[c * NS + f] : 0.0f;
        p0[0] = MFMA16(img[(s_ * 2 + 0) * 64 + lane], x, p0[0]);
        p0[1] = MFMA16(img[(s_ * 2 + 1) * 64 + lane], x, p0[1]);
    }
    p1[0] = *(const f32x4*)&img[O_B1 + 4 * q]; p1[1] = *(const f32x4*)&img[O_B1 + 16 + 4 * q];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) p0[cb][rr] = tanh_fast(p0[cb][rr]);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        p1[0] = MFMA16(img[O_PF1 + (kk * 2 + 0) * 64 + lane], p0[kk >> 2][kk & 3], p1[0]);
        p1[1] = MFMA16(img[O_PF1 + (kk * 2 + 1) * 64 + lane], p0[kk >> 2][kk & 3], p1[1]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) p1[cb][rr] = tanh_fast(p1[cb][rr]);
    f32x4 m0 = *(const f32x4*)&img[O_B2 + 4 * q], m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
        m0 = MFMA16(img[O_PF2 + kk * 64 + lane], p1[kk >> 2][kk & 3], m0);
        m1 = MFMA16(img[O_PF2 + (kk + 1) * 64 + lane], p1[(kk + 1) >> 2][(kk + 1) & 3], m1);
    }
    const f32x4 mu = m0 + m1;
    if (!active) return;
    const size_t tb = (size_t)tt * RK_STRIDE(r) + RK_ENV(r, b);
    // state part of the normalised, dropped input: lane (c, q) writes its dims q, q + 4, ...
#pragma unroll
    for (int j = 0; j < NSQ; ++j) { const int i = q + 4 * j; if (i < NS && i >= NDROP) store_x<SC1X>(&st.X[(size_t)b * st.ldx + i - NDROP], (ST[c * NS + i] - pl.smn[j]) / pl.ssd[j]); }     // training.py:228,146-151
    if (q == 0) for (int j = C::NIN; j < st.ldx; ++j) store_x<SC1X>(&st.X[(size_t)b * st.ldx + j], (st.xone && j == C::NIN) ? 1.0f : 0.0f);   // pad columns of the 16-byte aligned rows (layer 0 contracts over ldx)
    // action dims 4q .. 4q+3 of this lane = Philox chunks 2q, 2q+1 (chunk 0 = the step block), exactly as k_big_pre
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int d0 = 4 * q + 2 * h;
        if (d0 >= NA) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = d0 + j;
            if (d >= NA) continue;
            const float m = mu[2 * h + j];
            float a = m;
            if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * NA + d] : pl.zn[2 * h + j], __expf(fmaxf(pl.lsd[2 * h + j], LOG_MIN_STD)), m);
            r.act[tb * NA + d] = a; r.mean[tb * NA + d] = m;
            const float ac = fminf(fmaxf(a, -1.0f), 1.0f);          // env_helpers.py:599
            st.U[(size_t)b * NA + d] = ac;
            store_x<SC1X>(&st.X[(size_t)b * st.ldx + (NS - NDROP) + d], (ac - pl.amn[2 * h + j]) / pl.asd[2 * h + j]);
        }
    }
}
