// MFMA fast path of the fused imagined rollout (VectorizedSampler.obtain_samples, samplers/
// vectorized_sampler.py:45-116 over VecSimpleEnv.step, env_helpers.py:597-635) for small MLPs
// (2 hidden layers, widths <= 64): C1 / the BASELINE.json headline shape.
//
// Mapping to CDNA4
//   * one workgroup = one tile of 16 imagined envs; wave w of the workgroup owns dynamics head w
//     (K <= 8 waves).  Every wave also evaluates the (tiny) policy redundantly, so the only
//     cross-wave traffic is the K head outputs, exchanged through LDS with ONE barrier per step.
//   * every layer is computed TRANSPOSED on the f32 matrix core, H^T[unit][env] = W^T . X^T, with
//     v_mfma_f32_16x16x4_f32: A = weights (lane l holds W[in = 4s + (l>>4)][out = 16cb + (l&15)]),
//     B = activations (lane l holds x[in = 4s + (l>>4)] of env l&15).  The D fragment of layer n
//     (lane l: units 16cb + 4(l>>4) + r, env l&15) IS the B operand of layer n+1 when that layer's
//     k-steps are enumerated as (cb, r) and its weight fragment is permuted to match -- activations
//     never leave registers between layers.  f32 MFMA is an exact fmaf chain (guide section 3), so the
//     numerics are plain fp32.
//   * all weights of the wave's head + the policy are register-resident for the whole rollout
//     (~120 VGPRs); biases live in LDS and enter as the MFMA C operand.
//   * state of the tile is kept per wave in LDS in [env][ns] order == the global layout of one
//     time step of the trajectory, so the obs store is a fully coalesced linear copy.
#include "mfma_common.h"
#include <cstdlib>

template <int ENV, int DH, int PH>
__global__ void __launch_bounds__(512) k_rollout_mfma(RolloutK r, int K, const float* __restrict__ dynp,
                                                      const float* __restrict__ theta, const float* __restrict__ norm) {
    using C = Cfg<ENV, DH, PH>;
    constexpr int NS = C::NS, NA = C::NA, NSP = C::NSP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = lane & 15, q = lane >> 4;
    const int b0 = blockIdx.x * 16, b = b0 + e;
    const bool active = b < r.B;
    const uint64_t genv = r.stream_offset + (uint64_t)b;
    if (r.stop != nullptr && *r.stop != 0) return;           // the sampling loop already ended (metrpo_sampler_progress)
    float* W = lds + wave * C::W_TOTAL;                                  // this wave's private region
    float* ST = W + C::W_ST;  float* NX = W + C::W_NX;  float* ACT = W + C::W_ACT;
    float* NXT = lds + K * C::W_TOTAL;                                   // [2][K][16][NSP] exchange buffer

    // ---------------- one-time: weight fragments -> registers, biases -> LDS ----------------------
    const float* __restrict__ pk = dynp + (size_t)wave * C::PD;
    float wd0[C::NIN_KS][C::DH_CB], wd1[C::DH_CB * 4][C::DH_CB], wd2[C::DH_CB * 4][C::OUT_CB];
#pragma unroll
    for (int s = 0; s < C::NIN_KS; ++s)
#pragma unroll
        for (int cb = 0; cb < C::DH_CB; ++cb) {
            const int i = 4 * s + q, o = 16 * cb + e;
            wd0[s][cb] = (i < C::NIN && o < DH) ? pk[C::dW0 + i * DH + o] : 0.0f;
        }
#pragma unroll
    for (int kk = 0; kk < C::DH_CB * 4; ++kk) {
        const int i = 16 * (kk >> 2) + 4 * q + (kk & 3);
#pragma unroll
        for (int cb = 0; cb < C::DH_CB; ++cb) {
            const int o = 16 * cb + e;
            wd1[kk][cb] = (i < DH && o < DH) ? pk[C::dW1 + i * DH + o] : 0.0f;
        }
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb) {
            const int o = 16 * cb + e;
            wd2[kk][cb] = (i < DH && o < NS) ? pk[C::dW2 + i * NS + o] : 0.0f;
        }
    }
    float wp0[C::NS_KS][C::PH_CB], wp1[C::PH_CB * 4][C::PH_CB], wp2[C::PH_CB * 4];
#pragma unroll
    for (int s = 0; s < C::NS_KS; ++s)
#pragma unroll
        for (int cb = 0; cb < C::PH_CB; ++cb) {
            const int i = 4 * s + q, o = 16 * cb + e;
            wp0[s][cb] = (i < NS && o < PH) ? theta[C::pW0 + i * PH + o] : 0.0f;
        }
#pragma unroll
    for (int kk = 0; kk < C::PH_CB * 4; ++kk) {
        const int i = 16 * (kk >> 2) + 4 * q + (kk & 3);
#pragma unroll
        for (int cb = 0; cb < C::PH_CB; ++cb) {
            const int o = 16 * cb + e;
            wp1[kk][cb] = (i < PH && o < PH) ? theta[C::pW1 + i * PH + o] : 0.0f;
        }
        wp2[kk] = (i < PH && e < NA) ? theta[C::pW2 + i * NA + e] : 0.0f;
    }
    for (int i = lane; i < C::BD; i += 64) {
        W[C::W_BD0 + i] = (i < DH) ? pk[C::db0 + i] : 0.0f;
        W[C::W_BD1 + i] = (i < DH) ? pk[C::db1 + i] : 0.0f;
    }
    for (int i = lane; i < NSP; i += 64) W[C::W_BD2 + i] = (i < NS) ? pk[C::db2 + i] : 0.0f;
    for (int i = lane; i < C::BP; i += 64) {
        W[C::W_BP0 + i] = (i < PH) ? theta[C::pb0 + i] : 0.0f;
        W[C::W_BP1 + i] = (i < PH) ? theta[C::pb1 + i] : 0.0f;
    }
    if (lane < 16) W[C::W_BP2 + lane] = (lane < NA) ? theta[C::pb2 + lane] : 0.0f;
    // per-lane constants: policy sigma for its 4 action dims, input normalisers for its k-step features
    float sig[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sig[rr] = (4 * q + rr < NA) ? expf(fmaxf(theta[C::pLS + 4 * q + rr], LOG_MIN_STD)) : 0.0f;
    float nmean[C::NIN_KS], nstd[C::NIN_KS];
    int nsrc[C::NIN_KS];                                   // >= 0: state feature, < 0: -(action dim + 1), INT_MIN: padding
#pragma unroll
    for (int s = 0; s < C::NIN_KS; ++s) {
        const int i = 4 * s + q;
        int f = 0;
        if (i < NS - C::NDROP) { f = i + C::NDROP; nsrc[s] = f; }
        else if (i < C::NIN) { f = NS + (i - (NS - C::NDROP)); nsrc[s] = -(i - (NS - C::NDROP)) - 1; }
        else { nsrc[s] = -1000000; }
        nmean[s] = (i < C::NIN) ? norm[f] : 0.0f;
        nstd[s] = (i < C::NIN) ? 1.0f / norm[(NS + NA) + f] : 1.0f;   // reciprocal: (x - mean) * (1/std), <= 1 ulp from the division
    }
    f32x4 dmean[C::OUT_CB], dstd[C::OUT_CB];
#pragma unroll
    for (int cb = 0; cb < C::OUT_CB; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int dim = 16 * cb + 4 * q + rr;
            dmean[cb][rr] = (dim < NS) ? norm[2 * (NS + NA) + dim] : 0.0f;
            dstd[cb][rr] = (dim < NS) ? norm[2 * (NS + NA) + NS + dim] : 0.0f;
        }

    // ---------------- vec_env.reset(): initial state + cur_model_idx (env_helpers.py:585-595) ------
    const bool resume = r.init_obs != nullptr;
    int cur_model = 0, ts = 0;
    {
        int row = 0;
        if (active && !resume) {
            const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
            row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
            cur_model = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, K);
        }
        if (active && resume) { cur_model = r.init_model[b]; ts = r.init_ts[b]; }      // continuation of a chunked rollout
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                if (dim < NS) ST[e * NS + dim] = resume ? r.init_obs[(size_t)(active ? b : 0) * NS + dim] : r.pool[(size_t)row * NS + dim];
            }
    }
    wave_lds_sync();

    for (int t = 0; t < r.T; ++t) {
        const size_t tb = (size_t)t * r.B + b;
        // ---- obs[t] = state before the step: linear, fully coalesced copy of the tile (wave 0) -----
        if (wave == 0) {
            const size_t base = ((size_t)t * r.B + b0) * NS;
            const int lim = min(16, r.B - b0) * NS;
            for (int i = lane; i < lim; i += 64) r.obs[base + i] = ST[i];
        }
        // ---- policy.get_actions: mean = MLP(s), a = mean + exp(log_std) * eps ----------------------
        f32x4 p0[C::PH_CB], p1[C::PH_CB];
#pragma unroll
        for (int cb = 0; cb < C::PH_CB; ++cb) p0[cb] = *(const f32x4*)&W[C::W_BP0 + 16 * cb + 4 * q];
#pragma unroll
        for (int s = 0; s < C::NS_KS; ++s) {
            const int f = 4 * s + q;
            const float x = (f < NS) ? ST[e * NS + f] : 0.0f;
#pragma unroll
            for (int cb = 0; cb < C::PH_CB; ++cb) p0[cb] = MFMA16(wp0[s][cb], x, p0[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < C::PH_CB; ++cb) {
            p1[cb] = *(const f32x4*)&W[C::W_BP1 + 16 * cb + 4 * q];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p0[cb][rr] = tanh_fast(p0[cb][rr]);
        }
#pragma unroll
        for (int kk = 0; kk < C::PH_CB * 4; ++kk)
#pragma unroll
            for (int cb = 0; cb < C::PH_CB; ++cb) p1[cb] = MFMA16(wp1[kk][cb], p0[kk >> 2][kk & 3], p1[cb]);
#pragma unroll
        for (int cb = 0; cb < C::PH_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p1[cb][rr] = tanh_fast(p1[cb][rr]);
        f32x4 m0 = *(const f32x4*)&W[C::W_BP2 + 4 * q], m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < C::PH_CB * 4; kk += 2) {              // two accumulators: 40-cycle dependent latency
            m0 = MFMA16(wp2[kk], p1[kk >> 2][kk & 3], m0);
            m1 = MFMA16(wp2[kk + 1], p1[(kk + 1) >> 2][(kk + 1) & 3], m1);
        }
        const f32x4 mu = m0 + m1;
        const uint4 dstep = rng_draw(r.seed, genv, r.t0 + t, RNG_STEP, 0);
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (!r.determ && r.eps == nullptr) {            // lane q owns action dims 4q..4q+3 = chunks 2q, 2q+1 (chunk 0 = dstep)
            const uint4 b0k = (q == 0) ? dstep : ((NA > 4) ? rng_draw(r.seed, genv, r.t0 + t, RNG_STEP, 2 * q) : dstep);
            normal2(b0k.x, b0k.y, z[0], z[1]);
            if (NA > 2) { const uint4 b1k = rng_draw(r.seed, genv, r.t0 + t, RNG_STEP, 2 * q + 1); normal2(b1k.x, b1k.y, z[2], z[3]); }
        }
        float su2 = 0.0f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int d = 4 * q + rr;
            if (d < NA) {
                float a = mu[rr];
                if (!r.determ) {
                    const float zz = (r.eps != nullptr) ? (active ? r.eps[tb * NA + d] : 0.0f) : z[rr];
                    a = fmaf(zz, sig[rr], a);
                }
                if (wave == 0 && active) { r.act[tb * NA + d] = a; r.mean[tb * NA + d] = mu[rr]; }
                const float ac = fminf(fmaxf(a, -1.0f), 1.0f);        // np.clip(actions, *bounds), env_helpers.py:599
                ACT[e * NA + d] = ac;
                su2 = fmaf(ac, ac, su2);
            }
        }
        su2 = xor_sum(su2);
        wave_lds_sync();
        // ---- dynamics head `wave`: normalise, drop columns, 3 layers (training.py:218-269) ----------
        f32x4 h0[C::DH_CB], h1[C::DH_CB];
#pragma unroll
        for (int cb = 0; cb < C::DH_CB; ++cb) h0[cb] = *(const f32x4*)&W[C::W_BD0 + 16 * cb + 4 * q];
#pragma unroll
        for (int s = 0; s < C::NIN_KS; ++s) {
            float x = 0.0f;
            if (nsrc[s] >= 0) x = ST[e * NS + nsrc[s]];
            else if (nsrc[s] > -1000000) x = ACT[e * NA + (-nsrc[s] - 1)];
            x = (nsrc[s] > -1000000) ? (x - nmean[s]) * nstd[s] : 0.0f;      // (xgu - in_mean)/in_std, training.py:228
#pragma unroll
            for (int cb = 0; cb < C::DH_CB; ++cb) h0[cb] = MFMA16(wd0[s][cb], x, h0[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < C::DH_CB; ++cb) {
            h1[cb] = *(const f32x4*)&W[C::W_BD1 + 16 * cb + 4 * q];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) h0[cb][rr] = fmaxf(h0[cb][rr], 0.0f);
        }
#pragma unroll
        for (int kk = 0; kk < C::DH_CB * 4; ++kk)
#pragma unroll
            for (int cb = 0; cb < C::DH_CB; ++cb) h1[cb] = MFMA16(wd1[kk][cb], h0[kk >> 2][kk & 3], h1[cb]);
#pragma unroll
        for (int cb = 0; cb < C::DH_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) h1[cb][rr] = fmaxf(h1[cb][rr], 0.0f);
        f32x4 oa[C::OUT_CB], ob[C::OUT_CB];
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb) { oa[cb] = *(const f32x4*)&W[C::W_BD2 + 16 * cb + 4 * q]; ob[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kk = 0; kk < C::DH_CB * 4; kk += 2)
#pragma unroll
            for (int cb = 0; cb < C::OUT_CB; ++cb) {
                oa[cb] = MFMA16(wd2[kk][cb], h1[kk >> 2][kk & 3], oa[cb]);
                ob[cb] = MFMA16(wd2[kk + 1][cb], h1[(kk + 1) >> 2][(kk + 1) & 3], ob[cb]);
            }
        const int par = t & 1;
        float* nxt_w = NXT + ((size_t)(par * K + wave) * 16 + e) * NSP;
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb) {
            f32x4 o = oa[cb] + ob[cb], sv;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                sv[rr] = (dim < NS) ? ST[e * NS + dim] : 0.0f;
                o[rr] = fmaf(dstd[cb][rr], o[rr], dmean[cb][rr]) + sv[rr];      // diff_mean + diff_std*out + s, training.py:257
            }
            *(f32x4*)&nxt_w[16 * cb + 4 * q] = o;
        }
        __syncthreads();                                                     // all K heads of step t are in NXT[par]
        // ---- get_next_observation selection (env_helpers.py:617-634), redundantly in every wave -----
        ts += 1;
        int sel = cur_model;
        if (r.sam_mode == METRPO_SAM_STEP_RAND)
            sel = (r.model_idx != nullptr) ? (active ? r.model_idx[tb] : 0) : rng_index(dstep.z, K);
        if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
        const float* nxt_all = NXT + (size_t)par * K * 16 * NSP;
        f32x4 nx[C::OUT_CB];
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb) {
            const int off = e * NSP + 16 * cb + 4 * q;
            if (r.sam_mode == METRPO_SAM_STEP_RAND || r.sam_mode == METRPO_SAM_EPS_RAND || r.sam_mode == METRPO_SAM_ONE_MODEL) {
                nx[cb] = *(const f32x4*)&nxt_all[(size_t)sel * 16 * NSP + off];
            } else {
                f32x4 m = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < K; ++k) m += *(const f32x4*)&nxt_all[(size_t)k * 16 * NSP + off];
                m /= (float)K;
                if (r.sam_mode == METRPO_SAM_MODEL_MEAN) nx[cb] = m;
                else if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
                    f32x4 var = {0.f, 0.f, 0.f, 0.f};
                    for (int k = 0; k < K; ++k) { const f32x4 d = *(const f32x4*)&nxt_all[(size_t)k * 16 * NSP + off] - m; var += d * d; }
                    float zz[4] = {0.f, 0.f, 0.f, 0.f};
                    if (r.sel_noise == nullptr) normal4(rng_draw(r.seed, genv, r.t0 + t, RNG_SELNOISE, 4 * cb + q), zz);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int dim = 16 * cb + 4 * q + rr;
                        const float nz = (r.sel_noise != nullptr) ? ((active && dim < NS) ? r.sel_noise[tb * NS + dim] : 0.0f) : zz[rr];
                        nx[cb][rr] = fmaf(nz, sqrtf(var[rr] / (float)K), m[rr]);
                    }
                } else {                                                      // model_med: np.median over K
                    const int r_lo = (K - 1) / 2, r_hi = K / 2;
                    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
                    for (int k = 0; k < K; ++k) {
                        const f32x4 xk = *(const f32x4*)&nxt_all[(size_t)k * 16 * NSP + off];
                        int rank[4] = {0, 0, 0, 0};
                        for (int j = 0; j < K; ++j) {
                            const f32x4 xj = *(const f32x4*)&nxt_all[(size_t)j * 16 * NSP + off];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) rank[rr] += (xj[rr] < xk[rr]) || (xj[rr] == xk[rr] && j < k);
                        }
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) { if (rank[rr] == r_lo) lo[rr] = xk[rr]; if (rank[rr] == r_hi) hi[rr] = xk[rr]; }
                    }
                    nx[cb] = (lo + hi) * 0.5f;
                }
            }
        }
        // ---- reward = -cost_np_vec(s, a_clipped, s') (:601) and is_done (:603) ----------------------
        float pen = 0.0f; int fin = 1;
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                if (dim < NS) {
                    NX[e * NS + dim] = nx[cb][rr];
                    if (ENV == METRPO_ENV_HOPPER && dim >= 2) pen += fmaxf(fabsf(nx[cb][rr]) - 100.0f, 0.0f);
                    if (ENV == METRPO_ENV_ANT) fin &= isfinite(nx[cb][rr]) ? 1 : 0;
                }
            }
        wave_lds_sync();
        const float* xn = NX + e * NS;
        float cost = 0.0f;
        bool dn = false;
        if (ENV == METRPO_ENV_SWIMMER) cost = -(xn[5] - 1e-2f * (su2 / (float)NA));
        else if (ENV == METRPO_ENV_HALF_CHEETAH) cost = -fminf(fmaxf(xn[9] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
        else if (ENV == METRPO_ENV_SNAKE) cost = -(xn[7] - 1e-2f * 0.5f * su2);
        else if (ENV == METRPO_ENV_HOPPER) {
            pen = xor_sum(pen);
            cost = -(xn[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0], 0.0f) - 10.0f * fmaxf(fabsf(xn[1]) - 0.2f, 0.0f) - pen);
        } else if (ENV == METRPO_ENV_ANT) {
            cost = -(xn[15] - 1e-2f * 0.5f * su2 + 0.05f);
            int f2 = fin & __shfl_xor(fin, 16, 64);
            f2 &= __shfl_xor(f2, 32, 64);
            const float zc = xn[2];
            dn = !((zc >= 0.2f) && (zc <= 1.0f) && (f2 != 0));
        }
        dn = dn || (ts >= r.H);                                               // :604
        if (wave == 0 && q == 0 && active) { r.rew[tb] = -cost; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts - 1; }
        // ---- reset(dones) (:585-595) or advance ----------------------------------------------------
        int row = 0;
        if (dn) {
            if (active) {
                const size_t rb = (size_t)(t + 1) * r.B + b;
                row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
                cur_model = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
            }
            ts = 0;
        }
#pragma unroll
        for (int cb = 0; cb < C::OUT_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                if (dim < NS) ST[e * NS + dim] = dn ? r.pool[(size_t)row * NS + dim] : nx[cb][rr];
            }
        wave_lds_sync();
    }
    if (wave == 0 && r.last_obs != nullptr) {
        const int lim = min(16, r.B - b0) * NS;
        for (int i = lane; i < lim; i += 64) r.last_obs[(size_t)b0 * NS + i] = ST[i];
    }
    if (wave == 0 && q == 0 && active) {
        if (r.last_ts != nullptr) r.last_ts[b] = ts;
        if (r.last_model != nullptr) r.last_model[b] = cur_model;
    }
}

// -------------------------------------------------------------------------------------------------
typedef void (*mfma_kernel_t)(RolloutK, int, const float*, const float*, const float*);
struct MfmaEntry { int env, dh, ph; mfma_kernel_t kern; int w_total, nsp; };

#define ENTRY(ENVID, DH, PH) {ENVID, DH, PH, k_rollout_mfma<ENVID, DH, PH>, Cfg<ENVID, DH, PH>::W_TOTAL, Cfg<ENVID, DH, PH>::NSP}
static const MfmaEntry kTable[] = {
    ENTRY(METRPO_ENV_SWIMMER, 64, 32),
    ENTRY(METRPO_ENV_HALF_CHEETAH, 64, 32),
    ENTRY(METRPO_ENV_HOPPER, 64, 32),
    ENTRY(METRPO_ENV_SNAKE, 64, 32),
    ENTRY(METRPO_ENV_ANT, 64, 32),
    ENTRY(METRPO_ENV_SWIMMER, 32, 32),
};

// index into kTable of the fused kernels' shape class (env dims, hidden widths, activations), whatever K is; -1: none
int mfma_shape_config(const metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    if (pd.dyn.n_layers != 3 || pd.pol.n_layers != 3) return -1;
    if (pd.dyn.act[0] != METRPO_ACT_RELU || pd.dyn.act[1] != METRPO_ACT_RELU) return -1;
    if (pd.dyn.dims[1] != pd.dyn.dims[2] || pd.pol.dims[1] != pd.pol.dims[2]) return -1;
    const int n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    for (int i = 0; i < n; ++i) {
        const MfmaEntry& en = kTable[i];
        if (en.env != pd.env || en.dh != pd.dyn.dims[1] || en.ph != pd.pol.dims[1]) continue;
        // the template's env dims must be the ctx dims (custom ns/na/n_drop -> generic path)
        bool ok = false;
        switch (pd.env) {
        case METRPO_ENV_SWIMMER: ok = pd.ns == 10 && pd.na == 2 && pd.n_drop == 2; break;
        case METRPO_ENV_HALF_CHEETAH: ok = pd.ns == 18 && pd.na == 6 && pd.n_drop == 1; break;
        case METRPO_ENV_HOPPER: ok = pd.ns == 11 && pd.na == 3 && pd.n_drop == 0; break;
        case METRPO_ENV_SNAKE: ok = pd.ns == 14 && pd.na == 4 && pd.n_drop == 2; break;
        case METRPO_ENV_ANT: ok = pd.ns == 29 && pd.na == 8 && pd.n_drop == 2; break;
        }
        if (ok) return i;
    }
    return -1;
}
// the head-per-wave kernel: a wave per head, at most 8 (its workgroup is <= 512 threads)
int mfma_select_config(metrpo_ctx* c) { return c->pd.K > 8 ? -1 : mfma_shape_config(c); }

// weights are read straight from the flat parameter buffers at kernel start: nothing to prepare
int mfma_prepare_dynamics(metrpo_ctx*, hipStream_t) { return METRPO_OK; }
int mfma_prepare_policy(metrpo_ctx*, hipStream_t) { return METRPO_OK; }

// *coop (out, optional): 1 when the cooperative kernel took the launch
int launch_rollout_mfma(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st, int* coop) {
    if (coop) *coop = 0;
    RolloutK r = make_rollout_k(a);
    if (c->coop_cfg >= 0 && c->rollout_variant != 1) {
        const int rc = launch_rollout_coop(c, c->coop_cfg, r, st);           // METRPO_EUNSUPPORTED: more than 5 heads without a CU of their own per workgroup (below)
        if (rc != METRPO_EUNSUPPORTED) { if (coop) *coop = 1; return rc; }
    } else if (c->coop_pad_cfg >= 0 && c->rollout_variant == 0) {            // two hidden layers narrower than 64: the same kernel on the zero-padded weights
        const int rc = launch_rollout_coop(c, c->coop_pad_cfg, r, st, true);
        if (rc != METRPO_EUNSUPPORTED) { if (coop) *coop = 1; return rc; }
    }
    if (c->mfma_cfg < 0) return METRPO_EUNSUPPORTED;
    const MfmaEntry& en = kTable[c->mfma_cfg];
    const int K = c->pd.K;
    size_t sh = sizeof(float) * ((size_t)K * en.w_total + 2 * (size_t)K * 16 * en.nsp);
    const int grid = (a->B + 15) / 16;
    hipLaunchKernelGGL(en.kern, dim3(grid), dim3(K * 64), sh, st, r, K, c->d_dyn, c->d_theta, c->d_norm);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
