// Generic (any layer widths that fit LDS) VALU kernels for the imagined environment:
//   k_policy_actions  -- [rllab] GaussianMLPPolicy.get_actions        (samplers/vectorized_sampler.py:63)
//   k_step            -- VecSimpleEnv.step arithmetic                  (env_helpers.py:597-635)
//   k_rollout_generic -- whole obtain_samples loop fused over T steps  (samplers/vectorized_sampler.py:45-116)
//   k_validation      -- build_policy_graph forward                    (model_based_rl.py:106-151)
// One thread = one imagined env; its activations live in LDS columns (device_common.h); there is
// no cross-thread communication, so the time loop runs inside the kernel with no barriers.
// The MFMA fast path for small nets is rollout_mfma.hip; this file is the reference-shaped
// fallback for every other configuration and the on-GPU cross-check for the fast path.
#include "device_common.h"

struct EnvBufs {        // column buffers of one thread block, all [rows][LD]
    float* S;           // ns      current state
    float* X;           // ns+na   normalised (s, clipped a)
    float* U;           // na      clipped action
    float* A;           // max_width ping
    float* Bq;          // max_width pong
    float* NXT;         // slots*ns head outputs (slots = K for averaging modes, else 1)
    float* NEXT;        // ns      selected next state
};

__host__ __device__ inline size_t envbufs_floats(const ProblemDesc& pd, int sam_mode, int LD) {
    const int slots = (sam_mode == METRPO_SAM_MODEL_MEAN_STD || sam_mode == METRPO_SAM_MODEL_MEAN ||
                       sam_mode == METRPO_SAM_MODEL_MED) ? pd.K : 1;
    const int mw = max(max(pd.dyn.max_width, pd.pol.max_width), pd.ns + pd.na);
    return (size_t)LD * (pd.ns + (pd.ns + pd.na) + pd.na + 2 * mw + slots * pd.ns + pd.ns);
}

__device__ __forceinline__ EnvBufs envbufs_carve(float* lds, const ProblemDesc& pd, int sam_mode, int LD) {
    const int slots = (sam_mode == METRPO_SAM_MODEL_MEAN_STD || sam_mode == METRPO_SAM_MODEL_MEAN ||
                       sam_mode == METRPO_SAM_MODEL_MED) ? pd.K : 1;
    const int mw = max(max(pd.dyn.max_width, pd.pol.max_width), pd.ns + pd.na);
    EnvBufs e;
    e.S = lds;
    e.X = e.S + pd.ns * LD;
    e.U = e.X + (pd.ns + pd.na) * LD;
    e.A = e.U + pd.na * LD;
    e.Bq = e.A + mw * LD;
    e.NXT = e.Bq + mw * LD;
    e.NEXT = e.NXT + slots * pd.ns * LD;
    return e;
}

// Evaluate the dynamics heads on (S, U) and select per sam_mode into e.NEXT.
//   training.py:218-269 forward per head; env_helpers.py:617-634 selection.
// sel: model index for step_rand / eps_rand (ignored otherwise); noise: per-dim N(0,1) draws in
// column layout for model_mean_std.  next_all (optional, global [K][B][ns]) receives every head.
__device__ __forceinline__ void dyn_heads_select(const ProblemDesc& pd, const float* __restrict__ dynp,
                                                 const float* __restrict__ norm, const EnvBufs& e, int sam_mode,
                                                 int sel, const float* noise_col, bool eval_all,
                                                 float* next_all, int B, int b, bool active, int LD, int tid) {
    const int ns = pd.ns, na = pd.na, K = pd.K;
    const float* in_mean = norm;
    const float* in_std = norm + (ns + na);
    const float* diff_mean = norm + 2 * (ns + na);
    const float* diff_std = diff_mean + ns;
    // xgu_norm = (xgu - in_mean) / in_std   (training.py:228)
    for (int i = 0; i < ns; ++i) e.X[i * LD + tid] = (e.S[i * LD + tid] - in_mean[i]) / in_std[i];
    for (int d = 0; d < na; ++d) e.X[(ns + d) * LD + tid] = (e.U[d * LD + tid] - in_mean[ns + d]) / in_std[ns + d];
    const bool averaging = (sam_mode == METRPO_SAM_MODEL_MEAN_STD || sam_mode == METRPO_SAM_MODEL_MEAN ||
                            sam_mode == METRPO_SAM_MODEL_MED);
    if (sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
    for (int k = 0; k < K; ++k) {
        const bool mine = (sel == k);
        if (!averaging && !eval_all && next_all == nullptr) {
            if (__ballot(mine && active) == 0ull) continue;      // no lane of this wave needs head k
        }
        const float* __restrict__ pk = dynp + (size_t)k * pd.dyn.n_params;
        // drop the leading n_drop normalised columns (training.py:146-151): pointer offset in column layout
        float* out = mlp_col(pd.dyn, pk, e.X + pd.n_drop * LD, e.A, e.Bq, LD, tid);
        float* dst = averaging ? (e.NXT + k * ns * LD) : e.NXT;
        for (int i = 0; i < ns; ++i) {
            // next = diff_mean + diff_std * out + s    (training.py:257)
            const float v = fmaf(diff_std[i], out[i * LD + tid], diff_mean[i]) + e.S[i * LD + tid];
            if (averaging || mine) dst[i * LD + tid] = v;
            if (next_all != nullptr && active) next_all[((size_t)k * B + b) * ns + i] = v;
        }
    }
    if (!averaging) {
        for (int i = 0; i < ns; ++i) e.NEXT[i * LD + tid] = e.NXT[i * LD + tid];
        return;
    }
    for (int i = 0; i < ns; ++i) {
        float m = 0.0f;
        for (int k = 0; k < K; ++k) m += e.NXT[(k * ns + i) * LD + tid];
        m /= (float)K;
        float v = m;
        if (sam_mode == METRPO_SAM_MODEL_MEAN_STD) {             // mean + N(0,1) * population std (:624-626)
            float var = 0.0f;
            for (int k = 0; k < K; ++k) { const float d = e.NXT[(k * ns + i) * LD + tid] - m; var = fmaf(d, d, var); }
            v = fmaf(noise_col[i * LD + tid], sqrtf(var / (float)K), m);
        } else if (sam_mode == METRPO_SAM_MODEL_MED) {           // np.median over K (:629-630)
            float lo = 0.0f, hi = 0.0f;
            const int r_lo = (K - 1) / 2, r_hi = K / 2;
            for (int k = 0; k < K; ++k) {
                const float xk = e.NXT[(k * ns + i) * LD + tid];
                int rank = 0;
                for (int j = 0; j < K; ++j) {
                    const float xj = e.NXT[(j * ns + i) * LD + tid];
                    rank += (xj < xk) || (xj == xk && j < k);
                }
                if (rank == r_lo) lo = xk;
                if (rank == r_hi) hi = xk;
            }
            v = 0.5f * (lo + hi);
        }
        e.NEXT[i * LD + tid] = v;
    }
}

// -------------------------------------------------------------------------------------------------
__global__ void k_policy_actions(ProblemDesc pd, const float* __restrict__ theta, const float* __restrict__ obs,
                                 const float* __restrict__ eps, int B, float* __restrict__ actions,
                                 float* __restrict__ mean_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid;
    const bool active = b < B;
    float* S = lds;
    float* A = S + pd.ns * LD;
    float* Bq = A + pd.pol.max_width * LD;
    for (int i = 0; i < pd.ns; ++i) S[i * LD + tid] = active ? obs[(size_t)b * pd.ns + i] : 0.0f;
    float* m = mlp_col(pd.pol, theta, S, A, Bq, LD, tid);
    if (!active) return;
    const float* __restrict__ log_std = theta + pd.pol.n_params;
    for (int d = 0; d < pd.na; ++d) {
        const float mu = m[d * LD + tid];
        mean_out[(size_t)b * pd.na + d] = mu;
        float a = mu;
        if (eps != nullptr) a = fmaf(eps[(size_t)b * pd.na + d], __expf(fmaxf(log_std[d], LOG_MIN_STD)), mu);
        actions[(size_t)b * pd.na + d] = a;
    }
}

__global__ void k_step(ProblemDesc pd, const float* __restrict__ dynp, const float* __restrict__ norm,
                       const float* __restrict__ s, const float* __restrict__ a, int B, int sam_mode,
                       const int32_t* __restrict__ model_idx, const float* __restrict__ noise,
                       float* __restrict__ s_next, float* __restrict__ reward, uint8_t* __restrict__ done,
                       float* next_all) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid;
    const bool active = b < B;
    EnvBufs e = envbufs_carve(lds, pd, sam_mode, LD);
    for (int i = 0; i < pd.ns; ++i) e.S[i * LD + tid] = active ? s[(size_t)b * pd.ns + i] : 0.0f;
    for (int d = 0; d < pd.na; ++d)   // np.clip(actions, *bounds) with bounds = +-1 (normalize()), env_helpers.py:599
        e.U[d * LD + tid] = active ? fminf(fmaxf(a[(size_t)b * pd.na + d], -1.0f), 1.0f) : 0.0f;
    int sel = 0;
    if ((sam_mode == METRPO_SAM_STEP_RAND || sam_mode == METRPO_SAM_EPS_RAND) && active) sel = model_idx[b];
    // model_mean_std noise is staged in NEXT (each dim is read before it is overwritten by the selection)
    float* noise_col = e.NEXT;
    if (sam_mode == METRPO_SAM_MODEL_MEAN_STD)
        for (int i = 0; i < pd.ns; ++i) noise_col[i * LD + tid] = active ? noise[(size_t)b * pd.ns + i] : 0.0f;
    dyn_heads_select(pd, dynp, norm, e, sam_mode, sel, noise_col, true, next_all, B, b, active, LD, tid);
    if (!active) return;
    for (int i = 0; i < pd.ns; ++i) s_next[(size_t)b * pd.ns + i] = e.NEXT[i * LD + tid];
    reward[b] = -env_cost(pd.env, pd.ns, pd.na, e.NEXT, e.U, LD, tid);           // env_helpers.py:601
    done[b] = env_is_done(pd.env, pd.ns, e.NEXT, LD, tid) ? 1 : 0;               // :603
}

// -------------------------------------------------------------------------------------------------

__global__ void k_rollout_generic(ProblemDesc pd, RolloutK r, const float* __restrict__ dynp,
                                  const float* __restrict__ theta, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid;
    const bool active = b < r.B;
    const int ns = pd.ns, na = pd.na, K = pd.K;
    EnvBufs e = envbufs_carve(lds, pd, r.sam_mode, LD);
    const uint64_t genv = r.stream_offset + (uint64_t)b;
    const float* __restrict__ log_std = theta + pd.pol.n_params;

    // vec_env.reset(): every env draws an initial state and a cur_model_idx (env_helpers.py:585-595)
    if (r.stop != nullptr && *r.stop != 0) return;           // the sampling loop already ended (metrpo_sampler_progress)
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
        const float* src = resume ? r.init_obs + (size_t)(active ? b : 0) * ns : r.pool + (size_t)row * ns;
        for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = active ? src[i] : 0.0f;
    }

    for (int t = 0; t < r.T; ++t) {
        const size_t tb = (size_t)t * r.B + b;
        // ---- policy.get_actions(obses) -------------------------------------------------------
        float* m = mlp_col(pd.pol, theta, e.S, e.A, e.Bq, LD, tid);
        const uint4 dstep = rng_draw(r.seed, genv, r.t0 + t, RNG_STEP, 0);
        for (int d0 = 0; d0 < na; d0 += 2) {
            float z[2] = {0.f, 0.f};
            if (!r.determ && r.eps == nullptr) {
                const uint4 blk = (d0 == 0) ? dstep : rng_draw(r.seed, genv, r.t0 + t, RNG_STEP, d0 >> 1);
                normal2(blk.x, blk.y, z[0], z[1]);
            }
            for (int d = d0; d < min(d0 + 2, na); ++d) {
                const float mu = m[d * LD + tid];
                float a = mu;
                if (!r.determ) {
                    const float zz = (r.eps != nullptr) ? (active ? r.eps[tb * na + d] : 0.0f) : z[d - d0];
                    a = fmaf(zz, __expf(fmaxf(log_std[d], LOG_MIN_STD)), mu);
                }
                if (active) { r.act[tb * na + d] = a; r.mean[tb * na + d] = mu; }
                e.U[d * LD + tid] = fminf(fmaxf(a, -1.0f), 1.0f);
            }
        }
        // ---- vec_env.step(actions) -----------------------------------------------------------
        ts += 1;
        int sel = cur_model;
        if (r.sam_mode == METRPO_SAM_STEP_RAND)
            sel = (r.model_idx != nullptr) ? (active ? r.model_idx[tb] : 0) : rng_index(dstep.z, K);
        float* noise_col = e.NEXT;
        if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
            for (int i0 = 0; i0 < ns; i0 += 4) {
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (r.sel_noise == nullptr) normal4(rng_draw(r.seed, genv, r.t0 + t, RNG_SELNOISE, i0 >> 2), z);
                for (int i = i0; i < min(i0 + 4, ns); ++i)
                    noise_col[i * LD + tid] = (r.sel_noise != nullptr) ? (active ? r.sel_noise[tb * ns + i] : 0.0f) : z[i - i0];
            }
        }
        if (active) for (int i = 0; i < ns; ++i) r.obs[tb * ns + i] = e.S[i * LD + tid];
        dyn_heads_select(pd, dynp, norm, e, r.sam_mode, sel, noise_col, r.eval_all != 0, nullptr, r.B, b, active, LD, tid);
        const float rew = -env_cost(pd.env, ns, na, e.NEXT, e.U, LD, tid);
        bool dn = env_is_done(pd.env, ns, e.NEXT, LD, tid);
        dn = dn || (ts >= r.H);                                               // env_helpers.py:604
        if (active) { r.rew[tb] = rew; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts - 1; }
        // ---- reset(dones) (env_helpers.py:585-595): new state from the pool, new cur_model_idx ----
        if (dn) {
            int row = 0;
            if (active) {
                const size_t rb = (size_t)(t + 1) * r.B + b;
                row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
                cur_model = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
            }
            for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = active ? r.pool[(size_t)row * ns + i] : 0.0f;
            ts = 0;
        } else {
            for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = e.NEXT[i * LD + tid];
        }
    }
    if (active && r.last_obs != nullptr)
        for (int i = 0; i < ns; ++i) r.last_obs[(size_t)b * ns + i] = e.S[i * LD + tid];
    if (active && r.last_ts != nullptr) r.last_ts[b] = ts;
    if (active && r.last_model != nullptr) r.last_model[b] = cur_model;
}

// -------------------------------------------------------------------------------------------------
// Validation rollout: grid.y = model i; deterministic clipped policy, model i fixed; per-step batch
// means are accumulated as  sum_b gamma^t * cost / Bv  into costs[i] (double atomics per block).
__global__ void k_validation(ProblemDesc pd, const float* __restrict__ dynp, const float* __restrict__ theta,
                             const float* __restrict__ norm, const float* __restrict__ s0, int Bv, int T,
                             double gamma, double* __restrict__ costs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[16];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid;
    const int model = blockIdx.y;
    const bool active = b < Bv;
    const int ns = pd.ns, na = pd.na;
    EnvBufs e = envbufs_carve(lds, pd, METRPO_SAM_EPS_RAND, LD);
    for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = active ? s0[(size_t)b * ns + i] : 0.0f;
    double acc = 0.0, g = 1.0;
    float dones = 0.0f;
    for (int t = 0; t < T; ++t) {
        float* m = mlp_col(pd.pol, theta, e.S, e.A, e.Bq, LD, tid);
        for (int d = 0; d < na; ++d) e.U[d * LD + tid] = fminf(fmaxf(m[d * LD + tid], -1.0f), 1.0f);   // :128
        dyn_heads_select(pd, dynp, norm, e, METRPO_SAM_EPS_RAND, model, nullptr, false, nullptr, Bv, b, true, LD, tid);
        float c = env_cost(pd.env, ns, na, e.NEXT, e.U, LD, tid);
        if (pd.env == METRPO_ENV_ANT) {                       // cost_tf(..., dones) then dones update (:134-137)
            c *= (1.0f - dones);
            dones = fmaxf(dones, env_is_done(pd.env, ns, e.NEXT, LD, tid) ? 1.0f : 0.0f);
        }
        if (active) acc += g * (double)c;
        g *= gamma;
        for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = e.NEXT[i * LD + tid];
    }
    const double tot = block_sum(acc, red);
    if (tid == 0) costs[(size_t)model * gridDim.x + blockIdx.x] = tot / (double)Bv;     // one partial per block; added in block order by k_det_cost_reduce
}

// -------------------------------------------------------------------------------------------------
static int pick_block(metrpo_ctx* c, size_t floats_per_thread, int B, int* block, size_t* shmem) {
    // largest power-of-two block (<= 64 lanes so a block is one wave: no barriers needed) whose LDS fits;
    // prefer >= 2 blocks per CU when it does not cost lanes.
    const size_t LDS_MAX = 160 * 1024;
    int bs = 64;
    while (bs > 1 && floats_per_thread * bs * sizeof(float) > LDS_MAX) bs >>= 1;
    if (floats_per_thread * bs * sizeof(float) > LDS_MAX)
        return set_err(c, METRPO_EUNSUPPORTED, "layer widths exceed the LDS budget of the generic kernel");
    *block = bs;
    *shmem = floats_per_thread * bs * sizeof(float);
    (void)B;
    return METRPO_OK;
}

template <typename Kern>
static int allow_lds(metrpo_ctx* c, Kern kern, size_t shmem) {
    if (shmem > 64 * 1024)
        HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    return METRPO_OK;
}

int launch_policy_actions(metrpo_ctx* c, const float* obs, const float* eps, int B, float* actions, float* mean,
                          hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    int bs; size_t sh;
    int rc = pick_block(c, (size_t)pd.ns + 2 * pd.pol.max_width, B, &bs, &sh);
    if (rc) return rc;
    if ((rc = allow_lds(c, k_policy_actions, sh))) return rc;
    hipLaunchKernelGGL(k_policy_actions, dim3((B + bs - 1) / bs), dim3(bs), sh, st, pd, c->d_theta, obs, eps, B,
                       actions, mean);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_step(metrpo_ctx* c, const float* s, const float* a, int B, int sam_mode, const int32_t* model_idx,
                const float* noise, float* s_next, float* reward, uint8_t* done, float* next_all, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    int bs; size_t sh;
    int rc = pick_block(c, envbufs_floats(pd, sam_mode, 1), B, &bs, &sh);
    if (rc) return rc;
    if ((rc = allow_lds(c, k_step, sh))) return rc;
    hipLaunchKernelGGL(k_step, dim3((B + bs - 1) / bs), dim3(bs), sh, st, pd, c->d_dyn, c->d_norm, s, a, B, sam_mode,
                       model_idx, noise, s_next, reward, done, next_all);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_rollout_generic(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    int bs; size_t sh;
    int rc = pick_block(c, envbufs_floats(pd, a->sam_mode, 1), a->B, &bs, &sh);
    if (rc) return rc;
    if ((rc = allow_lds(c, k_rollout_generic, sh))) return rc;
    RolloutK r = make_rollout_k(a);
    hipLaunchKernelGGL(k_rollout_generic, dim3((a->B + bs - 1) / bs), dim3(bs), sh, st, pd, r, c->d_dyn, c->d_theta,
                       c->d_norm);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_validation_cost(metrpo_ctx* c, const float* s0, int Bv, int T, double gamma, double* costs, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    if (c->det_cfg >= 0) {                                   // MFMA forward sweep (bptt_mfma.hip), costs only
        const int rc0 = ensure_detpart(c, Bv); if (rc0) return rc0;
        return launch_det_forward(c, c->det_cfg, s0, Bv, T, gamma, nullptr, nullptr, c->d_detpart, costs, st);
    }
    if (c->det_gemm) {                                       // large nets: the resident kernel's validation mode where its table has the shape, else the GEMM-path sweep (det_gemm.hip)
        if (ctx_opt(c, OPT_NO_RESIDENT_VALIDATION) == nullptr) {
            const int rcr = launch_validation_resident(c, s0, Bv, T, gamma, costs, st);
            if (rcr != METRPO_EUNSUPPORTED) return rcr;
        }
        return launch_dg_forward(c, s0, Bv, T, gamma, nullptr, nullptr, costs, st);
    }
    int bs; size_t sh;
    int rc = pick_block(c, envbufs_floats(pd, METRPO_SAM_EPS_RAND, 1), Bv, &bs, &sh);
    if (rc) return rc;
    if ((rc = allow_lds(c, k_validation, sh))) return rc;
    const int gx = (Bv + bs - 1) / bs;
    if ((rc = ensure_detpart_n(c, (size_t)pd.K * gx))) return rc;
    hipLaunchKernelGGL(k_validation, dim3(gx, pd.K), dim3(bs), sh, st, pd, c->d_dyn, c->d_theta,
                       c->d_norm, s0, Bv, T, gamma, c->d_detpart);
    return launch_det_cost_reduce(c, gx, c->d_detpart, costs, st);
}
