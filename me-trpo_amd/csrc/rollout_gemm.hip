// Step-wise rollout for LARGE dynamics networks (hidden >= 128: the params-file shapes 2x512 / 2x1024 / 3x1024 of
// BASELINE's C0-params-file, C2, C3, C4).  Same reference path as the fused kernels (samplers/vectorized_sampler.py:45-116,
// env_helpers.py:597-635, training.py:218-269), different mapping: at these widths ONE time step is already tens of
// GFLOP, so the time loop stays on the host (stream-ordered launches, no synchronisation) and every dynamics layer is
// a batched-over-heads GEMM on the f32 matrix core:
//     k_big_pre    thread per env: policy forward (tiny), action, clip, normalise + drop -> X[B][n_in]; writes obs/act/mean
//     k_gemm_bias_act   C[k] = act(A[k] . W[k] + b[k])   128x128(64) tiles, v_mfma_f32_32x32x2_f32, LDS double buffer
//     k_big_post   thread per env: de-normalise + residual, sam_mode selection over the K heads, reward, done, reset
// The weights are streamed from L2/HBM every step (K x 1-9 MB): the tile shape gives >= 128-fold reuse per fetched
// weight, which keeps the kernel MFMA-bound (AI ~ 60 flop/B at B = 2500).
#include "gemm_mfma.h"

// ------------------------------------------------------------------------------------------------
struct BigState { float* S; int* ts; int* cur_model; float* X; float* U; float* HA; float* HB; float* OUT; float* PART; };

// policy.get_actions + clip + normalise/drop; one thread per env; policy activations in LDS columns
__global__ void k_big_pre(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ theta, const float* __restrict__ norm,
                          BigState st) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid;
    const bool active = b < r.B;
    const int ns = pd.ns, na = pd.na;
    float* Sc = lds; float* A = Sc + ns * LD; float* Bq = A + pd.pol.max_width * LD;
    const uint64_t genv = r.stream_offset + (uint64_t)b;
    if (t == 0 && active) {                                      // vec_env.reset() (env_helpers.py:585-595)
        const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
        st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, pd.K);
        st.ts[b] = 0;
        for (int i = 0; i < ns; ++i) st.S[(size_t)b * ns + i] = r.pool[(size_t)row * ns + i];
    }
    for (int i = 0; i < ns; ++i) Sc[i * LD + tid] = active ? st.S[(size_t)b * ns + i] : 0.0f;
    float* m = mlp_col(pd.pol, theta, Sc, A, Bq, LD, tid);
    if (!active) return;
    const size_t tb = (size_t)t * r.B + b;
    const float* __restrict__ log_std = theta + pd.pol.n_params;
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    const uint4 dstep = rng_draw(r.seed, genv, t, RNG_STEP, 0);
    for (int i = 0; i < ns; ++i) {
        const float s = Sc[i * LD + tid];
        r.obs[tb * ns + i] = s;
        if (i >= pd.n_drop) st.X[(size_t)b * pd.nin + i - pd.n_drop] = (s - in_mean[i]) / in_std[i];       // training.py:228,146-151
    }
    for (int d0 = 0; d0 < na; d0 += 2) {
        float z[2] = {0.f, 0.f};
        if (!r.determ && r.eps == nullptr) {
            const uint4 blk = (d0 == 0) ? dstep : rng_draw(r.seed, genv, t, RNG_STEP, d0 >> 1);
            normal2(blk.x, blk.y, z[0], z[1]);
        }
        for (int d = d0; d < min(d0 + 2, na); ++d) {
            const float mu = m[d * LD + tid];
            float a = mu;
            if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * na + d] : z[d - d0], __expf(fmaxf(log_std[d], LOG_MIN_STD)), mu);
            r.act[tb * na + d] = a; r.mean[tb * na + d] = mu;
            const float ac = fminf(fmaxf(a, -1.0f), 1.0f);          // env_helpers.py:599
            st.U[(size_t)b * na + d] = ac;
            st.X[(size_t)b * pd.nin + (ns - pd.n_drop) + d] = (ac - in_mean[ns + d]) / in_std[ns + d];
        }
    }
}

// de-normalise + residual (training.py:257), selection (env_helpers.py:617-634), reward (:601), done (:603-604), reset (:585-595)
__global__ void k_big_post(ProblemDesc pd, RolloutK r, int t, const float* __restrict__ norm, BigState st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= r.B) return;
    const int ns = pd.ns, na = pd.na, K = pd.K;
    const size_t tb = (size_t)t * r.B + b;
    const uint64_t genv = r.stream_offset + (uint64_t)b;
    const float* diff_mean = norm + 2 * (ns + na); const float* diff_std = diff_mean + ns;
    const uint4 dstep = rng_draw(r.seed, genv, t, RNG_STEP, 0);
    int sel = st.cur_model[b];
    if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (r.model_idx != nullptr) ? r.model_idx[tb] : rng_index(dstep.z, K);
    if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
    const bool simple = (r.sam_mode == METRPO_SAM_STEP_RAND || r.sam_mode == METRPO_SAM_EPS_RAND || r.sam_mode == METRPO_SAM_ONE_MODEL);
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) { const float a = st.U[(size_t)b * na + d]; su2 = fmaf(a, a, su2); }
    float* S = st.S + (size_t)b * ns;
    auto head = [&](int k, int i) { return fmaf(diff_std[i], st.OUT[((size_t)k * r.B + b) * ns + i], diff_mean[i]) + S[i]; };
    float pen = 0.0f, key = 0.0f, h0v = 0.0f, h1v = 0.0f, zc = 0.0f, last = 0.0f;
    bool finite = true;
    // the next state is written back into S only after every dim has been selected (S is the residual base)
    float nxt_small[64];                                           // ns <= 64 enforced by the launcher
    for (int i = 0; i < ns; ++i) {
        float v;
        if (simple) v = head(sel, i);
        else {
            float m = 0.0f;
            for (int k = 0; k < K; ++k) m += head(k, i);
            m /= (float)K;
            v = m;
            if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
                float var = 0.0f;
                for (int k = 0; k < K; ++k) { const float d = head(k, i) - m; var = fmaf(d, d, var); }
                float z4[4];
                float nz;
                if (r.sel_noise != nullptr) nz = r.sel_noise[tb * ns + i];
                else { normal4(rng_draw(r.seed, genv, t, RNG_SELNOISE, i >> 2), z4); nz = z4[i & 3]; }
                v = fmaf(nz, sqrtf(var / (float)K), m);
            } else if (r.sam_mode == METRPO_SAM_MODEL_MED) {
                const int r_lo = (K - 1) / 2, r_hi = K / 2;
                float lo = 0.0f, hi = 0.0f;
                for (int k = 0; k < K; ++k) {
                    const float xk = head(k, i);
                    int rank = 0;
                    for (int j = 0; j < K; ++j) { const float xj = head(j, i); rank += (xj < xk) || (xj == xk && j < k); }
                    if (rank == r_lo) lo = xk;
                    if (rank == r_hi) hi = xk;
                }
                v = 0.5f * (lo + hi);
            }
        }
        nxt_small[i] = v;
        finite = finite && isfinite(v);
        if (i >= 2) pen += fmaxf(fabsf(v) - 100.0f, 0.0f);
        if (i == 0) h0v = v;
        if (i == 1) h1v = v;
        if (i == 2) zc = v;
        if (i == ns - 1) last = v;
        const int ki = (pd.env == METRPO_ENV_SWIMMER || pd.env == METRPO_ENV_HOPPER) ? 5 : (pd.env == METRPO_ENV_HALF_CHEETAH) ? 9
                       : (pd.env == METRPO_ENV_ANT) ? 15 : (pd.env == METRPO_ENV_SNAKE) ? 7 : -1;
        if (i == ki) key = v;
    }
    float cost = 0.0f;
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: cost = -(key - 1e-2f * (su2 / (float)na)); break;
    case METRPO_ENV_HALF_CHEETAH: cost = -fminf(fmaxf(key - 1e-1f * 0.5f * su2, -10.0f), 10.0f); break;
    case METRPO_ENV_ANT: cost = -(key - 1e-2f * 0.5f * su2 + 0.05f); break;
    case METRPO_ENV_HUMANOID: cost = (last - 1.5f) * (last - 1.5f) + 1e-2f * 1e-3f * su2; break;
    case METRPO_ENV_HOPPER: cost = -(key - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - h0v, 0.0f) - 10.0f * fmaxf(fabsf(h1v) - 0.2f, 0.0f) - pen); break;
    case METRPO_ENV_SNAKE: cost = -(key - 1e-2f * 0.5f * su2); break;
    }
    int ts = st.ts[b] + 1;
    bool dn = (pd.env == METRPO_ENV_ANT) ? !((zc >= 0.2f) && (zc <= 1.0f) && finite) : false;
    dn = dn || (ts >= r.H);
    r.rew[tb] = -cost; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts - 1;
    if (dn) {
        const size_t rb = (size_t)(t + 1) * r.B + b;
        const int row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
        st.cur_model[b] = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
        for (int i = 0; i < ns; ++i) S[i] = r.pool[(size_t)row * ns + i];
        ts = 0;
    } else {
        for (int i = 0; i < ns; ++i) S[i] = nxt_small[i];
    }
    st.ts[b] = ts;
    if (t == r.T - 1 && r.last_obs != nullptr) for (int i = 0; i < ns; ++i) r.last_obs[(size_t)b * ns + i] = S[i];
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

bool gemm_path_applicable(const metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    if (pd.ns > 64) return false;
    int minw = 1 << 30;
    for (int l = 1; l < pd.dyn.n_layers; ++l) minw = std::min(minw, pd.dyn.dims[l]);
    return pd.dyn.n_layers >= 2 && minw >= 128;
}

int launch_rollout_gemm(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int B = a->B, K = pd.K, L = pd.dyn.n_layers;
    int maxh = 0;
    for (int l = 1; l < L; ++l) maxh = std::max(maxh, pd.dyn.dims[l]);
    // workspace (floats): S, X, U, HA, HB, OUT + ints ts, cur_model
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };       // keep every sub-buffer 16-byte aligned
    const size_t nS = up4((size_t)B * pd.ns), nX = up4((size_t)B * pd.nin), nU = up4((size_t)B * pd.na), nH = up4((size_t)K * B * maxh), nO = up4((size_t)K * B * pd.ns);
    const size_t nP = up4(skinny_part_floats(B, pd.ns, pd.dyn.dims[L - 1], K));
    const size_t need = (nS + nX + nU + 2 * nH + nO + nP) * sizeof(float) + 2 * (size_t)B * sizeof(int) + 256;
    if (need > c->big_cap) {
        if (c->d_big) HIP_TRY(c, hipFree(c->d_big));
        c->d_big = nullptr; c->big_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_big, need));
        c->big_cap = need;
    }
    BigState bs;
    float* p = (float*)c->d_big;
    bs.S = p; p += nS; bs.X = p; p += nX; bs.U = p; p += nU; bs.HA = p; p += nH; bs.HB = p; p += nH; bs.OUT = p; p += nO; bs.PART = nP ? p : nullptr; p += nP;
    bs.ts = (int*)p; bs.cur_model = bs.ts + B;
    RolloutK r;
    r.B = a->B; r.T = a->T; r.H = a->H; r.sam_mode = a->sam_mode; r.determ = a->determ; r.eval_all = a->eval_all_heads;
    r.n_pool = a->n_pool; r.seed = a->seed; r.stream_offset = a->stream_offset; r.pool = a->d_pool; r.eps = a->d_eps;
    r.model_idx = a->d_model_idx; r.sel_noise = a->d_sel_noise; r.reset_idx = a->d_reset_idx;
    r.reset_model = a->d_reset_model; r.obs = a->d_obs; r.act = a->d_act; r.rew = a->d_rew; r.mean = a->d_mean;
    r.done = a->d_done; r.tpath = a->d_tpath; r.last_obs = a->d_last_obs;
    const int pbs = 64;
    const size_t psh = (size_t)(pd.ns + 2 * pd.pol.max_width) * pbs * sizeof(float);
    if (psh > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "policy too wide for k_big_pre");
    if (psh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_big_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psh));
    for (int t = 0; t < a->T; ++t) {
        hipLaunchKernelGGL(k_big_pre, dim3((B + pbs - 1) / pbs), dim3(pbs), psh, st, pd, r, t, c->d_theta, c->d_norm, bs);
        const float* in = bs.X; long long sIn = 0; int ldin = pd.nin;
        float* bufs[2] = {bs.HA, bs.HB};
        for (int l = 0; l < L; ++l) {
            const int Kd = pd.dyn.dims[l], N = pd.dyn.dims[l + 1];
            const bool lastl = (l == L - 1);
            float* out = lastl ? bs.OUT : bufs[l & 1];
            const long long sOut = (long long)B * N;
            const float* Wl = c->d_dyn + pd.dyn.w_off[l];
            const float* bl = c->d_dyn + pd.dyn.b_off[l];
            if (lastl && pd.dyn.act[l] == METRPO_ACT_IDENTITY)
                gemm_skinny_bias(in, sIn, ldin, Wl, pd.dyn.n_params, N, bl, pd.dyn.n_params, out, sOut, B, N, Kd, K, bs.PART, st);
            else gemm_launch(pd.dyn.act[l], in, sIn, ldin, Wl, pd.dyn.n_params, N, bl, pd.dyn.n_params, out, sOut, N, B, N, Kd, K, st);
            in = out; sIn = sOut; ldin = N;
        }
        hipLaunchKernelGGL(k_big_post, dim3((B + 127) / 128), dim3(128), 0, st, pd, r, t, c->d_norm, bs);
    }
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
