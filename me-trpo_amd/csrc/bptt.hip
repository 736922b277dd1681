// BPTT policy gradient through the unrolled imagined rollout ("next" row):
//   d/d theta  mean_i  sum_t gamma^t mean_b cost_tf(x_t, u_t, x_{t+1})     with  u_t = clip(policy(x_t)),  x_{t+1} = model_i(x_t, u_t)
//   build_policy_graph (model_based_rl.py:106-151), training_policy_cost = reduce_mean over models (:365), Adam with per-variable
//   tf.clip_by_norm (get_policy_optimizer :186-206, utils.py:262-276), 'bptt' branch of optimize_policy (:1181-1187).
//
// Three device stages, all stream-ordered, no host sync:
//   k_bptt_forward   thread = (env b, model i): the deterministic rollout of build_policy_graph; keeps every state x_t, the per-step
//                    weight gamma^t (1 - dones_t) / (B K) and the per-model costs (= k_validation + stores).
//   k_bptt_backward  same mapping, t = T-1 .. 0: recomputes the two MLPs of the step from x_t, pushes the state adjoint through
//                    cost -> dynamics head i (incl. the residual and the normalisers) -> clip -> policy input, and writes the adjoint
//                    of the policy MEAN for every (i, t, b).
//   policy VJP       sum over all (i, t, b) samples of J_policy(x_t)^T adjoint_mean: the gradient kernels of the TRPO update with
//                    the mean-adjoint supplied (PolK.gm) instead of derived from the surrogate loss -- MFMA path included.
// Generic in the layer widths (activations in LDS columns, one thread per trajectory).
#include <cmath>
#include "device_common.h"

struct BpttBufs { float *S, *XN, *U, *X, *PH, *DH, *GA, *GB, *LAM; };

__host__ __device__ inline int net_store_rows(const NetDesc& n, bool with_output) {
    int r = 0;
    for (int l = 1; l <= n.n_layers; ++l) if (l < n.n_layers || with_output) r += n.dims[l];
    return r;
}
__host__ __device__ inline size_t bptt_floats(const ProblemDesc& pd) {
    const int mw = max(max(pd.dyn.max_width, pd.pol.max_width), pd.ns + pd.na);
    return (size_t)(3 * pd.ns + pd.na + (pd.ns + pd.na) + net_store_rows(pd.pol, true) + net_store_rows(pd.dyn, false) + 2 * mw);
}
__device__ __forceinline__ BpttBufs bptt_carve(float* lds, const ProblemDesc& pd, int LD) {
    const int mw = max(max(pd.dyn.max_width, pd.pol.max_width), pd.ns + pd.na);
    BpttBufs e;
    e.S = lds; e.XN = e.S + pd.ns * LD; e.LAM = e.XN + pd.ns * LD; e.U = e.LAM + pd.ns * LD; e.X = e.U + pd.na * LD;
    e.PH = e.X + (pd.ns + pd.na) * LD; e.DH = e.PH + net_store_rows(pd.pol, true) * LD;
    e.GA = e.DH + net_store_rows(pd.dyn, false) * LD; e.GB = e.GA + mw * LD;
    return e;
}

// forward keeping the output of every layer in `store` (layer l+1 outputs at row offset sum_{j<=l} dims[j]); returns the last layer's
// rows.  with_output = false: the output layer goes to `out_last` instead (not kept).
__device__ __forceinline__ float* mlp_col_store(const NetDesc& net, const float* __restrict__ params, const float* in, float* store,
                                                float* out_last, int LD, int tid) {
    const float* cur = in;
    float* dst = store;
    for (int l = 0; l < net.n_layers; ++l) {
        float* o = (l == net.n_layers - 1 && out_last != nullptr) ? out_last : dst;
        dense_col(params + net.w_off[l], params + net.b_off[l], net.dims[l], net.dims[l + 1], net.act[l], cur, o, LD, tid);
        cur = o;
        dst += net.dims[l + 1] * LD;
    }
    return const_cast<float*>(cur);
}

// dout[j] = sum_i W[j][i] * din[i]   (W row-major n_in x n_out: the VJP of  out = in . W)
__device__ __forceinline__ void dense_col_T(const float* __restrict__ W, int n_in, int n_out, const float* din, float* dout, int LD, int tid) {
    for (int j = 0; j < n_in; ++j) {
        const float* __restrict__ w = W + (size_t)j * n_out;
        float s0 = 0.0f, s1 = 0.0f;
        int i = 0;
        for (; i + 1 < n_out; i += 2) { s0 = fmaf(w[i], din[i * LD + tid], s0); s1 = fmaf(w[i + 1], din[(i + 1) * LD + tid], s1); }
        if (i < n_out) s0 = fmaf(w[i], din[i * LD + tid], s0);
        dout[j * LD + tid] = s0 + s1;
    }
}

// d(w * cost)/du -> gU (written), d(w * cost)/dx_next -> added into G
__device__ __forceinline__ void cost_adjoint(int env, int ns, int na, const float* xn, const float* u, float w, float* gU, float* G, int LD, int tid) {
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) { const float a = u[d * LD + tid]; su2 = fmaf(a, a, su2); }
    float cu = 0.0f;                                         // gU[d] = cu * u[d]
    switch (env) {
    case METRPO_ENV_SWIMMER: G[5 * LD + tid] -= w; cu = w * 1e-2f * 2.0f / (float)na; break;
    case METRPO_ENV_HALF_CHEETAH: {
        const float inner = xn[9 * LD + tid] - 1e-1f * 0.5f * su2;
        const float p = (inner >= -10.0f && inner <= 10.0f) ? 1.0f : 0.0f;       // tf.clip_by_value passes the gradient inside [min, max]
        G[9 * LD + tid] -= w * p; cu = w * p * 1e-1f; break;
    }
    case METRPO_ENV_ANT: G[15 * LD + tid] -= w; cu = w * 1e-2f; break;
    case METRPO_ENV_HUMANOID: G[(ns - 1) * LD + tid] += w * 2.0f * (xn[(ns - 1) * LD + tid] - 1.5f); cu = w * 2e-5f; break;
    case METRPO_ENV_HOPPER: {
        G[5 * LD + tid] -= w; cu = w * 0.01f;
        if (0.45f - xn[0 * LD + tid] > 0.0f) G[0 * LD + tid] -= w * 10.0f;
        const float a1 = xn[1 * LD + tid];
        if (fabsf(a1) - 0.2f > 0.0f) G[1 * LD + tid] += w * 10.0f * (a1 > 0.0f ? 1.0f : -1.0f);
        for (int i = 2; i < ns; ++i) { const float v = xn[i * LD + tid]; if (fabsf(v) - 100.0f > 0.0f) G[i * LD + tid] += w * (v > 0.0f ? 1.0f : -1.0f); }
        break;
    }
    case METRPO_ENV_SNAKE: G[7 * LD + tid] -= w; cu = w * 1e-2f; break;
    }
    for (int d = 0; d < na; ++d) gU[d * LD + tid] = cu * u[d * LD + tid];
}

// XS [K][T+1][B][ns], WT [K][T][B], costs [K] (+=)
__global__ void k_bptt_forward(ProblemDesc pd, const float* __restrict__ dynp, const float* __restrict__ theta, const float* __restrict__ norm,
                               const float* __restrict__ s0, int B, int T, double gamma, float* __restrict__ XS, float* __restrict__ WT,
                               double* __restrict__ costs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[16];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid, model = blockIdx.y;
    const bool active = b < B;
    const int ns = pd.ns, na = pd.na, K = pd.K;
    BpttBufs e = bptt_carve(lds, pd, LD);
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    const float* diff_mean = norm + 2 * (ns + na); const float* diff_std = diff_mean + ns;
    const float* __restrict__ pk = dynp + (size_t)model * pd.dyn.n_params;
    float* xs = XS + (size_t)model * (T + 1) * B * ns;
    for (int i = 0; i < ns; ++i) { const float v = active ? s0[(size_t)b * ns + i] : 0.0f; e.S[i * LD + tid] = v; if (active) xs[(size_t)b * ns + i] = v; }
    double acc = 0.0, g = 1.0;
    float dones = 0.0f;
    for (int t = 0; t < T; ++t) {
        float* m = mlp_col(pd.pol, theta, e.S, e.GA, e.GB, LD, tid);
        for (int d = 0; d < na; ++d) e.U[d * LD + tid] = fminf(fmaxf(m[d * LD + tid], -1.0f), 1.0f);        // :128
        for (int i = 0; i < ns; ++i) e.X[i * LD + tid] = (e.S[i * LD + tid] - in_mean[i]) / in_std[i];
        for (int d = 0; d < na; ++d) e.X[(ns + d) * LD + tid] = (e.U[d * LD + tid] - in_mean[ns + d]) / in_std[ns + d];
        float* out = mlp_col(pd.dyn, pk, e.X + pd.n_drop * LD, e.GA, e.GB, LD, tid);
        for (int i = 0; i < ns; ++i) e.XN[i * LD + tid] = fmaf(diff_std[i], out[i * LD + tid], diff_mean[i]) + e.S[i * LD + tid];
        const float c = env_cost(pd.env, ns, na, e.XN, e.U, LD, tid);
        const float live = 1.0f - dones;
        if (pd.env == METRPO_ENV_ANT) dones = fmaxf(dones, env_is_done(pd.env, ns, e.XN, LD, tid) ? 1.0f : 0.0f);
        if (active) {
            acc += g * (double)(c * live);
            WT[((size_t)model * T + t) * B + b] = (float)(g * (double)live / ((double)B * (double)K));
            for (int i = 0; i < ns; ++i) xs[((size_t)(t + 1) * B + b) * ns + i] = e.XN[i * LD + tid];
        }
        g *= gamma;
        for (int i = 0; i < ns; ++i) e.S[i * LD + tid] = e.XN[i * LD + tid];
    }
    const double tot = block_sum(acc, red);
    if (tid == 0) costs[(size_t)model * gridDim.x + blockIdx.x] = tot / (double)B;      // one partial per block; added in block order by k_det_cost_reduce
}

// GM [K][T+1][B][na]: adjoint of the (pre-clip) policy mean for every sample; slice t = T stays zero
__global__ void k_bptt_backward(ProblemDesc pd, const float* __restrict__ dynp, const float* __restrict__ theta, const float* __restrict__ norm,
                                int B, int T, const float* __restrict__ XS, const float* __restrict__ WT, float* __restrict__ GM) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid, model = blockIdx.y;
    const bool active = b < B;
    const int ns = pd.ns, na = pd.na;
    const NetDesc& dn = pd.dyn; const NetDesc& pn = pd.pol;
    BpttBufs e = bptt_carve(lds, pd, LD);
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    const float* diff_std = norm + 2 * (ns + na) + ns;
    const float* __restrict__ pk = dynp + (size_t)model * dn.n_params;
    const float* xs = XS + (size_t)model * (T + 1) * B * ns;
    for (int i = 0; i < ns; ++i) e.LAM[i * LD + tid] = 0.0f;
    for (int t = T - 1; t >= 0; --t) {
        for (int i = 0; i < ns; ++i) {
            e.S[i * LD + tid] = active ? xs[((size_t)t * B + b) * ns + i] : 0.0f;
            e.XN[i * LD + tid] = active ? xs[((size_t)(t + 1) * B + b) * ns + i] : 0.0f;
        }
        const float w = active ? WT[((size_t)model * T + t) * B + b] : 0.0f;
        // ---- recompute the step: policy (all layer outputs kept), clip, normalise, dynamics hidden layers ----
        float* mu = mlp_col_store(pn, theta, e.S, e.PH, nullptr, LD, tid);
        for (int d = 0; d < na; ++d) e.U[d * LD + tid] = fminf(fmaxf(mu[d * LD + tid], -1.0f), 1.0f);
        for (int i = 0; i < ns; ++i) e.X[i * LD + tid] = (e.S[i * LD + tid] - in_mean[i]) / in_std[i];
        for (int d = 0; d < na; ++d) e.X[(ns + d) * LD + tid] = (e.U[d * LD + tid] - in_mean[ns + d]) / in_std[ns + d];
        {
            const float* cur = e.X + pd.n_drop * LD; float* dst = e.DH;
            for (int l = 0; l < dn.n_layers - 1; ++l) {
                dense_col(pk + dn.w_off[l], pk + dn.b_off[l], dn.dims[l], dn.dims[l + 1], dn.act[l], cur, dst, LD, tid);
                cur = dst; dst += dn.dims[l + 1] * LD;
            }
        }
        // ---- adjoints: G = lambda_{t+1} + w dc/dx_next ;  gU_cost = w dc/du (kept in XN's place after use) -------------------
        float* G = e.LAM;                                       // lambda_{t+1} is consumed here
        float* gUc = e.X;                                       // the normalised input is dead after the forward recompute
        cost_adjoint(pd.env, ns, na, e.XN, e.U, w, gUc, G, LD, tid);
        // dynamics VJP: delta_out = diff_std * G ; back through the layers (relu masks from DH)
        float* da = e.GA; float* db = e.GB;
        for (int i = 0; i < ns; ++i) da[i * LD + tid] = diff_std[i] * G[i * LD + tid];
        {
            int off = 0;
            for (int l = 1; l < dn.n_layers - 1; ++l) off += dn.dims[l];          // rows of the LAST hidden layer in DH
            for (int l = dn.n_layers - 1; l >= 0; --l) {
                dense_col_T(pk + dn.w_off[l], dn.dims[l], dn.dims[l + 1], da, db, LD, tid);
                if (l > 0) {
                    const float* h = e.DH + off * LD;                              // output of layer l-1 (post-activation)
                    for (int j = 0; j < dn.dims[l]; ++j) {
                        const float hv = h[j * LD + tid];
                        const float dact = (dn.act[l - 1] == METRPO_ACT_RELU) ? (hv > 0.0f ? 1.0f : 0.0f)
                                          : (dn.act[l - 1] == METRPO_ACT_TANH) ? (1.0f - hv * hv) : 1.0f;
                        db[j * LD + tid] *= dact;
                    }
                    if (l > 1) off -= dn.dims[l - 1];
                }
                float* tmp = da; da = db; db = tmp;
            }
        }
        // da = adjoint of the (dropped) normalised input [nin].  State part: residual + normaliser; action part -> gU
        for (int i = 0; i < ns; ++i) {
            float v = G[i * LD + tid];
            if (i >= pd.n_drop) v += da[(i - pd.n_drop) * LD + tid] / in_std[i];
            e.XN[i * LD + tid] = v;                             // gS (x_next is dead now)
        }
        for (int d = 0; d < na; ++d) {
            const float gu = gUc[d * LD + tid] + da[(ns - pd.n_drop + d) * LD + tid] / in_std[ns + d];
            const float m = mu[d * LD + tid];
            const float gm = (m >= -1.0f && m <= 1.0f) ? gu : 0.0f;             // tf.clip_by_value gradient
            db[d * LD + tid] = gm;
            if (active) GM[(((size_t)model * (T + 1) + t) * B + b) * na + d] = gm;
        }
        // policy input VJP: back through the layers with the stored activations
        {
            float* pa = db; float* pb = da;
            int off = 0;
            for (int l = 1; l < pn.n_layers - 1; ++l) off += pn.dims[l];          // rows of the last hidden layer in PH
            for (int l = pn.n_layers - 1; l >= 0; --l) {
                dense_col_T(theta + pn.w_off[l], pn.dims[l], pn.dims[l + 1], pa, pb, LD, tid);
                if (l > 0) {
                    const float* h = e.PH + off * LD;
                    for (int j = 0; j < pn.dims[l]; ++j) {
                        const float hv = h[j * LD + tid];
                        const float dact = (pn.act[l - 1] == METRPO_ACT_TANH) ? (1.0f - hv * hv)
                                          : (pn.act[l - 1] == METRPO_ACT_RELU) ? (hv > 0.0f ? 1.0f : 0.0f) : 1.0f;
                        pb[j * LD + tid] *= dact;
                    }
                    if (l > 1) off -= pn.dims[l - 1];
                }
                float* tmp = pa; pa = pb; pb = tmp;
            }
            for (int i = 0; i < ns; ++i) e.LAM[i * LD + tid] = e.XN[i * LD + tid] + pa[i * LD + tid];       // lambda_t
        }
    }
}

// tf.clip_by_norm per variable + tf.train.AdamOptimizer on theta (one block per variable: W_l, b_l; log_std has zero gradient here)
__global__ void k_policy_adam(int n_seg, const int* __restrict__ seg_off, const double* __restrict__ grad, float* __restrict__ theta,
                              float* __restrict__ am, float* __restrict__ av, float lr_t, float b1, float b2, float eps, double clip_val) {
    __shared__ double red[16];
    __shared__ double s_scale;
    const int s = blockIdx.x, lo = seg_off[s], hi = seg_off[s + 1];
    double nn = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) nn += grad[i] * grad[i];
    const double tot = block_sum(nn, red);
    if (threadIdx.x == 0) { const double n = sqrt(tot); s_scale = (clip_val > 0.0) ? clip_val / fmax(n, clip_val) : 1.0; }
    __syncthreads();
    const double sc = s_scale;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float g = (float)(grad[i] * sc);
        const float m1 = b1 * am[i] + (1.0f - b1) * g, v1 = b2 * av[i] + (1.0f - b2) * g * g;
        am[i] = m1; av[i] = v1;
        theta[i] = theta[i] - lr_t * m1 / (sqrtf(v1) + eps);
    }
}

// -------------------------------------------------------------------------------------------------
int launch_bptt_grad(metrpo_ctx* c, const float* init, int B, int T, double gamma, double* costs, double* grad, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, ns = pd.ns, na = pd.na;
    const size_t nXS = (((size_t)K * (T + 1) * B * ns) + 3) & ~(size_t)3, nWT = (((size_t)K * T * B) + 3) & ~(size_t)3,
                 nGM = (((size_t)K * (T + 1) * B * na) + 3) & ~(size_t)3;
    const size_t need = (nXS + nWT + nGM) * sizeof(float) + sizeof(double) * (size_t)(pd.P + 1 + K);
    if (c->det_cfg >= 0) { const int rc0 = ensure_detpart(c, B); if (rc0) return rc0; }
    if (need > c->bptt_cap) {
        ws_retire(c, c->d_bptt);
        c->d_bptt = nullptr; c->bptt_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_bptt, need));
        c->bptt_cap = need;
    }
    float* XS = (float*)c->d_bptt; float* WT = XS + nXS; float* GM = WT + nWT;
    double* gout = (double*)(GM + nGM); double* cst = gout + pd.P + 1;
    HIP_TRY(c, hipMemsetAsync(GM, 0, sizeof(float) * nGM, st));
    if (c->det_cfg >= 0) {                                   // MFMA sweeps (bptt_mfma.hip)
        int rc1 = launch_det_forward(c, c->det_cfg, init, B, T, gamma, XS, WT, c->d_detpart, cst, st);
        if (rc1) return rc1;
        if ((rc1 = launch_det_backward(c, c->det_cfg, B, T, XS, WT, GM, st))) return rc1;
        const int rc2 = launch_policy_vjp(c, XS, GM, (long long)K * (T + 1) * B, gout, st);
        if (rc2) return rc2;
        if (grad) HIP_TRY(c, hipMemcpyAsync(grad, gout + 1, sizeof(double) * pd.P, hipMemcpyDeviceToDevice, st));
        if (costs) HIP_TRY(c, hipMemcpyAsync(costs, cst, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
        return METRPO_OK;
    }
    if (c->det_gemm) {                                       // GEMM-path sweeps for large dynamics nets (det_gemm.hip)
        int rc1 = launch_dg_forward(c, init, B, T, gamma, XS, WT, cst, st);
        if (rc1) return rc1;
        if ((rc1 = launch_dg_backward(c, B, T, XS, WT, GM, st))) return rc1;
        const int rc2 = launch_policy_vjp(c, XS, GM, (long long)K * (T + 1) * B, gout, st);
        if (rc2) return rc2;
        if (grad) HIP_TRY(c, hipMemcpyAsync(grad, gout + 1, sizeof(double) * pd.P, hipMemcpyDeviceToDevice, st));
        if (costs) HIP_TRY(c, hipMemcpyAsync(costs, cst, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
        return METRPO_OK;
    }
    const size_t fpt = bptt_floats(pd);
    const size_t LDS_MAX = 160 * 1024;
    int bs = 64;
    while (bs > 1 && fpt * bs * sizeof(float) > LDS_MAX) bs >>= 1;
    if (fpt * bs * sizeof(float) > LDS_MAX) return set_err(c, METRPO_EUNSUPPORTED, "bptt: layer widths exceed the LDS budget of the generic kernel");
    const size_t sh = fpt * bs * sizeof(float);
    if (sh > 64 * 1024) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_bptt_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        HIP_TRY(c, hipFuncSetAttribute((const void*)k_bptt_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    }
    const dim3 grid((B + bs - 1) / bs, K);
    { const int rcp = ensure_detpart_n(c, (size_t)K * grid.x); if (rcp) return rcp; }
    hipLaunchKernelGGL(k_bptt_forward, grid, dim3(bs), sh, st, pd, c->d_dyn, c->d_theta, c->d_norm, init, B, T, gamma, XS, WT, c->d_detpart);
    { const int rcp = launch_det_cost_reduce(c, (int)grid.x, c->d_detpart, cst, st); if (rcp) return rcp; }
    hipLaunchKernelGGL(k_bptt_backward, grid, dim3(bs), sh, st, pd, c->d_dyn, c->d_theta, c->d_norm, B, T, XS, WT, GM);
    HIP_TRY(c, hipGetLastError());
    // policy-parameter gradient: sum over the K (T+1) B samples of J(x)^T gm  (gradient kernels of the TRPO update, mean-adjoint supplied)
    const int rc = launch_policy_vjp(c, XS, GM, (long long)K * (T + 1) * B, gout, st);
    if (rc) return rc;
    if (grad) HIP_TRY(c, hipMemcpyAsync(grad, gout + 1, sizeof(double) * pd.P, hipMemcpyDeviceToDevice, st));
    if (costs) HIP_TRY(c, hipMemcpyAsync(costs, cst, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
    return METRPO_OK;
}

int launch_policy_adam(metrpo_ctx* c, const double* grad, double lr, double b1, double b2, double eps, double clip_val, bool reset, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int P = pd.P, L = pd.pol.n_layers, nseg = 2 * L + 1;
    if (!c->d_pol_adam) {
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_pol_adam, sizeof(float) * 2 * (size_t)P + sizeof(int) * (nseg + 1)));
        HIP_TRY(c, hipMemset(c->d_pol_adam, 0, sizeof(float) * 2 * (size_t)P));
        int seg[2 * MAXL + 2];
        for (int l = 0; l < L; ++l) { seg[2 * l] = pd.pol.w_off[l]; seg[2 * l + 1] = pd.pol.b_off[l]; }
        seg[2 * L] = pd.pol.n_params; seg[2 * L + 1] = P;
        HIP_TRY(c, hipMemcpy((char*)c->d_pol_adam + sizeof(float) * 2 * (size_t)P, seg, sizeof(int) * (nseg + 1), hipMemcpyHostToDevice));
        c->pol_adam_t = 0;
    }
    float* am = (float*)c->d_pol_adam; float* av = am + P;
    if (reset) {
        HIP_TRY(c, hipMemsetAsync(am, 0, sizeof(float) * 2 * (size_t)P, st));
        c->pol_adam_t = 0;
        return METRPO_OK;
    }
    c->pol_adam_t += 1;
    const double lr_t = lr * std::sqrt(1.0 - std::pow(b2, (double)c->pol_adam_t)) / (1.0 - std::pow(b1, (double)c->pol_adam_t));
    hipLaunchKernelGGL(k_policy_adam, dim3(nseg), dim3(256), 0, st, nseg, (const int*)(av + P), grad, c->d_theta, am, av, (float)lr_t, (float)b1,
                       (float)b2, (float)eps, clip_val);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int ensure_detpart(metrpo_ctx* c, int B) { return ensure_detpart_n(c, (size_t)c->pd.K * (size_t)(4 * ((B + 63) / 64))); }
int ensure_detpart_n(metrpo_ctx* c, size_t n_doubles) {
    const size_t need = sizeof(double) * n_doubles;
    if (need > c->detpart_cap) {
        ws_retire(c, c->d_detpart);
        c->d_detpart = nullptr; c->detpart_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_detpart, need));
        c->detpart_cap = need;
    }
    return METRPO_OK;
}
