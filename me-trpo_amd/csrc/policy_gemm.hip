// GEMM path of the policy-side TRPO kernels for policy shapes the fused MFMA kernels (policy_mfma.hip) do not cover -- e.g.
// Humanoid's 100-50-25 (params/params-humanoid.json:6-8) at N = B*H = millions of samples.  Same arithmetic as
// policy_update.hip (algos/npo.py:68-75 graph; [rllab] DiagonalGaussian / PerlmutterHvp), expressed layer by layer as
// tall-skinny GEMMs over the N samples on the f32 matrix core (gemm_mfma.h) with the element-wise parts in the epilogues:
//
//   forward     H_{l+1} = tanh(H_l W_l + b_l)                                  EPI_BIAS_TANH          (kept for the whole CG solve)
//   tangent     T_{l+1} = ([H_l | T_l] [V_l ; W_l] + vb_l) * (1 - H_{l+1}^2)   EPI_DTANH, ONE GEMM per layer: H_l and T_l are the two
//                                                                              halves of one row, V_l is stacked on W_l
//   head        per-sample kernel: likelihood ratio / KL / d(objective)/d(mean) -> U [N][na]
//   back-prop   D_l = (D_{l+1} W_l^T) * (1 - H_l^2)                            EPI_DTANH with opB = W^T
//   gradients   G_l = H_l^T D_{l+1}, column sums = bias gradients              split-K partials over the samples (EPI_PARTIAL),
//                                                                              added in split order in float64 -> deterministic
// Every width is padded to a multiple of 4 floats (pads are zero and stay zero) so that all tiles take the 16-byte load path.
// HBM: one row costs (pad(ns) + 3 * sum pad(hidden) + 2 * pad(na)) floats -- 2.6 KB for Humanoid, 16 GB at N = 6.25 M (C4's
// per-GPU share): resident in the 288 GB of HBM for the whole update, nothing is recomputed per Fisher-vector product except
// the tangent / back-prop / gradient GEMMs.
#include "gemm_mfma.h"
#include "cg_device.h"

namespace {

struct PgLay {
    int L;                          // weight layers
    int d[MAXL + 1], dp[MAXL + 1];  // true / padded widths, d[0] = ns ... d[L] = na
    int w_off[MAXL], b_off[MAXL];   // offsets in theta (API layout)
    // padded weight block (floats): per layer W [dp_l][dp_{l+1}], b [dp_{l+1}]; tangent block: per layer VW [(l ? 2 : 1) * dp_l][dp_{l+1}]
    // (V stacked on W for l >= 1), vb [dp_{l+1}]
    size_t oW[MAXL], oB[MAXL], oVW[MAXL], oVB[MAXL], wfloats;
    // per-row buffers (floats per row) and their offsets in the row workspace
    size_t oX, oHT[MAXL], oMU, oU, oD[MAXL];
    size_t row_floats_total;
    // split-K partials of the gradient GEMMs
    int splits, kchunk; size_t part_stride[MAXL], oPart[MAXL], part_floats;
    size_t head_parts;              // per-block partials of the head kernels
};

inline int up4(int v) { return (v + 3) & ~3; }

PgLay pg_layout(const ProblemDesc& pd, long long N) {
    PgLay g = {};
    g.L = pd.pol.n_layers;
    for (int l = 0; l <= g.L; ++l) { g.d[l] = pd.pol.dims[l]; g.dp[l] = up4(g.d[l]); }
    size_t o = 0;
    for (int l = 0; l < g.L; ++l) {
        g.w_off[l] = pd.pol.w_off[l]; g.b_off[l] = pd.pol.b_off[l];
        g.oW[l] = o; o += (size_t)g.dp[l] * g.dp[l + 1];
        g.oB[l] = o; o += g.dp[l + 1];
        g.oVW[l] = o; o += (size_t)(l ? 2 : 1) * g.dp[l] * g.dp[l + 1];
        g.oVB[l] = o; o += g.dp[l + 1];
    }
    g.wfloats = (o + 3) & ~(size_t)3;
    // row-major [N][width] buffers, one after the other (each a multiple of 4 floats wide -> 16-byte aligned rows)
    size_t r = 0;                                            // floats per row, summed
    g.oX = r; r += g.dp[0];
    for (int l = 1; l < g.L; ++l) { g.oHT[l] = r; r += 2 * (size_t)g.dp[l]; }
    g.oMU = r; r += g.dp[g.L];
    g.oU = r; r += g.dp[g.L];
    for (int l = 1; l < g.L; ++l) { g.oD[l] = r; r += g.dp[l]; }
    g.row_floats_total = r;
    // gradient GEMMs contract over the N samples: many splits fill the chip and keep each float32 partial short (<= 2048 rows)
    long long want = std::max<long long>(256, (N + 2047) / 2048);
    want = std::min<long long>(want, 4096);
    g.kchunk = (int)((((N + want - 1) / want) + 15) & ~15LL);
    g.splits = (int)((N + g.kchunk - 1) / g.kchunk);
    size_t p = 0;
    for (int l = 0; l < g.L; ++l) {
        g.part_stride[l] = (((size_t)g.dp[l] * g.dp[l + 1] + g.dp[l + 1]) + 3) & ~(size_t)3;
        g.oPart[l] = p; p += (size_t)g.splits * g.part_stride[l];
    }
    g.part_floats = p;
    g.head_parts = 1024 * 40;                               // up to 1024 blocks x (loss, kl, weight, dls[na <= 32]) doubles as floats x2
    return g;
}

struct PgBufs { float* wts; float* rows; float* part; double* hparts; long long N; };

// buffer b (offset o floats per row, width w) of the row workspace: buffers are stored one after the other, each [N][w]
inline float* rowbuf(const PgBufs& B, const PgLay& g, size_t o_before_rows /* sum of widths of earlier buffers */) {
    return B.rows + o_before_rows * (size_t)B.N;
}

// ---------------------------------------------------------------------------------------------------------------------
// theta / v -> zero-padded weight blocks
__global__ void k_pg_pack(PgLay g, const float* __restrict__ theta, const float* __restrict__ v, float* __restrict__ w) {
    const int l = blockIdx.y;
    const int din = g.d[l], dout = g.d[l + 1], pin = g.dp[l], pout = g.dp[l + 1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pin * pout; i += gridDim.x * blockDim.x) {
        const int r = i / pout, c = i % pout;
        const bool in = (r < din && c < dout);
        const float wv = in ? theta[g.w_off[l] + r * dout + c] : 0.0f;
        w[g.oW[l] + i] = wv;
        if (v != nullptr) {
            const float vv = in ? v[g.w_off[l] + r * dout + c] : 0.0f;
            if (l == 0) w[g.oVW[l] + i] = vv;
            else { w[g.oVW[l] + i] = vv; w[g.oVW[l] + (size_t)pin * pout + i] = wv; }          // [V_l ; W_l]
        }
    }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < pout; c += gridDim.x * blockDim.x) {
        w[g.oB[l] + c] = (c < dout) ? theta[g.b_off[l] + c] : 0.0f;
        if (v != nullptr) w[g.oVB[l] + c] = (c < dout) ? v[g.b_off[l] + c] : 0.0f;
    }
}

__global__ void k_pg_pad_obs(const float* __restrict__ obs, long long N, int ns, int nsp, float* __restrict__ X) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * nsp) return;
    const long long n = i / nsp; const int c = (int)(i % nsp);
    X[i] = (c < ns) ? obs[n * ns + c] : 0.0f;
}

// per-sample head, gradient mode (npo.py:69,75): U = d loss / d mean, per-block partial sums of loss and d loss / d log_std
// part row (doubles): [0] loss, [1] kl, [2] valid weight, [3 .. 3+na) dls
__global__ void __launch_bounds__(256) k_pg_head(int mode, PolK k, int na, int nap, const float* __restrict__ MU, const float* __restrict__ log_std,
                                                 float* __restrict__ U, double* __restrict__ parts) {
    __shared__ double sh[16];
    double acc[35];
    for (int i = 0; i < 3 + na; ++i) acc[i] = 0.0;
    for (long long n = (long long)blockIdx.x * 256 + threadIdx.x; n < k.N; n += (long long)gridDim.x * 256) {
        const bool ok = (k.valid == nullptr || k.valid[n]);
        if (mode == 1) {                                     // FVP: U = tangent(mean) / (s^2 + eps/2) / N
            for (int d = 0; d < nap; ++d) {
                float u = 0.0f;
                if (ok && d < na) { const float ls = fmaxf(log_std[d], LOG_MIN_STD); u = MU[n * nap + d] / (expf(2.0f * ls) + 0.5f * KL_EPS) * k.inv_n; }
                U[n * nap + d] = u;
            }
            if (ok) acc[2] += (double)k.inv_n;
            continue;
        }
        if (k.gm != nullptr) {                               // VJP mode (bptt.hip): the mean-adjoint is an input
            for (int d = 0; d < nap; ++d) U[n * nap + d] = (ok && d < na) ? k.gm[n * na + d] : 0.0f;
            continue;
        }
        float llr = 0.0f, kl = 0.0f, zz[32];
        for (int d = 0; d < na; ++d) {
            zz[d] = 0.0f;
            if (!ok) continue;
            const float ls = fmaxf(log_std[d], LOG_MIN_STD), mu = MU[n * nap + d];
            const float ols = k.old_ls[(size_t)n * k.ls_stride + d], omu = k.old_mean[n * na + d], a = k.act[n * na + d];
            const float z = (a - mu) * expf(-ls), zo = (a - omu) * expf(-ols);
            llr += (ols - ls) + 0.5f * (zo * zo - z * z);
            zz[d] = z;
            if (mode == 2) { const float s2 = expf(2.0f * ls), os2 = expf(2.0f * ols), dm = omu - mu; kl += (dm * dm + os2 - s2) / (2.0f * s2 + KL_EPS) + ls - ols; }
        }
        const float la = ok ? expf(llr) * k.adv[n] : 0.0f;   // lr * adv
        acc[0] -= (double)(la * k.inv_n);                    // surr_loss = -mean(lr * adv)
        if (mode == 2) { acc[1] += (double)(kl * k.inv_n); continue; }
        const float w = -la * k.inv_n;
        for (int d = 0; d < nap; ++d) {
            float u = 0.0f;
            if (d < na) { const float is = expf(-fmaxf(log_std[d], LOG_MIN_STD)); u = w * zz[d] * is; acc[3 + d] += (double)(w * (zz[d] * zz[d] - 1.0f)); }
            U[n * nap + d] = u;
        }
    }
    for (int i = 0; i < 3 + na; ++i) {
        const double t = block_sum(acc[i], sh);
        if (threadIdx.x == 0) parts[(size_t)blockIdx.x * 40 + i] = t;
    }
}

// The same head for the Fisher-vector products (10 of the 12-13 launches of an update), one LANE per (sample, padded action dim): k_pg_head's thread per
// sample walks a row of nap floats per lane -- every load and store instruction touches 64 lines (0.6 TB/s at N = 6.25 M: 2.66 ms of a 18 ms product);
// elementwise, MU and U stream through at the copy rate.  Same expression per element; the valid weight is a sum of identical 1/N values (exact in float64
// whatever the grouping).  part rows as k_pg_head's: only [2] is non-zero.
__global__ void __launch_bounds__(256) k_pg_head_fvp(PolK k, int na, int nap, const float* __restrict__ MU, const float* __restrict__ log_std,
                                                     float* __restrict__ U, double* __restrict__ parts) {
    __shared__ double sh[16];
    double w = 0.0;
    const long long tot = (long long)k.N * nap;
    const long long stride = (long long)gridDim.x * 256, e0 = (long long)blockIdx.x * 256 + threadIdx.x;
    long long n = e0 / nap; int d = (int)(e0 - n * nap);                       // (sample, dim) walk along with e: no 64-bit division per element
    const long long step_n = stride / nap; const int step_d = (int)(stride - step_n * nap);
    for (long long e = e0; e < tot; e += stride, n += step_n, d += step_d) {
        if (d >= nap) { d -= nap; ++n; }
        const bool ok = (k.valid == nullptr || k.valid[n]);
        float u = 0.0f;
        if (ok && d < na) { const float ls = fmaxf(log_std[d], LOG_MIN_STD); u = MU[e] / (expf(2.0f * ls) + 0.5f * KL_EPS) * k.inv_n; }
        U[e] = u;
        if (ok && d == 0) w += (double)k.inv_n;
    }
    const double t = block_sum(w, sh);
    if (threadIdx.x == 0) for (int i = 0; i < 3 + na; ++i) parts[(size_t)blockIdx.x * 40 + i] = (i == 2) ? t : 0.0;
}

// ordered (deterministic) assembly of the result vector in float64:
//   mode 0: out[0] = loss, out[1 + p] = g[p];   mode 1: out[p] = (H v)[p];   mode 2: out[0] = loss, out[1] = kl
__global__ void __launch_bounds__(256) k_pg_assemble(int mode, PgLay g, int P, int n_params, int nblk_head, const float* __restrict__ part,
                                                     const double* __restrict__ hparts, const float* __restrict__ theta, const double* __restrict__ v,
                                                     double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    auto head_sum = [&](int col) {
        double t = 0.0; int b = 0;
        for (; b + 8 <= nblk_head; b += 8) {
            double t8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t8[j] = hparts[(size_t)(b + j) * 40 + col];
#pragma unroll
            for (int j = 0; j < 8; ++j) t += t8[j];
        }
        for (; b < nblk_head; ++b) t += hparts[(size_t)b * 40 + col];
        return t;
    };
    if (mode == 2) { if (p < 2) out[p] = head_sum(p); return; }
    if (p == 0 && mode == 0) out[0] = head_sum(0);
    if (p >= P) return;
    double val = 0.0;
    if (p >= n_params) {                                     // log_std rows
        const int d = p - n_params;
        const double raw = (double)theta[p];
        if (mode == 0) val = (raw > (double)LOG_MIN_STD) ? head_sum(3 + d) : 0.0;
        else {
            const double s2 = exp(2.0 * fmax(raw, (double)LOG_MIN_STD));
            const double c = 4.0 * s2 * (2.0 * s2 - 1e-8) / ((2.0 * s2 + 1e-8) * (2.0 * s2 + 1e-8));
            val = (raw > (double)LOG_MIN_STD) ? c * v[p] * head_sum(2) : 0.0;
        }
    } else {
        int l = 0;
        while (l + 1 < g.L && p >= g.w_off[l + 1]) ++l;
        const int dout = g.d[l + 1], pout = g.dp[l + 1];
        size_t idx;
        if (p < g.b_off[l]) { const int q = p - g.w_off[l]; idx = (size_t)(q / dout) * pout + (q % dout); }
        else idx = (size_t)g.dp[l] * pout + (p - g.b_off[l]);                               // column sums follow the M x N partial
        const float* pl = part + g.oPart[l] + idx;
        int s = 0;
        for (; s + 8 <= g.splits; s += 8) {                    // eight loads in flight, added in split order (one dependent load per split: 90 us at 190 splits)
            float t8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t8[j] = pl[(size_t)(s + j) * g.part_stride[l]];
#pragma unroll
            for (int j = 0; j < 8; ++j) val += (double)t8[j];
        }
        for (; s < g.splits; ++s) val += (double)pl[(size_t)s * g.part_stride[l]];
    }
    out[(mode == 0 ? 1 : 0) + p] = val;
}

__global__ void k_pg_tail(CgTail t) {
    __shared__ double sh[16];
    cg_tail_run(t, sh);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
bool policy_gemm_applicable(const metrpo_ctx* c, long long N) {
    if (c->pol_path == 2) return true;                       // forced (test hook)
    if (c->pol_path == 0) return false;                      // generic forced
    if (f3_active(c)) return false;                          // fused three-hidden-layer kernels (policy_fused3.hip)
    // no fused kernel for this shape, and either enough rows to fill GEMM tiles or a policy large enough that the generic kernels' per-launch cost
    // (thread-per-parameter phases: 1.2 ms per launch for Humanoid's 12 275 weights, whatever N) exceeds the whole GEMM-path update (1.4 ms at N = 64)
    return c->pol_mfma < 0 && (N >= 8192 || c->pd.pol.n_params >= 4096);
}

static int pg_ensure(metrpo_ctx* c, const PgLay& g, long long N, PgBufs* B) {
    const size_t need = (g.wfloats + g.row_floats_total * (size_t)N + g.part_floats) * sizeof(float) + g.head_parts * sizeof(double) + 256;
    if (need > c->pg_cap) {
        if (c->d_pg) { ws_retire(c, c->d_pg); c->d_pg = nullptr; c->pg_cap = 0; }
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_pg, need));
        c->pg_cap = need;
        c->pg_fwd_rows = -1;
    }
    B->hparts = (double*)c->d_pg;
    B->wts = (float*)(B->hparts + g.head_parts);
    B->rows = B->wts + g.wfloats;
    B->part = B->rows + g.row_floats_total * (size_t)N;
    B->N = N;
    return METRPO_OK;
}

// mode 0 grad (+ VJP when c->vjp_gm), 1 fvp, 2 loss/kl.  Writes `out` like k_finalize and then runs `tail` (may be NULL).
int policy_gemm_run(metrpo_ctx* c, int mode, const metrpo_batch* b, const PolK& k0, const float* theta, const float* vf, const double* v64,
                    double* out, const CgTail* tail, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const long long N = b->N;
    if (N > 2000000000LL) return set_err(c, METRPO_EUNSUPPORTED, "policy_gemm: N too large");
    const PgLay g = pg_layout(pd, N);
    PgBufs B;
    int rc = pg_ensure(c, g, N, &B); if (rc) return rc;
    const int L = g.L, na = pd.na, nap = g.dp[L];
    PolK k = k0; k.gm = c->vjp_gm;
    auto RB = [&](size_t o) { return B.rows + o * (size_t)N; };
    float* X = RB(g.oX); float* MU = RB(g.oMU); float* U = RB(g.oU);
    const long long CH = 4000000;                            // rows per GEMM launch (grid.y <= 65535 even with 64-row tiles)

    hipLaunchKernelGGL(k_pg_pack, dim3(8, L), dim3(256), 0, st, g, theta, (mode == 1) ? vf : (const float*)nullptr, B.wts);
    // forward (cached across the Fisher-vector products of one CG solve: same theta, same observations)
    const bool have_fwd = (mode == 1) && c->hcache_on && c->pg_fwd_rows == N && c->pg_fwd_obs == b->d_obs;
    if (!have_fwd) {
        {
            const long long tot = N * g.dp[0];
            hipLaunchKernelGGL(k_pg_pad_obs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b->d_obs, N, g.d[0], g.dp[0], X);
        }
        for (long long r0 = 0; r0 < N; r0 += CH) {
            const int rows = (int)std::min<long long>(CH, N - r0);
            for (int l = 0; l < L - 1; ++l) {
                const float* A = (l == 0) ? X + (size_t)r0 * g.dp[0] : RB(g.oHT[l]) + (size_t)r0 * 2 * g.dp[l];
                const int lda = (l == 0) ? g.dp[0] : 2 * g.dp[l];
                GemmEpi ep = {}; ep.bias = B.wts + g.oB[l];
                gemm_auto<EPI_BIAS_TANH, false, false>(A, 0, lda, B.wts + g.oW[l], 0, g.dp[l + 1], RB(g.oHT[l + 1]) + (size_t)r0 * 2 * g.dp[l + 1], 0,
                                                      2 * g.dp[l + 1], rows, g.dp[l + 1], g.dp[l], 1, ep, st);
            }
        }
        c->pg_fwd_rows = (mode == 0 && c->hcache_on) ? N : -1; c->pg_fwd_obs = b->d_obs;
    }
    for (long long r0 = 0; r0 < N; r0 += CH) {
        const int rows = (int)std::min<long long>(CH, N - r0);
        if (mode != 1) {                                     // mean = H_{L-1} W_{L-1} + b
            GemmEpi ep = {}; ep.bias = B.wts + g.oB[L - 1];
            const float* A = (L == 1) ? X + (size_t)r0 * g.dp[0] : RB(g.oHT[L - 1]) + (size_t)r0 * 2 * g.dp[L - 1];
            gemm_auto<EPI_BIAS_ID, false, false>(A, 0, (L == 1) ? g.dp[0] : 2 * g.dp[L - 1], B.wts + g.oW[L - 1], 0, nap, MU + (size_t)r0 * nap, 0, nap,
                                                rows, nap, g.dp[L - 1], 1, ep, st);
        } else {                                             // tangent chain; the last GEMM leaves the tangent of the mean in MU
            for (int l = 0; l < L; ++l) {
                const float* A = (l == 0) ? X + (size_t)r0 * g.dp[0] : RB(g.oHT[l]) + (size_t)r0 * 2 * g.dp[l];
                const int lda = (l == 0) ? g.dp[0] : 2 * g.dp[l], Kd = (l == 0) ? g.dp[0] : 2 * g.dp[l];
                GemmEpi ep = {}; ep.bias = B.wts + g.oVB[l];
                if (l == L - 1) {
                    gemm_auto<EPI_BIAS_ID, false, false>(A, 0, lda, B.wts + g.oVW[l], 0, nap, MU + (size_t)r0 * nap, 0, nap, rows, nap, Kd, 1, ep, st);
                } else {
                    float* HTn = RB(g.oHT[l + 1]) + (size_t)r0 * 2 * g.dp[l + 1];
                    ep.mask = HTn; ep.ldm = 2 * g.dp[l + 1];                                   // (1 - H_{l+1}^2)
                    gemm_auto<EPI_DTANH, false, false>(A, 0, lda, B.wts + g.oVW[l], 0, g.dp[l + 1], HTn + g.dp[l + 1], 0, 2 * g.dp[l + 1],
                                                       rows, g.dp[l + 1], Kd, 1, ep, st);
                }
            }
        }
    }
    // per-sample head
    const bool head_elem = (mode == 1 && k.gm == nullptr);
    const int nblk = head_elem ? (int)std::min<long long>(1024, (N * nap + 2047) / 2048) : (int)std::min<long long>(1024, (N + 255) / 256);
    if (head_elem) hipLaunchKernelGGL(k_pg_head_fvp, dim3(nblk), dim3(256), 0, st, k, na, nap, MU, theta + pd.pol.n_params, U, B.hparts);
    else hipLaunchKernelGGL(k_pg_head, dim3(nblk), dim3(256), 0, st, mode, k, na, nap, MU, theta + pd.pol.n_params, U, B.hparts);
    if (mode != 2) {
        // back-prop D_l = (D_{l+1} W_l^T) * (1 - H_l^2), l = L-1 .. 1
        for (long long r0 = 0; r0 < N; r0 += CH) {
            const int rows = (int)std::min<long long>(CH, N - r0);
            for (int l = L - 1; l >= 1; --l) {
                const float* Dn = (l == L - 1) ? U + (size_t)r0 * nap : RB(g.oD[l + 1]) + (size_t)r0 * g.dp[l + 1];
                GemmEpi ep = {};
                ep.mask = RB(g.oHT[l]) + (size_t)r0 * 2 * g.dp[l]; ep.ldm = 2 * g.dp[l];
                gemm_auto<EPI_DTANH, false, true>(Dn, 0, g.dp[l + 1], B.wts + g.oW[l], 0, g.dp[l + 1], RB(g.oD[l]) + (size_t)r0 * g.dp[l], 0, g.dp[l],
                                                  rows, g.dp[l], g.dp[l + 1], 1, ep, st);
            }
        }
        // gradients G_l = A_l^T D_{l+1} (+ column sums), split over the samples
        for (int l = 0; l < L; ++l) {
            const float* A = (l == 0) ? X : RB(g.oHT[l]);
            const int lda = (l == 0) ? g.dp[0] : 2 * g.dp[l];
            const float* Dn = (l == L - 1) ? U : RB(g.oD[l + 1]);
            GemmEpi ep = {};
            ep.part = B.part + g.oPart[l]; ep.stridePart = (long long)g.part_stride[l]; ep.splits = g.splits; ep.kchunk = g.kchunk;
            gemm_mfma_launch<1, 1, EPI_PARTIAL, true, false>(A, 0, lda, Dn, 0, g.dp[l + 1], nullptr, 0, g.dp[l + 1], g.dp[l], g.dp[l + 1], (int)N, 1, ep, st);
        }
    }
    hipLaunchKernelGGL(k_pg_assemble, dim3((pd.P + 1 + 255) / 256), dim3(256), 0, st, mode, g, pd.P, pd.pol.n_params, nblk, B.part, B.hparts, theta, v64, out);
    if (tail && tail->op != 0) hipLaunchKernelGGL(k_pg_tail, dim3(1), dim3(1024), 0, st, *tail);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
