// Policy-side kernels of the NPO/TRPO update (algos/npo.py:68-111; [rllab] ConjugateGradientOptimizer,
// PerlmutterHvp, DiagonalGaussian):
//   k_loss_grad  -- surrogate loss + flat gradient          (f_loss / f_grad)
//   k_fvp        -- Hessian(mean_kl) . v  (Gauss-Newton form, exact at theta_old; derivation in DESIGN.md)
//   k_loss_kl    -- surrogate loss + mean KL at a trial theta (f_loss_constraint)
//   k_finalize   -- fixed-order (deterministic) float64 reduction of the per-block partial rows
//
// Generic VALU formulation: a block of PT threads owns tiles of PT samples.  Phase A is thread-per-
// sample (forward / tangent / back-prop in LDS columns, weights via scalar loads); phase B is
// thread-per-parameter (weight-gradient outer products over the tile, rows padded to PT+1 floats so
// that lanes holding different units hit different LDS banks).  Each block accumulates into its own
// row of a global partial buffer; k_finalize sums rows in block order -> bitwise reproducible.
#include "device_common.h"
#ifdef FIN_TIMING
__device__ unsigned long long g_fin_phase[512][8];
#define CG_MARK(i) { if (threadIdx.x == 0 && blockIdx.x < 512) g_fin_phase[blockIdx.x][i] = __builtin_readcyclecounter(); }
#endif
#include "cg_device.h"



// offsets (in rows) of the per-layer activation buffers h_0 (= input) .. h_{L-1}; h_L (mean) lives in DM
__device__ __forceinline__ int hrow(const NetDesc& net, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += net.dims[i];
    return o;
}

// forward pass storing every layer input; returns nothing, mean is written to DM rows [0,na)
template <int PT>
__device__ __forceinline__ void forward_store(const NetDesc& net, const float* __restrict__ th, float* H, float* DM, int tid) {
    constexpr int PLD = PT + 1;
    for (int l = 0; l < net.n_layers; ++l) {
        float* dst = (l == net.n_layers - 1) ? DM : (H + hrow(net, l + 1) * PLD);
        dense_col(th + net.w_off[l], th + net.b_off[l], net.dims[l], net.dims[l + 1], net.act[l], H + hrow(net, l) * PLD,
                  dst, PLD, tid);
    }
}

template <int PT>
__device__ __forceinline__ bool load_obs_tile(const PolK& k, int ns, long long base, float* H, int tid) {
    constexpr int PLD = PT + 1;
    const long long n = base + tid;
    const bool ok = (n < k.N) && (k.valid == nullptr || k.valid[n]);
    for (int i = 0; i < ns; ++i) H[i * PLD + tid] = (n < k.N) ? k.obs[n * ns + i] : 0.0f;
    return ok;
}

// Phase B for one layer: part[w_off + i*n_out + j] += sum_n h_l[i][n] * d[j][n];  part[b_off + j] += sum_n d[j][n]
template <int PT>
__device__ __forceinline__ void accum_layer(const NetDesc& net, int l, const float* Hl, const float* D, float* part, int tid) {
    constexpr int PLD = PT + 1;
    const int n_in = net.dims[l], n_out = net.dims[l + 1];
    for (int p = tid; p < n_in * n_out; p += PT) {
        const int i = p / n_out, j = p - i * n_out;
        const float* hr = Hl + i * PLD;
        const float* dr = D + j * PLD;
        float s = 0.0f;
#pragma unroll 8
        for (int n = 0; n < PT; ++n) s = fmaf(hr[n], dr[n], s);
        part[net.w_off[l] + p] += s;
    }
    for (int j = tid; j < n_out; j += PT) {
        const float* dr = D + j * PLD;
        float s = 0.0f;
#pragma unroll 8
        for (int n = 0; n < PT; ++n) s += dr[n];
        part[net.b_off[l] + j] += s;
    }
}

// delta_l[i] = (sum_j W_l[i][j] * delta_{l+1}[j]) * (1 - h_l[i]^2), written in place over h_l (tanh hidden layers)
template <int PT>
__device__ __forceinline__ void backprop_layer(const NetDesc& net, int l, const float* __restrict__ th, float* Hl, const float* D, int tid) {
    constexpr int PLD = PT + 1;
    const int n_in = net.dims[l], n_out = net.dims[l + 1];
    const float* __restrict__ W = th + net.w_off[l];
    for (int i = 0; i < n_in; ++i) {
        float s = 0.0f;
        const float* __restrict__ wr = W + (size_t)i * n_out;
        for (int j = 0; j < n_out; ++j) s = fmaf(wr[j], D[j * PLD + tid], s);
        const float h = Hl[i * PLD + tid];
        Hl[i * PLD + tid] = s * (1.0f - h * h);
    }
}

// shared tail of grad and fvp: DM holds d(objective)/d(mean) per sample (already scaled, zero if invalid)
template <int PT>
__device__ __forceinline__ void backward_accumulate(const NetDesc& net, const float* __restrict__ th, float* H, float* DM, float* part, int tid) {
    constexpr int PLD = PT + 1;
    const float* D = DM;
    for (int l = net.n_layers - 1; l >= 0; --l) {
        float* Hl = H + hrow(net, l) * PLD;
        __syncthreads();
        accum_layer<PT>(net, l, Hl, D, part, tid);
        __syncthreads();
        if (l > 0) { backprop_layer<PT>(net, l, th, Hl, D, tid); D = Hl; }
    }
}

// partial row layout: [0, P) gradient in theta order (log_std slots at pol.n_params..P), then
//   [P] = loss (or kl-side scalar), [P+1] = second scalar, [P+2] = valid-sample weight (count*inv_n)
#define PART_EXTRA 3

template <int PT>
__global__ void __launch_bounds__(PT) k_loss_grad(ProblemDesc pd, PolK k, const float* __restrict__ theta, float* __restrict__ partials) {
    constexpr int PLD = PT + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[16];
    const NetDesc& net = pd.pol;
    const int tid = threadIdx.x, ns = pd.ns, na = pd.na, P = pd.P;
    int hrows = 0;
    for (int l = 0; l < net.n_layers; ++l) hrows += net.dims[l];
    float* H = lds;
    float* DM = H + hrows * PLD;
    float* part = partials + (size_t)blockIdx.x * (P + PART_EXTRA);
    for (int p = tid; p < P + PART_EXTRA; p += PT) part[p] = 0.0f;
    const float* __restrict__ raw_ls = theta + net.n_params;
    double loss_acc = 0.0;
    float dls_acc[32];            // na <= 32 enforced by the launcher
#pragma unroll
    for (int d = 0; d < 32; ++d) dls_acc[d] = 0.0f;
    for (long long base = (long long)blockIdx.x * PT; base < k.N; base += (long long)gridDim.x * PT) {
        __syncthreads();
        const bool ok = load_obs_tile<PT>(k, ns, base, H, tid);
        forward_store<PT>(net, theta, H, DM, tid);
        const long long n = base + tid;
        if (k.gm != nullptr) {                                  // VJP mode (bptt.hip): d objective / d mean supplied
            for (int d = 0; d < na; ++d) DM[d * PLD + tid] = ok ? k.gm[n * na + d] : 0.0f;
            backward_accumulate<PT>(net, theta, H, DM, part, tid);
            continue;
        }
        float w = 0.0f;
        if (ok) {
            float llr = 0.0f;                                   // logli_new - logli_old
            for (int d = 0; d < na; ++d) {
                const float ls = fmaxf(raw_ls[d], LOG_MIN_STD), ols = k.old_ls[(size_t)n * k.ls_stride + d];
                const float a = k.act[n * na + d];
                const float z = (a - DM[d * PLD + tid]) * expf(-ls);
                const float zo = (a - k.old_mean[n * na + d]) * expf(-ols);
                llr += (ols - ls) + 0.5f * (zo * zo - z * z);
            }
            const float lr = expf(llr);                          // likelihood_ratio_sym (npo.py:69)
            const float la = lr * k.adv[n];
            loss_acc -= (double)la * (double)k.inv_n;            // surr_loss = -mean(lr*adv) (npo.py:75)
            w = -la * k.inv_n;
        }
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            if (d < na) {
                const float ls = fmaxf(raw_ls[d], LOG_MIN_STD);
                const float inv_std = expf(-ls);
                const float z = ok ? (k.act[n * na + d] - DM[d * PLD + tid]) * inv_std : 0.0f;
                DM[d * PLD + tid] = w * z * inv_std;             // d loss / d mean
                dls_acc[d] += w * (z * z - 1.0f);                // d loss / d log_std
            }
        }
        backward_accumulate<PT>(net, theta, H, DM, part, tid);
    }
    // block-reduce the per-thread scalars into the partial row
    const double l = block_sum(loss_acc, red);
    if (tid == 0) part[P] = (float)l;
    for (int d = 0; d < na; ++d) {
        const double s = block_sum((double)dls_acc[d], red);
        if (tid == 0) part[net.n_params + d] = (raw_ls[d] > LOG_MIN_STD) ? (float)s : 0.0f;
    }
}

// tangent forward: dpre_{l+1} = dh_l W_l + h_l V_l + vb_l ; dh_{l+1} = dpre * (1 - h_{l+1}^2)
template <int PT>
__device__ __forceinline__ void tangent_layer(const NetDesc& net, int l, const float* __restrict__ th, const float* __restrict__ v,
                                              const float* Hl, const float* dHl /*nullptr for l==0*/, const float* Hn /* h_(l+1), or nullptr for the output layer */,
                                              float* dst, int tid) {
    constexpr int PLD = PT + 1;
    const int n_in = net.dims[l], n_out = net.dims[l + 1];
    const float* __restrict__ W = th + net.w_off[l];
    const float* __restrict__ V = v + net.w_off[l];
    const float* __restrict__ vb = v + net.b_off[l];
    for (int j0 = 0; j0 < n_out; j0 += DENSE_JB) {
        float acc[DENSE_JB];
        const int nj = min(DENSE_JB, n_out - j0);
#pragma unroll
        for (int jj = 0; jj < DENSE_JB; ++jj) acc[jj] = (jj < nj) ? vb[j0 + jj] : 0.0f;
        for (int i = 0; i < n_in; ++i) {
            const float h = Hl[i * PLD + tid];
            const float dh = (dHl != nullptr) ? dHl[i * PLD + tid] : 0.0f;
#pragma unroll
            for (int jj = 0; jj < DENSE_JB; ++jj)
                if (jj < nj) acc[jj] = fmaf(h, V[(size_t)i * n_out + j0 + jj], fmaf(dh, W[(size_t)i * n_out + j0 + jj], acc[jj]));
        }
#pragma unroll
        for (int jj = 0; jj < DENSE_JB; ++jj)
            if (jj < nj) {
                float o = acc[jj];
                if (Hn != nullptr) { const float hn = Hn[(j0 + jj) * PLD + tid]; o *= (1.0f - hn * hn); }
                dst[(j0 + jj) * PLD + tid] = o;
            }
    }
}

template <int PT>
__global__ void __launch_bounds__(PT) k_fvp(ProblemDesc pd, PolK k, const float* __restrict__ theta, const float* __restrict__ v,
                                            float* __restrict__ partials) {
    constexpr int PLD = PT + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[16];
    const NetDesc& net = pd.pol;
    const int tid = threadIdx.x, ns = pd.ns, na = pd.na, P = pd.P, L = net.n_layers;
    int hrows = 0;
    for (int l = 0; l < L; ++l) hrows += net.dims[l];
    float* H = lds;                       // h_0..h_{L-1}
    float* DM = H + hrows * PLD;          // mean, then u
    float* dH = DM + na * PLD;            // tangents dh_1..dh_{L-1}, indexed with hrow(l) - dims[0]
    float* part = partials + (size_t)blockIdx.x * (P + PART_EXTRA);
    for (int p = tid; p < P + PART_EXTRA; p += PT) part[p] = 0.0f;
    const float* __restrict__ raw_ls = theta + net.n_params;
    double wsum = 0.0;
    for (long long base = (long long)blockIdx.x * PT; base < k.N; base += (long long)gridDim.x * PT) {
        __syncthreads();
        const bool ok = load_obs_tile<PT>(k, ns, base, H, tid);
        forward_store<PT>(net, theta, H, DM, tid);
        for (int l = 0; l < L; ++l) {
            const float* Hl = H + hrow(net, l) * PLD;
            const float* dHl = (l == 0) ? nullptr : dH + (hrow(net, l) - net.dims[0]) * PLD;
            const float* Hn = (l == L - 1) ? nullptr : H + hrow(net, l + 1) * PLD;
            float* dst = (l == L - 1) ? DM : dH + (hrow(net, l + 1) - net.dims[0]) * PLD;
            tangent_layer<PT>(net, l, theta, v, Hl, dHl, Hn, dst, tid);
        }
        for (int d = 0; d < na; ++d) {
            const float ls = fmaxf(raw_ls[d], LOG_MIN_STD);
            const float s2 = expf(2.0f * ls);
            // d2 KL / d mean^2 = 2 / (2 s^2 + eps) = 1 / (s^2 + eps/2)
            DM[d * PLD + tid] = ok ? DM[d * PLD + tid] / (s2 + 0.5f * KL_EPS) * k.inv_n : 0.0f;
        }
        if (ok) wsum += (double)k.inv_n;
        backward_accumulate<PT>(net, theta, H, DM, part, tid);
    }
    const double wtot = block_sum(wsum, red);
    if (tid == 0) part[P + 2] = (float)wtot;
}

template <int PT>
__global__ void __launch_bounds__(PT) k_loss_kl(ProblemDesc pd, PolK k, const float* __restrict__ theta, float* __restrict__ partials) {
    constexpr int PLD = PT + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double red[16];
    if (k.skip != nullptr && k.skip[0] >= 0.0) return;        // speculative line-search trial after the search stopped
    const NetDesc& net = pd.pol;
    const int tid = threadIdx.x, ns = pd.ns, na = pd.na;
    float* S = lds;
    float* A = S + ns * PLD;
    float* Bq = A + net.max_width * PLD;
    const float* __restrict__ raw_ls = theta + net.n_params;
    double loss_acc = 0.0, kl_acc = 0.0;
    for (long long base = (long long)blockIdx.x * PT; base < k.N; base += (long long)gridDim.x * PT) {
        const bool ok = load_obs_tile<PT>(k, ns, base, S, tid);
        const float* m = mlp_col(net, theta, S, A, Bq, PLD, tid);
        if (ok) {
            const long long n = base + tid;
            float llr = 0.0f, kl = 0.0f;
            for (int d = 0; d < na; ++d) {
                const float ls = fmaxf(raw_ls[d], LOG_MIN_STD), ols = k.old_ls[(size_t)n * k.ls_stride + d];
                const float mu = m[d * PLD + tid], omu = k.old_mean[n * na + d], a = k.act[n * na + d];
                const float z = (a - mu) * expf(-ls), zo = (a - omu) * expf(-ols);
                llr += (ols - ls) + 0.5f * (zo * zo - z * z);
                const float s2 = expf(2.0f * ls), os2 = expf(2.0f * ols);
                const float dm = omu - mu;
                kl += (dm * dm + os2 - s2) / (2.0f * s2 + KL_EPS) + ls - ols;      // DiagonalGaussian.kl_sym
            }
            loss_acc -= (double)(expf(llr) * k.adv[n]) * (double)k.inv_n;
            kl_acc += (double)kl * (double)k.inv_n;
        }
    }
    const double l = block_sum(loss_acc, red);
    const double q = block_sum(kl_acc, red);
    if (tid == 0) { partials[blockIdx.x * 2] = (float)l; partials[blockIdx.x * 2 + 1] = (float)q; }
}

// out[p] = sum over partial rows (fixed order) of column col(p), float64.
// mode 0: grad -> out[0] = loss (column P), out[1+p] = g[p]
// mode 1: fvp  -> out[p] = Hv[p] for the mean net; log_std rows get c(s) * v_ls * weight (column P+2)
// mode 2: loss/kl -> out[0], out[1] from columns (lk_col, lk_col+1)
#define FIN_C 32
#ifdef FIN_TIMING      // developer instrumentation (SRC=policy_update.hip tools/build_variant.sh ftiming -DFIN_TIMING; tools/fin_phases.py)
#define FT_MARK(i) { if (threadIdx.x == 0 && tail.op == 1 && blockIdx.x < 512) g_fin_phase[blockIdx.x][i] = __builtin_readcyclecounter(); }
extern "C" int32_t metrpo_debug_fin_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_phase), sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : -1; }
#else
#define FT_MARK(i)
#endif
__global__ void __launch_bounds__(1024) k_finalize(ProblemDesc pd, int mode, int nrows, int stride, int lk_col,
                                                   const float* __restrict__ partials, const float* __restrict__ theta,
                                                   const double* __restrict__ v, double* __restrict__ out, CgTail tail, XchgK xc) {
    // block = FIN_C output columns x (1024 / FIN_C) row slices (latency-bound sum: many small blocks); slice s adds rows
    // s, s+NSL, ... and the slice sums are added in slice order: deterministic.  32 columns = one 128-byte line per row read.
    constexpr int NSL = 1024 / FIN_C;
    __shared__ double sh[NSL][FIN_C + 1];
    if (tail.op == 4 && tail.ls[0] >= 0.0) {                  // speculative line-search trial after the search stopped (every block reads the same cell)
        // a sharded run keeps its exchanges in step: the slot protocol (xchg_device.h) counts on every sequence number being used by every rank
        if (xc.world > 1 && blockIdx.x == 0 && threadIdx.x < 2) { xchg_push(xc, (int)threadIdx.x, 0.0); (void)xchg_pull_sum(xc, (int)threadIdx.x); }
        if (tail.pub_dst != nullptr && blockIdx.x == 0) ls_publish(tail.scal, tail.pub_dst, tail.pub_stamp);      // the search's outcome still has to reach the host
        return;
    }
    FT_MARK(0)
    CgPre pre;
    cg_prefetch(tail, pre);                 // every block (nobody knows yet who arrives last); these vectors are not written by this launch
    const int P = pd.P;
    const int nout = (mode == 0) ? P + 1 : (mode == 1) ? P : 2;
    const int lc = threadIdx.x % FIN_C, sl = threadIdx.x / FIN_C;
    const int p = blockIdx.x * FIN_C + lc;
    int col = p;
    if (mode == 0) col = (p == 0) ? P : p - 1;
    if (mode == 2) col = lk_col + p;
    const bool lsrow = (mode == 1 && p >= pd.pol.n_params && p < nout);
    if (lsrow) col = P + 2;                                   // valid-sample weight column
    double a = 0.0;
    if (p < nout) {
        int b = sl;
        for (; b + 3 * NSL < nrows; b += 4 * NSL) {             // four independent loads in flight per thread
            const float v0 = partials[(size_t)b * stride + col], v1 = partials[(size_t)(b + NSL) * stride + col];
            const float v2 = partials[(size_t)(b + 2 * NSL) * stride + col], v3 = partials[(size_t)(b + 3 * NSL) * stride + col];
            a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
        }
        for (; b < nrows; b += NSL) a += (double)partials[(size_t)b * stride + col];
    }
    FT_MARK(1)
    sh[sl][lc] = a;
    __syncthreads();
    if (sl == 0 && p < nout) {
        double t = 0.0;
        for (int w = 0; w < NSL; ++w) t += sh[w][lc];
        if (lsrow) {
            // Hessian of mean KL w.r.t. log_std at theta_old: 4 s^2 (2 s^2 - eps) / (2 s^2 + eps)^2  (-> 2 as eps -> 0)
            const double raw = (double)theta[p];
            const double s2 = exp(2.0 * fmax(raw, (double)LOG_MIN_STD));
            const double c = 4.0 * s2 * (2.0 * s2 - 1e-8) / ((2.0 * s2 + 1e-8) * (2.0 * s2 + 1e-8));
            t = (raw > (double)LOG_MIN_STD) ? c * v[p] * t : 0.0;
        }
        // a fused tail reads `out` in ANOTHER workgroup (the last to arrive): write-through (sc1) stores, drained before the arrival ticket, instead of a
        // cache-wide release per workgroup (buffer_wbl2 x 45 workgroups at C1, x 391 for the 100-50-25 policy: 25-49 us of that reduction).
        // This relies on gfx942 / gfx950 lowering a relaxed agent-scope atomic store to an sc1 write-through store (visible beyond this XCD's L2 once vmcnt
        // drains).  EVERYTHING the closing workgroup reads that another workgroup of this launch wrote must travel this way -- today exactly: `out` (here) and the
        // exchange packets (xchg_push: agent-scope stores).  theta, v, partials and the CG state are written by EARLIER launches.  A plain store added to that list
        // would be a silent stale read across XCDs: on any other target the build stops here instead of guessing.
#if !defined(__gfx942__) && !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "k_finalize's fence-free hand-over is written for gfx942 / gfx950 (sc1 write-through stores); other targets need fence(release, agent) before the ticket"
#endif
        if (tail.op != 0 || xc.world > 1) __hip_atomic_store(out + p, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[p] = t;
        if (xc.world > 1) xchg_push(xc, p, t);                // sharded run: this rank's share goes straight into every rank's receive slot
    }
    if (tail.op == 0 && xc.world <= 1) return;
    // ---- fused tail: the last block to arrive owns the complete `out` vector: it adds the ranks' shares (one-shot exchange,
    //      xchg_device.h; the packets of the other blocks have been under way since they were produced) and runs the CG vector step ----
    __shared__ unsigned int s_last;
    __shared__ double cgsh[16];
    FT_MARK(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FT_MARK(3)
    if (threadIdx.x == 0) {
        const unsigned int tk = __hip_atomic_fetch_add(tail.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (tk == gridDim.x - 1) ? 1u : 0u;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    FT_MARK(4)
    if (!s_last) return;
    if (xc.world > 1) {
        for (int i = threadIdx.x; i < nout; i += blockDim.x) out[i] = xchg_pull_sum(xc, i);
        __syncthreads();
    }
    if (tail.op != 0) cg_tail_run(tail, cgsh, &pre);
    if (threadIdx.x == 0) __hip_atomic_store(tail.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

__global__ void k_d2f(const double* __restrict__ in, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---------------------------------------------------------------------------------------------
static int ensure_partials(metrpo_ctx* c, int nrows) {
    const size_t need = (size_t)nrows * (c->pd.P + PART_EXTRA);
    if (need > c->partials_cap) {
        ws_retire(c, c->d_partials);
        c->d_partials = nullptr; c->partials_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_partials, need * sizeof(float)));
        c->partials_cap = need;
    }
    return METRPO_OK;
}

static int fill_polk(metrpo_ctx* c, const metrpo_batch* b, PolK* k, bool need_targets) {
    if (!b || !b->d_obs) return set_err(c, METRPO_ENULL, "batch/d_obs is NULL");
    if (b->N <= 0) return set_err(c, METRPO_EINVAL, "batch N must be positive");
    if (c->pd.na > 32) return set_err(c, METRPO_EUNSUPPORTED, "na > 32");
    if (need_targets && (!b->d_act || !b->d_adv || !b->d_old_mean || !b->d_old_log_std))
        return set_err(c, METRPO_ENULL, "batch pointer is NULL");
    k->obs = b->d_obs; k->act = b->d_act; k->adv = b->d_adv; k->old_mean = b->d_old_mean; k->old_ls = b->d_old_log_std;
    k->ls_stride = b->old_log_std_stride; k->valid = b->d_valid; k->N = b->N; k->inv_n = (float)b->inv_n_global;
    k->gm = nullptr; k->img_map = nullptr; k->hcache = nullptr; k->imgval = nullptr; k->skip = nullptr;
    return METRPO_OK;
}

static void finalize(metrpo_ctx* c, int mode, int nrows, int stride, int lk_col, const double* v, double* out, hipStream_t st,
                     const CgTail* tail = nullptr) {
    const int nout = (mode == 0) ? c->pd.P + 1 : (mode == 1) ? c->pd.P : 2;
    CgTail none; none.op = 0; none.ticket = c->d_ticket; none.vpos = nullptr; none.imgval = nullptr; none.ls = nullptr; none.pub_dst = nullptr;
    // inside a fused update of a sharded run (run_trpo_update raises xg_fuse) the reduction carries the cross-rank sum in its tail
    const XchgK xc = (c->xg_fuse && c->xg_world > 1) ? xchg_next(c) : xchg_none();
    hipLaunchKernelGGL(k_finalize, dim3((nout + FIN_C - 1) / FIN_C), dim3(1024), 0, st, c->pd, mode, nrows, stride, lk_col,
                       c->d_partials, c->d_theta, v, out, tail ? *tail : none, xc);
}

// generic kernels: pick the largest sample tile (threads per block) whose LDS columns fit
template <int PT>
static int launch_generic(metrpo_ctx* c, int mode, const PolK& k, const float* theta, const float* vf, int* nrows, hipStream_t st) {
    const NetDesc& net = c->pd.pol;
    int hrows = 0; for (int l = 0; l < net.n_layers; ++l) hrows += net.dims[l];
    size_t rows = (mode == 0) ? hrows + c->pd.na : (mode == 1) ? hrows + c->pd.na + (hrows - net.dims[0])
                                                              : (size_t)c->pd.ns + 2 * net.max_width;
    const size_t sh = rows * (PT + 1) * sizeof(float);
    if (sh > 160 * 1024) return METRPO_EUNSUPPORTED;
    const long long tiles = (k.N + PT - 1) / PT;
    const int g = (int)std::max<long long>(1, std::min<long long>(tiles, (long long)c->n_sm * 2));
    int rc = ensure_partials(c, g); if (rc) return rc;
    *nrows = g;
    if (mode == 0) {
        if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_loss_grad<PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_loss_grad<PT>, dim3(g), dim3(PT), sh, st, c->pd, k, theta, c->d_partials);
    } else if (mode == 1) {
        if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_fvp<PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_fvp<PT>, dim3(g), dim3(PT), sh, st, c->pd, k, theta, vf, c->d_partials);
    } else {
        if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_loss_kl<PT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        hipLaunchKernelGGL(k_loss_kl<PT>, dim3(g), dim3(PT), sh, st, c->pd, k, theta, c->d_partials);
    }
    return METRPO_OK;
}

// runs mode on the fastest available path; on return the partial rows are in c->d_partials
static int run_mode(metrpo_ctx* c, int mode, const metrpo_batch* b, const PolK& k, const float* theta, const float* vf,
                    int* nrows, int* stride, int* lk_col, hipStream_t st) {
    const int P = c->pd.P;
    if (c->pol_mfma >= 0) {
        const long long tiles = (b->N + 15) / 16;
        // one 8-wave block per CU = 2 waves per SIMD (measured in round 1: 2 and 3 waves per SIMD run at the same speed, 1 and 4
        // are slower) and only n_sm partial rows for k_finalize
        // (the loss + KL evaluation of the line search needs no transpose tiles and half the registers: two blocks per CU)
        // small batches (the params files' N = 50 000 - 60 000: 1.5 tiles per wave on a full grid): fewer, fuller blocks -- a launch's fixed cost does not
        // shrink with the tile count, but k_finalize reads one partial row per block (upd_tiles_per_wave, api.hip)
        const long long per_block = 8ll * std::max(1, c->upd_tiles_per_wave);
        const int g = (int)std::max<long long>(1, std::min<long long>((tiles + per_block - 1) / per_block, (long long)c->n_sm * (mode == 2 ? 2 : 1)));
        int rc = ensure_partials(c, g); if (rc) return rc;
        *nrows = g; *stride = P + PART_EXTRA; *lk_col = P;
        c->ls_skip = k.skip;
        rc = policy_mfma_launch(c, c->pol_mfma, mode, b, theta, vf, c->d_partials, g, st);
        c->ls_skip = nullptr;
        return rc;
    }
    if (f3_active(c)) {
        // policy_fused3.hip: one block per CU; the back-prop kernel runs 4 waves per block (one per SIMD, 272 accumulator registers each), the
        // forward / tangent kernels 8; all write the same `g` partial rows
        const long long tiles = (b->N + 15) / 16;
        const long long per_block = 4ll * std::max(1, c->upd_tiles_per_wave);
        const int g = (int)std::max<long long>(1, std::min<long long>((tiles + per_block - 1) / per_block, (long long)c->n_sm));
        int rc = ensure_partials(c, g); if (rc) return rc;
        *nrows = g; *stride = P + PART_EXTRA; *lk_col = P;
        c->ls_skip = k.skip;
        rc = policy_f3_launch(c, mode, b, theta, vf, c->d_partials, g, st);
        c->ls_skip = nullptr;
        return rc;
    }
    *stride = (mode == 2) ? 2 : P + PART_EXTRA; *lk_col = 0;
    int rc = launch_generic<128>(c, mode, k, theta, vf, nrows, st);
    if (rc == METRPO_EUNSUPPORTED) rc = launch_generic<64>(c, mode, k, theta, vf, nrows, st);
    if (rc == METRPO_EUNSUPPORTED) rc = launch_generic<32>(c, mode, k, theta, vf, nrows, st);
    if (rc == METRPO_EUNSUPPORTED) return set_err(c, rc, "policy too wide for the update kernels' LDS tile");
    return rc;
}

int launch_loss_grad(metrpo_ctx* c, const metrpo_batch* b, double* out, hipStream_t st, const CgTail* tail) {
    PolK k; int rc = fill_polk(c, b, &k, true); if (rc) return rc;
    if (policy_gemm_applicable(c, b->N)) return policy_gemm_run(c, 0, b, k, c->d_theta, nullptr, nullptr, out, tail, st);
    int nrows, stride, lk;
    if ((rc = run_mode(c, 0, b, k, c->d_theta, nullptr, &nrows, &stride, &lk, st))) return rc;
    finalize(c, 0, nrows, stride, lk, nullptr, out, st, tail);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

// sum_n J_policy(obs_n)^T gm_n -> out[1 .. P] (out[0] = 0): the gradient kernels with the mean-adjoint supplied (bptt.hip)
int launch_policy_vjp(metrpo_ctx* c, const float* obs, const float* gm, long long N, double* out, hipStream_t st) {
    if (!obs || !gm || !out) return set_err(c, METRPO_ENULL, "policy_vjp: NULL pointer");
    if (c->pd.na > 32) return set_err(c, METRPO_EUNSUPPORTED, "na > 32");
    metrpo_batch b = {};
    b.d_obs = obs; b.N = N; b.inv_n_global = 1.0;
    PolK k = {};
    k.obs = obs; k.N = N; k.inv_n = 1.0f; k.gm = gm;
    int nrows, stride, lk;
    c->vjp_gm = gm;
    if (policy_gemm_applicable(c, N)) {
        const int rcg = policy_gemm_run(c, 0, &b, k, c->d_theta, nullptr, nullptr, out, nullptr, st);
        c->vjp_gm = nullptr;
        return rcg;
    }
    const int rc = run_mode(c, 0, &b, k, c->d_theta, nullptr, &nrows, &stride, &lk, st);
    c->vjp_gm = nullptr;
    if (rc) return rc;
    finalize(c, 0, nrows, stride, lk, nullptr, out, st);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_fvp(metrpo_ctx* c, const metrpo_batch* b, const double* v, double* hv, hipStream_t st) {
    if (!v || !hv) return set_err(c, METRPO_ENULL, "v/hv is NULL");
    const int P = c->pd.P;
    hipLaunchKernelGGL(k_d2f, dim3((P + 127) / 128), dim3(128), 0, st, v, c->d_vf, P);
    return launch_fvp_f32(c, b, c->d_vf, v, hv, st);
}

int launch_fvp_f32(metrpo_ctx* c, const metrpo_batch* b, const float* vf, const double* v, double* hv, hipStream_t st) {
    return launch_fvp_tail(c, b, vf, v, hv, nullptr, st);
}

int launch_fvp_tail(metrpo_ctx* c, const metrpo_batch* b, const float* vf, const double* v, double* hv, const CgTail* tail, hipStream_t st) {
    PolK k; int rc = fill_polk(c, b, &k, false); if (rc) return rc;
    if (policy_gemm_applicable(c, b->N)) return policy_gemm_run(c, 1, b, k, c->d_theta, vf, v, hv, tail, st);
    int nrows, stride, lk;
    const bool timed = ctx_opt(c, OPT_TIME_FVP) != nullptr && c->fvp_ev_n + 2 <= 32;      // diagnostics: metrpo_debug_fvp_us
    if (timed) {
        for (; c->fvp_ev_made < 32; ++c->fvp_ev_made) HIP_TRY(c, hipEventCreate(&c->fvp_ev[c->fvp_ev_made]));
        HIP_TRY(c, hipEventRecord(c->fvp_ev[c->fvp_ev_n], st));
    }
    if ((rc = run_mode(c, 1, b, k, c->d_theta, vf, &nrows, &stride, &lk, st))) return rc;
    if (timed) { HIP_TRY(c, hipEventRecord(c->fvp_ev[c->fvp_ev_n + 1], st)); c->fvp_ev_n += 2; }
    finalize(c, 1, nrows, stride, lk, v, hv, st, tail);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_loss_kl(metrpo_ctx* c, const metrpo_batch* b, const float* theta, double* out, hipStream_t st, const CgTail* decide) {
    PolK k; int rc = fill_polk(c, b, &k, true); if (rc) return rc;
    if (policy_gemm_applicable(c, b->N)) {
        if (decide) return set_err(c, METRPO_EUNSUPPORTED, "device-side line search: not on the GEMM update path");
        return policy_gemm_run(c, 2, b, k, theta ? theta : c->d_theta, nullptr, nullptr, out, nullptr, st);
    }
    if (decide) k.skip = decide->ls;
    int nrows, stride, lk;
    if ((rc = run_mode(c, 2, b, k, theta ? theta : c->d_theta, nullptr, &nrows, &stride, &lk, st))) return rc;
    finalize(c, 2, nrows, stride, lk, nullptr, out, st, decide);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
