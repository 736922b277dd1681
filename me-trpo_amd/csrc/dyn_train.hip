// Ensemble dynamics training and normaliser statistics (rank 2):
//   metrpo_dyn_train_step   one sess.run([dynamics_opt_op, dynamics_loss])          model_based_rl.py:961-971
//                           loss graph :23-71 (per-model mean_b sum_d (y_pred - y)^2), optimizers :154-183
//                           (tf.train.AdamOptimizer on the prediction loss + SGD on the regulariser)
//   metrpo_dyn_eval_losses  dynamics_losses on np.tile(validation, n_models)                       :933-945, :977-983
//   metrpo_rms_accumulate   RunningMeanStd.update sums                                      running_mean_std.py:35-42
// K independent MLPs are trained as ONE batched problem: every layer's forward, input-gradient and weight-gradient
// product is a GEMM batched over the K heads on the f32 matrix core (gemm_mfma.h); the Adam update of a weight matrix
// is the epilogue of its own gradient GEMM (the gradient never touches HBM), relu' is the epilogue of the
// input-gradient GEMM.  Model i trains on samples i, K+i, 2K+i, ... of the (batch_size*K)-row block (utils.get_ith_tensor).
#include "gemm_mfma.h"

struct TrainWs {             // carved from ctx->d_train
    float* Xn; float* H[MAXL]; float* OUT; float* dZa; float* dZb; float* part; float* am; float* av; double* loss; double* lpart; size_t rows;
};

// normalise + drop columns (training.py:228,146-151).  train: model k reads sample b*K + k; eval: every model reads sample b
__global__ void k_train_prep(ProblemDesc pd, const float* __restrict__ norm, const float* __restrict__ x, long long n_rows, int rows,
                             int shared, float* __restrict__ Xn) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)pd.K * rows * pd.nin;
    if (i >= total) return;
    const int c = (int)(i % pd.nin);
    const long long kb = i / pd.nin;
    const int b = (int)(kb % rows), k = (int)(kb / rows);
    const long long n = shared ? b : (long long)b * pd.K + k;
    const int f = c + pd.n_drop;                       // column of [s, a]
    float v = 0.0f;
    if (n < n_rows) v = (x[n * (pd.ns + pd.na) + f] - norm[f]) / norm[(pd.ns + pd.na) + f];
    Xn[i] = v;
}

// prediction, loss and d(loss)/d(out): pred = diff_mean + diff_std*out + s (training.py:257)
__global__ void k_train_out(ProblemDesc pd, const float* __restrict__ norm, const float* __restrict__ x, const float* __restrict__ y,
                            long long n_rows, int rows, int shared, double inv_n, const float* __restrict__ OUT,
                            float* __restrict__ dZ /* may be null (eval) */, double* __restrict__ loss, double* lpart) {
    __shared__ double red[16];
    const int k = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int ns = pd.ns;
    const float* diff_mean = norm + 2 * (ns + pd.na); const float* diff_std = diff_mean + ns;
    double acc = 0.0;
    if (b < rows) {
        const long long n = shared ? b : (long long)b * pd.K + k;
        const bool ok = n < n_rows;
        const size_t o = ((size_t)k * rows + b) * ns;
        for (int d = 0; d < ns; ++d) {
            float diff = 0.0f;
            if (ok) diff = fmaf(diff_std[d], OUT[o + d], diff_mean[d]) + x[n * (ns + pd.na) + d] - y[n * ns + d];
            acc += (double)diff * (double)diff;
            if (dZ != nullptr) dZ[o + d] = (float)(2.0 * (double)diff * (double)diff_std[d] * inv_n);
        }
    }
    const double t = block_sum(acc, red);
    // a model's loss = its workgroups' sums added in workgroup order by whichever of them finishes last (float64 atomics add in arrival order:
    // the loss the early-stopping comparisons see would differ in the last bit from run to run).  lpart = [K] tickets (8 bytes each, zero when
    // idle), then [K][gridDim.x] sums.
    if (threadIdx.x == 0) {
        unsigned int* ticket = (unsigned int*)(lpart + k);
        double* sums = lpart + gridDim.y + (size_t)k * gridDim.x;
        sums[blockIdx.x] = t;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            double a = 0.0;
            for (unsigned int j = 0; j < gridDim.x; ++j) a += __hip_atomic_load(sums + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss[k] += a * inv_n;                                   // chunks of an evaluation accumulate in stream order
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// regulariser value: constant * sum_l (l2_loss(W) + l2_loss(b)) per model (training.py:271-282); one block per model
__global__ void k_reg_loss(int Pd, const float* __restrict__ params, double constant, double* __restrict__ loss) {
    __shared__ double red[16];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < Pd; i += blockDim.x) { const double w = params[(size_t)k * Pd + i]; acc += 0.5 * w * w; }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(&loss[k], constant * t);
}

__global__ void k_rms_accumulate(const float* __restrict__ x, long long n, int dim, double* __restrict__ rsum, double* __restrict__ rsumsq) {
    // one block per column chunk of 64 columns x 4 row slices; per-column double sums, fixed order inside a block
    __shared__ double s1[4][65], s2[4][65];
    const int lc = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lc;
    double a = 0.0, q = 0.0;
    if (col < dim)
        for (long long r = sl; r < n; r += 4) { const double v = x[r * dim + col]; a += v; q += v * v; }
    s1[sl][lc] = a; s2[sl][lc] = q;
    __syncthreads();
    if (sl == 0 && col < dim) {
        rsum[col] += (s1[0][lc] + s1[1][lc]) + (s1[2][lc] + s1[3][lc]);
        rsumsq[col] += (s2[0][lc] + s2[1][lc]) + (s2[2][lc] + s2[3][lc]);
    }
}

// ------------------------------------------------------------------------------------------------

// Split-K decision for the weight-gradient GEMM of one layer (M = n_in, N = n_out, contraction over the batch rows).
struct SplitK { int splits, kchunk; long long stride; };
static SplitK choose_split(int M, int N, int rows, int heads) {
    const int bm = (M > 64) ? 128 : 64, bn = (N > 64) ? 128 : 64;
    const int blocks = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * heads;
    SplitK s = {1, rows, 0};
    if (blocks >= 128 || rows < 256) return s;
    int want = std::min(std::min((256 + blocks - 1) / blocks, (rows + 127) / 128), 16);
    if (want <= 1) return s;
    s.kchunk = ((rows + want - 1) / want + 15) & ~15;
    s.splits = (rows + s.kchunk - 1) / s.kchunk;
    s.stride = (((long long)M * N + N) + 3) & ~3LL;
    return s;
}

// Sum the split-K partials in split order and apply tf.train.AdamOptimizer (+ SGD on the regulariser) to W_l (i < M*N) and b_l.
__global__ void k_adam_apply(int splits, int heads, long long stridePart, const float* __restrict__ part, int MN, int N,
                             float* __restrict__ W, float* __restrict__ am, float* __restrict__ av, float* __restrict__ bvec,
                             float* __restrict__ bam, float* __restrict__ bav, long long strideP, float lr_t, float b1, float b2, float eps,
                             float decay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, head = blockIdx.y;
    if (i >= MN + N) return;
    float g = 0.0f;
    for (int s = 0; s < splits; ++s) g += part[((size_t)s * heads + head) * stridePart + i];
    float *pw, *pm, *pv;
    if (i < MN) { const size_t o = (size_t)head * strideP + i; pw = W + o; pm = am + o; pv = av + o; }
    else { const size_t o = (size_t)head * strideP + (i - MN); pw = bvec + o; pm = bam + o; pv = bav + o; }
    const float m1 = b1 * *pm + (1.0f - b1) * g, v1 = b2 * *pv + (1.0f - b2) * g * g;
    *pm = m1; *pv = v1;
    const float w = *pw;
    *pw = w - lr_t * m1 / (sqrtf(v1) + eps) - decay * w;
}

static int ensure_train_ws(metrpo_ctx* c, int rows, TrainWs* ws) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    int maxw = pd.ns;
    size_t hsum = 0;
    for (int l = 1; l < L; ++l) { maxw = std::max(maxw, pd.dyn.dims[l]); hsum += up4((size_t)K * rows * pd.dyn.dims[l]); }
    const size_t nXn = up4((size_t)K * rows * pd.nin), nOut = up4((size_t)K * rows * pd.ns), nZ = up4((size_t)K * rows * maxw);
    const size_t nP = up4((size_t)K * pd.dyn.n_params);
    // Adam moments live in their own allocation (they persist across steps and batch sizes)
    if (!c->d_adam) {
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_adam, 2 * nP * sizeof(float) + 64 * sizeof(double)));
        HIP_TRY(c, hipMemset(c->d_adam, 0, 2 * nP * sizeof(float) + 64 * sizeof(double)));
        c->adam_t = 0;
    }
    size_t nPart = 0;
    for (int l = 0; l < L; ++l) {
        const SplitK sk = choose_split(pd.dyn.dims[l], pd.dyn.dims[l + 1], rows, K);
        if (sk.splits > 1) nPart = std::max(nPart, (size_t)sk.splits * (size_t)K * (size_t)sk.stride);
    }
    nPart = std::max(nPart, skinny_part_floats(rows, pd.ns, pd.dyn.dims[L - 1], K));      // forward output layer (train_forward)
    const size_t need = (nXn + hsum + nOut + 2 * nZ + nPart) * sizeof(float);
    if (need > c->train_cap) {
        ws_retire(c, c->d_train);
        c->d_train = nullptr; c->train_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_train, need));
        c->train_cap = need;
    }
    const size_t nLp = (size_t)K * (1 + (size_t)(rows + 127) / 128);
    if (nLp > c->train_part_cap) {
        ws_retire(c, c->d_train_part);
        c->d_train_part = nullptr; c->train_part_cap = 0;
        const size_t cap = std::max<size_t>(nLp, 2048);
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_train_part, cap * sizeof(double)));
        HIP_TRY(c, hipMemset(c->d_train_part, 0, cap * sizeof(double)));          // tickets start at zero; every launch leaves them there
        c->train_part_cap = cap;
    }
    ws->lpart = c->d_train_part;
    float* p = (float*)c->d_train;
    ws->Xn = p; p += nXn;
    for (int l = 1; l < L; ++l) { ws->H[l] = p; p += up4((size_t)K * rows * pd.dyn.dims[l]); }
    ws->H[0] = ws->Xn;
    ws->OUT = p; p += nOut; ws->dZa = p; p += nZ; ws->dZb = p; p += nZ; ws->part = p;
    ws->am = (float*)c->d_adam; ws->av = ws->am + nP; ws->loss = (double*)(ws->av + nP);
    ws->rows = rows;
    return METRPO_OK;
}

// forward of all K heads on the prepared inputs; hidden activations kept in ws->H[l] for the backward pass
static void train_forward(metrpo_ctx* c, const TrainWs& ws, int rows, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    for (int l = 0; l < L; ++l) {
        const int Kd = pd.dyn.dims[l], N = pd.dyn.dims[l + 1];
        float* out = (l == L - 1) ? ws.OUT : ws.H[l + 1];
        GemmEpi ep = {};
        ep.bias = c->d_dyn + pd.dyn.b_off[l]; ep.strideBias = pd.dyn.n_params;
        const float* Wl = c->d_dyn + pd.dyn.w_off[l];
        // output layer (ns <= 64 columns, contraction over the hidden width): split-K partials + ordered reduce where that pays (gemm_skinny_bias; one
        // 64-column tile per 64 rows walks the whole K axis on a fifth of the CUs: 25.8 us of a 210 us step at 2 x 512, batch 1000)
        if (l == L - 1) gemm_skinny_bias(ws.H[l], (long long)rows * Kd, Kd, Wl, pd.dyn.n_params, N, ep.bias, ep.strideBias, out, (long long)rows * N, rows, N, Kd, K, ws.part, st);
        else if (pd.dyn.act[l] == METRPO_ACT_RELU) gemm_auto<EPI_BIAS_RELU, false, false>(ws.H[l], (long long)rows * Kd, Kd, Wl, pd.dyn.n_params, N, out, (long long)rows * N, N, rows, N, Kd, K, ep, st);
        else gemm_auto<EPI_BIAS_TANH, false, false>(ws.H[l], (long long)rows * Kd, Kd, Wl, pd.dyn.n_params, N, out, (long long)rows * N, N, rows, N, Kd, K, ep, st);
    }
}

int launch_dyn_train_step(metrpo_ctx* c, const float* x, const float* y, const metrpo_train_params* tp, double* loss_out, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers, rows = tp->batch_size;
    for (int l = 0; l < L - 1; ++l)
        if (pd.dyn.act[l] != METRPO_ACT_RELU) return set_err(c, METRPO_EUNSUPPORTED, "dyn_train: only relu hidden layers (all shipped params files)");
    TrainWs ws;
    int rc = ensure_train_ws(c, rows, &ws);
    if (rc) return rc;
    const long long n_rows = (long long)rows * K;
    HIP_TRY(c, hipMemsetAsync(ws.loss, 0, sizeof(double) * (K + 1), st));
    {
        const long long total = (long long)K * rows * pd.nin;
        hipLaunchKernelGGL(k_train_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pd, c->d_norm, x, n_rows, rows, 0, ws.Xn);
    }
    train_forward(c, ws, rows, st);
    hipLaunchKernelGGL(k_train_out, dim3((rows + 127) / 128, K), dim3(128), 0, st, pd, c->d_norm, x, y, n_rows, rows, 0, 1.0 / (double)rows,
                       ws.OUT, ws.dZa, ws.loss, ws.lpart);
    if (tp->reg_constant != 0.0) hipLaunchKernelGGL(k_reg_loss, dim3(K), dim3(256), 0, st, pd.dyn.n_params, c->d_dyn, tp->reg_constant, ws.loss);
    // Adam step (tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t), epsilon outside the correction)
    c->adam_t += 1;
    const double lr_t = tp->lr * std::sqrt(1.0 - std::pow(tp->beta2, (double)c->adam_t)) / (1.0 - std::pow(tp->beta1, (double)c->adam_t));
    const float decay = (float)(tp->lr * tp->reg_constant);
    float* dz = ws.dZa; float* dz_next = ws.dZb;
    for (int l = L - 1; l >= 0; --l) {
        const int n_in = pd.dyn.dims[l], n_out = pd.dyn.dims[l + 1];
        float* Wl = c->d_dyn + pd.dyn.w_off[l];
        if (l > 0) {            // dH_{l-1} = dZ_l . W_l^T, masked by relu'(H_{l-1})  -- uses W_l BEFORE its update
            GemmEpi ep = {};
            ep.mask = ws.H[l]; ep.strideMask = (long long)rows * n_in; ep.ldm = n_in;
            gemm_auto<EPI_RELU_MASK, false, true>(dz, (long long)rows * n_out, n_out, Wl, pd.dyn.n_params, n_out, dz_next, (long long)rows * n_in, n_in,
                                                 rows, n_in, n_out, K, ep, st);
        }
        {                       // dW_l = H_{l-1}^T . dZ_l with the Adam update as epilogue (W_l updated in place)
            GemmEpi ep = {};
            ep.am = ws.am + pd.dyn.w_off[l]; ep.av = ws.av + pd.dyn.w_off[l]; ep.strideAdam = pd.dyn.n_params;
            ep.bvec = c->d_dyn + pd.dyn.b_off[l]; ep.bam = ws.am + pd.dyn.b_off[l]; ep.bav = ws.av + pd.dyn.b_off[l];   // b_l: column sums of dZ_l
            ep.lr_t = (float)lr_t; ep.beta1 = (float)tp->beta1; ep.beta2 = (float)tp->beta2; ep.eps = (float)tp->eps; ep.decay = decay;
            const SplitK sk = choose_split(n_in, n_out, rows, K);
            if (sk.splits > 1) {
                ep.part = ws.part; ep.stridePart = sk.stride; ep.splits = sk.splits; ep.kchunk = sk.kchunk;
                gemm_auto<EPI_PARTIAL, true, false>(ws.H[l], (long long)rows * n_in, n_in, dz, (long long)rows * n_out, n_out, nullptr, 0, n_out,
                                                   n_in, n_out, rows, K, ep, st);
                const int tot = n_in * n_out + n_out;
                hipLaunchKernelGGL(k_adam_apply, dim3((tot + 255) / 256, K), dim3(256), 0, st, sk.splits, K, sk.stride, ws.part, n_in * n_out, n_out,
                                   Wl, ep.am, ep.av, ep.bvec, ep.bam, ep.bav, (long long)pd.dyn.n_params, ep.lr_t, ep.beta1, ep.beta2, ep.eps, decay);
            } else {
                gemm_auto<EPI_ADAM, true, false>(ws.H[l], (long long)rows * n_in, n_in, dz, (long long)rows * n_out, n_out, Wl, pd.dyn.n_params, n_out,
                                                n_in, n_out, rows, K, ep, st);
            }
        }
        std::swap(dz, dz_next);
    }
    if (loss_out) HIP_TRY(c, hipMemcpyAsync(loss_out, ws.loss, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_dyn_eval_losses(metrpo_ctx* c, const float* x, const float* y, long long n, double reg_constant, double* losses, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K;
    const int CH = 8192;                                  // rows per pass (bounds the activation workspace)
    TrainWs ws;
    int rc = ensure_train_ws(c, (int)std::min<long long>(n, CH), &ws);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(ws.loss, 0, sizeof(double) * (K + 1), st));
    for (long long r0 = 0; r0 < n; r0 += CH) {
        const int rows = (int)std::min<long long>(CH, n - r0);
        const long long total = (long long)K * rows * pd.nin;
        hipLaunchKernelGGL(k_train_prep, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pd, c->d_norm, x + r0 * (pd.ns + pd.na), n - r0,
                           rows, 1, ws.Xn);
        train_forward(c, ws, rows, st);
        hipLaunchKernelGGL(k_train_out, dim3((rows + 127) / 128, K), dim3(128), 0, st, pd, c->d_norm, x + r0 * (pd.ns + pd.na), y + r0 * pd.ns,
                           n - r0, rows, 1, 1.0 / (double)n, ws.OUT, (float*)nullptr, ws.loss, ws.lpart);
    }
    if (reg_constant != 0.0) hipLaunchKernelGGL(k_reg_loss, dim3(K), dim3(256), 0, st, pd.dyn.n_params, c->d_dyn, reg_constant, ws.loss);
    HIP_TRY(c, hipMemcpyAsync(losses, ws.loss, sizeof(double) * K, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_rms_accumulate(metrpo_ctx* c, const float* x, long long n, int dim, double* rsum, double* rsumsq, hipStream_t st) {
    hipLaunchKernelGGL(k_rms_accumulate, dim3((dim + 63) / 64), dim3(256), 0, st, x, n, dim, rsum, rsumsq);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
