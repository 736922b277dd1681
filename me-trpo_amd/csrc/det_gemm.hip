// Per-model DETERMINISTIC rollout of build_policy_graph (model_based_rl.py:106-151) for LARGE dynamics networks (hidden >= 128: the
// params-file shapes 2x512 / 2x1024 / 3x1024): forward sweep (= per-model validation costs, metrpo_validation_cost, and the stored
// trajectory of the BPTT update) and reverse sweep, with every dynamics layer and layer-adjoint a batched-over-models GEMM on the f32
// matrix core (gemm_mfma.h).  Same recursion as bptt.hip / bptt_mfma.hip; mapping as in rollout_gemm.hip: the time loop stays on the
// host (stream-ordered launches, no synchronisation), row r = (model k, env b):
//   forward step   k_dg_pre (policy, clip, normalise)  ->  L GEMMs (bias + relu)  ->  k_dg_post (residual, cost, dones, weights, x_{t+1})
//   reverse step   k_dg_pre  ->  L-1 GEMMs (hidden activations)  ->  k_dg_mid (cost adjoint, diff_std)  ->  L adjoint GEMMs
//                  (dZ . W^T with the relu mask of the layer below as epilogue)  ->  k_dg_back (normalisers, residual, clip gate,
//                  mean-adjoint output, policy input VJP)
#include "gemm_mfma.h"
#include "mfma_common.h"
#include "policy_chain3.h"

struct DgState {
    float *S, *X, *U, *MU, *OUT, *G, *DZa, *DZb, *LAM, *DONES, *PART;
    float* PIMG;                        // k_dg_pre_mfma3: the policy's fragment image (policy_chain3.h), built at the start of a sweep
    float* H[MAXL];
    double* ACC;
    // forward sweep: output layer left as partials (split-K, or one per 64 / 128-column block of the last hidden layer when that layer's launch
    // contracts its relu tile with the output weights, gemm_mfma.h EPI_RELU_OUT); k_dg_post adds them and the bias.  0 splits: OUT holds the layer.
    int out_splits; long long out_stride; const float* out_bias; long long out_bias_stride;
};

// thread = row (k, b): policy forward of the current state, clipped action, pre-clip mean, normalised + dropped dynamics input
__global__ void k_dg_pre(ProblemDesc pd, int B, const float* __restrict__ theta, const float* __restrict__ norm, const float* __restrict__ xs_t,
                         long long xs_model_stride, DgState st) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid, k = blockIdx.y;
    const bool active = b < B;
    const int ns = pd.ns, na = pd.na;
    const size_t row = (size_t)k * B + b;
    float* Sc = lds; float* A = Sc + ns * LD; float* Bq = A + pd.pol.max_width * LD;
    const float* src = (xs_t != nullptr) ? xs_t + (size_t)k * xs_model_stride + (size_t)b * ns : st.S + row * ns;
    for (int i = 0; i < ns; ++i) Sc[i * LD + tid] = active ? src[i] : 0.0f;
    float* m = mlp_col(pd.pol, theta, Sc, A, Bq, LD, tid);
    if (!active) return;
    const float* in_mean = norm; const float* in_std = norm + (ns + na);
    for (int i = 0; i < ns; ++i) {
        const float s = Sc[i * LD + tid];
        if (xs_t != nullptr) st.S[row * ns + i] = s;
        if (i >= pd.n_drop) st.X[row * pd.nin + i - pd.n_drop] = (s - in_mean[i]) / in_std[i];
    }
    for (int d = 0; d < na; ++d) {
        const float mu = m[d * LD + tid];
        const float ac = fminf(fmaxf(mu, -1.0f), 1.0f);                                   // model_based_rl.py:128
        st.MU[row * na + d] = mu; st.U[row * na + d] = ac;
        st.X[row * pd.nin + (ns - pd.n_drop) + d] = (ac - in_mean[ns + d]) / in_std[ns + d];
    }
}

// MFMA variant of k_dg_pre for the 2x32 tanh policies: a wave evaluates the policy of a 16-row tile of one model as the transposed MFMA
// chain of the fused kernels (30 MFMAs) instead of 64 threads walking three dense layers each.  grid = (ceil(B/64), K) blocks of 4 waves.
template <int ENV>
__global__ void __launch_bounds__(256) k_dg_pre_mfma(ProblemDesc pd, int B, const float* __restrict__ theta, const float* __restrict__ norm,
                                                     const float* __restrict__ xs_t, long long xs_model_stride, DgState st) {
    using C = Cfg<ENV, 64, 32>;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NIN = C::NIN, PH = 32, NS_KS = C::NS_KS;
    constexpr int O_PF1 = NS_KS * 2 * 64, O_PF2 = O_PF1 + 16 * 64, O_B0 = O_PF2 + 8 * 64, O_B1 = O_B0 + 32, O_B2 = O_B1 + 32, IMG = O_B2 + 16;
    __shared__ __attribute__((aligned(16))) float lds[IMG + 4 * 16 * NS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int k = blockIdx.y;
    const int b0 = (blockIdx.x * 4 + wave) * 16, b = b0 + c;
    const bool active = b < B;
    float* ST = lds + IMG + wave * 16 * NS;
    for (int i = tid; i < IMG; i += 256) {
        float w = 0.0f;
        const int ln = i & 63, cc = ln & 15, qq = ln >> 4;
        if (i < O_PF1) { const int f = i >> 6, s_ = f >> 1, cb = f & 1, in = 4 * s_ + qq; if (in < NS) w = theta[C::pW0 + in * PH + 16 * cb + cc]; }
        else if (i < O_PF2) { const int f = (i - O_PF1) >> 6, kk = f >> 1, cb = f & 1; w = theta[C::pW1 + (16 * (kk >> 2) + 4 * qq + (kk & 3)) * PH + 16 * cb + cc]; }
        else if (i < O_B0) { const int kk = (i - O_PF2) >> 6; if (cc < NA) w = theta[C::pW2 + (16 * (kk >> 2) + 4 * qq + (kk & 3)) * NA + cc]; }
        else if (i < O_B1) w = theta[C::pb0 + (i - O_B0)];
        else if (i < O_B2) w = theta[C::pb1 + (i - O_B1)];
        else { const int d = i - O_B2; if (d < NA) w = theta[C::pb2 + d]; }
        lds[i] = w;
    }
    const size_t row0 = (size_t)k * B + b0;
    const float* src = (xs_t != nullptr) ? xs_t + (size_t)k * xs_model_stride + (size_t)b0 * NS : st.S + row0 * NS;
    const int lim = min(16, max(0, B - b0)) * NS;
    for (int i = lane; i < 16 * NS; i += 64) {
        const float v = (i < lim) ? src[i] : 0.0f;
        ST[i] = v;
        if (xs_t != nullptr && i < lim) st.S[row0 * NS + i] = v;
    }
    __syncthreads();
    f32x4 p0[2], p1[2];
    p0[0] = *(const f32x4*)&lds[O_B0 + 4 * q]; p0[1] = *(const f32x4*)&lds[O_B0 + 16 + 4 * q];
#pragma unroll
    for (int s_ = 0; s_ < NS_KS; ++s_) {
        const int f = 4 * s_ + q;
        const float x = (f < NS) ? ST[c * NS + f] : 0.0f;
        p0[0] = MFMA16(lds[(s_ * 2 + 0) * 64 + lane], x, p0[0]);
        p0[1] = MFMA16(lds[(s_ * 2 + 1) * 64 + lane], x, p0[1]);
    }
    p1[0] = *(const f32x4*)&lds[O_B1 + 4 * q]; p1[1] = *(const f32x4*)&lds[O_B1 + 16 + 4 * q];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) p0[cb][rr] = tanh_fast(p0[cb][rr]);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        p1[0] = MFMA16(lds[O_PF1 + (kk * 2 + 0) * 64 + lane], p0[kk >> 2][kk & 3], p1[0]);
        p1[1] = MFMA16(lds[O_PF1 + (kk * 2 + 1) * 64 + lane], p0[kk >> 2][kk & 3], p1[1]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) p1[cb][rr] = tanh_fast(p1[cb][rr]);
    f32x4 m0 = *(const f32x4*)&lds[O_B2 + 4 * q], m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
        m0 = MFMA16(lds[O_PF2 + kk * 64 + lane], p1[kk >> 2][kk & 3], m0);
        m1 = MFMA16(lds[O_PF2 + (kk + 1) * 64 + lane], p1[(kk + 1) >> 2][(kk + 1) & 3], m1);
    }
    const f32x4 mu = m0 + m1;
    if (!active) return;
    const size_t row = (size_t)k * B + b;
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    for (int i = q; i < NS; i += 4) if (i >= NDROP) st.X[row * NIN + i - NDROP] = (ST[c * NS + i] - in_mean[i]) / in_std[i];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int d = 4 * q + rr;
        if (d < NA) {
            const float ac = fminf(fmaxf(mu[rr], -1.0f), 1.0f);                           // model_based_rl.py:128
            st.MU[row * NA + d] = mu[rr]; st.U[row * NA + d] = ac;
            st.X[row * NIN + (NS - NDROP) + d] = (ac - in_mean[NS + d]) / in_std[NS + d];
        }
    }
}

// MFMA variant of k_dg_pre for three-hidden-layer tanh policies (Humanoid's 100-50-25; policy_chain3.h): the thread-per-row k_dg_pre walks that
// policy's 12 275 weights through scalar loads, 253 us per step whatever the batch -- 70 % of a validation-cost evaluation at the
// params-humanoid.json shape.  grid = (ceil(B/64), K) blocks of 4 waves, dynamic LDS = image + state tiles.
template <int NS, int NA, int NDROP, int W1, int W2, int W3>
__global__ void __launch_bounds__(256) k_dg_pre_mfma3(ProblemDesc pd, int B, const float* __restrict__ theta, const float* __restrict__ norm,
                                                      const float* __restrict__ xs_t, long long xs_model_stride, DgState st) {
    using PC = P3<NS, NA, W1, W2, W3>;
    constexpr int CO = PC::CO, IMG = PC::IMG, NIN = NS - NDROP + NA;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int k = blockIdx.y;
    const int b0 = (blockIdx.x * 4 + wave) * 16, b = b0 + c;
    const bool active = b < B;
    float* ST = lds + IMG + wave * 16 * NS;
    PC::load_image(lds, st.PIMG, tid);
    const size_t row0 = (size_t)k * B + b0;
    const float* src = (xs_t != nullptr) ? xs_t + (size_t)k * xs_model_stride + (size_t)b0 * NS : st.S + row0 * NS;
    const int lim = min(16, max(0, B - b0)) * NS;
    for (int i = lane; i < 16 * NS; i += 64) {
        const float v = (i < lim) ? src[i] : 0.0f;
        ST[i] = v;
        if (xs_t != nullptr && i < lim) st.S[row0 * NS + i] = v;
    }
    __syncthreads();
    f32x4 mu[CO];
    PC::forward(lds, ST, lane, c, q, mu);
    if (!active) return;
    const size_t row = (size_t)k * B + b;
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    for (int i = q; i < NS; i += 4) if (i >= NDROP) st.X[row * NIN + i - NDROP] = (ST[c * NS + i] - in_mean[i]) / in_std[i];
#pragma unroll
    for (int cb = 0; cb < CO; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int d = 16 * cb + 4 * q + rr;
            if (d < NA) {
                const float ac = fminf(fmaxf(mu[cb][rr], -1.0f), 1.0f);                       // model_based_rl.py:128
                st.MU[row * NA + d] = mu[cb][rr]; st.U[row * NA + d] = ac;
                st.X[row * NIN + (NS - NDROP) + d] = (ac - in_mean[NS + d]) / in_std[NS + d];
            }
        }
}

typedef void (*dg_pre_mfma_t)(ProblemDesc, int, const float*, const float*, const float*, long long, DgState);
static dg_pre_mfma_t dg_pre_mfma_select(const metrpo_ctx* c, size_t* dyn_lds = nullptr) {
    const ProblemDesc& pd = c->pd;
    if (dyn_lds) *dyn_lds = 0;
    if (dyn_lds && pd.env == METRPO_ENV_HUMANOID && pd.ns == 55 && pd.na == 21 && pd.n_drop == 0 && pd.pol.n_layers == 4 && pd.pol.dims[1] == 100 &&
        pd.pol.dims[2] == 50 && pd.pol.dims[3] == 25 && pd.pol.act[0] == METRPO_ACT_TANH && pd.pol.act[1] == METRPO_ACT_TANH && pd.pol.act[2] == METRPO_ACT_TANH) {
        *dyn_lds = sizeof(float) * (size_t)(P3<55, 21, 100, 50, 25>::IMG + 4 * 16 * 55);
        return k_dg_pre_mfma3<55, 21, 0, 100, 50, 25>;
    }
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH) return nullptr;
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: return (pd.ns == 10 && pd.na == 2 && pd.n_drop == 2) ? k_dg_pre_mfma<METRPO_ENV_SWIMMER> : nullptr;
    case METRPO_ENV_HALF_CHEETAH: return (pd.ns == 18 && pd.na == 6 && pd.n_drop == 1) ? k_dg_pre_mfma<METRPO_ENV_HALF_CHEETAH> : nullptr;
    case METRPO_ENV_ANT: return (pd.ns == 29 && pd.na == 8 && pd.n_drop == 2) ? k_dg_pre_mfma<METRPO_ENV_ANT> : nullptr;
    case METRPO_ENV_HOPPER: return (pd.ns == 11 && pd.na == 3 && pd.n_drop == 0) ? k_dg_pre_mfma<METRPO_ENV_HOPPER> : nullptr;
    case METRPO_ENV_SNAKE: return (pd.ns == 14 && pd.na == 4 && pd.n_drop == 2) ? k_dg_pre_mfma<METRPO_ENV_SNAKE> : nullptr;
    }
    return nullptr;
}

__device__ __forceinline__ float dg_cost(int env, int ns, int na, const float* xn, const float* u) {
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) su2 = fmaf(u[d], u[d], su2);
    switch (env) {
    case METRPO_ENV_SWIMMER: return -(xn[5] - 1e-2f * (su2 / (float)na));
    case METRPO_ENV_HALF_CHEETAH: return -fminf(fmaxf(xn[9] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
    case METRPO_ENV_ANT: return -(xn[15] - 1e-2f * 0.5f * su2 + 0.05f);
    case METRPO_ENV_HUMANOID: { const float h = xn[ns - 1] - 1.5f; return h * h + 1e-2f * 1e-3f * su2; }
    case METRPO_ENV_HOPPER: {
        float pen = 0.0f;
        for (int i = 2; i < ns; ++i) pen += fmaxf(fabsf(xn[i]) - 100.0f, 0.0f);
        return -(xn[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0], 0.0f) - 10.0f * fmaxf(fabsf(xn[1]) - 0.2f, 0.0f) - pen);
    }
    case METRPO_ENV_SNAKE: return -(xn[7] - 1e-2f * 0.5f * su2);
    }
    return 0.0f;
}

// forward: x' = diff_mean + diff_std * out + x, cost, dones, weights, trajectory
constexpr int DG_POST_ROWS = 32, DG_POST_THREADS = 256;
__global__ __launch_bounds__(DG_POST_THREADS) void k_dg_post(ProblemDesc pd, int B, int T, int t, double gpow, const float* __restrict__ norm, DgState st,
                                                             float* __restrict__ XS, float* __restrict__ WT) {
    // phase 1, all threads: the block's DG_POST_ROWS x ns output-layer values, coalesced (and summed over the partials when the layer was left as
    // partials), into LDS; phase 2, one thread per row: residual, cost, dones, weights
    __shared__ float so[DG_POST_ROWS * 65];
    const int ns = pd.ns, na = pd.na, K = pd.K, k = blockIdx.y;
    const int b0 = blockIdx.x * DG_POST_ROWS, nr = min(DG_POST_ROWS, B - b0), nel = nr * ns, lds = ns | 1;
    for (int e = threadIdx.x; e < nel; e += DG_POST_THREADS) {
        const int r = e / ns, i = e - r * ns;
        float o;
        if (st.out_splits == 0) o = st.OUT[((size_t)k * B + b0) * ns + e];
        else {
            o = st.out_bias[(size_t)k * st.out_bias_stride + i];
            const float* part = st.PART + (size_t)k * st.out_stride + (size_t)b0 * ns + e;
            const size_t sps = (size_t)K * st.out_stride;
#pragma unroll 4
            for (int sp = 0; sp < st.out_splits; ++sp) o += part[sp * sps];
        }
        so[r * lds + i] = o;
    }
    __syncthreads();
    if ((int)threadIdx.x >= nr) return;
    const int b = b0 + threadIdx.x;
    const size_t row = (size_t)k * B + b;
    const float* diff_mean = norm + 2 * (ns + na); const float* diff_std = diff_mean + ns;
    float xn[64];                                                   // ns <= 64 enforced by the launcher
    bool fin = true;
    for (int i = 0; i < ns; ++i) { xn[i] = fmaf(diff_std[i], so[threadIdx.x * lds + i], diff_mean[i]) + st.S[row * ns + i]; fin = fin && isfinite(xn[i]); }
    const float c = dg_cost(pd.env, ns, na, xn, st.U + row * na);
    const float dones = st.DONES[row], live = 1.0f - dones;
    if (pd.env == METRPO_ENV_ANT) st.DONES[row] = fmaxf(dones, ((xn[2] >= 0.2f) && (xn[2] <= 1.0f) && fin) ? 0.0f : 1.0f);
    st.ACC[row] += gpow * (double)(c * live);
    if (WT != nullptr) WT[((size_t)k * T + t) * B + b] = (float)(gpow * (double)live / ((double)B * (double)K));
    for (int i = 0; i < ns; ++i) {
        st.S[row * ns + i] = xn[i];
        if (XS != nullptr) XS[((size_t)k * (T + 1) + (t + 1)) * B * ns + (size_t)b * ns + i] = xn[i];
    }
}

// costs[k] = sum_b ACC[k][b] / B in index order (deterministic)
__global__ void k_dg_costs(int B, const double* __restrict__ acc, double* __restrict__ costs) {
    __shared__ double sh[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) s += acc[(size_t)k * B + b];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 256; ++i) t += sh[i]; costs[k] = t / (double)B; }
}

// reverse: G = lambda_{t+1} + w dc/dx' ; dZ_out = diff_std * G
__global__ void k_dg_mid(ProblemDesc pd, int B, int T, int t, const float* __restrict__ norm, const float* __restrict__ XS,
                         const float* __restrict__ WT, DgState st) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (b >= B) return;
    const int ns = pd.ns, na = pd.na;
    const size_t row = (size_t)k * B + b;
    const float* diff_std = norm + 2 * (ns + na) + ns;
    const float* xn = XS + ((size_t)k * (T + 1) + (t + 1)) * B * ns + (size_t)b * ns;
    const float* u = st.U + row * na;
    const float w = WT[((size_t)k * T + t) * B + b];
    float g[64];
    for (int i = 0; i < ns; ++i) g[i] = st.LAM[row * ns + i];
    float su2 = 0.0f;
    for (int d = 0; d < na; ++d) su2 = fmaf(u[d], u[d], su2);
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: g[5] -= w; break;
    case METRPO_ENV_HALF_CHEETAH: { const float inner = xn[9] - 1e-1f * 0.5f * su2; if (inner >= -10.0f && inner <= 10.0f) g[9] -= w; break; }
    case METRPO_ENV_ANT: g[15] -= w; break;
    case METRPO_ENV_HUMANOID: g[ns - 1] += w * 2.0f * (xn[ns - 1] - 1.5f); break;
    case METRPO_ENV_HOPPER:
        g[5] -= w;
        if (0.45f - xn[0] > 0.0f) g[0] -= w * 10.0f;
        if (fabsf(xn[1]) - 0.2f > 0.0f) g[1] += w * 10.0f * (xn[1] > 0.0f ? 1.0f : -1.0f);
        for (int i = 2; i < ns; ++i) if (fabsf(xn[i]) - 100.0f > 0.0f) g[i] += w * (xn[i] > 0.0f ? 1.0f : -1.0f);
        break;
    case METRPO_ENV_SNAKE: g[7] -= w; break;
    }
    for (int i = 0; i < ns; ++i) { st.G[row * ns + i] = g[i]; st.DZa[row * ns + i] = diff_std[i] * g[i]; }
}

// reverse: dX (adjoint of the normalised, dropped dynamics input) -> state / action adjoints, clip gate, mean-adjoint out, policy input VJP
__global__ void k_dg_back(ProblemDesc pd, int B, int T, int t, const float* __restrict__ theta, const float* __restrict__ norm,
                          const float* __restrict__ XS, const float* __restrict__ WT, const float* __restrict__ DX, DgState st,
                          float* __restrict__ GM) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = blockDim.x, tid = threadIdx.x;
    const int b = blockIdx.x * blockDim.x + tid, k = blockIdx.y;
    const bool active = b < B;
    const int ns = pd.ns, na = pd.na;
    const NetDesc& pn = pd.pol;
    const size_t row = (size_t)k * B + b;
    int prow = 0;
    for (int l = 1; l <= pn.n_layers; ++l) prow += pn.dims[l];
    float* Sc = lds; float* PH = Sc + ns * LD; float* pa = PH + prow * LD; float* pb = pa + pn.max_width * LD;
    for (int i = 0; i < ns; ++i) Sc[i * LD + tid] = active ? st.S[row * ns + i] : 0.0f;
    // policy forward keeping every layer's output (needed by the input VJP)
    {
        const float* cur = Sc; float* dst = PH;
        for (int l = 0; l < pn.n_layers; ++l) {
            dense_col(theta + pn.w_off[l], theta + pn.b_off[l], pn.dims[l], pn.dims[l + 1], pn.act[l], cur, dst, LD, tid);
            cur = dst; dst += pn.dims[l + 1] * LD;
        }
    }
    const float* in_std = norm + (ns + na);
    const float* xn = XS + ((size_t)k * (T + 1) + (t + 1)) * B * ns + (size_t)b * ns;
    const float w = active ? WT[((size_t)k * T + t) * B + b] : 0.0f;
    float su2 = 0.0f;
    if (active) for (int d = 0; d < na; ++d) { const float a = st.U[row * na + d]; su2 = fmaf(a, a, su2); }
    float cu = 0.0f;                                                // d(w cost)/du_d = cu * u_d
    switch (pd.env) {
    case METRPO_ENV_SWIMMER: cu = w * 1e-2f * 2.0f / (float)na; break;
    case METRPO_ENV_HALF_CHEETAH: { const float inner = active ? xn[9] - 1e-1f * 0.5f * su2 : 0.0f; cu = (inner >= -10.0f && inner <= 10.0f) ? w * 1e-1f : 0.0f; break; }
    case METRPO_ENV_ANT: cu = w * 1e-2f; break;
    case METRPO_ENV_HUMANOID: cu = w * 2e-5f; break;
    case METRPO_ENV_HOPPER: cu = w * 0.01f; break;
    case METRPO_ENV_SNAKE: cu = w * 1e-2f; break;
    }
    const float* dx = DX + row * pd.nin;
    for (int d = 0; d < na; ++d) {
        float gm = 0.0f;
        if (active) {
            const float gu = cu * st.U[row * na + d] + dx[(ns - pd.n_drop) + d] / in_std[ns + d];
            const float mu = st.MU[row * na + d];
            gm = (mu >= -1.0f && mu <= 1.0f) ? gu : 0.0f;          // tf.clip_by_value gradient
            GM[(((size_t)k * (T + 1) + t) * B + b) * na + d] = gm;
        }
        pa[d * LD + tid] = gm;
    }
    // policy input VJP
    {
        int hoff = 0;                                                // rows of the last hidden layer's output in PH
        for (int l = 1; l < pn.n_layers - 1; ++l) hoff += pn.dims[l];
        float* a_ = pa; float* b_ = pb;
        for (int l = pn.n_layers - 1; l >= 0; --l) {
            const float* __restrict__ W = theta + pn.w_off[l];
            const int n_in = pn.dims[l], n_out = pn.dims[l + 1];
            for (int j = 0; j < n_in; ++j) {
                const float* __restrict__ wr = W + (size_t)j * n_out;
                float s = 0.0f;
                for (int i = 0; i < n_out; ++i) s = fmaf(wr[i], a_[i * LD + tid], s);
                b_[j * LD + tid] = s;
            }
            if (l > 0) {
                const float* h = PH + hoff * LD;                     // output of layer l-1
                for (int j = 0; j < n_in; ++j) {
                    const float hv = h[j * LD + tid];
                    const float dact = (pn.act[l - 1] == METRPO_ACT_TANH) ? (1.0f - hv * hv) : (pn.act[l - 1] == METRPO_ACT_RELU) ? (hv > 0.0f ? 1.0f : 0.0f) : 1.0f;
                    b_[j * LD + tid] *= dact;
                }
                if (l > 1) hoff -= pn.dims[l - 1];
            }
            float* tmp = a_; a_ = b_; b_ = tmp;
        }
        if (active)
            for (int i = 0; i < ns; ++i) {
                float v = st.G[row * ns + i];                        // residual connection
                if (i >= pd.n_drop) v += dx[i - pd.n_drop] / in_std[i];
                st.LAM[row * ns + i] = v + a_[i * LD + tid];         // lambda_t
            }
    }
}

// ------------------------------------------------------------------------------------------------
bool gemm_path_applicable(const metrpo_ctx* c);


bool det_gemm_applicable(const metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    if (!gemm_path_applicable(c)) return false;
    for (int l = 0; l < pd.dyn.n_layers - 1; ++l) if (pd.dyn.act[l] != METRPO_ACT_RELU) return false;
    return true;
}

// defer_out (forward sweeps: nobody reads the last hidden layer or OUT afterwards): the last hidden layer's launch contracts its relu tile with the
// output weights when the tile shape allows it, otherwise the output layer's split-K partials are left un-reduced; either way k_dg_post adds them
static int dg_fuse_tile(const metrpo_ctx* c, int B) {
    const ProblemDesc& pd = c->pd;
    const int L = pd.dyn.n_layers;
    if (L < 2 || pd.dyn.act[L - 2] != METRPO_ACT_RELU || pd.dyn.act[L - 1] != METRPO_ACT_IDENTITY || ctx_opt(c, OPT_NO_FUSED_OUT) != nullptr) return 0;
    return gemm_fused_out_tile(B, pd.dyn.dims[L - 1], pd.K, pd.ns);
}
static int dg_workspace(metrpo_ctx* c, int B, DgState* s) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    const size_t R = (size_t)K * B;
    int maxw = std::max(pd.nin, pd.ns);
    size_t nH = 0;
    for (int l = 1; l < L; ++l) { maxw = std::max(maxw, pd.dyn.dims[l]); nH += up4(R * pd.dyn.dims[l]); }
    const size_t nS = up4(R * pd.ns), nX = up4(R * pd.nin), nU = up4(R * pd.na), nZ = up4(R * maxw), nD = up4(R);
    size_t nP = 0;
    for (int l = 0; l < L; ++l) nP = std::max(nP, up4(skinny_part_floats(B, pd.dyn.dims[l + 1], pd.dyn.dims[l], K)));
    if (const int ft = dg_fuse_tile(c, B)) nP = std::max(nP, up4(gemm_fused_out_part_floats(B, pd.dyn.dims[L - 1], K, pd.ns, ft)));
    const size_t nPimg = up4((size_t)pre_mfma3_image_floats<55, 21, 100, 50, 25>());     // the one three-hidden-layer policy with an MFMA pre-step (dg_pre_mfma_select): 67 KB
    const size_t need = (4 * nS + nX + 2 * nU + nH + 2 * nZ + nD + nP + nPimg) * sizeof(float) + R * sizeof(double) + 64;
    if (need > c->dg_cap) {
        ws_retire(c, c->d_dg);
        c->d_dg = nullptr; c->dg_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_dg, need));
        c->dg_cap = need;
    }
    float* p = (float*)c->d_dg;
    s->S = p; p += nS; s->OUT = p; p += nS; s->G = p; p += nS; s->LAM = p; p += nS; s->X = p; p += nX; s->U = p; p += nU; s->MU = p; p += nU;
    for (int l = 1; l < L; ++l) { s->H[l - 1] = p; p += up4(R * pd.dyn.dims[l]); }
    s->DZa = p; p += nZ; s->DZb = p; p += nZ; s->DONES = p; p += nD; s->PART = nP ? p : nullptr; p += nP; s->PIMG = p; p += nPimg;
    s->ACC = (double*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    return METRPO_OK;
}

static void dg_forward_layers(metrpo_ctx* c, DgState& s, int B, int n_layers_to_run, bool defer_out, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    const float* in = s.X; int ldin = pd.nin;
    s.out_splits = 0; s.out_stride = 0; s.out_bias = c->d_dyn + pd.dyn.b_off[L - 1]; s.out_bias_stride = pd.dyn.n_params;
    const int fuse_tile = (defer_out && n_layers_to_run == L) ? dg_fuse_tile(c, B) : 0;
    for (int l = 0; l < n_layers_to_run; ++l) {
        const int Kd = pd.dyn.dims[l], N = pd.dyn.dims[l + 1];
        float* out = (l == L - 1) ? s.OUT : s.H[l];
        GemmEpi ep = {};
        ep.bias = c->d_dyn + pd.dyn.b_off[l]; ep.strideBias = pd.dyn.n_params;
        const float* Wl = c->d_dyn + pd.dyn.w_off[l];
        if (fuse_tile && l == L - 2) {
            gemm_relu_fused_out(fuse_tile, in, (long long)B * Kd, ldin, Wl, pd.dyn.n_params, N, ep.bias, ep.strideBias, c->d_dyn + pd.dyn.w_off[L - 1],
                                pd.dyn.n_params, pd.ns, B, N, Kd, K, s.PART, st, &s.out_splits, &s.out_stride);
            break;
        }
        SkinnyDefer df = {0, 0};
        gemm_skinny_bias(in, (long long)B * Kd, ldin, Wl, pd.dyn.n_params, N, ep.bias, ep.strideBias, out, (long long)B * N, B, N, Kd, K, s.PART, st,
                         (l == L - 1) ? 0 : 1, (defer_out && l == L - 1) ? &df : nullptr);
        if (defer_out && l == L - 1) { s.out_splits = df.splits; s.out_stride = df.stridePart; }
        in = out; ldin = N;
    }
}

// forward sweep; XS / WT may be NULL (validation cost only)
int launch_dg_forward(metrpo_ctx* c, const float* s0, int B, int T, double gamma, float* XS, float* WT, double* costs, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    DgState s;
    int rc = dg_workspace(c, B, &s); if (rc) return rc;
    const size_t R = (size_t)K * B;
    for (int k = 0; k < K; ++k) {
        HIP_TRY(c, hipMemcpyAsync(s.S + (size_t)k * B * pd.ns, s0, sizeof(float) * (size_t)B * pd.ns, hipMemcpyDeviceToDevice, st));
        if (XS) HIP_TRY(c, hipMemcpyAsync(XS + (size_t)k * (T + 1) * B * pd.ns, s0, sizeof(float) * (size_t)B * pd.ns, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(c, hipMemsetAsync(s.DONES, 0, sizeof(float) * R, st));
    HIP_TRY(c, hipMemsetAsync(s.ACC, 0, sizeof(double) * R, st));
    const int pbs = 64;
    const size_t psh = (size_t)(pd.ns + 2 * pd.pol.max_width) * pbs * sizeof(float);
    if (psh > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "policy too wide for k_dg_pre");
    if (psh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_dg_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psh));
    double g = 1.0;
    size_t pre_lds = 0;
    const dg_pre_mfma_t pre_mfma = dg_pre_mfma_select(c, &pre_lds);
    if (pre_lds) {
        hipLaunchKernelGGL((k_pre_mfma3_image<55, 21, 100, 50, 25>), dim3((unsigned)((P3<55, 21, 100, 50, 25>::IMG + 255) / 256)), dim3(256), 0, st, c->d_theta, s.PIMG);
        if (pre_lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pre_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre_lds));
    }
    for (int t = 0; t < T; ++t) {
        if (pre_mfma) hipLaunchKernelGGL(pre_mfma, dim3((B + 63) / 64, K), dim3(256), pre_lds, st, pd, B, c->d_theta, c->d_norm, (const float*)nullptr, 0LL, s);
        else hipLaunchKernelGGL(k_dg_pre, dim3((B + pbs - 1) / pbs, K), dim3(pbs), psh, st, pd, B, c->d_theta, c->d_norm, (const float*)nullptr, 0LL, s);
        dg_forward_layers(c, s, B, L, true, st);
        hipLaunchKernelGGL(k_dg_post, dim3((B + DG_POST_ROWS - 1) / DG_POST_ROWS, K), dim3(DG_POST_THREADS), 0, st, pd, B, T, t, g, c->d_norm, s, XS, WT);
        g *= gamma;
    }
    hipLaunchKernelGGL(k_dg_costs, dim3(K), dim3(256), 0, st, B, s.ACC, costs);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_dg_backward(metrpo_ctx* c, int B, int T, const float* XS, const float* WT, float* GM, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    const int K = pd.K, L = pd.dyn.n_layers;
    DgState s;
    int rc = dg_workspace(c, B, &s); if (rc) return rc;
    const size_t R = (size_t)K * B;
    HIP_TRY(c, hipMemsetAsync(s.LAM, 0, sizeof(float) * R * pd.ns, st));
    const int pbs = 64;
    const size_t psh = (size_t)(pd.ns + 2 * pd.pol.max_width) * pbs * sizeof(float);
    int prow = 0;
    for (int l = 1; l <= pd.pol.n_layers; ++l) prow += pd.pol.dims[l];
    const size_t bsh = (size_t)(pd.ns + prow + 2 * pd.pol.max_width) * pbs * sizeof(float);
    if (psh > 160 * 1024 || bsh > 160 * 1024) return set_err(c, METRPO_EUNSUPPORTED, "policy too wide for the GEMM-path sweeps");
    if (psh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_dg_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psh));
    if (bsh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)k_dg_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bsh));
    const long long xs_model = (long long)(T + 1) * B * pd.ns;
    size_t pre_lds = 0;
    const dg_pre_mfma_t pre_mfma = dg_pre_mfma_select(c, &pre_lds);
    if (pre_lds) {
        hipLaunchKernelGGL((k_pre_mfma3_image<55, 21, 100, 50, 25>), dim3((unsigned)((P3<55, 21, 100, 50, 25>::IMG + 255) / 256)), dim3(256), 0, st, c->d_theta, s.PIMG);
        if (pre_lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pre_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre_lds));
    }
    for (int t = T - 1; t >= 0; --t) {
        if (pre_mfma) hipLaunchKernelGGL(pre_mfma, dim3((B + 63) / 64, K), dim3(256), pre_lds, st, pd, B, c->d_theta, c->d_norm, XS + (size_t)t * B * pd.ns, xs_model, s);
        else hipLaunchKernelGGL(k_dg_pre, dim3((B + pbs - 1) / pbs, K), dim3(pbs), psh, st, pd, B, c->d_theta, c->d_norm,
                                XS + (size_t)t * B * pd.ns, xs_model, s);
        dg_forward_layers(c, s, B, L - 1, false, st);                             // hidden activations only
        hipLaunchKernelGGL(k_dg_mid, dim3((B + 127) / 128, K), dim3(128), 0, st, pd, B, T, t, c->d_norm, XS, WT, s);
        float* dz = s.DZa; float* dzn = s.DZb;
        for (int l = L - 1; l >= 0; --l) {
            const int n_in = pd.dyn.dims[l], n_out = pd.dyn.dims[l + 1];
            const float* Wl = c->d_dyn + pd.dyn.w_off[l];
            GemmEpi ep = {};
            if (l > 0) {
                ep.mask = s.H[l - 1]; ep.strideMask = (long long)B * n_in; ep.ldm = n_in;
                gemm_auto<EPI_RELU_MASK, false, true>(dz, (long long)B * n_out, n_out, Wl, pd.dyn.n_params, n_out, dzn, (long long)B * n_in, n_in, B, n_in,
                                                    n_out, K, ep, st);
            } else {
                gemm_auto<EPI_PLAIN, false, true>(dz, (long long)B * n_out, n_out, Wl, pd.dyn.n_params, n_out, dzn, (long long)B * n_in, n_in, B, n_in, n_out,
                                                K, ep, st);
            }
            float* tmp = dz; dz = dzn; dzn = tmp;
        }
        hipLaunchKernelGGL(k_dg_back, dim3((B + pbs - 1) / pbs, K), dim3(pbs), bsh, st, pd, B, T, t, c->d_theta, c->d_norm, XS, WT, dz, s, GM);
    }
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
