// MFMA path of the per-model DETERMINISTIC rollout of build_policy_graph (model_based_rl.py:106-151) for the small-MLP shapes
// (dynamics 2x64 relu, policy 2x32 tanh): forward sweep (= per-model validation costs, metrpo_validation_cost, and the
// stored trajectory of the BPTT update) and reverse sweep (adjoint of the policy mean for every (model, t, env); bptt.hip has
// the generic-width version and the description of the recursion).
//
// Mapping: grid.y = model i, a WAVE owns one tile of 16 envs of that model for the whole horizon (no cross-wave traffic, no
// barriers after the prologue).  Every layer and every layer-adjoint is a TRANSPOSED MFMA chain (v_mfma_f32_16x16x4_f32):
//   forward   H^T[unit][env]  = W^T . X^T      A = W^T fragments, B = activations in D layout of the previous layer
//   adjoint   dX^T[in][env]   = W . dH^T       A = W   fragments, B = deltas      in D layout of the next     layer
// so activations and deltas never leave registers between layers.  Constant factors are folded into the fragments: diff_std
// into the first adjoint layer, 1/in_std and the column drop into the last one, which is split into a state part (rows = state
// dims: lands in the layout of the residual path) and an action part (rows = action dims).
// Forward dynamics fragments are register-resident; adjoint fragments and the policy's live in an LDS image shared by the 4
// waves of a workgroup (same model).
#include "mfma_common.h"

enum { DET_FWD = 0, DET_BWD = 1 };

template <int ENV>
struct DetL {
    using C = Cfg<ENV, 64, 32>;
    static constexpr int NS = C::NS, NA = C::NA, OUT_CB = C::OUT_CB, NS_KS = C::NS_KS, NIN_KS = C::NIN_KS;
    static constexpr int RK = (NA < 4) ? NA : 4;                       // k-steps of the first policy-adjoint layer that are not padding
    // LDS image (floats): fragment tables [k-step][col-block][64 lanes], then biases
    static constexpr int O_PF0 = 0, O_PF1 = O_PF0 + NS_KS * 2 * 64, O_PF2 = O_PF1 + 8 * 2 * 64, O_PB2 = O_PF2 + 8 * 64,
                         O_PB1 = O_PB2 + RK * 2 * 64, O_PB0 = O_PB1 + 8 * 2 * 64, O_DB2 = O_PB0 + 8 * OUT_CB * 64,
                         O_DB1 = O_DB2 + 4 * OUT_CB * 4 * 64, O_DAS = O_DB1 + 16 * 4 * 64, O_DAA = O_DAS + 16 * OUT_CB * 64,
                         O_BD0 = O_DAA + 16 * 64, O_BD1 = O_BD0 + 64, O_BD2 = O_BD1 + 64, O_BP0 = O_BD2 + 16 * OUT_CB,
                         O_BP1 = O_BP0 + 32, O_BP2 = O_BP1 + 32, IMG = O_BP2 + 16;
    static constexpr int WV = ((16 * NS * 2 + 16 * NA + 3) / 4) * 4;     // per wave: ST | NX | ACT
    static constexpr int TOTAL = IMG + 4 * WV;
};

// cost of one env from its row-major tiles (env_helpers.py:601 / com_*_env.py cost_np_vec == cost_tf per sample)
template <int ENV>
__device__ __forceinline__ float cost_row(const float* xn, const float* u, int NS, int NA) {
    float su2 = 0.0f;
    for (int d = 0; d < NA; ++d) su2 = fmaf(u[d], u[d], su2);
    if (ENV == METRPO_ENV_SWIMMER) return -(xn[5] - 1e-2f * (su2 / (float)NA));
    if (ENV == METRPO_ENV_HALF_CHEETAH) return -fminf(fmaxf(xn[9] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
    if (ENV == METRPO_ENV_SNAKE) return -(xn[7] - 1e-2f * 0.5f * su2);
    if (ENV == METRPO_ENV_ANT) return -(xn[15] - 1e-2f * 0.5f * su2 + 0.05f);
    if (ENV == METRPO_ENV_HOPPER) {
        float pen = 0.0f;
        for (int i = 2; i < NS; ++i) pen += fmaxf(fabsf(xn[i]) - 100.0f, 0.0f);
        return -(xn[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0], 0.0f) - 10.0f * fmaxf(fabsf(xn[1]) - 0.2f, 0.0f) - pen);
    }
    return 0.0f;
}

template <int ENV, int MODE>
__global__ void __launch_bounds__(256, 2) k_det_mfma(int K, int B, int T, double gamma, const float* __restrict__ dynp,
                                                    const float* __restrict__ theta, const float* __restrict__ norm,
                                                    const float* __restrict__ s0, float* __restrict__ XS, float* __restrict__ WT,
                                                    float* __restrict__ GM, double* __restrict__ cost_part) {
    using L = DetL<ENV>;
    using C = typename L::C;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NIN = C::NIN, DH = 64, PH = 32, OUT_CB = C::OUT_CB, NS_KS = C::NS_KS,
                  NIN_KS = C::NIN_KS, RK = L::RK;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int model = blockIdx.y;
    const int b0 = (blockIdx.x * 4 + wave) * 16, b = b0 + c;
    const bool active = b < B;
    const float* __restrict__ pk = dynp + (size_t)model * C::PD;
    float* IMG = lds;
    float* ST = lds + L::IMG + wave * L::WV; float* NX = ST + 16 * NS; float* ACT = NX + 16 * NS;
    const float* in_std = norm + (NS + NA);
    const float* dstd_p = norm + 2 * (NS + NA) + NS;

    // ---------------- LDS image: fragment tables (see header) and biases; each element written by one thread ----------------
    auto chained = [&](int kk, int qq) { return 16 * (kk >> 2) + 4 * qq + (kk & 3); };
    for (int i = tid; i < L::IMG; i += 256) {
        float w = 0.0f;
        const int ln = i & 63, cc = ln & 15, qq = ln >> 4;
        if (i < L::O_PF1) { const int f = i >> 6, s = f >> 1, cb = f & 1, in = 4 * s + qq; if (in < NS) w = theta[C::pW0 + in * PH + 16 * cb + cc]; }
        else if (i < L::O_PF2) { const int f = (i - L::O_PF1) >> 6, kk = f >> 1, cb = f & 1; w = theta[C::pW1 + chained(kk, qq) * PH + 16 * cb + cc]; }
        else if (i < L::O_PB2) { const int kk = (i - L::O_PF2) >> 6; if (cc < NA) w = theta[C::pW2 + chained(kk, qq) * NA + cc]; }
        else if (i < L::O_PB1) { const int f = (i - L::O_PB2) >> 6, r = f >> 1, cb = f & 1, d = 4 * qq + r; if (d < NA) w = theta[C::pW2 + (16 * cb + cc) * NA + d]; }
        else if (i < L::O_PB0) { const int f = (i - L::O_PB1) >> 6, kk = f >> 1, cb = f & 1; w = theta[C::pW1 + (16 * cb + cc) * PH + chained(kk, qq)]; }
        else if (i < L::O_DB2) { const int f = (i - L::O_PB0) >> 6, kk = f / OUT_CB, cb = f % OUT_CB, dim = 16 * cb + cc; if (dim < NS) w = theta[C::pW0 + dim * PH + chained(kk, qq)]; }
        else if (i < L::O_DB1) {                 // first adjoint layer of the dynamics: k-steps (cbd, r) over state dims, diff_std folded in
            const int f = (i - L::O_DB2) >> 6, ks = f >> 2, cb = f & 3, dim = 16 * (ks >> 2) + 4 * qq + (ks & 3);
            if (dim < NS) w = pk[C::dW2 + (16 * cb + cc) * NS + dim] * dstd_p[dim];
        }
        else if (i < L::O_DAS) { const int f = (i - L::O_DB1) >> 6, kk = f >> 2, cb = f & 3; w = pk[C::dW1 + (16 * cb + cc) * DH + chained(kk, qq)]; }
        else if (i < L::O_DAA) {                 // last adjoint layer, state rows: column drop and 1/in_std folded in
            const int f = (i - L::O_DAS) >> 6, kk = f / OUT_CB, cb = f % OUT_CB, dim = 16 * cb + cc;
            if (dim >= NDROP && dim < NS) w = pk[C::dW0 + (dim - NDROP) * DH + chained(kk, qq)] / in_std[dim];
        }
        else if (i < L::O_BD0) { const int kk = (i - L::O_DAA) >> 6; if (cc < NA) w = pk[C::dW0 + (NS - NDROP + cc) * DH + chained(kk, qq)] / in_std[NS + cc]; }
        else if (i < L::O_BD1) w = pk[C::db0 + (i - L::O_BD0)];
        else if (i < L::O_BD2) w = pk[C::db1 + (i - L::O_BD1)];
        else if (i < L::O_BP0) { const int u = i - L::O_BD2; if (u < NS) w = pk[C::db2 + u]; }
        else if (i < L::O_BP1) w = theta[C::pb0 + (i - L::O_BP0)];
        else if (i < L::O_BP2) w = theta[C::pb1 + (i - L::O_BP1)];
        else { const int d = i - L::O_BP2; if (d < NA) w = theta[C::pb2 + d]; }
        IMG[i] = w;
    }
#define TAB2(off, ks, cb, ncb) IMG[(off) + (((ks) * (ncb)) + (cb)) * 64 + lane]
    // ---------------- register-resident forward dynamics fragments ----------------
    float wd0[NIN_KS][4], wd1[16][4], wd2[MODE == DET_FWD ? 16 : 1][OUT_CB];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int s = 0; s < NIN_KS; ++s) { const int i = 4 * s + q; wd0[s][cb] = (i < NIN) ? pk[C::dW0 + i * DH + 16 * cb + c] : 0.0f; }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wd1[kk][cb] = pk[C::dW1 + (16 * (kk >> 2) + 4 * q + (kk & 3)) * DH + 16 * cb + c];
    }
    if (MODE == DET_FWD) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) { const int o = 16 * cb + c; wd2[kk][cb] = (o < NS) ? pk[C::dW2 + (16 * (kk >> 2) + 4 * q + (kk & 3)) * NS + o] : 0.0f; }
    }
    float nmean[NIN_KS], nstd[NIN_KS];
    int nsrc[NIN_KS];
#pragma unroll
    for (int s = 0; s < NIN_KS; ++s) {
        const int i = 4 * s + q;
        int f = 0;
        if (i < NS - NDROP) { f = i + NDROP; nsrc[s] = f; }
        else if (i < NIN) { f = NS + (i - (NS - NDROP)); nsrc[s] = -(i - (NS - NDROP)) - 1; }
        else { nsrc[s] = -1000000; }
        nmean[s] = (i < NIN) ? norm[f] : 0.0f;
        nstd[s] = (i < NIN) ? 1.0f / norm[(NS + NA) + f] : 1.0f;
    }
    f32x4 dmean[OUT_CB], dstd[OUT_CB];
#pragma unroll
    for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int dim = 16 * cb + 4 * q + r;
            dmean[cb][r] = (dim < NS) ? norm[2 * (NS + NA) + dim] : 0.0f;
            dstd[cb][r] = (dim < NS) ? norm[2 * (NS + NA) + NS + dim] : 0.0f;
        }
    const size_t xs_model = (size_t)model * (T + 1) * B * NS;
    __syncthreads();

    // one step of the forward computations shared by both modes: policy chain -> clip -> ACT; dynamics layers 0, 1
    f32x4 p0[2], p1[2], mu, h0[4], h1[4];
    auto step_forward = [&]() {
        p0[0] = *(const f32x4*)&IMG[L::O_BP0 + 4 * q]; p0[1] = *(const f32x4*)&IMG[L::O_BP0 + 16 + 4 * q];
#pragma unroll
        for (int s = 0; s < NS_KS; ++s) {
            const int f = 4 * s + q;
            const float x = (f < NS) ? ST[c * NS + f] : 0.0f;
            p0[0] = MFMA16(TAB2(L::O_PF0, s, 0, 2), x, p0[0]);
            p0[1] = MFMA16(TAB2(L::O_PF0, s, 1, 2), x, p0[1]);
        }
        p1[0] = *(const f32x4*)&IMG[L::O_BP1 + 4 * q]; p1[1] = *(const f32x4*)&IMG[L::O_BP1 + 16 + 4 * q];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) p0[cb][r] = tanh_fast(p0[cb][r]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            p1[0] = MFMA16(TAB2(L::O_PF1, kk, 0, 2), p0[kk >> 2][kk & 3], p1[0]);
            p1[1] = MFMA16(TAB2(L::O_PF1, kk, 1, 2), p0[kk >> 2][kk & 3], p1[1]);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) p1[cb][r] = tanh_fast(p1[cb][r]);
        f32x4 m0 = *(const f32x4*)&IMG[L::O_BP2 + 4 * q], m1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            m0 = MFMA16(IMG[L::O_PF2 + kk * 64 + lane], p1[kk >> 2][kk & 3], m0);
            m1 = MFMA16(IMG[L::O_PF2 + (kk + 1) * 64 + lane], p1[(kk + 1) >> 2][(kk + 1) & 3], m1);
        }
        mu = m0 + m1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int d = 4 * q + r; if (d < NA) ACT[c * NA + d] = fminf(fmaxf(mu[r], -1.0f), 1.0f); }      // :128
        wave_lds_sync();
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) h0[cb] = *(const f32x4*)&IMG[L::O_BD0 + 16 * cb + 4 * q];
#pragma unroll
        for (int s = 0; s < NIN_KS; ++s) {
            float x = 0.0f;
            if (nsrc[s] >= 0) x = ST[c * NS + nsrc[s]];
            else if (nsrc[s] > -1000000) x = ACT[c * NA + (-nsrc[s] - 1)];
            x = (nsrc[s] > -1000000) ? (x - nmean[s]) * nstd[s] : 0.0f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) h0[cb] = MFMA16(wd0[s][cb], x, h0[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            h1[cb] = *(const f32x4*)&IMG[L::O_BD1 + 16 * cb + 4 * q];
#pragma unroll
            for (int r = 0; r < 4; ++r) h0[cb][r] = fmaxf(h0[cb][r], 0.0f);
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) h1[cb] = MFMA16(wd1[kk][cb], h0[kk >> 2][kk & 3], h1[cb]);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[cb][r] = fmaxf(h1[cb][r], 0.0f);
    };
    const int lim = min(16, max(0, B - b0)) * NS;                            // floats of this tile in one [B][ns] slice

    if (MODE == DET_FWD) {
        // ------------------------------------------------ forward sweep ------------------------------------------------
        for (int i = lane; i < 16 * NS; i += 64) ST[i] = (i < lim) ? s0[(size_t)b0 * NS + i] : 0.0f;
        wave_lds_sync();
        double acc = 0.0, g = 1.0;
        float dones = 0.0f;
        for (int t = 0; t < T; ++t) {
            if (XS != nullptr) for (int i = lane; i < lim; i += 64) XS[xs_model + ((size_t)t * B + b0) * NS + i] = ST[i];
            step_forward();
            f32x4 oa[OUT_CB], ob[OUT_CB];
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) { oa[cb] = *(const f32x4*)&IMG[L::O_BD2 + 16 * cb + 4 * q]; ob[cb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2)
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb) {
                    oa[cb] = MFMA16(wd2[kk][cb], h1[kk >> 2][kk & 3], oa[cb]);
                    ob[cb] = MFMA16(wd2[kk + 1][cb], h1[(kk + 1) >> 2][(kk + 1) & 3], ob[cb]);
                }
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dim = 16 * cb + 4 * q + r;
                    if (dim < NS) NX[c * NS + dim] = fmaf(dstd[cb][r], oa[cb][r] + ob[cb][r], dmean[cb][r]) + ST[c * NS + dim];      // training.py:257
                }
            wave_lds_sync();
            if (q == 0) {                                        // one lane per env: cost, dones, weight
                const float cst = cost_row<ENV>(NX + c * NS, ACT + c * NA, NS, NA);
                const float live = 1.0f - dones;
                if (ENV == METRPO_ENV_ANT) {
                    bool fin = true;
                    for (int i = 0; i < NS; ++i) fin = fin && isfinite(NX[c * NS + i]);
                    const float z = NX[c * NS + 2];
                    dones = fmaxf(dones, ((z >= 0.2f) && (z <= 1.0f) && fin) ? 0.0f : 1.0f);
                }
                if (active) {
                    acc += g * (double)(cst * live);
                    if (WT != nullptr) WT[((size_t)model * T + t) * B + b] = (float)(g * (double)live / ((double)B * (double)K));
                }
            }
            g *= gamma;
            for (int i = lane; i < 16 * NS; i += 64) ST[i] = NX[i];
            wave_lds_sync();
        }
        if (XS != nullptr) for (int i = lane; i < lim; i += 64) XS[xs_model + ((size_t)T * B + b0) * NS + i] = ST[i];
        // deterministic cost reduction: lanes -> wave (shuffles) -> one slot per (model, tile)
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) cost_part[(size_t)model * (gridDim.x * 4) + blockIdx.x * 4 + wave] = acc / (double)B;
    } else {
        // ------------------------------------------------ reverse sweep ------------------------------------------------
        f32x4 lam[OUT_CB];
#pragma unroll
        for (int cb = 0; cb < OUT_CB; ++cb) lam[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = T - 1; t >= 0; --t) {
            for (int i = lane; i < 16 * NS; i += 64) {
                ST[i] = (i < lim) ? XS[xs_model + ((size_t)t * B + b0) * NS + i] : 0.0f;
                NX[i] = (i < lim) ? XS[xs_model + ((size_t)(t + 1) * B + b0) * NS + i] : 0.0f;
            }
            const float w = active ? WT[((size_t)model * T + t) * B + b] : 0.0f;
            wave_lds_sync();
            step_forward();
            // ---- cost adjoint in D layout: G = lambda_{t+1} + w dc/dx_next (state dims), gU = w dc/du (action dims) ----
            f32x4 G[OUT_CB], gU = {0.f, 0.f, 0.f, 0.f};
            float cu = 0.0f;
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) G[cb] = lam[cb];
            auto addG = [&](int dim, float v) {                  // add v to the lane/register that owns state dim `dim`
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (16 * cb + 4 * q + r == dim) G[cb][r] += v;
            };
            if (ENV == METRPO_ENV_SWIMMER) { addG(5, -w); cu = w * 1e-2f * 2.0f / (float)NA; }
            else if (ENV == METRPO_ENV_SNAKE) { addG(7, -w); cu = w * 1e-2f; }
            else if (ENV == METRPO_ENV_ANT) { addG(15, -w); cu = w * 1e-2f; }
            else if (ENV == METRPO_ENV_HALF_CHEETAH) {
                float su2 = 0.0f;
                for (int d = 0; d < NA; ++d) su2 = fmaf(ACT[c * NA + d], ACT[c * NA + d], su2);
                const float inner = NX[c * NS + 9] - 1e-1f * 0.5f * su2;
                const float p = (inner >= -10.0f && inner <= 10.0f) ? 1.0f : 0.0f;
                addG(9, -w * p); cu = w * p * 1e-1f;
            } else if (ENV == METRPO_ENV_HOPPER) {
                cu = w * 0.01f;
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dim = 16 * cb + 4 * q + r;
                        if (dim < NS) {
                            const float v = NX[c * NS + dim];
                            float gsum = 0.0f;
                            if (dim == 5) gsum -= w;
                            if (dim == 0 && 0.45f - v > 0.0f) gsum -= w * 10.0f;
                            if (dim == 1 && fabsf(v) - 0.2f > 0.0f) gsum += w * 10.0f * (v > 0.0f ? 1.0f : -1.0f);
                            if (dim >= 2 && fabsf(v) - 100.0f > 0.0f) gsum += w * (v > 0.0f ? 1.0f : -1.0f);
                            G[cb][r] += gsum;
                        }
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int d = 4 * q + r; if (d < NA) gU[r] = cu * ACT[c * NA + d]; }
            // ---- dynamics adjoint chain: d2 = W2.(diff_std G) * relu'(h1); d1 = W1.d2 * relu'(h0); gS += W0s.d1 ; gU += W0a.d1 ----
            f32x4 d2[4], d1[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) d2[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4 * OUT_CB; ++ks)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) d2[cb] = MFMA16(TAB2(L::O_DB2, ks, cb, 4), G[ks >> 2][ks & 3], d2[cb]);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                d1[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) d2[cb][r] = (h1[cb][r] > 0.0f) ? d2[cb][r] : 0.0f;
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) d1[cb] = MFMA16(TAB2(L::O_DB1, kk, cb, 4), d2[kk >> 2][kk & 3], d1[cb]);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[cb][r] = (h0[cb][r] > 0.0f) ? d1[cb][r] : 0.0f;
            f32x4 gS[OUT_CB];
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) gS[cb] = G[cb];                       // residual connection x' = x + ...
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb) gS[cb] = MFMA16(TAB2(L::O_DAS, kk, cb, OUT_CB), d1[kk >> 2][kk & 3], gS[cb]);
                gU = MFMA16(IMG[L::O_DAA + kk * 64 + lane], d1[kk >> 2][kk & 3], gU);
            }
            // ---- clip gate (tf.clip_by_value passes the gradient inside [min, max]) and the mean-adjoint output ----
            f32x4 gm;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 4 * q + r;
                gm[r] = (d < NA && mu[r] >= -1.0f && mu[r] <= 1.0f) ? gU[r] : 0.0f;
                if (d < NA && active) GM[(((size_t)model * (T + 1) + t) * B + b) * NA + d] = gm[r];
            }
            // ---- policy input adjoint chain ----
            f32x4 e1[2], e0[2];
            e1[0] = e1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                e1[0] = MFMA16(TAB2(L::O_PB2, r, 0, 2), gm[r], e1[0]);
                e1[1] = MFMA16(TAB2(L::O_PB2, r, 1, 2), gm[r], e1[1]);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                e0[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) e1[cb][r] *= fmaf(-p1[cb][r], p1[cb][r], 1.0f);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                e0[0] = MFMA16(TAB2(L::O_PB1, kk, 0, 2), e1[kk >> 2][kk & 3], e0[0]);
                e0[1] = MFMA16(TAB2(L::O_PB1, kk, 1, 2), e1[kk >> 2][kk & 3], e0[1]);
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) e0[cb][r] *= fmaf(-p0[cb][r], p0[cb][r], 1.0f);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb) gS[cb] = MFMA16(TAB2(L::O_PB0, kk, cb, OUT_CB), e0[kk >> 2][kk & 3], gS[cb]);
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) lam[cb] = gS[cb];                     // lambda_t
            wave_lds_sync();
        }
    }
#undef TAB2
}

// fixed-order sum of the per-tile cost partials of one model
// err (may be NULL): time-out cell of the launch that wrote the partials (val_err_cell: the resident validation launches; cleared in front of them).  A
// resident validation launch whose grid was not co-resident leaves invalid partial sums; callers of metrpo_validation_cost read no error cell, so the
// costs themselves carry the failure: NaN instead of plausible garbage feeding model selection and early stopping.  Every other producer of partials
// passes NULL: an unrelated rollout's sticky time-out must not turn valid costs into NaN.
__global__ void k_det_cost_reduce(int n_part, const double* __restrict__ part, double* __restrict__ costs, const double* __restrict__ err) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n_part; ++i) s += part[(size_t)blockIdx.x * n_part + i];
        if (err != nullptr && *err != 0.0) s = __builtin_nan("");
        costs[blockIdx.x] = s;
    }
}

// costs[k] = the n_part partials of model k added in index order (also the generic forward kernels' block sums: rollout_generic.hip, bptt.hip)
int launch_det_cost_reduce(metrpo_ctx* c, int n_part, const double* part, double* costs, hipStream_t st, const double* err) {
    hipLaunchKernelGGL(k_det_cost_reduce, dim3(c->pd.K), dim3(64), 0, st, n_part, part, costs, err);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

// -------------------------------------------------------------------------------------------------
typedef void (*det_kernel_t)(int, int, int, double, const float*, const float*, const float*, const float*, float*, float*, float*, double*);
struct DetEntry { int env; det_kernel_t fwd, bwd; int lds_floats; };
#define DENTRY(E) {E, k_det_mfma<E, DET_FWD>, k_det_mfma<E, DET_BWD>, DetL<E>::TOTAL}
static const DetEntry kDet[] = {DENTRY(METRPO_ENV_SWIMMER), DENTRY(METRPO_ENV_HALF_CHEETAH), DENTRY(METRPO_ENV_HOPPER), DENTRY(METRPO_ENV_SNAKE),
                                DENTRY(METRPO_ENV_ANT)};

// table index or -1: same shape conditions as the MFMA rollouts (dynamics 2x64 relu, policy 2x32 tanh, known env dims)
int det_mfma_select(const metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    const bool exact = c->mfma_cfg >= 0 && pd.dyn.n_layers == 3 && pd.dyn.dims[1] == 64 && pd.dyn.dims[2] == 64;
    const bool padded = c->coop_cfg < 0 && c->coop_pad_cfg >= 0;            // two hidden layers of at most 64 units: the zero-padded copy in the 64 x 64 layout (api.hip)
    if ((!exact && !padded) || pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32)
        return -1;
    for (int i = 0; i < (int)(sizeof(kDet) / sizeof(kDet[0])); ++i)
        if (kDet[i].env == pd.env) return i;
    return -1;
}

// part: >= K * tiles4 doubles of scratch (tiles4 = 4 * ceil(B/64)).  XS / WT may be NULL (validation cost only).
int launch_det_forward(metrpo_ctx* c, int idx, const float* s0, int B, int T, double gamma, float* XS, float* WT, double* part, double* costs,
                       hipStream_t st) {
    const DetEntry& en = kDet[idx];
    const size_t sh = sizeof(float) * (size_t)en.lds_floats;
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)en.fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const int gx = (B + 63) / 64;
    const float* dyn = c->d_dyn;
    if (c->det_padded) { const int rc = launch_pad_dyn(c, st); if (rc) return rc; dyn = c->d_dyn_pad; }
    hipLaunchKernelGGL(en.fwd, dim3(gx, c->pd.K), dim3(256), sh, st, c->pd.K, B, T, gamma, dyn, c->d_theta, c->d_norm, s0, XS, WT,
                       (float*)nullptr, part);
    hipLaunchKernelGGL(k_det_cost_reduce, dim3(c->pd.K), dim3(64), 0, st, gx * 4, part, costs, (const double*)nullptr);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_det_backward(metrpo_ctx* c, int idx, int B, int T, const float* XS, const float* WT, float* GM, hipStream_t st) {
    const DetEntry& en = kDet[idx];
    const size_t sh = sizeof(float) * (size_t)en.lds_floats;
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)en.bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const float* dyn = c->d_dyn;
    if (c->det_padded) { const int rc = launch_pad_dyn(c, st); if (rc) return rc; dyn = c->d_dyn_pad; }
    hipLaunchKernelGGL(en.bwd, dim3((B + 63) / 64, c->pd.K), dim3(256), sh, st, c->pd.K, B, T, 1.0, dyn, c->d_theta, c->d_norm,
                       (const float*)nullptr, const_cast<float*>(XS), const_cast<float*>(WT), GM, (double*)nullptr);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
