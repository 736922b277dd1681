// Resident rollout for SMALL batches of 2 x 512 dynamics ensembles (the reference's own params-swimmer.json: K = 5, B = 100 envs,
// three rounds of 200 steps): the whole time loop in ONE launch, weights resident in LDS, the steps chained through 8-byte packets.
// Same reference path as the other rollout kernels (samplers/vectorized_sampler.py:45-116, env_helpers.py:597-635, training.py:218-269).
//
// Why: at B = 100 a step of the step-wise GEMM path (rollout_gemm.hip) is four dependent launches of 4-19 us that each re-read the
// weights from L2 and leave most CUs idle; 600 steps take 9 ms whatever the launch mechanism.  Here
//   * a COMPUTE workgroup owns (round, model k, slice of WS hidden-2 units): W0 (all of it), its W1 columns and its W2 rows sit in
//     LDS as MFMA fragments for the whole rollout (100 KB).  Wave w owns env tile w (16 envs): per step it recomputes hidden layer 0
//     for its tile (DH/16 x NIN_KS MFMAs -- cheaper than fetching the activations from another CU), contracts it with the slice
//     (WS/16 x DH/4 MFMAs, the layer-0 results feeding the next MFMA from registers) and emits the slice's contribution to the
//     output layer: [ns x 16 envs] partial sums;
//   * a POST wave owns (round, env tile): policy forward (the MFMA chain of k_big_pre_mfma), action noise, normalised input -> X
//     packets; then it adds the DH/WS partials of each env's selected model in slice order, applies the residual, reward, done,
//     reset, and writes the trajectory rows.  R x ceil(B/16) post waves on the CUs the compute grid leaves free;
//   * hand-over in both directions by {32 data bits | 32-bit step stamp} packets in an uncached exchange region (agent-scope relaxed
//     8-byte stores / loads: no fence, no flag, no grid barrier; xchg_device.h uses the same idea between GPUs).  A wave only waits
//     for ITS tile, so the two waves of a SIMD drift apart and one computes while the other waits for its hand-over.
// All K heads are evaluated every step (as the reference's graph does); only simple sampling modes (step_rand / eps_rand / one_model).
// Everything else (B > 128, other widths, Ant-sized inputs, model_mean_std / model_med, chunks with a stop flag) stays on rollout_gemm.hip.
#include "mfma_common.h"

struct ResidentK {
    int R, round0, rounds_total, NT, NSL, U, PW, steps;   // rounds of this launch, first round, env tiles, slices, compute blocks, post waves per block, steps per round
    unsigned int seq0;                                    // packets of local step tau carry seq0 + tau + 1
    unsigned long long* X; unsigned long long* P; unsigned int* abort_cell;
    double* err;
};

__device__ __forceinline__ unsigned long long res_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void res_st(unsigned long long* p, unsigned int seq, float v) {
    __hip_atomic_store(p, ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded wait bookkeeping of a polling wave: true -> give up (time limit, or another wave of this launch already gave up)
struct ResSpin {
    int spins = 0; unsigned long long t0 = 0;
    __device__ __forceinline__ bool give_up(const ResidentK& z) {
        ++spins;
        __builtin_amdgcn_s_sleep(4);
        if ((spins & 63) != 0) return false;
        if (__hip_atomic_load(z.abort_cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == z.seq0 + 1u) return true;
        if (t0 == 0) { t0 = wall_clock64(); return false; }
        if (wall_clock64() - t0 > 200000000ull) {             // 2 s at 100 MHz: a workgroup of this launch is not running
            __hip_atomic_store(z.abort_cell, z.seq0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *z.err = 1.0;
            return true;
        }
        return false;
    }
};

// ---- compute role ---------------------------------------------------------------------------------------------------------------------
template <int NS, int NIN, int DH, int WS>
__device__ __forceinline__ void resident_compute(const ProblemDesc& pd, const ResidentK& z, const float* __restrict__ dyn, float* lds) {
    constexpr int NIN_KS = cdiv(NIN, 4), KS4 = cdiv(NIN_KS, 4), J = DH / 16, MT = WS / 16, OUT_CB = cdiv(NS, 16), NSP = 16 * OUT_CB;
    constexpr int O_W1 = 0, O_W2 = O_W1 + J * KS4 * 256, O_W3 = O_W2 + MT * J * 256, O_B1 = O_W3 + OUT_CB * MT * 256, O_B2 = O_B1 + DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int K = pd.K, NSL = z.NSL;
    const int u = blockIdx.x, rho = u / (K * NSL), k = (u / NSL) % K, sl = u % NSL, col0 = sl * WS;
    const float* __restrict__ W = dyn + (size_t)k * pd.dyn.n_params;
    const float* __restrict__ W0 = W + pd.dyn.w_off[0];
    const float* __restrict__ W1 = W + pd.dyn.w_off[1];
    const float* __restrict__ W2 = W + pd.dyn.w_off[2];
    // fragment images.  MFMA16(a, b, acc): a = A[m = lane & 15][k = lane >> 4], b = B[k = lane >> 4][n = lane & 15], acc[r] = D[4 (lane >> 4) + r][n]:
    // a layer's output register r of lane (c, q) is unit 4q + r of its 16-unit tile, i.e. exactly the B operand of k-slot q of the
    // next layer's MFMA number r -- so the next layer's A fragments are stored in that order and nothing is ever transposed.
    for (int i = tid; i < J * KS4 * 256; i += 512) {                 // [j][g][lane][e]: W0[input 4 (4g + e) + q][unit 16 j + c]
        const int e = i & 3, ln = (i >> 2) & 63, g = (i >> 8) % KS4, j = i / (256 * KS4);
        const int in = 4 * (4 * g + e) + (ln >> 4);
        lds[O_W1 + i] = (in < NIN) ? W0[(size_t)in * DH + 16 * j + (ln & 15)] : 0.0f;
    }
    for (int i = tid; i < MT * J * 256; i += 512) {                  // [mt][j][lane][r]: W1[unit 16 j + 4q + r][column col0 + 16 mt + c]
        const int r = i & 3, ln = (i >> 2) & 63, j = (i >> 8) % J, mt = i / (256 * J);
        lds[O_W2 + i] = W1[(size_t)(16 * j + 4 * (ln >> 4) + r) * DH + col0 + 16 * mt + (ln & 15)];
    }
    for (int i = tid; i < OUT_CB * MT * 256; i += 512) {             // [ocb][mt][lane][r]: W2[unit col0 + 16 mt + 4q + r][dim 16 ocb + c]
        const int r = i & 3, ln = (i >> 2) & 63, mt = (i >> 8) % MT, ocb = i / (256 * MT);
        const int dim = 16 * ocb + (ln & 15);
        lds[O_W3 + i] = (dim < NS) ? W2[(size_t)(col0 + 16 * mt + 4 * (ln >> 4) + r) * NS + dim] : 0.0f;
    }
    for (int i = tid; i < DH; i += 512) lds[O_B1 + i] = W[pd.dyn.b_off[0] + i];
    for (int i = tid; i < WS; i += 512) lds[O_B2 + i] = W[pd.dyn.b_off[1] + col0 + i];
    __syncthreads();
    if (wave >= z.NT) return;
    const unsigned long long* xp = z.X + ((size_t)(rho * z.NT + wave) * (4 * NIN_KS)) * 16 + c;
    unsigned long long* pp = z.P + ((((size_t)(rho * z.NT + wave) * K + k) * NSL + sl) * NSP) * 16 + c;
    const f32x4* W1I = (const f32x4*)(lds + O_W1) + lane;
    const f32x4* W2I = (const f32x4*)(lds + O_W2) + lane;
    const f32x4* W3I = (const f32x4*)(lds + O_W3) + lane;
    for (int tau = 0; tau < z.steps; ++tau) {
        const unsigned int seq = z.seq0 + (unsigned int)tau + 1u;
        float x[NIN_KS];
        {
            ResSpin sp;
            for (;;) {
                unsigned long long pk[NIN_KS];
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < NIN_KS; ++kk) pk[kk] = res_ld(xp + (4 * kk + q) * 16);
#pragma unroll
                for (int kk = 0; kk < NIN_KS; ++kk) { ok = ok && ((unsigned int)(pk[kk] >> 32) == seq); x[kk] = __uint_as_float((unsigned int)pk[kk]); }
                if (__all(ok)) break;
                if (sp.give_up(z)) return;
            }
        }
        f32x4 a2[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { a2[mt][0] = *(const f32x4*)&lds[O_B2 + 16 * mt + 4 * q]; a2[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        auto layer0 = [&](int j) {
            f32x4 h = *(const f32x4*)&lds[O_B1 + 16 * j + 4 * q];
#pragma unroll
            for (int g = 0; g < KS4; ++g) {
                const f32x4 wf = W1I[(j * KS4 + g) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (4 * g + e < NIN_KS) h = MFMA16(wf[e], x[4 * g + e], h);
            }
            return h;
        };
        f32x4 hn = layer0(0);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = relu1(hn[r]);
            if (j + 1 < J) hn = layer0(j + 1);                        // the next tile's short dependent chain sits between this tile's independent MFMAs
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 wf = W2I[(mt * J + j) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) a2[mt][j & 1] = MFMA16(wf[r], h[r], a2[mt][j & 1]);
            }
        }
        f32x4 o[OUT_CB];
#pragma unroll
        for (int ocb = 0; ocb < OUT_CB; ++ocb) o[ocb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 h2;
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[r] = relu1(a2[mt][0][r] + a2[mt][1][r]);
#pragma unroll
            for (int ocb = 0; ocb < OUT_CB; ++ocb) {
                const f32x4 wf = W3I[(ocb * MT + mt) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[ocb] = MFMA16(wf[r], h2[r], o[ocb]);
            }
        }
#pragma unroll
        for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int dim = 16 * ocb + 4 * q + r; if (dim < NS) res_st(pp + dim * 16, seq, o[ocb][r]); }
    }
}

// ---- post role ------------------------------------------------------------------------------------------------------------------------
template <int ENV>
__device__ __forceinline__ void resident_post(const ProblemDesc& pd, const RolloutK& r, const ResidentK& z, const float* __restrict__ dyn,
                                              const float* __restrict__ theta, const float* __restrict__ norm, float* lds) {
    using C = Cfg<ENV, 64, 32>;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NIN = C::NIN, PH = 32, NS_KS = C::NS_KS, NIN_KS = C::NIN_KS, OUT_CB = C::OUT_CB, NSP = C::NSP;
    constexpr int O_PF1 = NS_KS * 2 * 64, O_PF2 = O_PF1 + 16 * 64, O_B0 = O_PF2 + 8 * 64, O_B1 = O_B0 + 32, O_B2 = O_B1 + 32, IMG = ((O_B2 + 16 + 3) / 4) * 4;
    constexpr int PW_LDS = 2 * 16 * NS + 2 * 16 * NA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    for (int i = tid; i < IMG; i += 512) {                            // policy fragment image (layout of k_big_pre_mfma, rollout_gemm.hip)
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
    __syncthreads();
    const int g = ((int)blockIdx.x - z.U) * z.PW + wave;
    if (wave >= z.PW || g >= z.R * z.NT) return;
    const int rho = g / z.NT, w = g % z.NT, round = z.round0 + rho;
    const int K = pd.K, NSL = z.NSL, B = r.B;
    float* ST = lds + IMG + wave * PW_LDS; float* NX = ST + 16 * NS; float* UA = NX + 16 * NS; float* XA = UA + 16 * NA;
    const int b0 = w * 16, b = b0 + c;
    const bool active = b < B;
    const int bc = active ? b : 0;
    const uint64_t genv = r.stream_offset + (uint64_t)bc;
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    const float* diff_mean = norm + 2 * (NS + NA); const float* diff_std = diff_mean + NS;
    const float* __restrict__ log_std = theta + C::pLS;
    const int lim = min(16, max(0, B - b0)) * NS;
    unsigned long long* xp = z.X + ((size_t)(rho * z.NT + w) * (4 * NIN_KS)) * 16 + c;
    const unsigned long long* pbase = z.P + ((size_t)(rho * z.NT + w) * K) * NSL * NSP * 16 + c;
    // ---- state at the first step of this round (every lane of an env computes the env's scalars redundantly: no broadcast needed)
    int ts = 0, cur_model = 0;
    {
        int row = 0;
        if (round == 0 && r.init_obs != nullptr) { cur_model = r.init_model[bc]; ts = r.init_ts[bc]; row = -1; }
        else if (round == 0) {                                        // vec_env.reset() (env_helpers.py:585-595)
            const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
            row = (r.reset_idx != nullptr) ? r.reset_idx[bc] : rng_index(d0.x, r.n_pool);
            cur_model = (r.reset_model != nullptr) ? r.reset_model[bc] : rng_index(d0.y, K);
        } else {                                                      // the reset that ends step round * steps - 1 (k_big_post's reset branch)
            const int t_prev = round * z.steps - 1;
            const uint4 dp = rng_draw(r.seed, genv, r.t0 + t_prev, RNG_STEP, 0);
            const size_t rb = (size_t)(t_prev + 1) * B + bc;
            row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dp.w, r.n_pool);
            cur_model = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dp.z, K);
        }
        for (int i = q; i < NS; i += 4)
            ST[c * NS + i] = !active ? 0.0f : (row < 0 ? r.init_obs[(size_t)bc * NS + i] : r.pool[(size_t)row * NS + i]);
    }
    wave_lds_sync();
    for (int tau = 0; tau < z.steps; ++tau) {
        const unsigned int seq = z.seq0 + (unsigned int)tau + 1u;
        const int t_loc = round * z.steps + tau;                      // row of the trajectory tensors; draws are keyed by r.t0 + t_loc
        const size_t tb = (size_t)t_loc * B + bc;
        // ---- policy.get_actions (MFMA chain of k_big_pre_mfma)
        f32x4 p0[2], p1[2];
        p0[0] = *(const f32x4*)&lds[O_B0 + 4 * q]; p0[1] = *(const f32x4*)&lds[O_B0 + 16 + 4 * q];
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) {
            const int f = 4 * s_ + q;
            const float xs = (f < NS) ? ST[c * NS + f] : 0.0f;
            p0[0] = MFMA16(lds[(s_ * 2 + 0) * 64 + lane], xs, p0[0]);
            p0[1] = MFMA16(lds[(s_ * 2 + 1) * 64 + lane], xs, p0[1]);
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
        // ---- actions: lane (c, q) owns action dims 4q .. 4q+3 = Philox chunks 2q, 2q+1 (chunk 0 = the step block), as k_big_pre_mfma
        const uint4 dstep = rng_draw(r.seed, genv, r.t0 + t_loc, RNG_STEP, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d0 = 4 * q + 2 * h;
            if (d0 >= NA) continue;
            float zz[2] = {0.f, 0.f};
            if (!r.determ && r.eps == nullptr) {
                const uint4 blk = (d0 == 0) ? dstep : rng_draw(r.seed, genv, r.t0 + t_loc, RNG_STEP, d0 >> 1);
                normal2(blk.x, blk.y, zz[0], zz[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = d0 + j;
                if (d >= NA) continue;
                const float m = mu[2 * h + j];
                float a = m;
                if (!r.determ) a = fmaf((r.eps != nullptr) ? r.eps[tb * NA + d] : zz[j], __expf(fmaxf(log_std[d], LOG_MIN_STD)), m);
                if (active) { r.act[tb * NA + d] = a; r.mean[tb * NA + d] = m; }
                const float ac = fminf(fmaxf(a, -1.0f), 1.0f);            // env_helpers.py:599
                UA[c * NA + d] = ac;
                XA[c * NA + d] = (ac - in_mean[NS + d]) / in_std[NS + d];
            }
        }
        wave_lds_sync();
        // ---- normalised, dropped input of the dynamics nets (training.py:228,146-151) -> X packets: element 4 kk + q of env c
#pragma unroll
        for (int kk = 0; kk < NIN_KS; ++kk) {
            const int f = 4 * kk + q;
            float v = 0.0f;
            if (active && f < NS - NDROP) v = (ST[c * NS + f + NDROP] - in_mean[f + NDROP]) / in_std[f + NDROP];
            else if (active && f < NIN) v = XA[c * NA + f - (NS - NDROP)];
            res_st(xp + f * 16, seq, v);
        }
        if (lim > 0) { const size_t base = ((size_t)t_loc * B + b0) * NS; for (int i = lane; i < lim; i += 64) r.obs[base + i] = ST[i]; }
        // ---- which head this env follows this step (env_helpers.py:617-634)
        int sel = cur_model;
        if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (r.model_idx != nullptr) ? r.model_idx[tb] : rng_index(dstep.z, K);
        if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
        // ---- output layer of head `sel`: bias + the slices' partial sums in slice order, dims 16 ocb + 4q + rr of env c
        const float* __restrict__ b2 = dyn + (size_t)sel * pd.dyn.n_params + pd.dyn.b_off[2];
        const unsigned long long* pq = pbase + (size_t)sel * NSL * NSP * 16;
        float outv[OUT_CB][4];
        {
            ResSpin sp;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int dim = 16 * ocb + 4 * q + rr;
                        float acc = (dim < NS) ? b2[dim] : 0.0f;
                        if (active && dim < NS) {
                            for (int s0 = 0; s0 < NSL; s0 += 8) {
                                unsigned long long pk[8];
#pragma unroll
                                for (int s1 = 0; s1 < 8; ++s1) pk[s1] = (s0 + s1 < NSL) ? res_ld(pq + ((size_t)(s0 + s1) * NSP + dim) * 16) : 0ull;
#pragma unroll
                                for (int s1 = 0; s1 < 8; ++s1)
                                    if (s0 + s1 < NSL) { ok = ok && ((unsigned int)(pk[s1] >> 32) == seq); acc += __uint_as_float((unsigned int)pk[s1]); }
                            }
                        }
                        outv[ocb][rr] = acc;
                    }
                if (__all(ok)) break;
                if (sp.give_up(z)) return;
            }
        }
        // ---- de-normalise + residual (training.py:257)
#pragma unroll
        for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * ocb + 4 * q + rr;
                if (dim < NS) NX[c * NS + dim] = fmaf(diff_std[dim], outv[ocb][rr], diff_mean[dim]) + ST[c * NS + dim];
            }
        wave_lds_sync();
        // ---- reward (env_helpers.py:601), done (:603-604), reset (:585-595); per-env scalars on every lane of the env
        const float* xn = NX + c * NS;
        float su2 = 0.0f;
#pragma unroll
        for (int d = 0; d < NA; ++d) { const float a = UA[c * NA + d]; su2 = fmaf(a, a, su2); }
        float cost = 0.0f;
        if constexpr (ENV == METRPO_ENV_SWIMMER) cost = -(xn[5] - 1e-2f * (su2 / (float)NA));
        else if constexpr (ENV == METRPO_ENV_HALF_CHEETAH) cost = -fminf(fmaxf(xn[9] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
        else if constexpr (ENV == METRPO_ENV_ANT) cost = -(xn[15] - 1e-2f * 0.5f * su2 + 0.05f);
        else if constexpr (ENV == METRPO_ENV_HOPPER) {
            float pen = 0.0f;
            for (int j = 2; j < NS; ++j) pen += fmaxf(fabsf(xn[j]) - 100.0f, 0.0f);
            cost = -(xn[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0], 0.0f) - 10.0f * fmaxf(fabsf(xn[1]) - 0.2f, 0.0f) - pen);
        } else if constexpr (ENV == METRPO_ENV_SNAKE) cost = -(xn[7] - 1e-2f * 0.5f * su2);
        bool dn = false;
        if constexpr (ENV == METRPO_ENV_ANT) {
            bool fin = true;
            for (int j = 0; j < NS; ++j) fin = fin && isfinite(xn[j]);
            dn = !((xn[2] >= 0.2f) && (xn[2] <= 1.0f) && fin);
        }
        int ts_new = ts + 1;
        dn = dn || (ts_new >= r.H);
        if (active && q == 0) { r.rew[tb] = -cost; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts_new - 1; }
        int row = -1;
        if (dn) {
            const size_t rb = (size_t)(t_loc + 1) * B + bc;
            row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
            cur_model = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
            ts_new = 0;
        }
        ts = ts_new;
        for (int i = q; i < NS; i += 4) ST[c * NS + i] = !active ? 0.0f : (row >= 0 ? r.pool[(size_t)row * NS + i] : NX[c * NS + i]);
        wave_lds_sync();
        if (round == z.rounds_total - 1 && tau == z.steps - 1 && active) {
            if (r.last_obs != nullptr) for (int i = q; i < NS; i += 4) r.last_obs[(size_t)b * NS + i] = ST[c * NS + i];
            if (q == 0) { if (r.last_ts != nullptr) r.last_ts[b] = ts; if (r.last_model != nullptr) r.last_model[b] = cur_model; }
        }
    }
}

template <int ENV, int DH, int WS>
__global__ void __launch_bounds__(512) k_rollout_resident(ProblemDesc pd, RolloutK r, ResidentK z, const float* __restrict__ dyn,
                                                          const float* __restrict__ theta, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using C = Cfg<ENV, 64, 32>;
    if (r.stop != nullptr && *r.stop != 0) return;                   // the sampling loop already ended (metrpo_sampler_progress); uniform over the grid
    if ((int)blockIdx.x < z.U) resident_compute<C::NS, C::NIN, DH, WS>(pd, z, dyn, lds);
    else resident_post<ENV>(pd, r, z, dyn, theta, norm, lds);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------
template <int ENV, int DH, int WS> static size_t resident_lds_bytes() {
    using C = Cfg<ENV, 64, 32>;
    constexpr int NIN_KS = C::NIN_KS, KS4 = cdiv(NIN_KS, 4), J = DH / 16, MT = WS / 16;
    const size_t comp = (size_t)(J * KS4 * 256 + MT * J * 256 + C::OUT_CB * MT * 256 + DH + WS) * sizeof(float);
    const size_t post = (size_t)((C::NS_KS * 2 + 24) * 64 + 84 + 8 * (2 * 16 * C::NS + 2 * 16 * C::NA)) * sizeof(float);
    return std::max(comp, post);
}
typedef void (*resident_kernel_t)(ProblemDesc, RolloutK, ResidentK, const float*, const float*, const float*);
struct ResidentEntry { int env, ns, na, n_drop, dh, ws; resident_kernel_t fn; size_t lds; };
#define RES_ENTRY(ENV, DH, WS) {ENV, EnvDim<ENV>::NS, EnvDim<ENV>::NA, EnvDim<ENV>::NDROP, DH, WS, k_rollout_resident<ENV, DH, WS>, resident_lds_bytes<ENV, DH, WS>()}
static const ResidentEntry* resident_table(int* n) {
    static const ResidentEntry tab[] = {
        RES_ENTRY(METRPO_ENV_SWIMMER, 512, 16), RES_ENTRY(METRPO_ENV_SWIMMER, 512, 32),
        RES_ENTRY(METRPO_ENV_HOPPER, 512, 32), RES_ENTRY(METRPO_ENV_SNAKE, 512, 32), RES_ENTRY(METRPO_ENV_HALF_CHEETAH, 512, 32),
    };
    *n = (int)(sizeof(tab) / sizeof(tab[0]));
    return tab;
}

// METRPO_EUNSUPPORTED: this shape / call stays on the step-wise path (rollout_gemm.hip)
int launch_rollout_resident(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    if (c->rollout_variant == 1 || getenv("METRPO_NO_RESIDENT") != nullptr) return METRPO_EUNSUPPORTED;
    if (pd.dyn.n_layers != 3 || pd.dyn.dims[1] != pd.dyn.dims[2] || pd.dyn.act[0] != METRPO_ACT_RELU || pd.dyn.act[1] != METRPO_ACT_RELU ||
        pd.dyn.act[2] != METRPO_ACT_IDENTITY) return METRPO_EUNSUPPORTED;
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH || pd.pol.act[1] != METRPO_ACT_TANH)
        return METRPO_EUNSUPPORTED;
    if (!(a->sam_mode == METRPO_SAM_STEP_RAND || a->sam_mode == METRPO_SAM_EPS_RAND || a->sam_mode == METRPO_SAM_ONE_MODEL)) return METRPO_EUNSUPPORTED;
    if (a->B > 128) return METRPO_EUNSUPPORTED;
    const int B = a->B, K = pd.K, H = a->H, DH = pd.dyn.dims[1], NT = (B + 15) / 16;
    // rounds of a horizon-terminated rollout are independent given the counter-based draws (see launch_rollout_gemm): they run side by side
    int R = 1;
    if (H > 0 && a->T % H == 0 && a->T / H >= 2 && pd.env != METRPO_ENV_ANT && a->t0 == 0 && a->d_init_obs == nullptr && a->d_stop == nullptr &&
        getenv("METRPO_SEQ_ROUNDS") == nullptr) R = a->T / H;
    const int steps = a->T / R;
    int n = 0;
    const ResidentEntry* tab = resident_table(&n);
    // narrowest slice (most CUs, least work per step) whose grid -- all rounds side by side -- still fits the chip; if even the widest
    // slice does not fit, the largest divisor of R that does, the remaining rounds as further launches
    const ResidentEntry* pick = nullptr; int Rg = 0, PW = 0;
    auto fits = [&](int ws, int rg, int* pw_out) {
        for (int pw = 1; pw <= 8; pw *= 2)
            if (rg * K * (DH / ws) + (rg * NT + pw - 1) / pw <= c->n_sm) { *pw_out = pw; return true; }
        return false;
    };
    auto entry = [&](int ws) -> const ResidentEntry* {
        for (int i = 0; i < n; ++i)
            if (tab[i].env == pd.env && tab[i].ns == pd.ns && tab[i].na == pd.na && tab[i].n_drop == pd.n_drop && tab[i].dh == DH && tab[i].ws == ws) return &tab[i];
        return nullptr;
    };
    const char* ws_env = getenv("METRPO_RESIDENT_WS");                // test hook: pin the slice width (results are bit-identical only at equal widths)
    for (int ws = 16; ws <= 32 && !pick; ws *= 2) {
        if (ws_env != nullptr && atoi(ws_env) != ws) continue;
        const ResidentEntry* e = entry(ws);
        if (e && fits(ws, R, &PW)) { pick = e; Rg = R; }
    }
    if (!pick) {
        const ResidentEntry* e = entry(32);
        for (int rg = R - 1; e && rg >= 1 && !pick; --rg)
            if (R % rg == 0 && fits(32, rg, &PW)) { pick = e; Rg = rg; }
    }
    if (!pick) return METRPO_EUNSUPPORTED;
    const int NSL = DH / pick->ws, OUT_CB = (pd.ns + 15) / 16, NIN_KS = (pd.nin + 3) / 4;
    const size_t nX = (size_t)Rg * NT * 4 * NIN_KS * 16, nP = (size_t)Rg * NT * K * NSL * 16 * OUT_CB * 16;
    const size_t need = (nX + nP + 32) * sizeof(unsigned long long);
    if (need > c->res_cap) {
        if (c->d_res) HIP_TRY(c, hipFree(c->d_res));
        c->d_res = nullptr; c->res_cap = 0;
        HIP_TRY(c, hipExtMallocWithFlags(&c->d_res, need, hipDeviceMallocUncached));
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, need, st));
        c->res_cap = need; c->res_seq = 0;
    }
    if ((unsigned long long)c->res_seq + (unsigned long long)(R / Rg) * (steps + 1) >= 0xfffffff0ull) {   // stamps would wrap: start over on a clean region
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, c->res_cap, st));
        c->res_seq = 0;
    }
    if (pick->lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pick->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick->lds));
    RolloutK rk = make_rollout_k(a);
    for (int round0 = 0; round0 < R; round0 += Rg) {
        ResidentK z;
        z.R = Rg; z.round0 = round0; z.rounds_total = R; z.NT = NT; z.NSL = NSL; z.U = Rg * K * NSL; z.PW = PW; z.steps = steps;
        z.seq0 = c->res_seq; c->res_seq += (unsigned int)steps + 1u;
        z.abort_cell = (unsigned int*)c->d_res;
        z.X = (unsigned long long*)c->d_res + 32; z.P = z.X + nX;
        z.err = comm_err_cell(c) + 1;                               // scal[S_ROLLERR]
        const int grid = z.U + (Rg * NT + PW - 1) / PW;
        hipLaunchKernelGGL(pick->fn, dim3(grid), dim3(512), pick->lds, st, pd, rk, z, c->d_dyn, c->d_theta, c->d_norm);
    }
    HIP_TRY(c, hipGetLastError());
    c->last_rollout_kernel = 4;
    return METRPO_OK;
}
