#pragma once
// Cooperative-heads MFMA rollout (second generation of rollout_mfma.hip; same arithmetic, same reference
// sites: samplers/vectorized_sampler.py:45-116, env_helpers.py:597-635, training.py:218-269).
//
// Why: with one head per wave (rollout_mfma.hip) a K=5 workgroup has 5 waves on 4 SIMDs, one SIMD carries two
// heads (2 x 92 MFMAs per step) and, at ~230 VGPRs, only ONE such workgroup fits a CU -- the 313 tiles of the
// headline shape (B = 5000) then run in two sequential rounds on 256 CUs (measured: B=4096 0.64 ms, B=5000 1.23 ms).
// Here a workgroup is exactly 4 waves (one per SIMD) for a 16-env tile and EVERY head is split across them:
//     wave w owns hidden col-block w (16 of the 64 units) of layers 0 and 1 of all K heads, and the K-slice w of
//     layer 2 (its own 16 hidden units), i.e. K x (3 + 16 + 4) = 115 MFMAs per step instead of 92 / 184,
// balanced over the 4 SIMDs, and two workgroups fit a CU (8 waves, 2 per SIMD).  Costs: the layer-0 activations and the layer-2
// partial sums cross waves through LDS -> three barriers per step (action, layer-0 activations, layer-2 partials).
// Round 2: the policy (32 MFMAs, 16 tanh per lane) runs on wave 0 only; waves 1-3 meanwhile run the state-only k-steps of layer 0
// (their own col-block and wave 0's) and wave 1 the next step's Philox draws; launches with up to 1.55 tiles per CU run ONE workgroup
// per CU with the tile-steps dealt out evenly and tiles migrating between workgroups (see the kernel and launch_rollout_coop);
// this file is compiled with -amdgpu-mfma-vgpr-form (Makefile) so that MFMA results stay in the vector half of the register file.
#include "mfma_common.h"

// Developer instrumentation (tools/build_variant.sh timing -DCOOP_TIMING=0xFFF): per-phase shader-clock sums of the four waves of
// workgroup 0, read back with metrpo_debug_coop_phases (tools/coop_phases.py).  Not part of the shipped library.
#if defined(COOP_TIMING) && defined(COOP_MAIN_TU)
// s_memtime (the SHADER_CYCLES hardware register reads 0 on gfx950); all four waves of workgroup 0.  Every mark also drains the wave's
// LDS queue (s_memtime returns through lgkmcnt), so phases that overlap LDS latency with later work look longer than they are.
__device__ unsigned long long g_coop_phase[4][16];
#define PH_NOW() __builtin_readcyclecounter()
#define PH_DECL unsigned long long ph_t = PH_NOW(); unsigned long long ph_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(i) { if ((COOP_TIMING >> (i)) & 1) { const unsigned long long n_ = PH_NOW(); ph_acc[i] += n_ - ph_t; ph_t = n_; } }   // -DCOOP_TIMING=<bit mask of live marks>
#define PH_DUMP { if (lane == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < 14; ++i_) g_coop_phase[wave][i_] = ph_acc[i_]; }
extern "C" int32_t metrpo_debug_coop_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_phase), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1; }
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_DUMP
#endif

#ifndef COOP_SKIP
#define COOP_SKIP 0        // developer experiments (tools/build_variant.sh): bit mask of step parts to leave out; results are then meaningless
#endif
template <int ENV, int K>
struct Coop {
    using C = Cfg<ENV, 64, 32>;
    static constexpr int NS = C::NS, NA = C::NA, NSP = C::NSP;
    static constexpr int P0KS = 4 * C::OUT_CB;                        // policy layer-0 k-steps (cb_in, rr): input dim 16 cb_in + 4 q + rr, the D layout of the state
    static constexpr int NPF = P0KS * 2 + 16 + 8;                     // policy weight fragments: wp0[P0KS][2], wp1[8][2], wp2[8]
    // LDS map (floats)
    static constexpr int WV = ((16 * NSP + 16 * NS + 16 * NA + 3) / 4) * 4;   // per wave: ST (rows padded to NSP: one unpredicated 16-byte access per lane) | NX | ACT
    static constexpr int O_BD0 = 4 * WV, O_BD1 = O_BD0 + K * 64, O_BD2 = O_BD1 + K * 64, O_BP0 = O_BD2 + K * NSP,
                         O_BP1 = O_BP0 + 32, O_BP2 = O_BP1 + 32, O_PW = O_BP2 + 16, O_H0 = O_PW + NPF * 64,
                         O_PART = O_H0 + K * 1024, O_RNG = O_PART + K * 4 * 16 * NSP, O_WX = O_RNG + 2 * 16 * 20,
                         O_AK = O_WX + 3 * ((K + 2) / 3) * C::NIN_KS * 64, TOTAL = O_AK + 2 * 16 * NA;   // O_AK: unclipped action | mean of the step          // O_WX: col-block-0 layer-0 fragments of the helper waves (two-tile kernel)
};

// ONE: instantiation for launches with at most one tile per CU -- the whole register file (512 lanes-wide registers per SIMD) belongs to
// one workgroup, so the half-cheetah / Ant weight fragments (150 VGPRs) stop spilling to scratch.
// DRAWS: instantiation for launches that SUPPLY draws (eps / model_idx / sel_noise / reset_idx / reset_model: the parity tests).  The
// production instantiation (Philox draws) has no global load in the step loop outside the rare reset branch, hence no s_waitcnt
// vmcnt in it: a vmcnt wait also waits for the step's stores, whose acknowledge latency then lands on the critical path.
template <int ENV, int K, bool ONE, bool DRAWS>
__global__ void __launch_bounds__(256, ONE ? 1 : 2) k_rollout_coop(RolloutK r, const float* __restrict__ dynp,
                                                      const float* __restrict__ theta, const float* __restrict__ norm) {
    using L = Coop<ENV, K>;
    using C = typename L::C;
    constexpr int NS = C::NS, NA = C::NA, NSP = C::NSP, DH = 64, PH = 32, OUT_CB = C::OUT_CB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // wave = hidden col-block
    const int e = lane & 15, q = lane >> 4;
    if (r.stop != nullptr && *r.stop != 0) return;           // the sampling loop already ended (metrpo_sampler_progress)
    // ---------------- which (tile, step range) pieces this workgroup runs ------------------------------------------------------
    // !ONE: workgroup = tile, all T steps.  ONE: the launch has at most one workgroup per CU and the tiles x T tile-steps are dealt
    // out evenly in tile-major order (wrap-around rule): workgroup i owns the linear range [i Q, (i+1) Q), i.e. the END of tile
    // c_first (steps s_first..T-1), whole tiles, and the BEGINNING of tile c_last (steps 0..e_last-1).  It runs the beginning piece
    // first, then the whole tiles, then the ending piece, whose state it takes over from workgroup i-1 through a hand-over slot in
    // HBM (flag = launch epoch, agent-scope release / acquire).  Workgroup i-1 runs that tile's beginning before anything else and
    // waits for nobody first, so with in-order dispatch the wait cannot deadlock; Q >= T keeps a tile on at most two workgroups.
    int c_first = blockIdx.x, c_last = blockIdx.x, s_first = 0, e_last = r.T;
    if (ONE) {
        const long long total = (long long)((r.B + 15) / 16) * r.T;
        const long long Q = (total + gridDim.x - 1) / gridDim.x;
        const long long lo = (long long)blockIdx.x * Q, hi = (lo + Q < total) ? lo + Q : total;
        if (lo >= hi) return;
        c_first = (int)(lo / r.T); s_first = (int)(lo % r.T); c_last = (int)((hi - 1) / r.T); e_last = (int)((hi - 1) % r.T) + 1;
    }
    const int npc = c_last - c_first + 1;
    int b0 = 0, b = 0;
    bool active = false;
    uint64_t genv = 0;
    float* ST = lds + wave * L::WV;  float* NX = ST + 16 * NSP;
    float* ACT = lds + 16 * NSP + 16 * NS;                                         // clipped actions of the tile: written by wave 0 (the policy wave), read by all

    float* BD0 = lds + L::O_BD0; float* BD1 = lds + L::O_BD1; float* BD2 = lds + L::O_BD2;
    float* BP0 = lds + L::O_BP0; float* BP1 = lds + L::O_BP1; float* BP2 = lds + L::O_BP2;
    float* PW = lds + L::O_PW; float* H0 = lds + L::O_H0; float* PART = lds + L::O_PART;
    float* AK = lds + L::O_AK;                                          // [action | mean][16 envs][na], staged for wave 0's coalesced stores
    float* RNGB = lds + L::O_RNG;                                       // [2 parities][16 envs][dstep (4 x u32) | z of q-lane 0..3 (4 floats each)]

    // ---------------- one-time: dynamics fragments -> registers ----------------------------------
    float wd0[K][C::NIN_KS], wd1[K][16], wd2[K][4][OUT_CB];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float* __restrict__ pk = dynp + (size_t)k * C::PD;
#pragma unroll
        for (int s = 0; s < C::NIN_KS; ++s) { const int i = 4 * s + q; wd0[k][s] = (i < C::NIN) ? pk[C::dW0 + i * DH + 16 * wave + e] : 0.0f; }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { const int i = 16 * (kk >> 2) + 4 * q + (kk & 3); wd1[k][kk] = pk[C::dW1 + i * DH + 16 * wave + e]; }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) { const int i = 16 * wave + 4 * q + rr, o = 16 * cb + e; wd2[k][rr][cb] = (o < NS) ? pk[C::dW2 + i * NS + o] : 0.0f; }
    }
    // Layer 0 of col-block 0 is NOT computed by wave 0 (the policy wave, whose step is the longest): waves 1, 2, 3 take it for heads
    // {0,1}, {2,3}, {4,..} on top of their own col-block -- they idle while the policy is evaluated anyway.
    constexpr int XH = (K + 2) / 3;                                    // heads of col-block 0 per helper wave
    float wx0[ONE ? XH : 1][C::NIN_KS];                                // registers when the workgroup owns the CU, else an LDS image (no room)
    float* WXL = lds + L::O_WX + (wave >= 1 ? wave - 1 : 0) * XH * C::NIN_KS * 64 + lane;
#pragma unroll
    for (int j = 0; j < XH; ++j) {
        const int hx = XH * (wave - 1) + j;
        const float* __restrict__ pk = dynp + (size_t)((wave >= 1 && hx < K) ? hx : 0) * C::PD;
#pragma unroll
        for (int s = 0; s < C::NIN_KS; ++s) {
            const int i = 4 * s + q;
            const float w = (i < C::NIN && wave >= 1 && hx < K) ? pk[C::dW0 + i * DH + e] : 0.0f;
            if (ONE) wx0[j][s] = w; else if (wave >= 1) WXL[(j * C::NIN_KS + s) * 64] = w;
        }
    }
    // shared images: biases and policy fragments (each element written by exactly one thread)
    for (int i = tid; i < K * 64; i += 256) { const int k = i >> 6, u = i & 63; BD0[i] = dynp[(size_t)k * C::PD + C::db0 + u]; BD1[i] = dynp[(size_t)k * C::PD + C::db1 + u]; }
    for (int i = tid; i < K * NSP; i += 256) { const int k = i / NSP, u = i % NSP; BD2[i] = (u < NS) ? dynp[(size_t)k * C::PD + C::db2 + u] : 0.0f; }
    if (tid < 32) { BP0[tid] = theta[C::pb0 + tid]; BP1[tid] = theta[C::pb1 + tid]; }
    if (tid < 16) BP2[tid] = (tid < NA) ? theta[C::pb2 + tid] : 0.0f;
    for (int i = tid; i < L::NPF * 64; i += 256) {
        const int f = i >> 6, ln = i & 63, ee = ln & 15, qq = ln >> 4;
        float w = 0.0f;
        if (f < L::P0KS * 2) { const int j = f >> 1, cb = f & 1, in = 16 * (j >> 2) + 4 * qq + (j & 3); w = (in < NS) ? theta[C::pW0 + in * PH + 16 * cb + ee] : 0.0f; }
        else if (f < L::P0KS * 2 + 16) { const int g = f - L::P0KS * 2, kk = g >> 1, cb = g & 1, in = 16 * (kk >> 2) + 4 * qq + (kk & 3); w = theta[C::pW1 + in * PH + 16 * cb + ee]; }
        else { const int kk = f - L::P0KS * 2 - 16, in = 16 * (kk >> 2) + 4 * qq + (kk & 3); w = (ee < NA) ? theta[C::pW2 + in * NA + ee] : 0.0f; }
        PW[i] = w;
    }
    const float* pw0 = PW + lane, *pw1 = PW + L::P0KS * 2 * 64 + lane, *pw2 = PW + (L::P0KS * 2 + 16) * 64 + lane;
    float sig[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sig[rr] = (4 * q + rr < NA) ? expf(fmaxf(theta[C::pLS + 4 * q + rr], LOG_MIN_STD)) : 0.0f;
    float nmean[C::NIN_KS], nstd[C::NIN_KS];
    int nsrc[C::NIN_KS];
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
    f32x4 dmean[OUT_CB], dstd[OUT_CB];
#pragma unroll
    for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int dim = 16 * cb + 4 * q + rr;
            dmean[cb][rr] = (dim < NS) ? norm[2 * (NS + NA) + dim] : 0.0f;
            dstd[cb][rr] = (dim < NS) ? norm[2 * (NS + NA) + NS + dim] : 0.0f;
        }
    // Per-step draws (Philox block 0 of RNG_STEP: action noise dims 0,1 | step model | reset row/model) are produced ONE STEP AHEAD.
    auto step_draws = [&](int tt, uint4& ds, float (&zz)[4]) {
        ds = rng_draw(r.seed, genv, tt, RNG_STEP, 0);
        {   // unconditional (no branch); unused when determ / draws are supplied.
            // lane q owns action dims 4q..4q+3 = chunks 2q, 2q+1 (chunk 0 = ds)
            const uint4 b0k = (q == 0) ? ds : ((NA > 4) ? rng_draw(r.seed, genv, tt, RNG_STEP, 2 * q) : ds);
            normal2(b0k.x, b0k.y, zz[0], zz[1]);
            if (NA > 2) { const uint4 b1k = rng_draw(r.seed, genv, tt, RNG_STEP, 2 * q + 1); normal2(b1k.x, b1k.y, zz[2], zz[3]); }
        }
    };
    constexpr bool LOCAL_REWARD = (ENV == METRPO_ENV_SWIMMER || ENV == METRPO_ENV_HALF_CHEETAH || ENV == METRPO_ENV_SNAKE);
    constexpr int RDIM = (ENV == METRPO_ENV_SWIMMER) ? 5 : (ENV == METRPO_ENV_HALF_CHEETAH) ? 9 : 7;   // reward reads next_state[RDIM]
    constexpr bool AHEAD = true;        // the draws of step t+1 are produced during step t (by wave 1, below) for every env family
    // One workgroup per CU and a small state (one col-block): the policy weight fragments and biases of this lane stay in registers for
    // the whole launch -- the policy chain is the step's critical path before B0 and otherwise starts with an LDS round trip.
    constexpr bool PWREG = ONE && OUT_CB == 1;
    float w0r[PWREG ? L::P0KS * 2 : 1], w1r[PWREG ? 16 : 1], w2r[PWREG ? 8 : 1];
    f32x4 bp0r[2], bp1r[2], bp2r;
    if (PWREG) {
        __syncthreads();                                                   // PW / BP images complete
#pragma unroll
        for (int f = 0; f < L::P0KS * 2; ++f) w0r[f] = pw0[f * 64];
#pragma unroll
        for (int f = 0; f < 16; ++f) w1r[f] = pw1[f * 64];
#pragma unroll
        for (int f = 0; f < 8; ++f) w2r[f] = pw2[f * 64];
        bp0r[0] = *(const f32x4*)&BP0[4 * q]; bp0r[1] = *(const f32x4*)&BP0[16 + 4 * q];
        bp1r[0] = *(const f32x4*)&BP1[4 * q]; bp1r[1] = *(const f32x4*)&BP1[16 + 4 * q];
        bp2r = *(const f32x4*)&BP2[4 * q];
    }
    PH_DECL
    for (int pc = 0; pc < npc; ++pc) {
    const int tile = (pc == npc - 1) ? c_first : (pc == 0 ? c_last : c_first + pc);
    const int t_begin = (tile == c_first) ? s_first : 0, t_end = (tile == c_last) ? e_last : r.T;
    b0 = tile * 16; b = b0 + e; active = b < r.B; genv = r.stream_offset + (uint64_t)b;
    // ---------------- vec_env.reset() (env_helpers.py:585-595), or the state handed over by the previous owner -------------
    const bool resume = r.init_obs != nullptr;
    int cur_model = 0, ts = 0;
    if (ONE && t_begin > 0) {
        // The producer is the previous workgroup, which runs this very piece FIRST and waits for nobody; workgroups are dispatched in
        // ascending order per XCD, so the smallest unfinished workgroup is always resident and the chain cannot deadlock.  Should the
        // platform ever break that assumption (CU masks, partition modes), the wait gives up after ~2 s of wall clock and raises the
        // ctx's sticky error cell (reported by the next metrpo_trpo_update / metrpo_comm_check) instead of hanging the GPU.
        if (tid == 0) {
            unsigned long long t0 = 0; int spins = 0;
            while (__hip_atomic_load(&r.mig_flag[tile], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != r.mig_epoch) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins == 4096) t0 = wall_clock64();
                if (spins > 4096 && (spins & 1023) == 0 && wall_clock64() - t0 > 200000000ull) { if (r.mig_err) *r.mig_err = 1.0; break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        cur_model = r.mig_model[b]; ts = r.mig_ts[b];
#pragma unroll
        for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                ST[e * NSP + dim] = (dim < NS) ? r.mig_obs[(size_t)b * NS + dim] : 0.0f;
            }
    } else {
        int row = 0;
        if (active && !resume) {
            const uint4 d0 = rng_draw(r.seed, genv, 0, RNG_RESET, 0);
            row = (DRAWS && r.reset_idx != nullptr) ? r.reset_idx[b] : rng_index(d0.x, r.n_pool);
            cur_model = (DRAWS && r.reset_model != nullptr) ? r.reset_model[b] : rng_index(d0.y, K);
        }
        if (active && resume) { cur_model = r.init_model[b]; ts = r.init_ts[b]; }      // continuation of a chunked rollout
#pragma unroll
        for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int dim = 16 * cb + 4 * q + rr;
                ST[e * NSP + dim] = (dim < NS) ? (resume ? r.init_obs[(size_t)(active ? b : 0) * NS + dim] : r.pool[(size_t)row * NS + dim]) : 0.0f;
            }
    }
    __syncthreads();

    f32x4 xc[OUT_CB];                                                  // the tile's current state in registers: lane (e, q) holds dims 16 cb + 4 q + r (0 beyond ns)
#pragma unroll
    for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { const int dim = 16 * cb + 4 * q + rr; xc[cb][rr] = ST[e * NSP + dim]; }
    uint4 dstep = make_uint4(0, 0, 0, 0); float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (AHEAD) step_draws(r.t0 + t_begin, dstep, z);
    // vmcnt(0) HERE: every global load above (weight fragments, normaliser, initial state) is complete before the step loop.  Without
    // it the compiler places the wait for those loop-invariant loads at their first use INSIDE the loop, where it is executed every
    // step and then also waits for the stores of the previous step (HBM acknowledge latency on the step's critical path).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int t = t_begin; t < t_end; ++t) {
        PH_MARK(9)
        const size_t trow = (size_t)t * r.B;                               // uniform row base; lane offsets stay 32-bit (scalar base + offset addressing)
        const unsigned ub = (unsigned)b;
        // ---- policy (redundant in every wave; weight fragments from LDS) -----------------------------
        // Weight fragments of a layer are fetched as one batch BEFORE the activation of the previous layer is evaluated (the LDS
        // latency hides under the tanh VALU work), and each tanh is issued right before the MFMA pair that consumes it, so the
        // matrix pipe works through k-step kk while the VALU evaluates the activation of k-step kk+1.
        // The policy (30 MFMAs, 16 tanh per lane) is evaluated by wave 0 ONLY and its clipped action handed to the other waves through
        // LDS (one more barrier per step).  Evaluating it redundantly in all 4 waves kept the step free of that barrier but cost
        // 90 of the 580 MFMAs and three quarters of the tanh VALU work of a tile-step -- issue slots a co-resident workgroup can use.
        if (!AHEAD) step_draws(r.t0 + t, dstep, z);                        // every wave needs the step's model / reset draws
        // dynamics layer 0, k-steps that read only the STATE (inputs 4s..4s+3 all below ns - n_drop): independent of the action, so
        // waves 1-3 run them while wave 0 evaluates the policy
        constexpr int KS_STATE = (NS - C::NDROP) / 4;
        f32x4 h0[K], hx0[XH];
        if (wave != 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) h0[k] = *(const f32x4*)&BD0[k * 64 + 16 * wave + 4 * q];
#pragma unroll
            for (int j = 0; j < XH; ++j) { const int hx = XH * (wave - 1) + j; hx0[j] = *(const f32x4*)&BD0[(hx < K ? hx : 0) * 64 + 4 * q]; }
#pragma unroll
            for (int s = 0; s < KS_STATE; ++s) {
                const float x = (ST[e * NSP + nsrc[s]] - nmean[s]) * nstd[s];             // training.py:228
#pragma unroll
                for (int k = 0; k < K; ++k) { if (COOP_SKIP & 32) h0[k][0] += x; else h0[k] = MFMA16(wd0[k][s], x, h0[k]); }
#pragma unroll
                for (int j = 0; j < XH; ++j) { if (COOP_SKIP & 32) hx0[j][0] += x; else hx0[j] = MFMA16(ONE ? wx0[ONE ? j : 0][s] : WXL[(j * C::NIN_KS + s) * 64], x, hx0[j]); }
            }
        }
        if (AHEAD && wave == 1 && !(COOP_SKIP & 16)) {
            // Next step's Philox block + Box-Muller for the 16 envs of the tile, produced by wave 1 while wave 0 evaluates the policy
            // (waves 1-3 have nothing else to do until the action exists) and handed over through LDS, double-buffered by step
            // parity: every wave picks its copy up after barrier B2.  (It used to be dealt out in pieces between the layer-1 MFMAs
            // of every wave: 16x redundant, and ~1000 cycles of VALU inside the phase two co-resident workgroups fight over.)
            uint4 dn; float zn[4] = {0.f, 0.f, 0.f, 0.f};
            step_draws(r.t0 + t + 1, dn, zn);
            float* dst = RNGB + (((t + 1) & 1) * 16 + e) * 20;
            if (q == 0) *(uint4*)dst = dn;
            *(float4*)(dst + 4 + 4 * q) = make_float4(zn[0], zn[1], zn[2], zn[3]);          // lane q owns action dims 4q .. 4q+3
        }
        if (wave == 0) {
            f32x4 p0[2], p1[2];
            if (COOP_SKIP & 1) { p0[0] = p0[1] = p1[0] = p1[1] = xc[0]; }
            f32x4 m0, m1 = {0.f, 0.f, 0.f, 0.f};
            if (!(COOP_SKIP & 1)) {
            {
                // layer 0 straight from the state registers xc (D layout of the previous step's output): k-step (cb_in, rr) contracts input
                // dims 16 cb_in + 4 q + rr, the weight fragments are stored in that order -- no LDS round trip between the end of a step
                // and the first policy MFMA of the next
                float w0[L::P0KS * 2];
#pragma unroll
                for (int f = 0; f < L::P0KS * 2; ++f) w0[f] = PWREG ? w0r[PWREG ? f : 0] : pw0[f * 64];
                if (PWREG) { p0[0] = bp0r[0]; p0[1] = bp0r[1]; } else { p0[0] = *(const f32x4*)&BP0[4 * q]; p0[1] = *(const f32x4*)&BP0[16 + 4 * q]; }
#pragma unroll
                for (int j = 0; j < L::P0KS; ++j) {
                    p0[0] = MFMA16(w0[2 * j], xc[j >> 2][j & 3], p0[0]);
                    p0[1] = MFMA16(w0[2 * j + 1], xc[j >> 2][j & 3], p0[1]);
                }
            }
            {
                // Straight-line code plus an explicit issue pipeline (sched_group_barrier): each tanh (v_mul, v_exp, v_add, v_rcp, v_fma:
                // ~45 cycles of VALU / transcendental issue) is placed behind the MFMA pair (layer 1) or MFMA (layer 2) of the PREVIOUS
                // k-step, so that it runs while the matrix pipe works.  Left to itself the compiler evaluates all eight tanh of a layer
                // first and then lets every MFMA wait for the final fma of its operand.
                float w1[16], w2[8];
#pragma unroll
                for (int f = 0; f < 16; ++f) w1[f] = PWREG ? w1r[PWREG ? f : 0] : pw1[f * 64];
#pragma unroll
                for (int f = 0; f < 8; ++f) w2[f] = PWREG ? w2r[PWREG ? f : 0] : pw2[f * 64];
                if (PWREG) { p1[0] = bp1r[0]; p1[1] = bp1r[1]; m0 = bp2r; }
                else { p1[0] = *(const f32x4*)&BP1[4 * q]; p1[1] = *(const f32x4*)&BP1[16 + 4 * q]; m0 = *(const f32x4*)&BP2[4 * q]; }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const float hv = (COOP_SKIP & 64) ? p0[kk >> 2][kk & 3] : tanh_fast(p0[kk >> 2][kk & 3]);
                    p1[0] = MFMA16(w1[2 * kk], hv, p1[0]);
                    p1[1] = MFMA16(w1[2 * kk + 1], hv, p1[1]);
                }
#pragma unroll
                for (int kk = 0; kk < 8; kk += 2) {
                    const float ha = (COOP_SKIP & 64) ? p1[kk >> 2][kk & 3] : tanh_fast(p1[kk >> 2][kk & 3]);
                    m0 = MFMA16(w2[kk], ha, m0);
                    const float hb = (COOP_SKIP & 64) ? p1[(kk + 1) >> 2][(kk + 1) & 3] : tanh_fast(p1[(kk + 1) >> 2][(kk + 1) & 3]);
                    m1 = MFMA16(w2[kk + 1], hb, m1);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 64, 0);                // every LDS read of the block (inputs, weight fragments, biases) up front
                __builtin_amdgcn_sched_group_barrier(0x008, L::P0KS * 2, 0);       // layer 0
                __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);                 // tanh of k-step 0
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {                                    // layer 1, k-step kk: the wave cannot issue past an MFMA the pipe
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // has no room for, so the tanh of k-step kk+1 (the last one: layer 2's
                    __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);             // first) is split around the pair's second MFMA instead of queuing
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // behind it
                    __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // layer 2, k-step kk
                    if (kk < 7) __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
                }
            }
            } else m0 = xc[0];
            const f32x4 mu = m0 + m1;
            PH_MARK(0)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int d = 4 * q + rr;
                if (d < NA) {
                    float a = mu[rr];
                    if (!r.determ) {
                        const float zz = (DRAWS && r.eps != nullptr) ? (active ? (r.eps + trow * NA)[ub * NA + d] : 0.0f) : z[rr];
                        a = fmaf(zz, sig[rr], a);
                    }
                    AK[e * NA + d] = a; AK[16 * NA + e * NA + d] = mu[rr];
                    const float ac = fminf(fmaxf(a, -1.0f), 1.0f);            // env_helpers.py:599
                    ACT[e * NA + d] = ac;
                }
            }
        }
        __syncthreads();                                                   // B0: actions visible
        PH_MARK(1)
        // ---- dynamics layer 0: waves 1-3 finish their col-block and their share of col-block 0; wave 0 issues the step's global
        //      stores meanwhile (it has no layer-0 work, and nothing waits for a store: barriers wait for LDS traffic only) -------
        if (wave != 0) {
            float xin[C::NIN_KS];
#pragma unroll
            for (int s = KS_STATE; s < C::NIN_KS; ++s) {
                float x = 0.0f;
                if (nsrc[s] >= 0) x = ST[e * NSP + nsrc[s]];
                else if (nsrc[s] > -1000000) x = ACT[e * NA + (-nsrc[s] - 1)];
                xin[s] = (nsrc[s] > -1000000) ? (x - nmean[s]) * nstd[s] : 0.0f;          // training.py:228
            }
#pragma unroll
            for (int s = KS_STATE; s < C::NIN_KS; ++s) {
#pragma unroll
                for (int k = 0; k < K; ++k) { if (COOP_SKIP & 8) h0[k][0] += xin[s]; else h0[k] = MFMA16(wd0[k][s], xin[s], h0[k]); }
#pragma unroll
                for (int j = 0; j < XH; ++j) { if (COOP_SKIP & 8) hx0[j][0] += xin[s]; else hx0[j] = MFMA16(ONE ? wx0[ONE ? j : 0][s] : WXL[(j * C::NIN_KS + s) * 64], xin[s], hx0[j]); }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) h0[k][rr] = relu1(h0[k][rr]);
                *(f32x4*)&H0[k * 1024 + wave * 256 + q * 64 + e * 4] = h0[k];            // H0[k][cb][q][env][r]
            }
#pragma unroll
            for (int j = 0; j < XH; ++j) {
                const int hx = XH * (wave - 1) + j;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) hx0[j][rr] = relu1(hx0[j][rr]);
                if (hx < K) *(f32x4*)&H0[hx * 1024 + q * 64 + e * 4] = hx0[j];
            }
        } else if (!(COOP_SKIP & 256)) {
            {                                                              // unclipped action and mean: coalesced linear copies of the tile
                const size_t base_a = (trow + b0) * NA;
                const int lim_a = min(16, r.B - b0) * NA;
#pragma unroll
                for (int j = 0; j < (16 * NA + 63) / 64; ++j) {
                    const int i = lane + 64 * j;
                    if (i < lim_a) { (r.act + base_a)[(unsigned)i] = AK[i]; (r.mean + base_a)[(unsigned)i] = AK[16 * NA + i]; }
                }
            }
            const size_t base = (trow + b0) * NS;               // obs[t]: coalesced linear copy of the tile
            const int lim = min(16, r.B - b0) * NS;
            float ov[(16 * NS + 63) / 64];                                 // all LDS reads first: one register per store, no store waits for another
#pragma unroll
            for (int j = 0; j < (16 * NS + 63) / 64; ++j) { const int i = min(lane + 64 * j, 16 * NS - 1); ov[j] = ST[(i / NS) * NSP + i % NS]; }
#pragma unroll
            for (int j = 0; j < (16 * NS + 63) / 64; ++j) { const int i = lane + 64 * j; if (i < lim) (r.obs + base)[(unsigned)i] = ov[j]; }
        }
        PH_MARK(2)
        __syncthreads();                                                   // B1: layer-0 activations of all heads visible
        PH_MARK(3)
        float su2 = 0.0f;                                                  // sum_d clip(a_d)^2 of the own env (ACT is complete since B1)
#pragma unroll
        for (int d = 0; d < NA; ++d) { const float ac = ACT[e * NA + d]; su2 = fmaf(ac, ac, su2); }
        // ---- layer 1 (own col-block) and the layer-2 partial over the own 16 hidden units ----------------
        PH_MARK(10)
        __builtin_amdgcn_s_setprio(1);          // the matrix-heavy phase wins issue arbitration over a co-resident workgroup's VALU phases
        f32x4 h1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) h1[k] = *(const f32x4*)&BD1[k * 64 + 16 * wave + 4 * q];
        constexpr int NHB = ONE ? 2 : 1;                                   // ONE: layer-0 activations fetched one col-block ahead (registers to spare)
        f32x4 hb[NHB][K];
        if (ONE) {
#pragma unroll
            for (int k = 0; k < K; ++k) hb[0][k] = *(const f32x4*)&H0[k * 1024 + q * 64 + e * 4];
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (ONE) {
                if (cb < 3) {
#pragma unroll
                    for (int k = 0; k < K; ++k) hb[(cb + 1) % NHB][k] = *(const f32x4*)&H0[k * 1024 + (cb + 1) * 256 + q * 64 + e * 4];
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) hb[0][k] = *(const f32x4*)&H0[k * 1024 + cb * 256 + q * 64 + e * 4];
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                for (int k = 0; k < K; ++k) { if (COOP_SKIP & 4) h1[k] += hb[cb % NHB][k]; else h1[k] = MFMA16(wd1[k][cb * 4 + rr], hb[cb % NHB][k][rr], h1[k]); }
                __builtin_amdgcn_sched_barrier(0x94);                      // only SALU / VMEM / LDS instructions may cross
                if (ONE) pin_order(h1);                                    // and the K chains advance in lock-step (mfma_common.h)
            }
        }
        PH_MARK(11)
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) h1[k][rr] = relu1(h1[k][rr]);
        __builtin_amdgcn_sched_barrier(0);      // all ReLUs first: a v_max feeding the very next MFMA's operand stalls it (0.386 -> 0.373 ms at B = 4096)
        {
            f32x4 po[K][OUT_CB];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb) po[k][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
                    for (int k = 0; k < K; ++k) { if (COOP_SKIP & 2) po[k][cb] += h1[k]; else po[k][cb] = MFMA16(wd2[k][rr][cb], h1[k][rr], po[k][cb]); }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int cb = 0; cb < OUT_CB; ++cb) *(f32x4*)&PART[((k * 4 + wave) * 16 + e) * NSP + 16 * cb + 4 * q] = po[k][cb];
        }
        __builtin_amdgcn_s_setprio(0);
        PH_MARK(4)
        __syncthreads();                                                   // B2: all partial sums visible
        PH_MARK(5)
        // ---- selection (env_helpers.py:617-634): out_k = b2_k + sum_w partial ; next = dmean + dstd*out + s ----
        ts += 1;
        int sel = cur_model;
        if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (DRAWS && r.model_idx != nullptr) ? (active ? (r.model_idx + trow)[ub] : 0) : rng_index(dstep.z, K);
        if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
        const bool simple = (r.sam_mode == METRPO_SAM_STEP_RAND || r.sam_mode == METRPO_SAM_EPS_RAND || r.sam_mode == METRPO_SAM_ONE_MODEL);
        f32x4 nx[OUT_CB];
#pragma unroll
        for (int cb = 0; cb < OUT_CB; ++cb) {
            const int off = e * NSP + 16 * cb + 4 * q;
            const f32x4 sv = xc[cb];
            auto head = [&](int k) -> f32x4 {
                if (COOP_SKIP & 128) return sv * 0.999f;
                const float* pp = PART + (size_t)k * 4 * 16 * NSP + off;
                f32x4 o = *(const f32x4*)&BD2[k * NSP + 16 * cb + 4 * q];
                o += (*(const f32x4*)&pp[0] + *(const f32x4*)&pp[16 * NSP]) + (*(const f32x4*)&pp[2 * 16 * NSP] + *(const f32x4*)&pp[3 * 16 * NSP]);
                return dstd[cb] * o + dmean[cb] + sv;                     // training.py:257
            };
            if (simple) nx[cb] = head(sel);
            else {
                f32x4 hv[K], m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < K; ++k) { hv[k] = head(k); m += hv[k]; }
                m /= (float)K;
                if (r.sam_mode == METRPO_SAM_MODEL_MEAN) nx[cb] = m;
                else if (r.sam_mode == METRPO_SAM_MODEL_MEAN_STD) {
                    f32x4 var = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < K; ++k) { const f32x4 d = hv[k] - m; var += d * d; }
                    float zz[4] = {0.f, 0.f, 0.f, 0.f};
                    if (!DRAWS || r.sel_noise == nullptr) normal4(rng_draw(r.seed, genv, r.t0 + t, RNG_SELNOISE, 4 * cb + q), zz);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int dim = 16 * cb + 4 * q + rr;
                        const float nz = (DRAWS && r.sel_noise != nullptr) ? ((active && dim < NS) ? (r.sel_noise + trow * NS)[ub * NS + dim] : 0.0f) : zz[rr];
                        nx[cb][rr] = fmaf(nz, sqrtf(var[rr] / (float)K), m[rr]);
                    }
                } else {                                                  // model_med: np.median over K
                    constexpr int r_lo = (K - 1) / 2, r_hi = K / 2;
                    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        int rank[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int j = 0; j < K; ++j)
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) rank[rr] += (hv[j][rr] < hv[k][rr]) || (hv[j][rr] == hv[k][rr] && j < k);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) { if (rank[rr] == r_lo) lo[rr] = hv[k][rr]; if (rank[rr] == r_hi) hi[rr] = hv[k][rr]; }
                    }
                    nx[cb] = (lo + hi) * 0.5f;
                }
            }
        }
        PH_MARK(6)
        // ---- reward (:601), is_done (:603), horizon (:604) ----------------------------------------------
        float cost = 0.0f;
        bool dn = false;
        if (LOCAL_REWARD) {
            // the reward reads ONE next-state dim: the lane that owns it (q == RDIM/4 of col-block 0) evaluates and stores it from
            // registers -- no LDS round trip; is_done is the horizon only (uniform)
            const float xr = nx[RDIM / 16][RDIM & 3];
            if (ENV == METRPO_ENV_SWIMMER) cost = -(xr - 1e-2f * (su2 / (float)NA));
            else if (ENV == METRPO_ENV_HALF_CHEETAH) cost = -fminf(fmaxf(xr - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
            else cost = -(xr - 1e-2f * 0.5f * su2);
            dn = (ts >= r.H);
            if (wave == 2 && q == ((RDIM & 15) >> 2) && active) { (r.rew + trow)[ub] = -cost; (r.done + trow)[ub] = dn ? 1 : 0; (r.tpath + trow)[ub] = ts - 1; }
        } else {
            float pen = 0.0f; int fin = 1;
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb)
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
            if (ENV == METRPO_ENV_HOPPER) {
                pen = xor_sum(pen);
                cost = -(xn[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - xn[0], 0.0f) - 10.0f * fmaxf(fabsf(xn[1]) - 0.2f, 0.0f) - pen);
            } else if (ENV == METRPO_ENV_ANT) {
                cost = -(xn[15] - 1e-2f * 0.5f * su2 + 0.05f);
                int f2 = fin & __shfl_xor(fin, 16, 64);
                f2 &= __shfl_xor(f2, 32, 64);
                const float zc = xn[2];
                dn = !((zc >= 0.2f) && (zc <= 1.0f) && (f2 != 0));
            }
            dn = dn || (ts >= r.H);
            if (wave == 2 && q == 0 && active) { (r.rew + trow)[ub] = -cost; (r.done + trow)[ub] = dn ? 1 : 0; (r.tpath + trow)[ub] = ts - 1; }
        }
        PH_MARK(7)
        // ---- reset(dones) (:585-595) or advance; every wave keeps its own copy of the tile state ----------
        if (__any(dn)) {                                                   // rare: horizon reached / Ant fell
            int row = 0;
            if (dn) {
                if (active) {
                    row = (DRAWS && r.reset_idx != nullptr) ? (r.reset_idx + trow + r.B)[ub] : rng_index(dstep.w, r.n_pool);
                    cur_model = (DRAWS && r.reset_model != nullptr) ? (r.reset_model + trow + r.B)[ub] : rng_index16(dstep.z, K);
                }
                ts = 0;
            }
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int dim = 16 * cb + 4 * q + rr;
                    if (dim < NS) { const float xv = dn ? r.pool[(unsigned)(row * NS + dim)] : nx[cb][rr]; ST[e * NSP + dim] = xv; xc[cb][rr] = xv; }
                }
        } else {
#pragma unroll
            for (int cb = 0; cb < OUT_CB; ++cb) {
                xc[cb] = nx[cb];
                *(f32x4*)&ST[e * NSP + 16 * cb + 4 * q] = nx[cb];              // padded dims: dstd = dmean = 0 and zero weights keep them 0
            }
        }
        if (AHEAD) {                                                       // written by wave 1 before B0 of this step
            const float* src = RNGB + (((t + 1) & 1) * 16 + e) * 20;
            dstep = *(const uint4*)src;
            const float4 zf = *(const float4*)(src + 4 + 4 * q);
            z[0] = zf.x; z[1] = zf.y; z[2] = zf.z; z[3] = zf.w;
        }
        wave_lds_sync();
        PH_MARK(8)
    }
    if (ONE && t_end < r.T) {                                              // hand the tile over (another workgroup runs steps t_end..T-1)
        if (wave == 0) {
            const int lim = 16 * NS;
            for (int i = lane; i < lim; i += 64) r.mig_obs[(size_t)b0 * NS + i] = ST[(i / NS) * NSP + i % NS];
            if (q == 0) { r.mig_ts[b] = ts; r.mig_model[b] = cur_model; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) __hip_atomic_store(&r.mig_flag[tile], r.mig_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (wave == 0 && r.last_obs != nullptr) {
            const int lim = min(16, r.B - b0) * NS;
            for (int i = lane; i < lim; i += 64) r.last_obs[(size_t)b0 * NS + i] = ST[(i / NS) * NSP + i % NS];
        }
        if (wave == 0 && q == 0 && active) {
            if (r.last_ts != nullptr) r.last_ts[b] = ts;
            if (r.last_model != nullptr) r.last_model[b] = cur_model;
        }
    }
    __syncthreads();                                                       // the next piece re-initialises the tile state in LDS
    }
    PH_DUMP
}


// -------------------------------------------------------------------------------------------------
typedef void (*coop_kernel_t)(RolloutK, const float*, const float*, const float*);
template <int ENV> static constexpr bool coop_two_per_cu_spills() { return EnvDim<ENV>::NS > 16; }   // see the launch rule in rollout_coop.hip
// kern[one workgroup per CU][draws supplied]; kern[0][*] == nullptr: the shape only has the one-workgroup-per-CU instantiation (K > 5)
struct CoopEntry { int env, K; coop_kernel_t kern[2][2]; int lds_floats; bool always_one; double pair, rem_slope; };
#define CENTRY(ENVID, KK) {ENVID, KK, {{k_rollout_coop<ENVID, KK, false, false>, k_rollout_coop<ENVID, KK, false, true>}, \
                                       {k_rollout_coop<ENVID, KK, true, false>, k_rollout_coop<ENVID, KK, true, true>}}, Coop<ENVID, KK>::TOTAL, coop_two_per_cu_spills<ENVID>(), \
                                       (ENVID == METRPO_ENV_SWIMMER) ? 1.50 : (ENVID == METRPO_ENV_SNAKE) ? 1.58 : 1.65, (ENVID == METRPO_ENV_SWIMMER) ? 0.0 : 1.0}
// K = 6 ... 10 heads (round 6): K x 23 weight fragments per wave need the WHOLE register file of a SIMD (512 registers: one workgroup per CU) -- only that instantiation
// exists; each K is a translation unit of its own (rollout_coop_k<K>.hip: these kernels compile for ~30 s apiece)
// (the LDS map grows with K x padded state width: Ant holds 8 heads in a CU's 160 KB, half-cheetah 9, the 16-dim envs 10 -- larger K: no kernel, nullptr)
template <int ENV, int K, bool DRAWS> constexpr coop_kernel_t coop_one_kernel() {
    if constexpr (Coop<ENV, K>::TOTAL * sizeof(float) <= 160 * 1024) return k_rollout_coop<ENV, K, true, DRAWS>; else return nullptr;
}
#define CENTRY_ONE(ENVID, KK) {ENVID, KK, {{nullptr, nullptr}, {coop_one_kernel<ENVID, KK, false>(), coop_one_kernel<ENVID, KK, true>()}}, Coop<ENVID, KK>::TOTAL, true, 0.0, 0.0}
#define COOP_WIDE_TABLE(NAME, KK) extern const CoopEntry NAME[5]; const CoopEntry NAME[5] = { CENTRY_ONE(METRPO_ENV_SWIMMER, KK), CENTRY_ONE(METRPO_ENV_HALF_CHEETAH, KK), \
    CENTRY_ONE(METRPO_ENV_HOPPER, KK), CENTRY_ONE(METRPO_ENV_SNAKE, KK), CENTRY_ONE(METRPO_ENV_ANT, KK) };
