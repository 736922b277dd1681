// MFMA fast path of the policy-side TRPO kernels for the 2-hidden-layer tanh policies the reference
// ships (params-*.json "policy.hidden_layers": [32, 32]): surrogate loss + gradient, Fisher-vector
// product, loss + mean-KL.  Same arithmetic as policy_update.hip (algos/npo.py:68-75 graph; [rllab]
// DiagonalGaussian / PerlmutterHvp), mapped to v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains):
//
//   * a wave owns tiles of 16 samples; weight fragments come from an LDS image the block builds once (the compiler keeps what fits in
//     registers and re-reads the rest per tile);
//   * forward, tangent-forward and back-prop run TRANSPOSED (H^T[unit][sample] = W^T X^T): the D
//     fragment of one layer is the B operand of the next when the k-steps are enumerated (cb, r)
//     (same trick as rollout_mfma.hip), so these chains never leave registers;
//   * the weight gradients  G[i][j] = sum_n a[i][n] d[j][n]  contract over SAMPLES, i.e. over the lane
//     index of the D fragments, so h0, h1 and the layer-1 deltas take one 16x16 transpose through LDS (read back 16 bytes at a time), the
//     layer-0 deltas are produced in that orientation directly (MFMA with swapped operands), and G accumulates in MFMA accumulators
//     across all tiles of the wave;
//   * waves -> block partial (LDS, fixed order) -> global partial row -> k_finalize (fixed order,
//     float64): bitwise reproducible.
#include <type_traits>
#include "device_common.h"
#include "cg_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PKFMA(a, b, c) __builtin_elementwise_fma((a), (b), (c))     // v_pk_fma_f32: both action dims of the VALU output layer per instruction
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define PART_EXTRA 3
#ifndef POL_SKIP
#define POL_SKIP 0              // developer experiments (tools/build_variant.sh): bit mask of tile stages to leave out; results are then meaningless
#endif
#ifndef POL_SPLIT_R
// The two waves that share a SIMD do not run at the same speed: the one launched first (waves 0-3 of the block) gets through its tiles ~17 %
// faster (tools/pol_phases.py: 110k vs 129k cycles for 16 tiles each), and then idles while the other finishes alone.  The tiles of a SIMD's
// pair are therefore dealt (R+1)/2 : (R-1)/2 in rounds of R (odd); a fixed assignment, so results stay bitwise reproducible.  R = 13 (7 : 6)
// measured best of {9, 13, 17}: FVP 65.1 -> 63.5 us, gradient 75.1 -> 72.4 us at N = 500 000.  0 = equal shares.
#define POL_SPLIT_R 13
#endif
#ifndef POL_SPLIT_R_FVP
// The cached-activation FVP and the loss / KL evaluation are lighter per tile and the gap between the two waves of a SIMD is wider there (102 k vs
// 115 k cycles at 7 : 6): 4 : 3 measured best of {5, 7, 9, 11, 13} (FVP 63.5 -> 62.2 us, evaluation 34.4 -> 32.8 us under the kernel tracer; the
// gradient kernel keeps 7 : 6, 71.5 vs 72.0).
#define POL_SPLIT_R_FVP 7
#endif
#ifndef POL_DEFER_S7
// 1: the sample-contracted weight-gradient products of a tile (S7: 24 matrix instructions at 2 x 32, na <= 2) are issued ONE TILE LATER, in four groups placed
// inside the next tile's vector-ALU stretch (output layer, tanh' factors, deltas: ~95 instructions with no matrix instruction of their own) -- their operands
// wait in 28 registers.  Same products in the same order: the same sums bit for bit.
#define POL_DEFER_S7 0      // measured (round 5): Fisher-vector product 59.8 -> 60.5 us at C1 -- the stretch is already covered by the SIMD's other wave; off
#endif
#ifndef NWAVES
#define NWAVES 8                // waves per block: one block per CU (2 waves per SIMD), the weight image is shared by all 8
#endif
constexpr int cdiv_(int a, int b) { return (a + b - 1) / b; }
static_assert(POL_SPLIT_R == 0 || (NWAVES == 8 && (POL_SPLIT_R & 1) == 1 && POL_SPLIT_R >= 3 && (POL_SPLIT_R_FVP & 1) == 1 && POL_SPLIT_R_FVP >= 3), "the uneven deal pairs wave w with wave w + 4");
// Developer instrumentation (SRC=policy_mfma.hip tools/build_variant.sh ptiming -DPOL_TIMING): s_memtime at the phase boundaries of the
// cached-activation FVP kernel, waves of workgroup 0, read back with metrpo_debug_pol_phases (tools/pol_phases.py).  Not in the shipped library.
#ifdef POL_TIMING
__device__ unsigned long long g_pol_phase[16][8];
#define PT_MARK(i) { if (MODE_ == MODE_FVPC && blockIdx.x == 0 && lane == 0) g_pol_phase[wave][i] = __builtin_readcyclecounter(); }
extern "C" int32_t metrpo_debug_pol_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pol_phase), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1; }
#else
#define PT_MARK(i)
#endif
// CGP_TIMING (developer build: SRC=policy_mfma.hip tools/build_variant.sh cgptiming -DCGP_TIMING; tools/cgp_phases.py): s_memrealtime (100 MHz) of thread 0 of
// workgroups 0 and gridDim-1 at the phase boundaries of every iteration of the persistent CG solve
#ifdef CGP_TIMING
__device__ unsigned long long g_cgp_phase[2][16][8];
#define CT_MARK(i) { if (tid == 0 && it < 16 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_cgp_phase[blockIdx.x == 0 ? 0 : 1][it][i] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int32_t metrpo_debug_cgp_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cgp_phase), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -1; }
#else
#define CT_MARK(i)
#endif


__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// sum over the four q-lanes of a sample (lanes c, c + 16, c + 32, c + 48) on the vector ALU: v_permlane16_swap (odd rows of one copy <-> even rows of the other:
// the two copies then hold v[row] and v[row ^ 1] between them) and v_permlane32_swap (upper half <-> lower half), gfx950 -- the xor-16 / xor-32 butterfly
// (v + v[lane ^ 16]) + (..)[lane ^ 32] bit for bit (the additions commute), without the two ds_bpermute round trips per sum that sat on every tile's
// dependent chain (4 per tile in the Fisher-vector product)
__device__ __forceinline__ float xsum_q(float v) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const u32x2_ b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// sum over the 16 lanes c of a row, on the VALU's DPP path (quad swaps, then half-row and row mirrors: after each step the lanes that are
// exchanged hold equal partial sums, so this is the xor-1, 2, 4, 8 butterfly bit for bit).  The epilogue runs ~45 of these per wave: as
// ds_bpermute chains (__shfl_xor) they were 6 us of every launch, measured with tools/pol_phases.py.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float xsum_c(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    return v;
}

enum { MODE_GRAD = 0, MODE_FVP = 1, MODE_LOSSKL = 2, MODE_FVPC = 3, MODE_CGP = 4 };
// POL_PRIO (experiment, round 4): issue priorities of the two waves of a SIMD.  1: the younger wave of every SIMD (waves 4 .. 7) runs at priority 1 for the whole
// kernel; 2: a wave raises its priority for the vector-ALU stretch of a tile (S4 / S5: output layer, tanh' factors, deltas) and drops it for the matrix runs.
#ifndef POL_PRIO
#define POL_PRIO 0
#endif
// POL_H0R = 1 (experiment, round 4): MODE_FVPC reads only h1 from the cache and recomputes h0 (one layer: NS_KS x HB MFMAs + 4 HB tanh per lane): -128 of
// 302 B per sample from HBM, +8 % MFMAs.  Measured SLOWER at C1 (tools/variant_update.py h0r, twice: update 0.846 vs 0.798 ms = +4.8 us per
// product): the product is bound by its dependent issue chain, not by HBM; the extra layer + tanh at the head of every tile's chain costs more than the bytes.
#ifndef POL_H0R
#define POL_H0R 0
#endif
// MODE_FVPC: Fisher-vector product with the hidden activations h0, h1 = tanh(.) read from the cache the gradient kernel of the
// same (theta, batch) wrote (PolK::hcache) instead of being recomputed: all 10 products of a CG solve share theta and the
// observations, so the forward pass (22 of the 100 MFMAs of a tile and all 16 tanh per lane) is done once per update, not 11 times.
// Cache layout per 16-sample tile: [h0 cb0 | h0 cb1 | .. | h1 cb0 | ..][64 lanes] float4 in the MFMA D layout -- each wave
// instruction reads or writes 1 KB contiguously.

// LDS weight image, shared by the NWAVES waves of a block (filled once per block).  Tables with a col-block index store the HB
// col-block fragments of one k-step adjacently per lane, so one ds_read_b64 (HB = 2) feeds both MFMAs of that k-step.
template <int NS, int NA, int PH>
struct PolImg {
    static constexpr int NS_KS = cdiv_(NS, 4), HB = cdiv_(PH, 16), KK = HB * 4;
    // offsets in floats; [rows][64 lanes][HB] tables first, then [rows][64] tables
    static constexpr int O_W0F = 0, O_W1F = O_W0F + NS_KS * 64 * HB, O_V0F = O_W1F + KK * 64 * HB, O_V1F = O_V0F + NS_KS * 64 * HB,
                         O_W2B = O_V1F + KK * 64 * HB, O_W1B = O_W2B + 4 * 64 * HB, O_W2F = O_W1B + KK * 64 * HB,
                         O_V2F = O_W2F + KK * 64, TOTAL = O_V2F + KK * 64;
};

// MODE_CGP: ALL Fisher-vector products of a CG solve and the vector steps between them in ONE launch (one block per CU, all co-resident).  Per
// iteration: the cached-activation product exactly as MODE_FVPC runs it (same tile deal, same partial rows) -> grid barrier -> the float64 column
// sums in k_finalize's order, spread over the blocks -> the last block to arrive runs the krylov.cg step (cg_device.h: cgv_step_body, bit for bit the
// fused tail's) and releases the others, which then copy the new tangent tables.  What a launch-per-product solve spends on ten kernel starts /
// drains, ten reductions' launches and their tails' ticket waits stays inside: see DESIGN.md section 4d.
constexpr int CGP_RMAX = 3;         // MODE_CGP: P <= 3 * 1024 (Ant's 2 x 32 policy: 2 288)
struct CgpArgs {
    CgTail tail;                    // op 1 fields: x r p z step scal gout pf vpos imgval, P, reg, tol, max_kl, implicit_hd
    int n_it, n_params;
    unsigned int* bar;              // [0] arrivals behind the products, [1] arrivals behind the column sums, [2] CG steps released; zero at launch
    const float* theta_ls;          // raw log_std parameters (theta + n_params)
    float* partials;                // the launch's partial rows (no __restrict__: read across workgroups)
    unsigned long long timeout;     // s_memrealtime ticks (100 MHz) a block waits at a barrier before it gives up (scal[S_COMMERR] = 2)
};
#define LDA(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ bool cgp_wait(const unsigned int* ctr, unsigned int target, unsigned long long timeout) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) return false;
    }
    return true;
}

template <int NS, int NA, int PH, int MODE_>
__device__ __forceinline__ void pol_body(const PolK& k, const float* __restrict__ theta, const float* __restrict__ v_in, float* __restrict__ partials_in,
                                         const CgpArgs* cgp) {
    // MODE_CGP: the tangent vector and the other blocks' partial rows CHANGE during the launch (written by other workgroups between the grid barriers):
    // no __restrict__ on these two, or their loads may legally be hoisted out of the iteration loop
    const float* v = v_in; float* partials = partials_in;
    if constexpr (MODE_ == MODE_CGP) { v = cgp->tail.pf; partials = cgp->partials; }
    using I = PolImg<NS, NA, PH>;
    constexpr bool PERSIST = (MODE_ == MODE_CGP);
    constexpr bool CACHED = (MODE_ == MODE_FVPC || PERSIST);
    constexpr int MODE = CACHED ? MODE_FVP : MODE_;
    constexpr int NS_KS = I::NS_KS, NSI = cdiv_(NS, 16), HB = I::HB, KK = I::KK;
    constexpr int pW0 = 0, pb0 = NS * PH, pW1 = pb0 + PH, pb1 = pW1 + PH * PH, pW2 = pb1 + PH, pb2 = pW2 + PH * NA,
                  pLS = pb2 + NA, P = pLS + NA, ROW = P + PART_EXTRA;
    // transpose tile T[unit][sample]: k-step s of S7 contracts, in lane (unit, q), sample 4q + s, so a lane's four samples are CONTIGUOUS:
    // one ds_read_b128 per tile and array instead of four scalar reads (S7's LDS reads were 11 of an FVP's 70 us).  The same sample <-> k-step
    // assignment is what an MFMA with swapped operands produces (S6), so the layer-0 deltas need no tile at all.  Row stride 20 floats keeps
    // the 16-byte reads aligned (2-way bank conflicts on both sides).
    constexpr int TS = 20, TILE = 16 * TS;
    constexpr int WTL = (3 * HB + (NA <= 2 ? 0 : 1)) * TILE;   // per-wave transpose tiles: h0, h1, d1 (HB each) and, with the output layer on the MFMA, u
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index in an SGPR: tile indices and base pointers stay scalar
    const int c = lane & 15, q = lane >> 4;
    if (POL_PRIO == 1 && wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);
    if (POL_PRIO == 3 && wave < NWAVES / 2) __builtin_amdgcn_s_setprio(1);
    if (MODE_ == MODE_LOSSKL && k.skip != nullptr && k.skip[0] >= 0.0) return;      // speculative line-search trial after the search stopped
    float* IMG = lds;
    float* TL = lds + I::TOTAL + wave * WTL;
    PT_MARK(0)
    const long long ntiles = (k.N + 15) / 16;
    const f32x4* __restrict__ hc = (const f32x4*)k.hcache;
    // a state-independent old log_std (stride 0: the reference's GaussianMLPPolicy) is one value per action dim for the whole batch: its loads
    // and its two exponentials per sample leave the tile loop (the loss / KL kernel is VALU-bound: 340 instructions per tile, 39 transcendental)
    const bool ols_const = (MODE != MODE_FVP) && k.ls_stride == 0 && k.old_ls != nullptr;
    // Everything a tile reads from HBM (observations in both layouts, valid flag, cached activations, and for the loss modes the old
    // distribution / action / advantage) is fetched ONE TILE AHEAD into registers: consumed in the iteration that issued them, the
    // valid flag and the observation loads each put a full HBM round trip (~2000 cycles) on the wave's critical path, per tile.
    struct TileIn {
        float xB[NS_KS]; float xTs[4][NSI]; float ols[4], omu[4], act[4], adv; f32x4 gmv; int vld, nrem; f32x4 h[2 * HB];
    };
    // `tile` is wave-uniform (SGPRs): scalar base pointers + 32-bit lane offsets, and every load is issued unconditionally on a clamped
    // (always valid) address; what lies outside the batch or in the feature padding is zeroed when the tile is CONSUMED (mask_tile; a
    // select at fetch time would wait for the load).  As written before (bounds-checked 64-bit per-lane addresses) the fetch was 17
    // exec-masked branches and ~150 instructions per tile.
    auto fetch = [&](long long tile_, TileIn& in) {
        const long long tile = (tile_ < ntiles) ? tile_ : ntiles - 1;
        const long long n0 = tile * 16;
        const int nrem = (int)((k.N - n0 < 16) ? k.N - n0 : 16);            // samples of this tile that exist (>= 1)
        in.nrem = nrem;
        const int cl = (c < nrem) ? c : nrem - 1;
        const float* __restrict__ ob = k.obs + n0 * NS;
#pragma unroll
        for (int s = 0; s < NS_KS; ++s) { const int f = 4 * s + q; in.xB[s] = ob[cl * NS + ((f < NS) ? f : NS - 1)]; }
        if (MODE != MODE_LOSSKL) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int sl = 4 * q + s, slc = (sl < nrem) ? sl : nrem - 1;     // lane q holds samples 4q .. 4q+3 (k-step s of S7 covers samples 4q+s)
#pragma unroll
                for (int ci = 0; ci < NSI; ++ci) { const int f = 16 * ci + c; in.xTs[s][ci] = ob[slc * NS + ((f < NS) ? f : NS - 1)]; }
            }
        }
        in.vld = (k.valid == nullptr) ? 1 : (int)k.valid[n0 + cl];
        if (MODE != MODE_FVP) {
            const long long nl = n0 + cl;
            if (MODE == MODE_GRAD && k.gm != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int d = 4 * q + r; in.gmv[r] = k.gm[nl * NA + ((d < NA) ? d : NA - 1)]; }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 4 * q + r, dc = (d < NA) ? d : NA - 1;
                    in.ols[r] = ols_const ? 0.f : k.old_ls[(size_t)nl * k.ls_stride + dc]; in.omu[r] = k.old_mean[nl * NA + dc]; in.act[r] = k.act[nl * NA + dc];
                }
                in.adv = k.adv[nl];
            }
        }
        if (CACHED) {
            const f32x4* __restrict__ hb_ = hc + tile * (2 * HB) * 64;
#pragma unroll
            for (int j = (POL_H0R ? HB : 0); j < 2 * HB; ++j) in.h[j] = hb_[j * 64 + lane];
        }
    };
    auto mask_tile = [&](TileIn& in) {
        // feature padding of the B-operand observations (k rows f >= NS meet zero weights, but 0 x garbage must stay 0)
#pragma unroll
        for (int s = 0; s < NS_KS; ++s) if (4 * s + 3 >= NS && 4 * s + q >= NS) in.xB[s] = 0.f;
        if (in.nrem == 16) return;                          // wave-uniform: only the batch's last tile is partial
        const bool inr = c < in.nrem;
#pragma unroll
        for (int s = 0; s < NS_KS; ++s) if (!inr) in.xB[s] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ci = 0; ci < NSI; ++ci) if (4 * q + s >= in.nrem) in.xTs[s][ci] = 0.f;
        if (!inr) { in.vld = 0; in.adv = 0.f; }
#pragma unroll
        for (int r = 0; r < 4; ++r) if (!inr) { in.ols[r] = 0.f; in.omu[r] = 0.f; in.act[r] = 0.f; in.gmv[r] = 0.f; }
    };
    TileIn nxt;

    // per-lane biases / output-layer weights first: their loads are in flight while the image's map -> gather round trips run
    // fragment accessors (this lane's element)
#define FRAG2(off, row, cb) IMG[(off) + ((row) * 64 + lane) * HB + (cb)]
#define FRAG1(off, row) IMG[(off) + (row) * 64 + lane]
    f32x4 b0f[HB], b1f[HB], b2f;
#pragma unroll
    for (int cb = 0; cb < HB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int u = 16 * cb + 4 * q + r; b0f[cb][r] = (u < PH) ? theta[pb0 + u] : 0.f; b1f[cb][r] = (u < PH) ? theta[pb1 + u] : 0.f; }
#pragma unroll
    for (int r = 0; r < 4; ++r) b2f[r] = (4 * q + r < NA) ? theta[pb2 + 4 * q + r] : 0.f;
    float ls[4], inv_std[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { ls[r] = (4 * q + r < NA) ? fmaxf(theta[pLS + 4 * q + r], LOG_MIN_STD) : 0.f; inv_std[r] = expf(-ls[r]); }
    float fisher_w[4];                                      // 1 / (std^2 + eps/2), hoisted out of the tile loop (exp + full-precision division per tile)
#pragma unroll
    for (int r = 0; r < 4; ++r) fisher_w[r] = 1.0f / (expf(2.f * ls[r]) + 0.5f * KL_EPS);
    float ols_c[4] = {0.f, 0.f, 0.f, 0.f}, eo_c[4] = {1.f, 1.f, 1.f, 1.f}, os2_c[4] = {1.f, 1.f, 1.f, 1.f};
    if (ols_const) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < NA) { ols_c[r] = k.old_ls[4 * q + r]; eo_c[r] = expf(-ols_c[r]); os2_c[r] = expf(2.f * ols_c[r]); }
    }
    const int n_it = PERSIST ? cgp->n_it : 1;
    for (int it = 0; it < n_it; ++it) {                     // PERSIST: one pass per CG iteration; everything that depends on the tangent vector is (re)loaded inside
    if constexpr (PERSIST) { CT_MARK(0) }
    f32x4 vb0f[HB], vb1f[HB], vb2f;
    if (MODE == MODE_FVP) {
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int u = 16 * cb + 4 * q + r; vb0f[cb][r] = (u < PH) ? v[pb0 + u] : 0.f; vb1f[cb][r] = (u < PH) ? v[pb1 + u] : 0.f; }
#pragma unroll
        for (int r = 0; r < 4; ++r) vb2f[r] = (4 * q + r < NA) ? v[pb2 + 4 * q + r] : 0.f;
    }
    // Output layer on the VALU when it is only 1-2 units wide (na <= 2): a 16-wide MFMA would be 87 % padding there.  Each lane keeps
    // the weights of its 8 hidden units (D layout) for every action dim; partial sums are combined over the 4 q-lanes of a sample.
    constexpr bool L2V = (NA <= 2);
    constexpr int NAV = L2V ? NA : 1;
    float w2l[HB][4][NAV], v2l[HB][4][NAV], gw2l[HB][4][NAV], b2l[NAV], vb2l[NAV], fisher_d[NAV];   // fisher_d: fisher_w by action dim, in every lane
    if (L2V) {
#pragma unroll
        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < NAV; ++d) {
                    const int u = 16 * cb + 4 * q + r;
                    w2l[cb][r][d] = (u < PH) ? theta[pW2 + u * NA + d] : 0.f;
                    v2l[cb][r][d] = (MODE == MODE_FVP && u < PH) ? v[pW2 + u * NA + d] : 0.f;
                    gw2l[cb][r][d] = 0.f;
                }
#pragma unroll
        for (int d = 0; d < NAV; ++d) {
            b2l[d] = theta[pb2 + d]; vb2l[d] = (MODE == MODE_FVP) ? v[pb2 + d] : 0.f;
            fisher_d[d] = 1.0f / (expf(2.f * fmaxf(theta[pLS + d], LOG_MIN_STD)) + 0.5f * KL_EPS);
        }
    }
    // ---------------- weight fragment image -> LDS (each element written by exactly one thread) ----------------
    // k.img_map[i] = source index of image element i in theta (bit 30 clear) or in v (bit 30 set), -1 = zero; built once on the
    // host (pol_image_map).  Map loads, gathers and LDS stores are issued in independent batches of IMG_U per thread: the
    // prologue costs ~2 L2 round trips instead of one dependent global load per element.
    constexpr int SPL = POL_SPLIT_R ? ((MODE_ == MODE_GRAD || MODE_ == MODE_FVP) ? POL_SPLIT_R : POL_SPLIT_R_FVP) : 0;      // rounds of the uneven tile deal (0: equal shares)
    const long long first_tile = SPL ? (long long)blockIdx.x * 4 + (wave & 3) + ((wave < 4) ? 0 : 1) * (long long)gridDim.x * 4 : (long long)blockIdx.x * NWAVES + wave;
    if (CACHED && k.imgval != nullptr) {
        // inside a fused CG solve the image already exists in global memory, element for element (weight entries: published by block 0 of the
        // gradient kernel of this theta; tangent entries: stored by the CG tail that produced this product's input vector): one coalesced copy
        // = one L2 round trip, instead of the map loads and the gathers that depend on them (9 k of the kernel's 130 k cycles at N = 500 000)
        static_assert(I::TOTAL % 4 == 0, "image tables are multiples of 64 floats");
        constexpr int NQ = I::TOTAL / 4, NIT = cdiv_(NQ, NWAVES * 64);
        float4 w4[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int i = it * NWAVES * 64 + tid; w4[it] = (i < NQ) ? ((const float4*)k.imgval)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
        asm volatile("" ::: "memory");
        fetch(first_tile, nxt);
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int i = it * NWAVES * 64 + tid; if (i < NQ) ((float4*)IMG)[i] = w4[it]; }
    } else {
        constexpr int IMG_U = 8;
#pragma unroll
        for (int i0 = 0; i0 < I::TOTAL; i0 += NWAVES * 64 * IMG_U) {
            int m[IMG_U]; float w[IMG_U];
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) { const int i = i0 + u * NWAVES * 64 + tid; m[u] = (i < I::TOTAL) ? k.img_map[i] : -1; }
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) {
                const int i = i0 + u * NWAVES * 64 + tid;
                bool use = m[u] >= 0;
                if (MODE != MODE_FVP && (m[u] & 0x40000000)) use = false;                       // tangent tables: FVP only
                if (MODE == MODE_LOSSKL && i >= I::O_W2B && i < I::O_W2F) use = false;          // back-prop tables unused
                if (CACHED && !POL_H0R && i < I::O_W1F) use = false;                            // W0 forward table unused
                w[u] = 0.f;
                if (use) w[u] = (m[u] & 0x40000000) ? v[m[u] & 0x3FFFFFFF] : theta[m[u]];
            }
            // the first tile's loads go out behind the last batch of gathers (vmcnt retires in order: issued any earlier, their HBM round
            // trip would hold up the map loads' return): they overlap the LDS stores, the barrier and the bias loads
            if (i0 + NWAVES * 64 * IMG_U >= I::TOTAL) { asm volatile("" ::: "memory"); fetch(first_tile, nxt); asm volatile("" ::: "memory"); }
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) { const int i = i0 + u * NWAVES * 64 + tid; if (i < I::TOTAL) IMG[i] = w[u]; }
        }
    }
    __syncthreads();
    if constexpr (PERSIST) { CT_MARK(1) }
    if (MODE_ == MODE_GRAD && k.imgval != nullptr && blockIdx.x == 0) {      // publish the image of this theta for the CG products that follow (tangent entries: zero here, the CG tails fill them)
        for (int i = tid; i < I::TOTAL; i += NWAVES * 64) k.imgval[i] = IMG[i];
    }
    PT_MARK(1)

    // ---------------- accumulators -------------------------------------------------------------------
    const f32x4 Z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gW0[NSI][HB], gW1[HB][HB], gW2[HB], gb0[HB], gb1[HB], gb2 = Z4;
#pragma unroll
    for (int a = 0; a < HB; ++a) {
        gW2[a] = Z4; gb0[a] = Z4; gb1[a] = Z4;
#pragma unroll
        for (int b = 0; b < HB; ++b) gW1[a][b] = Z4;
#pragma unroll
        for (int b = 0; b < NSI; ++b) gW0[b][a] = Z4;
    }
    float dls[4] = {0.f, 0.f, 0.f, 0.f};
    float acc0 = 0.f, acc1 = 0.f, accw = 0.f;               // loss, kl, valid weight (per-lane partials)

    constexpr bool DEFER = POL_DEFER_S7 && (NA <= 2) && MODE != MODE_LOSSKL;
    f32x4 qa0[HB], qb1[HB], qd0[HB]; float qxT[4][NSI];     // DEFER: S7 operands of the previous tile (zeros in front of the first: its run adds nothing)
#pragma unroll
    for (int cb = 0; cb < HB; ++cb) { qa0[cb] = Z4; qb1[cb] = Z4; qd0[cb] = Z4; }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int ci = 0; ci < NSI; ++ci) qxT[s_][ci] = 0.f;
    auto s7_step = [&](auto sc) {                           // k-step s of the pending tile's products (samples 4q + s)
        constexpr int s_ = decltype(sc)::value;
        if constexpr (DEFER) {
#pragma unroll
            for (int ci = 0; ci < HB; ++ci)
#pragma unroll
                for (int cj = 0; cj < HB; ++cj) gW1[ci][cj] = MFMA16(qa0[ci][s_], qb1[cj][s_], gW1[ci][cj]);
#pragma unroll
            for (int ci = 0; ci < NSI; ++ci)
#pragma unroll
                for (int cj = 0; cj < HB; ++cj) gW0[ci][cj] = MFMA16(qxT[s_][ci], qd0[cj][s_], gW0[ci][cj]);
        }
    };
#define S7_STEP(n) s7_step(std::integral_constant<int, (n)>{})
    // vmcnt(0) HERE: otherwise the wait for these first loads is placed inside the loop, at the top of every iteration, right
    // behind the prefetch of the next tile -- which it then waits for as well
    __builtin_amdgcn_s_waitcnt(0x0F70);
    PT_MARK(2)
    const long long sp_base = SPL ? (long long)blockIdx.x * 4 + (wave & 3) : (long long)blockIdx.x * NWAVES + wave, sp_stride = (long long)gridDim.x * (SPL ? 4 : NWAVES);
    auto sp_next = [&](long long m) -> long long {
        if constexpr (SPL == 0) return m + 1;
        else { const int ph = (int)(m % SPL); return (wave < 4) ? ((ph == SPL - 1) ? m + 1 : m + 2) : ((ph == SPL - 2) ? m + 3 : m + 2); }
    };
    for (long long m = (SPL && wave >= 4) ? 1 : 0, tile = sp_base + m * sp_stride; tile < ntiles; m = sp_next(m), tile = sp_base + m * sp_stride) {
        const long long n0 = tile * 16, n = n0 + c;
        const bool inr = n < k.N;
        if (POL_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        TileIn in = nxt;
        fetch(sp_base + sp_next(m) * sp_stride, nxt);
        asm volatile("" ::: "memory");                      // the loads are issued HERE (left alone, the compiler sinks them to the end of the iteration)
        mask_tile(in);
        const bool ok = inr && in.vld != 0;
        // The tile is processed as a few long MFMA runs with the VALU work of the neighbouring stages placed textually inside them
        // (it issues in the matrix pipe's shadow), and every activation is dropped into its own wave-private transpose tile the
        // moment it exists, so the three sample-contracted weight-gradient products run as ONE run after a single LDS sync.
        const float (&xB)[NS_KS] = in.xB;
        const float (&xTs)[4][NSI] = in.xTs;                // observations transposed ([feature 16ci + c][sample 4q + s]) for S7
        const int wpos = c;                                 // T[unit][sample]: lane (unit, q) of S7 reads its samples 4q .. 4q+3 as one 16-byte word
        float* T_H0 = TL, *T_H1 = TL + HB * TILE, *T_D1 = TL + 2 * HB * TILE, *T_UM = TL + 3 * HB * TILE;
        // ---- S1: layer 0, forward and (FVP) tangent  ------------------------------------------------------
        f32x4 h0[HB], h1[HB], t0[HB], t1[HB];
        constexpr bool H0C = CACHED && !POL_H0R;            // h0 comes from the cache
        if (CACHED) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) { if (H0C) h0[cb] = in.h[cb]; h1[cb] = in.h[HB + cb]; }
        }
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) { if (!H0C) h0[cb] = b0f[cb]; if (MODE == MODE_FVP) t0[cb] = vb0f[cb]; }
#pragma unroll
        for (int s = 0; s < NS_KS; ++s)
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) {
                if (!H0C) h0[cb] = MFMA16(FRAG2(I::O_W0F, s, cb), xB[s], h0[cb]);
                if (MODE == MODE_FVP) t0[cb] = MFMA16(FRAG2(I::O_V0F, s, cb), xB[s], t0[cb]);
            }
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
            if (!CACHED) h1[cb] = b1f[cb];
            if (MODE == MODE_FVP) t1[cb] = vb1f[cb];
            if (!H0C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h0[cb][r] = tanh_fast(h0[cb][r]);
            }
        }
        // ---- S2: layer 1 on h0: forward h1 += W1^T h0 and (FVP) t1 += V1^T h0;  VALU inside: t0 *= 1 - h0^2, h0 -> T_H0 ----
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) {
                if (!CACHED) h1[cb] = MFMA16(FRAG2(I::O_W1F, kk, cb), h0[kk >> 2][kk & 3], h1[cb]);
                if (MODE == MODE_FVP) { if (POL_SKIP & 4) t1[cb][0] += h0[kk >> 2][kk & 3]; else t1[cb] = MFMA16(FRAG2(I::O_V1F, kk, cb), h0[kk >> 2][kk & 3], t1[cb]); }
            }
        if (MODE != MODE_LOSSKL) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < ((POL_SKIP & 8) ? 0 : 4); ++r) T_H0[cb * TILE + (4 * q + r) * TS + wpos] = h0[cb][r];
        }
        if (MODE == MODE_FVP) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) t0[cb][r] *= fmaf(-h0[cb][r], h0[cb][r], 1.f);
            // ---- S3: t1 += W1^T t0;  VALU inside: tanh(h1) ------------------------------------------------
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int cb = 0; cb < HB; ++cb) { if (POL_SKIP & 4) t1[cb][1] += t0[kk >> 2][kk & 3]; else t1[cb] = MFMA16(FRAG2(I::O_W1F, kk, cb), t0[kk >> 2][kk & 3], t1[cb]); }
        }
        if (!CACHED) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) h1[cb][r] = tanh_fast(h1[cb][r]);
        }
        if (MODE != MODE_LOSSKL) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < (((POL_SKIP & 8) || NA <= 2) ? 0 : 4); ++r) T_H1[cb * TILE + (4 * q + r) * TS + wpos] = h1[cb][r];     // only the MFMA output layer (na > 2) reads it back
        }
        if (MODE == MODE_GRAD && k.hcache != nullptr) {     // publish the activations for the FVPs of this update
            f32x4* hw = (f32x4*)k.hcache + (size_t)tile * (2 * HB) * 64 + lane;
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) { hw[cb * 64] = h0[cb]; hw[(HB + cb) * 64] = h1[cb]; }
        }

        S7_STEP(0);
        if (POL_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        f32x4 um = Z4;                                      // d(objective)/d(mean) in D layout [d = 4q+r][sample c]
        float ual[NAV];                                     // FVP with the VALU output layer: the sample's mean-adjoint, already in all of its lanes
#pragma unroll
        for (int d = 0; d < NAV; ++d) ual[d] = 0.f;
        if (MODE == MODE_GRAD && k.gm != nullptr) {         // VJP mode (bptt.hip): the mean-adjoint is an input
#pragma unroll
            for (int r = 0; r < 4; ++r) um[r] = (ok && 4 * q + r < NA) ? in.gmv[r] : 0.f;
        } else if (MODE != MODE_FVP) {
            f32x4 mu = Z4;
            if (L2V) {
#pragma unroll
                for (int d = 0; d < NAV; ++d) {
                    float a = 0.f;
#pragma unroll
                    for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) a = fmaf(w2l[cb][r][d], h1[cb][r], a);
                    a = xsum_q(a) + b2l[d];
                    if (q == 0) mu[d] = a;                  // D layout: action dim d = 4q + r lives in lane q = 0, register d
                }
            } else {
                f32x4 m0 = b2f, m1 = Z4;
#pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    m0 = MFMA16(FRAG1(I::O_W2F, kk), h1[kk >> 2][kk & 3], m0);
                    m1 = MFMA16(FRAG1(I::O_W2F, kk + 1), h1[(kk + 1) >> 2][(kk + 1) & 3], m1);
                }
                mu = m0 + m1;
            }
            float llr = 0.f, kl = 0.f, zz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 4 * q + r;
                if (d < NA && ok) {
                    const float omu = in.omu[r], a = in.act[r];
                    float ols, eo, os2 = 0.f;
                    if (ols_const) { ols = ols_c[r]; eo = eo_c[r]; os2 = os2_c[r]; }          // wave-uniform branch
                    else {
                        ols = in.ols[r];
                        asm volatile("" : "+v"(ols));                                          // keeps the exponentials on this side of the branch
                        eo = expf(-ols);
                        if (MODE == MODE_LOSSKL) os2 = expf(2.f * ols);
                    }
                    const float z = (a - mu[r]) * inv_std[r], zo = (a - omu) * eo;
                    llr += (ols - ls[r]) + 0.5f * (zo * zo - z * z);
                    zz[r] = z;
                    if (MODE == MODE_LOSSKL) {
                        const float s2 = expf(2.f * ls[r]), dm = omu - mu[r];
                        kl += (dm * dm + os2 - s2) / (2.f * s2 + KL_EPS) + ls[r] - ols;
                    }
                }
            }
            llr = xsum_q(llr);                              // sum over action dims held by the 4 q-lanes of sample c
            const float la = ok ? expf(llr) * in.adv : 0.f;         // lr * adv
            if (q == 0) acc0 -= la * k.inv_n;               // surr_loss = -mean(lr*adv) (npo.py:75), once per sample
            if (MODE == MODE_LOSSKL) { acc1 += kl * k.inv_n; continue; }
            const float w = -la * k.inv_n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                um[r] = w * zz[r] * inv_std[r];             // d loss / d mean = w (a-mu)/std^2
                if (4 * q + r < NA) dls[r] += w * (zz[r] * zz[r] - 1.f);      // d loss / d log_std
            }
        } else {
            // ---- S4: tangent of the mean: m1 = V2^T h1 (VALU inside: t1 *= 1 - h1^2), then m0 = vb2 + W2^T t1 ----------
            f32x4 m0 = vb2f, m1 = Z4;
            if (!L2V) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) m1 = MFMA16(FRAG1(I::O_V2F, kk), h1[kk >> 2][kk & 3], m1);
            }
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) t1[cb][r] *= fmaf(-h1[cb][r], h1[cb][r], 1.f);
            if (L2V) {
                m0 = Z4;
                float av[NAV];
                if constexpr (NAV == 2) {
                    f32x2 a2 = {0.f, 0.f};
#pragma unroll
                    for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            a2 = PKFMA((f32x2{w2l[cb][r][0], w2l[cb][r][1]}), (f32x2{t1[cb][r], t1[cb][r]}),
                                       PKFMA((f32x2{v2l[cb][r][0], v2l[cb][r][1]}), (f32x2{h1[cb][r], h1[cb][r]}), a2));
                    av[0] = a2[0]; av[1] = a2[1];
                } else {
#pragma unroll
                    for (int d = 0; d < NAV; ++d) {
                        av[d] = 0.f;
#pragma unroll
                        for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) av[d] = fmaf(w2l[cb][r][d], t1[cb][r], fmaf(v2l[cb][r][d], h1[cb][r], av[d]));
                    }
                }
#pragma unroll
                for (int d = 0; d < NAV; ++d) {
                    const float a = xsum_q(av[d]) + vb2l[d];       // the butterfly leaves the same bits in all four q-lanes of the sample
                    if (q == 0) m0[d] = a;
                    ual[d] = ok ? a * fisher_d[d] * k.inv_n : 0.f;   // = um[d] of the sample's q = 0 lane: no broadcast needed in S5
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) m0 = MFMA16(FRAG1(I::O_W2F, kk), t1[kk >> 2][kk & 3], m0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)                     // d2 KL / d mean^2 = 1 / (s^2 + eps/2)
                um[r] = (ok && 4 * q + r < NA) ? (m0[r] + m1[r]) * fisher_w[r] * k.inv_n : 0.f;
            if (ok && q == 0) accw += k.inv_n;
        }
        S7_STEP(1);
        // ---- S5/S6: back-prop (transposed chain); deltas go straight into their transpose tiles ---------------------
        f32x4 d1[HB], d0n[HB];                              // d0n: layer-0 deltas in the OTHER orientation, [sample 4q+r][unit c] (see S6)
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) d1[cb] = Z4;
        if (L2V) {
            float ua[NAV];                                  // the sample's mean-adjoint, broadcast from its q = 0 lane to all 4 lanes
#pragma unroll
            for (int d = 0; d < NAV; ++d) ua[d] = (MODE == MODE_FVP) ? ual[d] : __shfl(um[d], c, 64);
#pragma unroll
            for (int cb = 0; cb < HB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int d = 0; d < NAV; ++d) d1[cb][r] = fmaf(w2l[cb][r][d], ua[d], d1[cb][r]);
                    // weight gradient of the output layer, per lane
                    if constexpr (NAV == 2) {
                        const f32x2 g2 = PKFMA((f32x2{h1[cb][r], h1[cb][r]}), (f32x2{ua[0], ua[1]}), (f32x2{gw2l[cb][r][0], gw2l[cb][r][1]}));
                        gw2l[cb][r][0] = g2[0]; gw2l[cb][r][1] = g2[1];
                    } else {
#pragma unroll
                        for (int d = 0; d < NAV; ++d) gw2l[cb][r][d] = fmaf(h1[cb][r], ua[d], gw2l[cb][r][d]);
                    }
                }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) T_UM[(4 * q + r) * TS + wpos] = um[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= NA) continue;                      // k-step r covers action dims r, 4+r, 8+r, 12+r: all padding when r >= na
#pragma unroll
                for (int cb = 0; cb < HB; ++cb) d1[cb] = MFMA16(FRAG2(I::O_W2B, r, cb), um[r], d1[cb]);
            }
        }
        S7_STEP(2);
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
            d0n[cb] = Z4;
#pragma unroll
            for (int r = 0; r < 4; ++r) { d1[cb][r] *= fmaf(-h1[cb][r], h1[cb][r], 1.f); T_D1[cb * TILE + (4 * q + r) * TS + wpos] = d1[cb][r]; }
        }
        S7_STEP(3);
        if (POL_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) { if (POL_SKIP & 4) d0n[cb][2] += d1[kk >> 2][kk & 3]; else d0n[cb] = MFMA16(d1[kk >> 2][kk & 3], FRAG2(I::O_W1B, kk, cb), d0n[cb]); }
        // S6 with the operands SWAPPED (A = the delta fragment, B = the very weight fragment the transposed chain uses as A): the product comes
        // out as D[sample 4q+r][unit c] -- units on the lane axis, which is what the sample-contracted products of S7 take as an operand.  The
        // layer-0 deltas therefore never go through LDS; their tanh' factor comes from the h0 tile S7 reads anyway (same orientation).
        gb2 += um;
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) gb1[cb] += d1[cb];
        // ---- S7: weight gradients G[i][j] += sum_n a[i][n] d[j][n] as one MFMA run ------------------------------------
        // D fragment [unit 16cb+4q+r][sample c] -> T[unit][sample]; k-step s of the MFMA covers samples 4q+s
        wave_sync_lds();
        f32x4 a1v[HB], a0v[HB], b1v[HB], buv = Z4;
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
            a1v[cb] = *(const f32x4*)&T_H1[cb * TILE + c * TS + 4 * q]; a0v[cb] = *(const f32x4*)&T_H0[cb * TILE + c * TS + 4 * q];
            b1v[cb] = *(const f32x4*)&T_D1[cb * TILE + c * TS + 4 * q];
        }
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d0n[cb][r] *= fmaf(-a0v[cb][r], a0v[cb][r], 1.f);
            gb0[cb] += d0n[cb];
        }
        if (!L2V) buv = *(const f32x4*)&T_UM[c * TS + 4 * q];
        if constexpr (DEFER) {                              // this tile's products run inside the next tile (or behind the loop)
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) { qa0[cb] = a0v[cb]; qb1[cb] = b1v[cb]; qd0[cb] = d0n[cb]; }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int ci = 0; ci < NSI; ++ci) qxT[s_][ci] = xTs[s_][ci];
        }
#pragma unroll
        for (int s = 0; s < ((DEFER || (POL_SKIP & 2)) ? 0 : 4); ++s) {
            const float bu = buv[s];
            float a1_[HB], a0_[HB], b1_[HB], b0_[HB], xT[NSI];
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) { a1_[cb] = a1v[cb][s]; a0_[cb] = a0v[cb][s]; b1_[cb] = b1v[cb][s]; b0_[cb] = d0n[cb][s]; }
#pragma unroll
            for (int ci = 0; ci < NSI; ++ci) xT[ci] = xTs[s][ci];
            if (POL_SKIP & 1) { gW1[0][0][0] += a0_[0] + b1_[0] + a1_[0] + b0_[0] + xT[0] + a0_[HB - 1] + b1_[HB - 1] + a1_[HB - 1] + b0_[HB - 1] + bu; continue; }
            if (!L2V) {
#pragma unroll
                for (int ci = 0; ci < HB; ++ci) gW2[ci] = MFMA16(a1_[ci], bu, gW2[ci]);
            }
#pragma unroll
            for (int ci = 0; ci < HB; ++ci)
#pragma unroll
                for (int cj = 0; cj < HB; ++cj) gW1[ci][cj] = MFMA16(a0_[ci], b1_[cj], gW1[ci][cj]);
#pragma unroll
            for (int ci = 0; ci < NSI; ++ci)
#pragma unroll
                for (int cj = 0; cj < HB; ++cj) gW0[ci][cj] = MFMA16(xT[ci], b0_[cj], gW0[ci][cj]);
        }
        wave_sync_lds();
    }
    S7_STEP(0); S7_STEP(1); S7_STEP(2); S7_STEP(3);       // the last tile's products
#undef S7_STEP
#undef FRAG2
#undef FRAG1

    // ---------------- epilogue: wave partial -> block partial (fixed order) -> global row -------------
    PT_MARK(3)
    __syncthreads();
    PT_MARK(4)
    float* RB = lds;                                        // [NWAVES][ROW] (weight image and transpose tiles are dead)
    float* row = RB + wave * ROW;
    // every column of the row is written exactly once below, except the log_std columns outside the gradient mode (zero there); the
    // loss / KL mode only produces the three scalar columns (and only those are summed and stored)
    if (MODE != MODE_GRAD && lane < NA) row[pLS + lane] = 0.f;
    wave_sync_lds();
    if (MODE != MODE_LOSSKL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int cj = 0; cj < HB; ++cj) {
                const int j = 16 * cj + c;
#pragma unroll
                for (int ci = 0; ci < NSI; ++ci) { const int i = 16 * ci + 4 * q + r; if (i < NS && j < PH) row[pW0 + i * PH + j] = gW0[ci][cj][r]; }
#pragma unroll
                for (int ci = 0; ci < HB; ++ci) { const int i = 16 * ci + 4 * q + r; if (i < PH && j < PH) row[pW1 + i * PH + j] = gW1[ci][cj][r]; }
            }
            if (L2V) {
#pragma unroll
                for (int ci = 0; ci < HB; ++ci)
#pragma unroll
                    for (int d = 0; d < NAV; ++d) {
                        const float sg = xsum_c(gw2l[ci][r][d]);               // sum over the 16 samples (lanes c) of the wave's tiles
                        const int i = 16 * ci + 4 * q + r;
                        if (c == 0 && i < PH) row[pW2 + i * NA + d] = sg;
                    }
            } else {
#pragma unroll
                for (int ci = 0; ci < HB; ++ci) { const int i = 16 * ci + 4 * q + r; if (i < PH && c < NA) row[pW2 + i * NA + c] = gW2[ci][r]; }
            }
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) {
                const float s1 = xsum_c(gb1[cb][r]);
                const int u = 16 * cb + 4 * q + r;
                if (c == 0 && u < PH) row[pb1 + u] = s1;
            }
            if (r == 0) {                                   // layer-0 bias gradient: accumulated as [sample 4q+r][unit c] -> sum over r, then over the q-lanes
#pragma unroll
                for (int cb = 0; cb < HB; ++cb) {
                    const float s0 = xsum_q((gb0[cb][0] + gb0[cb][1]) + (gb0[cb][2] + gb0[cb][3]));
                    if (q == 0 && 16 * cb + c < PH) row[pb0 + 16 * cb + c] = s0;
                }
            }
            const float s2 = xsum_c(gb2[r]), sl = xsum_c(dls[r]);
            if (c == 0 && 4 * q + r < NA) {
                row[pb2 + 4 * q + r] = s2;
                if (MODE == MODE_GRAD) row[pLS + 4 * q + r] = (theta[pLS + 4 * q + r] > LOG_MIN_STD) ? sl : 0.f;
            }
        }
    }
    {
        const float a0 = xsum_c(xsum_q(acc0)), a1 = xsum_c(xsum_q(acc1)), aw = xsum_c(xsum_q(accw));
        if (lane == 0) { row[P] = a0; row[P + 1] = a1; row[P + 2] = aw; }
    }
    __syncthreads();
    PT_MARK(5)
    float* out = partials + (size_t)blockIdx.x * ROW;
    for (int i = (MODE == MODE_LOSSKL ? P : 0) + tid; i < ROW; i += NWAVES * 64) {            // fixed pairwise order over the waves
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; w += 4) a += (RB[w * ROW + i] + RB[(w + 1) * ROW + i]) + (RB[(w + 2) * ROW + i] + RB[(w + 3) * ROW + i]);
        if constexpr (PERSIST) __hip_atomic_store(out + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1): summed by other workgroups, behind other L2s
        else out[i] = a;
    }
    PT_MARK(6)
    if constexpr (PERSIST) {
        CT_MARK(2)
        const CgpArgs& A = *cgp;
        const unsigned int nblk = gridDim.x;
        __shared__ unsigned int s_flag;
        // ---- barrier: every block's partial row is in memory
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Data that crosses workgroups inside the launch travels by agent-scope (sc1) stores and loads -- write-through, and never served from a line another
        // XCD's L2 still holds -- so the barriers carry no cache-wide release / acquire (256 workgroups x 3 of those per iteration cost more than the products:
        // measured 122 vs 66 us per iteration at C1).  The exception is the closing workgroup's CG step (plain stores: ONE release) and what every workgroup
        // reads of it afterwards (tangent tables, float copy of p: ONE acquire per workgroup and iteration, issued by one wave).
        if (tid == 0) {
            __hip_atomic_fetch_add(A.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = cgp_wait(A.bar, (unsigned int)(it + 1) * nblk, A.timeout) ? 1u : 0u;
        }
        __syncthreads();
        if (!s_flag) { if (tid == 0) A.tail.scal[S_COMMERR] = 2.0; return; }
        CT_MARK(3)
        // ---- column sums, float64, k_finalize's order (slice s adds rows s, s + 32, ..; the slice sums are added in slice order); 16 columns per block
        {
            constexpr int NSL = 32, FC = NWAVES * 64 / NSL;
            double* shd = (double*)lds;                     // [FC][NSL + 1]
            const int Pn = A.tail.P, sl = tid & 31;
            for (int o0 = (int)blockIdx.x * FC; o0 < Pn; o0 += (int)nblk * FC) {      // (block-uniform trip count: barriers inside)
            const int o = o0 + (tid >> 5);
            const bool lsrow = (o >= A.n_params && o < Pn);
            const int col = lsrow ? Pn + 2 : o;
            const int nrows = (int)nblk;
            double a = 0.0;
            if (o < Pn) {
                int b = sl;
                for (; b + 3 * NSL < nrows; b += 4 * NSL) {
                    const float v0 = LDA(partials + (size_t)b * ROW + col), v1 = LDA(partials + (size_t)(b + NSL) * ROW + col);
                    const float v2 = LDA(partials + (size_t)(b + 2 * NSL) * ROW + col), v3 = LDA(partials + (size_t)(b + 3 * NSL) * ROW + col);
                    a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
                }
                for (; b < nrows; b += NSL) a += (double)LDA(partials + (size_t)b * ROW + col);
            }
            shd[(tid >> 5) * (NSL + 1) + sl] = a;
            __syncthreads();
            if (sl == 0 && o < Pn) {
                double t = 0.0;
                for (int w = 0; w < NSL; ++w) t += shd[(tid >> 5) * (NSL + 1) + w];
                if (lsrow) {
                    const double raw = (double)A.theta_ls[o - A.n_params];
                    const double s2 = exp(2.0 * fmax(raw, (double)LOG_MIN_STD));
                    const double cc = 4.0 * s2 * (2.0 * s2 - 1e-8) / ((2.0 * s2 + 1e-8) * (2.0 * s2 + 1e-8));
                    t = (raw > (double)LOG_MIN_STD) ? cc * LDA(A.tail.p + o) * t : 0.0;
                }
                __hip_atomic_store(A.tail.z + o, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            }
        }
        CT_MARK(4)
        // ---- second barrier; the last block to arrive owns the complete product and runs the CG step, the others wait for its release
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned int tk = __hip_atomic_fetch_add(A.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = (tk == (unsigned int)(it + 1) * nblk - 1u) ? 1u : 0u;
        }
        __syncthreads();
        const bool closes = (s_flag != 0u);
        CT_MARK(5)
        __syncthreads();                                    // s_flag is written again below
        if (closes) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // z (sc1 stores of this iteration) and the previous closer's vectors, whatever this L2 held of them
            __syncthreads();
            double* cgsh = (double*)lds;
            const int last = (it == n_it - 1) ? 1 : 0;
            if (A.tail.P <= CG_R * 1024)
                cgv_step_body<1024 / (NWAVES * 64), CG_R>(A.tail.P, A.tail.reg, A.tail.tol, last, A.tail.x, A.tail.r, A.tail.p, A.tail.z,
                                                          PfOut{A.tail.pf, A.tail.vpos, A.tail.imgval}, A.tail.scal, cgsh);
            else
                cgv_step_body<1024 / (NWAVES * 64), CGP_RMAX>(A.tail.P, A.tail.reg, A.tail.tol, last, A.tail.x, A.tail.r, A.tail.p, A.tail.z,
                                                             PfOut{A.tail.pf, A.tail.vpos, A.tail.imgval}, A.tail.scal, cgsh);
            if (last && A.tail.implicit_hd) {
                __syncthreads();
                cgv_finish_implicit<1024 / (NWAVES * 64)>(A.tail.P, A.tail.max_kl, A.tail.x, A.tail.r, A.tail.gout + 1, A.tail.step, A.tail.scal, cgsh);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (last) {                                 // nobody reads the counters any more: leave them at zero for the next solve
                    __hip_atomic_store(A.bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(A.bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(A.bar + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else __hip_atomic_store(A.bar + 2, (unsigned int)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (it + 1 < n_it) {
            if (tid == 0) s_flag = cgp_wait(A.bar + 2, (unsigned int)(it + 1), A.timeout) ? 2u : 3u;
            __syncthreads();
            if (s_flag == 3u) { if (tid == 0) A.tail.scal[S_COMMERR] = 2.0; return; }
        }
        __syncthreads();
        CT_MARK(6)
        if (it + 1 < n_it) {
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the closer's tangent tables / float copy of p
            __syncthreads();
        }
        CT_MARK(7)
    }
    }                                                       // iterations
}

template <int NS, int NA, int PH, int MODE_>
__global__ void __launch_bounds__(NWAVES * 64, 1) k_policy_mfma(PolK k, const float* __restrict__ theta, const float* __restrict__ v,
                                                        float* __restrict__ partials) {
    pol_body<NS, NA, PH, MODE_>(k, theta, v, partials, nullptr);
}
template <int NS, int NA, int PH>
__global__ void __launch_bounds__(NWAVES * 64, 1) k_policy_cgp(PolK k, const float* __restrict__ theta, const float* __restrict__ v,
                                                       float* __restrict__ partials, CgpArgs cgp) {
    pol_body<NS, NA, PH, MODE_CGP>(k, theta, v, partials, &cgp);
}


// Host mirror of the image layout: element i of the LDS image <- theta[idx] / v[idx] / 0.
template <int NS, int NA, int PH>
static void pol_image_map(std::vector<int>& map) {
    using I = PolImg<NS, NA, PH>;
    constexpr int HB = I::HB;
    constexpr int pW0 = 0, pb0 = NS * PH, pW1 = pb0 + PH, pb1 = pW1 + PH * PH, pW2 = pb1 + PH;
    const int VFLAG = 0x40000000;
    map.assign(I::TOTAL, -1);
    for (int i = 0; i < I::TOTAL; ++i) {
        int m = -1;
        if (i < I::O_W2F) {                                  // [row][lane][cb] tables
            int t, base;
            if (i < I::O_W1F) { t = 0; base = I::O_W0F; } else if (i < I::O_V0F) { t = 1; base = I::O_W1F; }
            else if (i < I::O_V1F) { t = 2; base = I::O_V0F; } else if (i < I::O_W2B) { t = 3; base = I::O_V1F; }
            else if (i < I::O_W1B) { t = 4; base = I::O_W2B; } else { t = 5; base = I::O_W1B; }
            const int j = i - base, cb = j % HB, ln = (j / HB) & 63, row = j / (HB * 64), cc = ln & 15, qq = ln >> 4;
            const int flag = (t == 2 || t == 3) ? VFLAG : 0;
            if (t == 0 || t == 2) { const int in = 4 * row + qq, o = 16 * cb + cc; if (in < NS && o < PH) m = (pW0 + in * PH + o) | flag; }
            else if (t == 1 || t == 3) { const int in = 16 * (row >> 2) + 4 * qq + (row & 3), o = 16 * cb + cc; if (in < PH && o < PH) m = (pW1 + in * PH + o) | flag; }
            else if (t == 4) { const int ii = 16 * cb + cc, d = 4 * qq + row; if (ii < PH && d < NA) m = pW2 + ii * NA + d; }
            else { const int ii = 16 * cb + cc, jj = 16 * (row >> 2) + 4 * qq + (row & 3); if (ii < PH && jj < PH) m = pW1 + ii * PH + jj; }
        } else {                                             // [row][lane] tables: W2f, V2f
            const bool isv = i >= I::O_V2F;
            const int j = i - (isv ? I::O_V2F : I::O_W2F), ln = j & 63, row = j >> 6, cc = ln & 15, qq = ln >> 4;
            const int in = 16 * (row >> 2) + 4 * qq + (row & 3);
            if (in < PH && cc < NA) m = (pW2 + in * NA + cc) | (isv ? VFLAG : 0);
        }
        map[i] = m;
    }
}

// -------------------------------------------------------------------------------------------------
typedef void (*pol_kernel_t)(PolK, const float*, const float*, float*);
typedef void (*pol_cgp_t)(PolK, const float*, const float*, float*, CgpArgs);
struct PolEntry { int ns, na, ph; pol_kernel_t kern[4]; pol_cgp_t cgp; int lds_floats, lds_floats_eval; void (*build_map)(std::vector<int>&); };
template <int NS, int NA, int PH> constexpr int pol_lds() {
    constexpr int HB = cdiv_(PH, 16);
    constexpr int a = PolImg<NS, NA, PH>::TOTAL + NWAVES * (3 * HB + (NA <= 2 ? 0 : 1)) * 16 * 20;      // 20 = TS of the kernel's transpose tiles
    constexpr int P = NS * PH + PH + PH * PH + PH + PH * NA + NA + NA;
    constexpr int b = NWAVES * (P + PART_EXTRA);
    return a > b ? a : b;
}
template <int NS, int NA, int PH> constexpr int pol_lds_eval() {       // MODE_LOSSKL: weight image, then the epilogue's [NWAVES][ROW] rows
    constexpr int P = NS * PH + PH + PH * PH + PH + PH * NA + NA + NA;
    constexpr int a = PolImg<NS, NA, PH>::TOTAL, b = NWAVES * (P + PART_EXTRA);
    return a > b ? a : b;
}
#define PENTRY(NS, NA, PH) {NS, NA, PH, {k_policy_mfma<NS, NA, PH, 0>, k_policy_mfma<NS, NA, PH, 1>, k_policy_mfma<NS, NA, PH, 2>, k_policy_mfma<NS, NA, PH, 3>}, k_policy_cgp<NS, NA, PH>, pol_lds<NS, NA, PH>(), pol_lds_eval<NS, NA, PH>(), pol_image_map<NS, NA, PH>}
static const PolEntry kPol[] = {
    PENTRY(10, 2, 32),    // swimmer
    PENTRY(18, 6, 32),    // half-cheetah
    PENTRY(11, 3, 32),    // hopper
    PENTRY(14, 4, 32),    // snake
    PENTRY(29, 8, 32),    // ant
};

// returns the table index for this ctx or -1 (generic kernels in policy_update.hip)
int policy_mfma_select(const ProblemDesc& pd) {
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != pd.pol.dims[2]) return -1;
    for (int i = 0; i < (int)(sizeof(kPol) / sizeof(kPol[0])); ++i)
        if (kPol[i].ns == pd.ns && kPol[i].na == pd.na && kPol[i].ph == pd.pol.dims[1]) return i;
    return -1;
}

// gather map of ctx->pol_mfma's weight-fragment image, its inverse over the tangent entries (theta index -> image position) and the buffer
// the image VALUES of a CG solve are kept in (metrpo_ctx::d_pol_imgval); built on first use
int policy_mfma_image_buffers(metrpo_ctx* c) {
    const int idx = c->pol_mfma;
    if (idx < 0) return set_err(c, METRPO_EUNSUPPORTED, "no MFMA policy kernels for this shape");
    if (c->pol_img_idx == idx) return METRPO_OK;
    const PolEntry& en = kPol[idx];
    std::vector<int> map;
    en.build_map(map);
    std::vector<int> vpos((size_t)c->pd.P, -1);
    for (size_t i = 0; i < map.size(); ++i)
        if (map[i] >= 0 && (map[i] & 0x40000000)) {
            const int j = map[i] & 0x3FFFFFFF;
            if (j >= c->pd.P || vpos[j] != -1) return set_err(c, METRPO_EINVAL, "policy image map: tangent entry out of range or stored twice");
            vpos[j] = (int)i;
        }
    for (void** q : {(void**)&c->d_pol_img, (void**)&c->d_pol_vpos, (void**)&c->d_pol_imgval}) if (*q) { ws_retire(c, *q); *q = nullptr; }
    c->pol_img_idx = -1;
    HIP_TRY(c, ws_alloc(c, (void**)&c->d_pol_img, sizeof(int) * map.size()));
    HIP_TRY(c, hipMemcpy(c->d_pol_img, map.data(), sizeof(int) * map.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, ws_alloc(c, (void**)&c->d_pol_vpos, sizeof(int) * vpos.size()));
    HIP_TRY(c, hipMemcpy(c->d_pol_vpos, vpos.data(), sizeof(int) * vpos.size(), hipMemcpyHostToDevice));
    HIP_TRY(c, ws_alloc(c, (void**)&c->d_pol_imgval, sizeof(float) * map.size()));
    HIP_TRY(c, hipMemset(c->d_pol_imgval, 0, sizeof(float) * map.size()));
    c->pol_img_idx = idx;
    return METRPO_OK;
}

// launches mode `mode`; per-block rows of P+3 floats land in `partials`; returns the block count via *nblocks
int policy_mfma_launch(metrpo_ctx* c, int idx, int mode, const metrpo_batch* b, const float* theta, const float* v,
                       float* partials, int nblocks, hipStream_t st) {
    const PolEntry& en = kPol[idx];
    PolK k;
    k.obs = b->d_obs; k.act = b->d_act; k.adv = b->d_adv; k.old_mean = b->d_old_mean; k.old_ls = b->d_old_log_std;
    k.ls_stride = b->old_log_std_stride; k.valid = b->d_valid; k.N = b->N; k.inv_n = (float)b->inv_n_global;
    { const int rc = policy_mfma_image_buffers(c); if (rc) return rc; }
    k.img_map = (const int*)c->d_pol_img;
    k.gm = c->vjp_gm;
    k.skip = c->ls_skip;
    k.hcache = nullptr;
    k.imgval = (c->img_live && c->hcache_on && k.gm == nullptr && (mode == MODE_GRAD || mode == MODE_FVP)) ? c->d_pol_imgval : nullptr;
    if (c->hcache_on && (mode == MODE_GRAD || mode == MODE_FVP) && k.gm == nullptr) {      // set by run_trpo_update around one CG solve
        const size_t need = (size_t)((b->N + 15) / 16) * 2 * (size_t)cdiv_(en.ph, 16) * 64 * 4;
        if (need > c->hcache_cap) {
            if (c->d_hcache) { ws_retire(c, c->d_hcache); c->d_hcache = nullptr; c->hcache_cap = 0; }
            HIP_TRY(c, ws_alloc(c, (void**)&c->d_hcache, need * sizeof(float)));
            c->hcache_cap = need;
        }
        k.hcache = c->d_hcache;
        if (mode == MODE_FVP) mode = MODE_FVPC;
    }
    // loss + KL evaluation (line search): no transpose tiles, 128 VGPRs -> two blocks fit a CU (run_mode launches 2 x n_sm of them)
    size_t sh = sizeof(float) * (size_t)en.lds_floats;
    if (mode == MODE_LOSSKL) sh = sizeof(float) * (size_t)en.lds_floats_eval;
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)en.kern[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(en.kern[mode], dim3(nblocks), dim3(NWAVES * 64), sh, st, k, theta, v, partials);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

// Whole CG solve in one launch (MODE_CGP): n_it Fisher-vector products on the cached activations + the vector steps between them.  Requires what the
// fused launch-per-product solve has set up (run_trpo_update: hcache_on, img_live: the gradient kernel of this theta has written the activation cache
// and published the weight image, its tail the first tangent entries) and a grid the device holds at once.  METRPO_EUNSUPPORTED: take the per-launch path.
int policy_mfma_cg_persist(metrpo_ctx* c, int idx, const metrpo_batch* b, const float* theta, const CgTail& tl, int n_it, float* partials, int nblocks,
                           hipStream_t st) {
    const PolEntry& en = kPol[idx];
    if (!c->img_live || !c->hcache_on || c->d_hcache == nullptr || c->d_pol_imgval == nullptr || tl.vpos == nullptr) return METRPO_EUNSUPPORTED;
    if (tl.P > CGP_RMAX * 1024 || n_it < 1 || c->cgp_failed) return METRPO_EUNSUPPORTED;
    const size_t sh = sizeof(float) * (size_t)en.lds_floats;
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)en.cgp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    if (!grid_is_coresident(c, (const void*)en.cgp, NWAVES * 64, sh, nblocks, st)) return METRPO_EUNSUPPORTED;
    if (!c->d_cgp_bar) {
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_cgp_bar, 64));
        HIP_TRY(c, hipMemsetAsync(c->d_cgp_bar, 0, 64, st));      // the closing block of a solve leaves the counters at zero again
    }
    PolK k;
    k.obs = b->d_obs; k.act = b->d_act; k.adv = b->d_adv; k.old_mean = b->d_old_mean; k.old_ls = b->d_old_log_std;
    k.ls_stride = b->old_log_std_stride; k.valid = b->d_valid; k.N = b->N; k.inv_n = (float)b->inv_n_global;
    k.img_map = (const int*)c->d_pol_img; k.gm = nullptr; k.skip = nullptr; k.hcache = c->d_hcache; k.imgval = c->d_pol_imgval;
    CgpArgs a;
    a.tail = tl; a.n_it = n_it; a.n_params = c->pd.pol.n_params; a.bar = c->d_cgp_bar; a.theta_ls = theta + c->pd.pol.n_params; a.partials = partials;
    a.timeout = 200000000ull;                                 // 2 s
    hipLaunchKernelGGL(en.cgp, dim3(nblocks), dim3(NWAVES * 64), sh, st, k, theta, (const float*)tl.pf, partials, a);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
