// Fisher-vector product of the 2x32 tanh policies from CACHED activations, second generation: one wave per SIMD, explicit two-stage
// software pipeline (the Hessian-vector products of [rllab] PerlmutterHvp as called by krylov.cg from ConjugateGradientOptimizer.optimize,
// reached from algos/npo.py:111; Gauss-Newton form, DESIGN.md "FVP form").
//
// k_policy_mfma<MODE_FVPC> (policy_mfma.hip) runs two 240-register waves per SIMD, each a serial stream
//   tangent (38 MFMAs) -> output layer + back-prop on the VALU -> 16 MFMAs -> LDS transposes -> 24 sample-contracted MFMAs
// and relies on the two streams overlapping by themselves: the matrix pipe is 51 % busy, 9 400 cycles per pair of tiles against 5 000 of
// MFMA issue.  Here a workgroup is FOUR waves (one per SIMD, the whole 512-register file each) and the overlap is written down:
//   * every weight fragment the tile loop needs lives in registers (54 MFMA operands + the per-lane output-layer rows): no LDS weight
//     reads at all (27 ds_read_b64 per tile before, 8 us of a 62 us launch);
//   * iteration i runs  block A = tangent MFMAs of tile i, with the VALU stage of tile i-1 (tanh', output layer, mean-adjoint,
//     back-prop to layer 1, its LDS transpose writes) placed in their issue gaps (<= 5 single-issue instructions per 32-cycle MFMA,
//     MI355X_MICROARCH.md), then  block B = back-prop to layer 0 (16 MFMAs, swapped operands) and the 24 sample-contracted
//     weight-gradient MFMAs of tile i-1, whose transposed operands were written a whole block earlier;
//   * HBM inputs of tile i+1 (cached activations, observations in both layouts, valid flag) are requested at the top of iteration i.
// Arithmetic, accumulation order per wave and the partial-row layout are those of the first-generation kernel, so k_finalize and the
// CG tail are unchanged; per-tile results are bitwise those of k_policy_mfma<MODE_FVPC>, only the assignment of tiles to waves
// (4 waves per block instead of 8) and hence the float32 summation grouping differs.
#include <type_traits>
#include "device_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define PART_EXTRA 3
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#ifndef FVPC_SKIP
#define FVPC_SKIP 0             // developer experiments (results meaningless, only the clock is read): 2 = no prefetch in the steady state, 4 = no VALU stage, 8 = gaps not pinned, 16 = sched_group_barrier recipes
#endif

// Developer instrumentation (SRC=policy_fvpc.hip tools/build_variant.sh ftiming -DFVPC_TIMING): cycle counter of wave 0 of
// workgroup 0 at [0] kernel entry, [1] weights loaded / first tile waited for, [2] after the prologue block, then per pipeline step
// (up to 40) after block A and after block B, [3] after the tail, [4] after the epilogue; read with metrpo_debug_fvpc_phases
// (tools/fvpc_phases.py).  Not in the shipped library.
#ifdef FVPC_TIMING
__device__ unsigned long long g_fvpc_phase[8 + 2 * 40];
#define FT(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_fvpc_phase[(i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t metrpo_debug_fvpc_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fvpc_phase), sizeof(g_fvpc_phase)) == hipSuccess ? 0 : -1; }
#else
#define FT(i) do { } while (0)
#endif

namespace {
constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL> __device__ __forceinline__ float dpp_add_(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum_c(float v) {      // over the 16 lanes c of a row (xor 1, 2, 4, 8 butterfly on the DPP path)
    v = dpp_add_<0xB1>(v); v = dpp_add_<0x4E>(v); v = dpp_add_<0x141>(v); v = dpp_add_<0x140>(v);
    return v;
}
// sum over the four q-lanes (lanes c, c + 16, c + 32, c + 48) on the VALU: gfx950's row / half swaps instead of two dependent
// ds_bpermute round trips, each of which parks the wave -- and with it the MFMA stream -- at an s_waitcnt.  The two operands must be
// distinct registers (given the same value twice the compiler allocates ONE register and the swap degenerates), hence the opaque copy.
// Same association as v += xor16(v); v += xor32(v).
__device__ __forceinline__ float sum_q16(float v) {       // v + v of the lane 16 away (rows 0<->1, 2<->3)
    unsigned a = __float_as_uint(v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);       // {rows 0 0 2 2, rows 1 1 3 3}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_q32(float v) {       // v + v of the lane 32 away
    unsigned a = __float_as_uint(v), b = a;
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);       // {lower half twice, upper half twice}
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_q(float v) { return sum_q32(sum_q16(v)); }
__device__ __forceinline__ float sum_q_lds(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
}  // namespace

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
#define PIN() do { if (!(FVPC_SKIP & 1)) __builtin_amdgcn_sched_barrier(0); } while (0)

template <int NS, int NA, int PH>
__global__ void __launch_bounds__(256, 1) k_fvpc_pipe(PolK k, const float* __restrict__ theta, const float* __restrict__ v,
                                                      float* __restrict__ partials) {
    static_assert(PH == 32 && NA == 2, "output layer on the VALU, written out for two action dims; two 16-unit column blocks");
    constexpr int NS_KS = cdiv_c(NS, 4), NSI = cdiv_c(NS, 16), HB = 2, KK = 8;
    static_assert(NSI == 1, "observation fits one 16-feature block");
    constexpr int pW0 = 0, pb0 = NS * PH, pW1 = pb0 + PH, pb1 = pW1 + PH * PH, pW2 = pb1 + PH, pb2 = pW2 + PH * NA,
                  pLS = pb2 + NA, P = pLS + NA, ROW = P + PART_EXTRA;
    constexpr int TS = 20, TILE = 16 * TS;                 // transpose tile T[unit][sample], 16-byte aligned rows
    constexpr int WTL = 3 * HB * TILE;                     // per wave: h0 (two parities) and d1
    constexpr int LDSF = (4 * WTL > 4 * ROW) ? 4 * WTL : 4 * ROW;
    constexpr int NMA = NS_KS * HB + 2 * KK * HB;          // MFMAs of block A (38 for swimmer)
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    FT(0);
    float* TL = lds + wave * WTL;
    const long long N = k.N, ntiles = (N + 15) / 16;
    const f32x4 Z4 = {0.f, 0.f, 0.f, 0.f};
    const bool has_valid = k.valid != nullptr;

    // ---------------- tile inputs: buffer loads through per-tile descriptors ------------------------------------------------------
    // Every tile gets its own resource descriptors (scalar arithmetic only): base = the tile's first sample, size = what is left of
    // the batch.  Rows beyond the batch and the feature padding (offset 2^30) are then out of range and read as 0 -- no clamps, no
    // masks, no per-tile address arithmetic on the VALU; the per-lane offsets are loop invariants.
    struct TileIn { float xB[NS_KS]; float xTs[4]; int vld; f32x4 h[2 * HB]; };
    constexpr unsigned OOB = 0x40000000u;
    // layer-0 k-step s contracts, in k-row q, feature NS_KS q + s: a lane's NS_KS features are then CONTIGUOUS (one 12-byte load instead
    // of three; every VMEM issue costs the in-order wave 60-100 cycles here).  Slots beyond the row (features >= NS) read the next
    // sample's first values or, at the end of the batch, 0: they meet zero weight rows.
    static_assert(NS_KS == 3 && NS < 4 * NS_KS, "observation row loaded as one dwordx3; one spare feature slot carries the layer-0 bias");
    unsigned offT[4];
    const unsigned offB = (unsigned)(c * NS + NS_KS * q) * 4u;
#pragma unroll
    for (int s = 0; s < 4; ++s) offT[s] = (c < NS) ? (unsigned)((4 * q + s) * NS + c) * 4u : OOB;
    const unsigned offH = (unsigned)lane * 16u;
    struct TileSrc { __amdgpu_buffer_rsrc_t ro, rv, rh; };
    auto tile_src = [&](long long tile) {
        const long long n0 = tile * 16, left = (N > n0) ? N - n0 : 0;
        const long long lim = (left < 16) ? left : 16;
        TileSrc t;
        t.ro = __builtin_amdgcn_make_buffer_rsrc((void*)(k.obs + n0 * NS), 0, (int)(lim * NS * 4), 0x00020000);
        t.rv = __builtin_amdgcn_make_buffer_rsrc((void*)(k.valid + (has_valid ? n0 : 0)), 0, has_valid ? (int)lim : 0, 0x00020000);
        t.rh = __builtin_amdgcn_make_buffer_rsrc((void*)(k.hcache + tile * (2 * HB * 64 * 4)), 0, (tile < ntiles) ? 2 * HB * 64 * 16 : 0, 0x00020000);
        return t;
    };
    constexpr int NLOAD = 2 * HB + 1 + 4 + 1;               // loads of one tile: cached activations, observations (both layouts), valid flag
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    auto fetch_one = [&](auto i_, const TileSrc& t, TileIn& in) {
        constexpr int i = decltype(i_)::value;
        if constexpr (i < 2 * HB) in.h[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t.rh, offH + i * 1024, 0, 0));
        else if constexpr (i == 2 * HB) {
            const f32x3 x3 = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(t.ro, offB, 0, 0));
            in.xB[0] = x3[0]; in.xB[1] = x3[1]; in.xB[2] = x3[2];           // slot NS is set to 1 where it is consumed (one_slot)
        }
        else if constexpr (i < 2 * HB + 1 + 4) in.xTs[i - 2 * HB - 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(t.ro, offT[i - 2 * HB - 1], 0, 0));
        else in.vld = (int)__builtin_amdgcn_raw_buffer_load_b8(t.rv, (unsigned)c, 0, 0);
    };
    auto fetch = [&](long long tile, TileIn& in) {
        const TileSrc t = tile_src(tile);
        static_for<0, NLOAD>([&](auto i_) { fetch_one(i_, t, in); });
    };
    // 1 for lanes whose sample exists and is valid (folds the batch bound and the valid flag), else 0
    auto okf = [&](long long tile, const TileIn& in) {
        const long long left = N - tile * 16;
        const int nrem = (int)((left < 16) ? ((left > 0) ? left : 0) : 16);
        return (c < nrem && (!has_valid || in.vld != 0)) ? 1.f : 0.f;
    };

    // tiles of this wave: gw, gw + W, gw + 2W, ...
    const long long gw = (long long)blockIdx.x * 4 + wave, W = (long long)gridDim.x * 4;
    const long long ntw = (gw < ntiles) ? (ntiles - gw + W - 1) / W : 0;
    TileIn in[3];                                            // rotation: tile j of the wave lives in in[j % 3]
    fetch(gw, in[0]);
    fetch(gw + W, in[1]);

    // ---------------- register-resident weights -----------------------------------------------------------------------------
    float V0F[NS_KS][HB], V1F[KK][HB], W1F[KK][HB], W1B[KK][HB];
#pragma unroll
    for (int s = 0; s < NS_KS; ++s)
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {                   // slot NS = the constant-1 input: its weight row is the layer-0 bias of the direction
            const int fi = NS_KS * q + s;
            V0F[s][cb] = (fi < NS) ? v[pW0 + fi * PH + 16 * cb + c] : (fi == NS) ? v[pb0 + 16 * cb + c] : 0.f;
        }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
            const int u = 16 * (kk >> 2) + 4 * q + (kk & 3);            // the unit k-step kk contracts in this lane's k-row
            V1F[kk][cb] = v[pW1 + u * PH + 16 * cb + c];
            W1F[kk][cb] = theta[pW1 + u * PH + 16 * cb + c];
            W1B[kk][cb] = theta[pW1 + (16 * cb + c) * PH + u];
        }
    f32x4 vb1f[HB];
    float w2l[8][NA], v2l[8][NA], gw2l[8][NA], vb2l[NA], fisher_d[NA];       // element e = 4 cb + r  <->  unit 16 cb + 4 q + r
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int u = 16 * (e >> 2) + 4 * q + (e & 3);
        vb1f[e >> 2][e & 3] = v[pb1 + u];
#pragma unroll
        for (int d = 0; d < NA; ++d) { w2l[e][d] = theta[pW2 + u * NA + d]; v2l[e][d] = v[pW2 + u * NA + d]; gw2l[e][d] = 0.f; }
    }
#pragma unroll
    for (int d = 0; d < NA; ++d) {
        vb2l[d] = v[pb2 + d];
        fisher_d[d] = k.inv_n / (expf(2.f * fmaxf(theta[pLS + d], LOG_MIN_STD)) + 0.5f * KL_EPS);   // inv_n x d2 KL / d mean^2 = inv_n / (s^2 + eps/2)
    }
    const float q0f = (q == 0) ? 1.f : 0.f;

    // ---------------- accumulators --------------------------------------------------------------------------------------------
    f32x4 gW0[HB], gW1[HB][HB], gb0[HB], gb1[HB];
    float gb2[NA], accw = 0.f;
#pragma unroll
    for (int a = 0; a < HB; ++a) {
        gb0[a] = Z4; gb1[a] = Z4; gW0[a] = Z4;
#pragma unroll
        for (int b = 0; b < HB; ++b) gW1[a][b] = Z4;
    }
#pragma unroll
    for (int d = 0; d < NA; ++d) gb2[d] = 0.f;
    f32x4 t1s[3][HB];                                        // layer-1 tangent pre-activation of tile j in t1s[j % 3]
    float ual[NA] = {0.f, 0.f};                              // mean-adjoint of the tile whose VALU stage ran last (block B -> next block A)
    using std::integral_constant;
    using Yes = integral_constant<bool, true>;
    using No = integral_constant<bool, false>;
// The weight-gradient accumulators live for the whole kernel and no VALU instruction touches them before the epilogue: they are kept in
// the ACCUMULATION half of the register file by issuing their MFMAs as inline assembly ("+a").  The rest of the file is compiled with
// -amdgpu-mfma-vgpr-form (chain accumulators in vector registers: the VALU reads them, and every v_accvgpr move queues behind the MFMA
// in flight), which would otherwise also claim 24 vector registers for these and spill elsewhere.  An accumulator is re-used every
// fourth product (>= 128 cycles later) and its operands were written several gaps earlier: no hazard for the recogniser to miss.
#define MFMA16_ACC(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define TIE(...) do { if (!(FVPC_SKIP & 8)) asm volatile("" : __VA_ARGS__ :: "memory"); } while (0)

    // Per tile:  M1 (38 tangent MFMAs) -> V1a (tanh' of layer 1, tangent of the mean, q-lane sums, mean-adjoint) -> V1b (back-prop to
    // layer 1, output-layer weight gradient, transpose writes) -> M2 (16 MFMAs: back-prop to layer 0, swapped operands) -> V2 (tanh' of
    // layer 0) -> M3 (24 sample-contracted MFMAs).  A single in-order wave hides VALU / LDS / VMEM issues only in the shadow of its own
    // MFMAs, and only ~2-3 per 32-cycle MFMA (a dependent VALU instruction costs ~10 cycles here, a VMEM issue 60-100), so the VALU
    // stages are spread over BOTH MFMA blocks of the pipeline step:
    //   block A (step j) = M1(j)            with  loads(j + 1), tanh'(layer 0) of tile j, V1b(j - 1)
    //   block B (step j) = M2, M3 (j - 1)   with  V2(j - 1), V1a(j)
    // Every gap ends in an empty asm that "modifies" the accumulators and running values (TIE): the compiler must finish the gap's work
    // above it and start the next gap's below it; inside a gap it is free.  (sched_group_barrier recipes and sched_barrier pins did not
    // survive instruction selection's own ordering here.)
    // The VALU stages are written as lists of single instructions ("ops", each followed by an empty asm naming its result, which fixes
    // its place among the MFMAs' pins) and dealt out evenly over the gaps of a block.  Within a list, neighbouring ops are independent
    // (four partial chains): a dependent VALU instruction costs the lone wave ~10 cycles, an independent one its issue slot.
#define PINV(x) do { if (!(FVPC_SKIP & 8)) asm volatile("" : "+v"(x)); } while (0)
    float gq[4] = {0.f, 0.f, 0.f, 0.f}, gp[4] = {0.f, 0.f, 0.f, 0.f}, avp[NA][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, av[NA] = {0.f, 0.f};
    // ---- "own" list of tile C (after its layer-0 MFMAs): tanh' of layer 0 on the tangent; 8 VALU ops per column block
    constexpr int N_OWN = 16;
    auto own_op = [&](auto i_, TileIn& ci, f32x4 (&t0)[HB]) {
        constexpr int i = decltype(i_)::value, cb = i / 8, ph = (i % 8) / 4, r = i % 4;
        if constexpr (ph == 0) { PINV(ci.h[cb][r]); gq[r] = fmaf(-ci.h[cb][r], ci.h[cb][r], 1.f); PINV(gq[r]); }
        else { PINV(gq[r]); t0[cb][r] *= gq[r]; PINV(t0[cb][r]); }
    };
    // ---- V1b list of tile P: d1 = (W2 u)(1 - h1^2), output-layer weight gradient, layer-1 bias gradient; 28 VALU ops per column block
    constexpr int N_V1B = 56;
    auto v1b_op = [&](auto i_, TileIn& pi, f32x4 (&d1)[HB]) {
        constexpr int i = decltype(i_)::value, cb = i / 28, ph = (i % 28) / 4, r = i % 4, e = 4 * cb + r;
        if constexpr (ph == 0) { PINV(pi.h[HB + cb][r]); d1[cb][r] = w2l[e][0] * ual[0]; PINV(d1[cb][r]); }
        if constexpr (ph == 1) { PINV(pi.h[HB + cb][r]); gp[r] = fmaf(-pi.h[HB + cb][r], pi.h[HB + cb][r], 1.f); PINV(gp[r]); }
        if constexpr (ph == 2) { PINV(d1[cb][r]); d1[cb][r] = fmaf(w2l[e][1], ual[1], d1[cb][r]); PINV(d1[cb][r]); }
        if constexpr (ph == 3) { PINV(gw2l[e][0]); gw2l[e][0] = fmaf(pi.h[HB + cb][r], ual[0], gw2l[e][0]); PINV(gw2l[e][0]); }
        if constexpr (ph == 4) { PINV(gp[r]); d1[cb][r] *= gp[r]; PINV(d1[cb][r]); }
        if constexpr (ph == 5) { PINV(gw2l[e][1]); gw2l[e][1] = fmaf(pi.h[HB + cb][r], ual[1], gw2l[e][1]); PINV(gw2l[e][1]); }
        if constexpr (ph == 6) { PINV(gb1[cb][r]); gb1[cb][r] += d1[cb][r]; PINV(gb1[cb][r]); }
    };
    // ---- V2 list of tile P (after its back-prop MFMAs): tanh' of layer 0 from the transposed h0, bias gradient; 12 ops per column block
    constexpr int N_V2 = 24;
    auto v2_op = [&](auto i_, f32x4 (&a0v)[HB], f32x4 (&d0n)[HB]) {
        constexpr int i = decltype(i_)::value, cb = i / 12, ph = (i % 12) / 4, r = i % 4;
        if constexpr (ph == 0) { PINV(a0v[cb][r]); gq[r] = fmaf(-a0v[cb][r], a0v[cb][r], 1.f); PINV(gq[r]); }
        if constexpr (ph == 1) { PINV(gq[r]); d0n[cb][r] *= gq[r]; PINV(d0n[cb][r]); }
        if constexpr (ph == 2) { PINV(gb0[cb][r]); gb0[cb][r] += d0n[cb][r]; PINV(gb0[cb][r]); }
    };
    // ---- V1a list of tile C (after its tangent MFMAs): tanh' of layer 1, tangent of the mean into four partial sums per action dim,
    //      fold, q-lane sums, mean-adjoint
    constexpr int N_V1A = 2 * 28 + 6 + 4 + 5;
    auto v1a_op = [&](auto i_, TileIn& ci, const f32x4 (&t1)[HB], float okc) {
        constexpr int i = decltype(i_)::value;
        if constexpr (i < 56) {
            constexpr int cb = i / 28, ph = (i % 28) / 4, r = i % 4, e = 4 * cb + r;
            if constexpr (ph == 0) { PINV(ci.h[HB + cb][r]); gp[r] = fmaf(-ci.h[HB + cb][r], ci.h[HB + cb][r], 1.f); PINV(gp[r]); }
            if constexpr (ph == 1) { PINV(ci.h[HB + cb][r]); gq[r] = t1[cb][r] + vb1f[cb][r]; PINV(gq[r]); }
            if constexpr (ph == 2) { PINV(avp[0][r]); avp[0][r] = fmaf(v2l[e][0], ci.h[HB + cb][r], avp[0][r]); PINV(avp[0][r]); }
            if constexpr (ph == 3) { PINV(gq[r]); gp[r] *= gq[r]; PINV(gp[r]); }
            if constexpr (ph == 4) { PINV(avp[1][r]); avp[1][r] = fmaf(v2l[e][1], ci.h[HB + cb][r], avp[1][r]); PINV(avp[1][r]); }
            if constexpr (ph == 5) { PINV(avp[0][r]); avp[0][r] = fmaf(w2l[e][0], gp[r], avp[0][r]); PINV(avp[0][r]); }
            if constexpr (ph == 6) { PINV(avp[1][r]); avp[1][r] = fmaf(w2l[e][1], gp[r], avp[1][r]); PINV(avp[1][r]); }
        } else if constexpr (i < 62) {
            constexpr int k_ = i - 56;                       // fold the four partial sums: (0 + 1), (2 + 3) for both dims, then the pair
            if constexpr (k_ == 0) { PINV(avp[0][0]); avp[0][0] += avp[0][1]; PINV(avp[0][0]); }
            if constexpr (k_ == 1) { PINV(avp[1][0]); avp[1][0] += avp[1][1]; PINV(avp[1][0]); }
            if constexpr (k_ == 2) { PINV(avp[0][2]); avp[0][2] += avp[0][3]; PINV(avp[0][2]); }
            if constexpr (k_ == 3) { PINV(avp[1][2]); avp[1][2] += avp[1][3]; PINV(avp[1][2]); }
            if constexpr (k_ == 4) { PINV(avp[0][0]); av[0] = avp[0][0] + avp[0][2]; PINV(av[0]); }
            if constexpr (k_ == 5) { PINV(avp[1][0]); av[1] = avp[1][0] + avp[1][2]; PINV(av[1]); }
        } else if constexpr (i < 66) {
            constexpr int k_ = i - 62;
            if constexpr (k_ == 0) { PINV(av[0]); av[0] = sum_q16(av[0]); PINV(av[0]); }
            if constexpr (k_ == 1) { PINV(av[1]); av[1] = sum_q16(av[1]); PINV(av[1]); }
            if constexpr (k_ == 2) { PINV(av[0]); av[0] = sum_q32(av[0]); PINV(av[0]); }
            if constexpr (k_ == 3) { PINV(av[1]); av[1] = sum_q32(av[1]); PINV(av[1]); }
        } else {
            constexpr int k_ = i - 66;
            if constexpr (k_ == 0) { PINV(av[0]); ual[0] = (av[0] + vb2l[0]) * (fisher_d[0] * okc); PINV(ual[0]); }
            if constexpr (k_ == 1) { PINV(av[1]); ual[1] = (av[1] + vb2l[1]) * (fisher_d[1] * okc); PINV(ual[1]); }
            if constexpr (k_ == 2) { PINV(gb2[0]); gb2[0] = fmaf(q0f, ual[0], gb2[0]); PINV(gb2[0]); }
            if constexpr (k_ == 3) { PINV(gb2[1]); gb2[1] = fmaf(q0f, ual[1], gb2[1]); PINV(gb2[1]); }
            if constexpr (k_ == 4) {
                PINV(accw); accw = fmaf(q0f * okc, k.inv_n, accw); PINV(accw);
#pragma unroll
                for (int d = 0; d < NA; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) avp[d][r] = 0.f;
            }
        }
    };
    // ops [first(g), first(g + 1)) of an n-op list run in gap g of `gaps` gaps starting at `g0`
    auto deal = [](int n, int gaps, int g) { return (g <= 0) ? 0 : (g >= gaps) ? n : (n * g + gaps - 1) / gaps; };

    auto block_a = [&](auto rc_, auto cur_, auto prev_, auto next_, long long tile, float* T_H0c, float* T_D1, f32x4 (&d1)[HB]) {
        constexpr int RC = decltype(rc_)::value, RP = (RC + 2) % 3, RN = (RC + 1) % 3;
        constexpr bool CUR = decltype(cur_)::value, PREV = decltype(prev_)::value, NEXT = decltype(next_)::value;
        TileIn& ci = in[RC];
        TileIn& pi = in[RP];
        f32x4 t0[HB];
        f32x4 (&t1)[HB] = t1s[RC];
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) { t0[cb] = Z4; t1[cb] = Z4; d1[cb] = Z4; }      // biases: layer 0 rides in the contraction, layer 1 is added on the VALU
        const TileSrc nsrc = tile_src(tile + W);
        constexpr int F0 = NS_KS * HB + 2, FG = 12;         // "own" list: FG gaps from F0 (the layer-0 MFMAs have landed; done before the W1^T t0 products start)
        constexpr int G3 = 0, G3N = NMA - 1;                 // V1b list: all gaps but the last (its transpose writes trail by one gap)
        static_assert(F0 + FG <= NS_KS * HB + KK * HB, "t0 scaled before it is consumed");
        static_for<0, NMA>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (CUR) {
                if constexpr (g < NS_KS * HB) {
                    constexpr int s = g / HB, cb = g % HB;
                    if constexpr (cb == 0 && s == NS % NS_KS) ci.xB[s] = (q == NS / NS_KS) ? 1.f : ci.xB[s];      // the constant-1 input of the bias row
                    t0[cb] = MFMA16(V0F[s][cb], ci.xB[s], t0[cb]);
                }
                else if constexpr (g < NS_KS * HB + KK * HB) { constexpr int i = g - NS_KS * HB, kk = i / HB, cb = i % HB; t1[cb] = MFMA16(V1F[kk][cb], ci.h[kk >> 2][kk & 3], t1[cb]); }
                else { constexpr int i = g - NS_KS * HB - KK * HB, kk = i / HB, cb = i % HB; t1[cb] = MFMA16(W1F[kk][cb], t0[kk >> 2][kk & 3], t1[cb]); }
                TIE("+v"(t0[0]), "+v"(t0[1]), "+v"(t1[0]), "+v"(t1[1]));
                if constexpr (NEXT && g < NLOAD) fetch_one(g_, nsrc, in[RN]);      // one prefetch load of the next tile per gap
                if constexpr (g >= F0 && g < F0 + FG) {
                    constexpr int lo = (N_OWN * (g - F0) + FG - 1) / FG, hi = (N_OWN * (g - F0 + 1) + FG - 1) / FG;
                    static_for<lo, hi>([&](auto i_) { own_op(i_, ci, t0); });
                }
                if constexpr (g >= F0 + 4 && g < F0 + 12) {  // the eight transpose writes of h0, one per gap
                    constexpr int e = g - F0 - 4, cb = e >> 2, r = e & 3;
                    T_H0c[cb * TILE + (4 * q + r) * TS + c] = ci.h[cb][r];
                }
            }
            if constexpr (PREV) {
                if constexpr (g >= G3 && g < G3 + G3N) {
                    constexpr int lo = (N_V1B * (g - G3) + G3N - 1) / G3N, hi = (N_V1B * (g - G3 + 1) + G3N - 1) / G3N;
                    static_for<lo, hi>([&](auto i_) { v1b_op(i_, pi, d1); });
                    // transpose write of an element once its product with tanh' (op 16 + r of its column block) has been issued
                    static_for<lo, hi>([&](auto i_) {
                        constexpr int i = decltype(i_)::value, cb = i / 28, ph = (i % 28) / 4, r = i % 4;
                        if constexpr (ph == 4) T_D1[cb * TILE + (4 * q + r) * TS + c] = d1[cb][r];
                    });
                }
            }
            asm volatile("" ::: "memory");
        });
    };
    auto block_b = [&](auto rc_, auto prevm_, auto curv_, long long tile, f32x4 (&d1)[HB], const float* T_H0p, const float* T_D1) {
        constexpr int RC = decltype(rc_)::value, RP = (RC + 2) % 3;
        constexpr bool PREVM = decltype(prevm_)::value, CURV = decltype(curv_)::value;
        TileIn& ci = in[RC];
        TileIn& pi = in[RP];
        f32x4 (&t1)[HB] = t1s[RC];
        f32x4 a0v[HB], b1v[HB], d0n[HB];
        float okc = 0.f;
        if (CURV) okc = okf(tile, ci);
        lds_order();
        if (PREVM) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) {
                a0v[cb] = *(const f32x4*)&T_H0p[cb * TILE + c * TS + 4 * q];
                b1v[cb] = *(const f32x4*)&T_D1[cb * TILE + c * TS + 4 * q];
                d0n[cb] = Z4;
            }
        }
        constexpr int NMB = KK * HB + 4 * HB * HB + 4 * HB;  // 40 MFMAs
        constexpr int V2G = KK * HB + 2, V2N = 12;           // V2 list: 12 gaps behind the back-prop MFMAs, done before the layer-0 products start
        static_assert(V2G + V2N <= KK * HB + 4 * HB * HB, "d0 scaled before it is consumed");
        static_for<0, NMB>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            if constexpr (PREVM) {
                if constexpr (g < KK * HB) { constexpr int kk = g / HB, cb = g % HB; d0n[cb] = MFMA16(d1[kk >> 2][kk & 3], W1B[kk][cb], d0n[cb]); TIE("+v"(d0n[0]), "+v"(d0n[1])); }
                else if constexpr (g < KK * HB + 4 * HB * HB) { constexpr int i = g - KK * HB, s = i / (HB * HB), ci_ = (i / HB) % HB, cj = i % HB; MFMA16_ACC(gW1[ci_][cj], a0v[ci_][s], b1v[cj][s]); }
                else { constexpr int i = g - KK * HB - 4 * HB * HB, s = i / HB, cj = i % HB; MFMA16_ACC(gW0[cj], pi.xTs[s], d0n[cj][s]); }
                if constexpr (g >= V2G && g < V2G + V2N) {
                    constexpr int lo = (N_V2 * (g - V2G) + V2N - 1) / V2N, hi = (N_V2 * (g - V2G + 1) + V2N - 1) / V2N;
                    static_for<lo, hi>([&](auto i_) { v2_op(i_, a0v, d0n); });
                }
            }
            if constexpr (CURV) {
                constexpr int lo = (N_V1A * g + NMB - 1) / NMB, hi = (N_V1A * (g + 1) + NMB - 1) / NMB;
                static_for<lo, hi>([&](auto i_) { v1a_op(i_, ci, t1, okc); });
            }
            asm volatile("" ::: "memory");
        });
        lds_order();
    };
#undef PINV
    float* const T_D1 = TL + 2 * HB * TILE;
    auto h0_tile = [&](long long j) { return TL + (int)(j & 1) * HB * TILE; };
    // one pipeline step for wave-tile j (rotation slot R = j % 3)
    auto step = [&](auto r_, long long j) {
        f32x4 d1[HB];
        block_a(r_, Yes{}, Yes{}, integral_constant<bool, !(FVPC_SKIP & 2)>{}, gw + j * W, h0_tile(j), T_D1, d1);
        if (j <= 40) FT(8 + 2 * (j - 1));
        block_b(r_, Yes{}, integral_constant<bool, !(FVPC_SKIP & 4)>{}, gw + j * W, d1, h0_tile(j - 1), T_D1);
        if (j <= 40) FT(8 + 2 * (j - 1) + 1);
    };
    auto tail = [&](auto r_, long long j) {                  // wave-tile j = ntw does not exist: V1b, M2, V2, M3 of the last tile
        f32x4 d1[HB];
        block_a(r_, No{}, Yes{}, No{}, gw + j * W, h0_tile(j), T_D1, d1);
        block_b(r_, Yes{}, No{}, gw + j * W, d1, h0_tile(j - 1), T_D1);
    };

    if (ntw > 0) {
        {   // prologue: M1 and V1a of the first tile (its successor was requested above)
            f32x4 d1[HB];
            __builtin_amdgcn_s_waitcnt(0x0F70);
            FT(1);
            block_a(integral_constant<int, 0>{}, Yes{}, No{}, No{}, gw, h0_tile(0), T_D1, d1);
            block_b(integral_constant<int, 0>{}, No{}, Yes{}, gw, d1, h0_tile(0), T_D1);
            FT(2);
        }
        for (long long j = 1; j < ntw; j += 3) {
            step(integral_constant<int, 1>{}, j);
            if (j + 1 < ntw) step(integral_constant<int, 2>{}, j + 1);
            if (j + 2 < ntw) step(integral_constant<int, 0>{}, j + 2);
        }
        const int rl = (int)(ntw % 3);
        if (rl == 0) tail(integral_constant<int, 0>{}, ntw);
        else if (rl == 1) tail(integral_constant<int, 1>{}, ntw);
        else tail(integral_constant<int, 2>{}, ntw);
        FT(3);
    }
#undef TIE

    // ---------------- wave partial -> block partial (fixed order) -> global row: the layout of k_policy_mfma's epilogue -----------
    __syncthreads();
    float* RB = lds;                                         // [4][ROW]
    float* row = RB + wave * ROW;
    if (lane < NA) row[pLS + lane] = 0.f;                    // log_std rows of the product are formed by k_finalize
    lds_order();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int cj = 0; cj < HB; ++cj) {
            const int j = 16 * cj + c;
            { const int i = 4 * q + r; if (i < NS) row[pW0 + i * PH + j] = gW0[cj][r]; }
#pragma unroll
            for (int ci = 0; ci < HB; ++ci) { const int i = 16 * ci + 4 * q + r; row[pW1 + i * PH + j] = gW1[ci][cj][r]; }
        }
#pragma unroll
        for (int ci = 0; ci < HB; ++ci)
#pragma unroll
            for (int d = 0; d < NA; ++d) {
                const float sg = sum_c(gw2l[4 * ci + r][d]);
                if (c == 0) row[pW2 + (16 * ci + 4 * q + r) * NA + d] = sg;
            }
#pragma unroll
        for (int cb = 0; cb < HB; ++cb) {
            const float s1 = sum_c(gb1[cb][r]);
            if (c == 0) row[pb1 + 16 * cb + 4 * q + r] = s1;
        }
        if (r == 0) {
#pragma unroll
            for (int cb = 0; cb < HB; ++cb) {
                const float s0 = sum_q_lds((gb0[cb][0] + gb0[cb][1]) + (gb0[cb][2] + gb0[cb][3]));
                if (q == 0) row[pb0 + 16 * cb + c] = s0;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < NA; ++d) {
        const float s2 = sum_c(gb2[d]);                      // held by the q = 0 lanes
        if (lane == 0) row[pb2 + d] = s2;
    }
    {
        const float aw = sum_c(sum_q_lds(accw));
        if (lane == 0) { row[P] = 0.f; row[P + 1] = 0.f; row[P + 2] = aw; }
    }
    __syncthreads();
    float* out = partials + (size_t)blockIdx.x * ROW;
    for (int i = tid; i < ROW; i += 256) out[i] = (RB[i] + RB[ROW + i]) + (RB[2 * ROW + i] + RB[3 * ROW + i]);
    FT(4);
}

// ---- dispatch (policy_mfma.hip routes MODE_FVPC here when the shape has an instantiation) ----------------------------------------
typedef void (*fvpc_kernel_t)(PolK, const float*, const float*, float*);
struct FvpcEntry { int ns, na, ph; fvpc_kernel_t kern; };
static const FvpcEntry kFvpc[] = {
    {10, 2, 32, k_fvpc_pipe<10, 2, 32>},      // swimmer
};

int policy_fvpc_select(int ns, int na, int ph) {
    if (getenv("METRPO_FVPC_GEN1") != nullptr) return -1;    // test hook: first-generation kernel
    for (int i = 0; i < (int)(sizeof(kFvpc) / sizeof(kFvpc[0])); ++i)
        if (kFvpc[i].ns == ns && kFvpc[i].na == na && kFvpc[i].ph == ph) return i;
    return -1;
}

void policy_fvpc_launch(int idx, const PolK& k, const float* theta, const float* v, float* partials, int nblocks, hipStream_t st) {
    hipLaunchKernelGGL(kFvpc[idx].kern, dim3(nblocks), dim3(256), 0, st, k, theta, v, partials);
}
