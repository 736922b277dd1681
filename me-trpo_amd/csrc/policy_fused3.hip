// Fused MFMA update kernels for three-hidden-layer tanh policies -- Humanoid's 100-50-25 (params/params-humanoid.json:5-9), which the
// two-layer kernels of policy_mfma.hip do not cover and which until round 5 ran layer by layer on the GEMM path (policy_gemm.hip: ~190
// launches per update at N = 50 000, 0.23 of the matrix peak at N = 6.25 M).  Same arithmetic as policy_update.hip (algos/npo.py:68-75 graph;
// [rllab] DiagonalGaussian / PerlmutterHvp), mapped to v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains) as THREE kernels, because the
// fragment tables of one fused product (forward 70 KB + tangent 70 KB + back-prop 45 KB + the transpose tiles) do not fit one CU's 160 KB:
//
//   k_f3_fwd  forward chain (transposed: H^T[unit][sample] = W^T X^T; the D fragment of a layer is the B operand of the next), per-sample
//             head (likelihood ratio, KL, d loss / d mean) on the VALU; writes the tanh activations (cache, MFMA D layout, 1 KB per wave
//             store) and the mean-adjoint U; per-block loss / KL / d log_std sums
//   k_f3_jvp  tangent chain of the Fisher-vector product on the cached activations: T_{l+1} = (V_l^T H_l + W_l^T T_l) (1 - H_{l+1}^2),
//             U = tangent(mean) / (s^2 + eps/2) / N
//   k_f3_bwd  back-prop of U through the cached activations + all four weight-gradient products G_l += A_l^T D_{l+1} (contracted over the
//             samples: operands through wave-private LDS transpose tiles, the last delta produced in that orientation by an MFMA with
//             swapped operands) in 272 accumulator registers per wave (one wave per SIMD); waves -> block row (LDS, fixed order) -> global
//             partial row in theta's own layout -> k_finalize (policy_update.hip: float64, fixed order, fused CG tail / line-search
//             decision / cross-rank exchange -- everything the two-layer kernels have)
//
// Biases ride along as weights: every padded width has a spare unit (55 -> 56, 100 -> 112, 50 -> 64, 25 -> 32) that is held at 1.0, so
// the bias is the weight row of that unit, its gradient the matching row of G_l, its tangent the matching row of V_l -- no bias registers,
// no column sums.  tanh'(1.0-unit) = 1 - 1 = 0 keeps tangents and deltas of the constant unit at zero by themselves.
#include "device_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
struct F3K {                    // kernel arguments
    const float* obs; const float* act; const float* adv; const float* old_mean; const float* old_ls; int ls_stride;
    const uint8_t* valid; long long N; float inv_n; const double* skip;
    const float* theta; const float* v;
    const float* img;           // fragment tables (k_f3_image)
    f32x4* hc; f32x4* u;        // activation cache / mean-adjoint: [tile][block][64 lanes] 16-byte words in the MFMA D layout
    float* partials; int row_stride, P, ls_off;
    int w_off[4], b_off[4];
};
constexpr int cdiv3(int a, int b) { return (a + b - 1) / b; }
constexpr int cbp_of(int cb) { return cb <= 1 ? 1 : cb <= 2 ? 2 : cb <= 4 ? 4 : 8; }

__device__ __forceinline__ void wave_sync_lds3() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float xsum_q3(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
template <int CTRL> __device__ __forceinline__ float dpp_add3(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float xsum_c3(float v) {         // sum over the 16 lanes c of a row (xor-1, 2, 4, 8 butterfly on the DPP path)
    v = dpp_add3<0xB1>(v); v = dpp_add3<0x4E>(v); v = dpp_add3<0x141>(v); v = dpp_add3<0x140>(v);
    return v;
}

// Shape of one policy: D0 inputs, D1-D2-D3 tanh units, D4 action dims.  P* = padded widths (multiples of 16 with room for the constant unit),
// K0 = padded input width (multiple of 4), KS* = k-steps of a product contracting that width, CB* = 16-unit column blocks.
template <int D0_, int D1_, int D2_, int D3_, int D4_>
struct F3Shape {
    static constexpr int D0 = D0_, D1 = D1_, D2 = D2_, D3 = D3_, D4 = D4_;
    static constexpr int K0 = cdiv3(D0 + 1, 4) * 4, P1 = cdiv3(D1 + 1, 16) * 16, P2 = cdiv3(D2 + 1, 16) * 16, P3 = cdiv3(D3 + 1, 16) * 16,
                         P4 = cdiv3(D4, 16) * 16;
    static constexpr int KS0 = K0 / 4, KS1 = P1 / 4, KS2 = P2 / 4, KS3 = P3 / 4, KS4 = P4 / 4;
    static constexpr int CB1 = P1 / 16, CB2 = P2 / 16, CB3 = P3 / 16, CB4 = P4 / 16, XI = cdiv3(K0, 16);
    static constexpr int NHB = CB1 + CB2 + CB3;                       // activation blocks of a tile in the cache
    // fragment tables (floats): a product of KS k-steps into CB column blocks stores KS * cbp(CB) fragments per lane, four per 16-byte word
    static constexpr int F0 = KS0 * cbp_of(CB1) * 64, F1 = KS1 * cbp_of(CB2) * 64, F2 = KS2 * cbp_of(CB3) * 64, F3 = KS3 * cbp_of(CB4) * 64;
    static constexpr int B3 = KS4 * cbp_of(CB3) * 64, B2 = KS3 * cbp_of(CB2) * 64, B1 = KS2 * cbp_of(CB1) * 64;
    static constexpr int LDS_FWD = F0 + F1 + F2 + F3, LDS_JVP = 2 * (F0 + F1 + F2 + F3) - F0;
    static constexpr int IMG_F = 0, IMG_V = LDS_FWD, IMG_B = 2 * LDS_FWD, IMG_FLOATS = 2 * LDS_FWD + B3 + B2 + B1;      // global table image (k_f3_image)
    // per-wave transpose tiles of the back-prop kernel: U, H3, H2, H1, D3, D2
    static constexpr int TS = 20, TILE = 16 * TS, NTB = CB4 + CB3 + CB2 + CB1 + CB3 + CB2;
    static constexpr int BWD_WAVES = 4;
    // gradient accumulators in fragment order: [G0: XI x CB1 | G1: CB1 x CB2 | G2: CB2 x CB3 | G3: CB3 x CB4] blocks of 256 floats
    static constexpr int GB0 = 0, GB1 = GB0 + XI * CB1, GB2 = GB1 + CB1 * CB2, GB3 = GB2 + CB2 * CB3, GBN = GB3 + CB3 * CB4;
    static constexpr int LDS_BWD_LOOP = B3 + B2 + B1 + BWD_WAVES * NTB * TILE, LDS_BWD_EPI = 2 * GBN * 256;
    static constexpr int LDS_BWD = LDS_BWD_LOOP > LDS_BWD_EPI ? LDS_BWD_LOOP : LDS_BWD_EPI;
    static constexpr int NPAR = D0 * D1 + D1 + D1 * D2 + D2 + D2 * D3 + D3 + D3 * D4 + D4;
    static_assert(D0 % 4 != 0 || K0 > D0, "room for the constant input");
    static_assert(KS3 % 2 == 0 && KS4 % 2 == 0, "two-block products pack two k-steps per word");
};

// ---- fragment tables --------------------------------------------------------------------------------------------------------------------
// Word j of a table holds, per lane, the fragments of the MFMAs e = 4j .. 4j+3 of the product's (k-step kk, column block cb) sequence,
// e = kk * CBP + cb (a lane reads 16 bytes at a lane stride of 16 bytes: no bank conflicts).
// forward type (A operand of the transposed chain): Wx[k-row in][out = 16 cb + c], in = 4 kk + q for the input layer (B operand = observations,
//   lane (c, q) holds feature 4 kk + q of sample c), in = 16 (kk >> 2) + 4 q + (kk & 3) behind a hidden layer (B operand = register kk & 3 of that
//   layer's D fragment kk >> 2).  Row DIN of Wx is the bias (the constant unit).
// back-prop type: W[i = 16 cb + c][j = 16 (kk >> 2) + 4 q + (kk & 3)], i < DIN (rows of the constant / padding units are zero), j < DOUT.
// The tables are built ONCE per launch group by k_f3_image into a global image [F0 F1 F2 F3 | V0 V1 V2 V3 | B3 B2 B1] (gathers spread over the chip) and
// copied into every workgroup's LDS by 16-byte loads.  (Built in every workgroup -- 256 workgroups gathering 27 000 four-byte words each from the same
// 50 KB -- the prologue was 15-25 us of every launch, most of a Fisher-vector product at the params files' N = 50 000.)
template <int KS, int CB, bool L0, bool BWDT>
__device__ __forceinline__ float tab_value(int i, const float* __restrict__ W, const float* __restrict__ bias, int DIN, int DOUT) {
    constexpr int CBP = cbp_of(CB);
    const int e4 = i & 3, ln = (i >> 2) & 63, j = i >> 8;
    const int e = 4 * j + e4, kk = e / CBP, cb = e % CBP, c = ln & 15, q = ln >> 4;
    const int kidx = L0 ? 4 * kk + q : 16 * (kk >> 2) + 4 * q + (kk & 3), u = 16 * cb + c;
    if (cb >= CB) return 0.f;
    if (!BWDT) {
        if (u >= DOUT) return 0.f;
        if (kidx < DIN) return W[kidx * DOUT + u];
        return (kidx == DIN && bias != nullptr) ? bias[u] : 0.f;
    }
    return (u < DIN && kidx < DOUT) ? W[u * DOUT + kidx] : 0.f;
}
// LDS <- global, n floats (a multiple of 4), all NTH threads: every load issued before the first store
template <int N, int NTH> __device__ __forceinline__ void copy_tab(float* __restrict__ dst, const float* __restrict__ src, int tid) {
    static_assert(N % 4 == 0, "16-byte words");
    constexpr int NQ = N / 4, NIT = (NQ + NTH - 1) / NTH;
    float4 w4[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const int i = it * NTH + tid; w4[it] = (i < NQ) ? ((const float4*)src)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
    for (int it = 0; it < NIT; ++it) { const int i = it * NTH + tid; if (i < NQ) ((float4*)dst)[i] = w4[it]; }
}

// acc[cb] += sum over the KS k-steps of  table fragment (A)  x  bop(kk) (B)   -- or, SWAP: bop(kk) as A and the table fragment as B, which
// delivers the product in the other orientation, D[sample 4q+r][unit c]
template <int KS, int CB, bool SWAP, class BOP>
__device__ __forceinline__ void chain(const float* __restrict__ tab, int lane, BOP bop, f32x4 (&acc)[CB]) {
    constexpr int CBP = cbp_of(CB), NQ = KS * CBP / 4;
    static_assert((KS * CBP) % 4 == 0, "whole words");
    const f32x4* __restrict__ t4 = (const f32x4*)tab;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const f32x4 w = t4[j * 64 + lane];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int e = 4 * j + e4, kk = e / CBP, cb = e % CBP;
            if (cb < CB) acc[cb] = SWAP ? MFMA16(bop(kk), w[e4], acc[cb]) : MFMA16(w[e4], bop(kk), acc[cb]);
        }
    }
}

template <int CB> __device__ __forceinline__ void zero_acc(f32x4 (&a)[CB]) {
#pragma unroll
    for (int i = 0; i < CB; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// register (cb, r) of lane (., q) holds unit 16 cb + 4 q + r: hold unit U at 1.0
template <int U, int CB> __device__ __forceinline__ void set_const_unit(f32x4 (&h)[CB], int q) {
    if (q == (U % 16) / 4) h[U / 16][U % 4] = 1.0f;
}
template <int CB> __device__ __forceinline__ void tanh_acc(f32x4 (&h)[CB]) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[cb][r] = tanh_fast(h[cb][r]);
}
template <int CB> __device__ __forceinline__ void dtanh_mul(f32x4 (&t)[CB], const f32x4 (&h)[CB]) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[cb][r] *= fmaf(-h[cb][r], h[cb][r], 1.f);
}

// observations of a tile as the B operand of the input layer: lane (c, q), k-step s -> feature 4 s + q of sample c (clamped addresses; masked by fix_xb)
template <class S> __device__ __forceinline__ void fetch_xb(const F3K& k, long long tile, int c, int q, float (&xb)[S::KS0], int& nrem) {
    const long long n0 = tile * 16;
    nrem = (int)((k.N - n0 < 16) ? k.N - n0 : 16);
    const int cl = (c < nrem) ? c : nrem - 1;
    const float* __restrict__ ob = k.obs + n0 * S::D0 + cl * S::D0;
#pragma unroll
    for (int s = 0; s < S::KS0; ++s) { const int f = 4 * s + q; xb[s] = ob[(f < S::D0) ? f : S::D0 - 1]; }
}
template <class S> __device__ __forceinline__ void fix_xb(float (&xb)[S::KS0], int q) {
#pragma unroll
    for (int s = 0; s < S::KS0; ++s) if (4 * s + 3 >= S::D0) { const int f = 4 * s + q; if (f == S::D0) xb[s] = 1.0f; else if (f > S::D0) xb[s] = 0.f; }
}

enum { F3_GRAD = 0, F3_LOSSKL = 2, F3_CACHE = 3 };
enum { IMG_WHAT_F = 1, IMG_WHAT_V = 2, IMG_WHAT_B = 4 };
// fragment tables of theta (F: forward, B: back-prop) and of the tangent vector v (V) -> the global image
template <class S>
__global__ void __launch_bounds__(256) k_f3_image(F3K k, float* __restrict__ img, int what) {
    const float* __restrict__ th = k.theta; const float* __restrict__ v = k.v;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < S::IMG_FLOATS; i += gridDim.x * 256) {
        float val;
        if (i < S::IMG_V) {
            if (!(what & IMG_WHAT_F)) continue;
            const int j = i - S::IMG_F;
            if (j < S::F0) val = tab_value<S::KS0, S::CB1, true, false>(j, th + k.w_off[0], th + k.b_off[0], S::D0, S::D1);
            else if (j < S::F0 + S::F1) val = tab_value<S::KS1, S::CB2, false, false>(j - S::F0, th + k.w_off[1], th + k.b_off[1], S::D1, S::D2);
            else if (j < S::F0 + S::F1 + S::F2) val = tab_value<S::KS2, S::CB3, false, false>(j - S::F0 - S::F1, th + k.w_off[2], th + k.b_off[2], S::D2, S::D3);
            else val = tab_value<S::KS3, S::CB4, false, false>(j - S::F0 - S::F1 - S::F2, th + k.w_off[3], th + k.b_off[3], S::D3, S::D4);
        } else if (i < S::IMG_B) {
            if (!(what & IMG_WHAT_V)) continue;
            const int j = i - S::IMG_V;
            if (j < S::F0) val = tab_value<S::KS0, S::CB1, true, false>(j, v + k.w_off[0], v + k.b_off[0], S::D0, S::D1);
            else if (j < S::F0 + S::F1) val = tab_value<S::KS1, S::CB2, false, false>(j - S::F0, v + k.w_off[1], v + k.b_off[1], S::D1, S::D2);
            else if (j < S::F0 + S::F1 + S::F2) val = tab_value<S::KS2, S::CB3, false, false>(j - S::F0 - S::F1, v + k.w_off[2], v + k.b_off[2], S::D2, S::D3);
            else val = tab_value<S::KS3, S::CB4, false, false>(j - S::F0 - S::F1 - S::F2, v + k.w_off[3], v + k.b_off[3], S::D3, S::D4);
        } else {
            if (!(what & IMG_WHAT_B)) continue;
            const int j = i - S::IMG_B;
            if (j < S::B3) val = tab_value<S::KS4, S::CB3, false, true>(j, th + k.w_off[3], nullptr, S::D3, S::D4);
            else if (j < S::B3 + S::B2) val = tab_value<S::KS3, S::CB2, false, true>(j - S::B3, th + k.w_off[2], nullptr, S::D2, S::D3);
            else val = tab_value<S::KS2, S::CB1, false, true>(j - S::B3 - S::B2, th + k.w_off[1], nullptr, S::D1, S::D2);
        }
        img[i] = val;
    }
}


// =========================================================================================================================================
// forward + head
template <class S, int MODE, int NW>
__global__ void __launch_bounds__(NW * 64) k_f3_fwd(F3K k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, q = lane >> 4;
    if (MODE == F3_LOSSKL && k.skip != nullptr && k.skip[0] >= 0.0) return;        // speculative line-search trial after the search stopped
    float* T0 = lds; float* T1 = T0 + S::F0; float* T2 = T1 + S::F1; float* T3 = T2 + S::F2;
    const float* __restrict__ th = k.theta;
    copy_tab<S::LDS_FWD, NW * 64>(lds, k.img + S::IMG_F, tid);
    // per-lane constants of the head: this lane's action dims d = 16 cb + 4 q + r
    float ls[S::CB4][4], inv_std[S::CB4][4];
#pragma unroll
    for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * cb + 4 * q + r;
            ls[cb][r] = (MODE != F3_CACHE && d < S::D4) ? fmaxf(th[k.ls_off + d], LOG_MIN_STD) : 0.f;
            inv_std[cb][r] = expf(-ls[cb][r]);
        }
    const long long ntiles = (k.N + 15) / 16;
    struct In { float xb[S::KS0]; float omu[S::CB4][4], act[S::CB4][4], ols[S::CB4][4]; float adv; int vld, nrem; };
    auto fetch = [&](long long tile_, In& in) {
        const long long tile = (tile_ < ntiles) ? tile_ : ntiles - 1;
        fetch_xb<S>(k, tile, c, q, in.xb, in.nrem);
        const long long nl = tile * 16 + ((c < in.nrem) ? c : in.nrem - 1);
        in.vld = (k.valid == nullptr) ? 1 : (int)k.valid[nl];
        if (MODE != F3_CACHE) {
#pragma unroll
            for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * cb + 4 * q + r, dc = (d < S::D4) ? d : S::D4 - 1;
                    in.omu[cb][r] = k.old_mean[nl * S::D4 + dc]; in.act[cb][r] = k.act[nl * S::D4 + dc];
                    in.ols[cb][r] = k.old_ls[(size_t)nl * k.ls_stride + dc];
                }
            in.adv = k.adv[nl];
        }
    };
    In nxt;
    const long long tstride = (long long)gridDim.x * NW;
    long long tile = (long long)blockIdx.x * NW + wave;
    fetch(tile, nxt);
    __syncthreads();
    float acc0 = 0.f, acc1 = 0.f;
    float dls[S::CB4][4];
#pragma unroll
    for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dls[cb][r] = 0.f;

    for (; tile < ntiles; tile += tstride) {
        In in = nxt;
        fetch(tile + tstride, nxt);
        asm volatile("" ::: "memory");
        fix_xb<S>(in.xb, q);
        const bool ok = (c < in.nrem) && in.vld != 0;
        f32x4 h1[S::CB1], h2[S::CB2], h3[S::CB3];
        zero_acc(h1);
        chain<S::KS0, S::CB1, false>(T0, lane, [&](int kk) { return in.xb[kk]; }, h1);
        tanh_acc(h1); set_const_unit<S::D1>(h1, q);
        zero_acc(h2);
        chain<S::KS1, S::CB2, false>(T1, lane, [&](int kk) { return h1[kk >> 2][kk & 3]; }, h2);
        tanh_acc(h2); set_const_unit<S::D2>(h2, q);
        zero_acc(h3);
        chain<S::KS2, S::CB3, false>(T2, lane, [&](int kk) { return h2[kk >> 2][kk & 3]; }, h3);
        tanh_acc(h3); set_const_unit<S::D3>(h3, q);
        if (MODE != F3_LOSSKL) {
            f32x4* __restrict__ hw = k.hc + (size_t)tile * S::NHB * 64 + lane;
#pragma unroll
            for (int cb = 0; cb < S::CB1; ++cb) hw[cb * 64] = h1[cb];
#pragma unroll
            for (int cb = 0; cb < S::CB2; ++cb) hw[(S::CB1 + cb) * 64] = h2[cb];
#pragma unroll
            for (int cb = 0; cb < S::CB3; ++cb) hw[(S::CB1 + S::CB2 + cb) * 64] = h3[cb];
        }
        if (MODE == F3_CACHE) continue;
        f32x4 mu[S::CB4];
        zero_acc(mu);
        chain<S::KS3, S::CB4, false>(T3, lane, [&](int kk) { return h3[kk >> 2][kk & 3]; }, mu);
        // ---- head (npo.py:69-75; DiagonalGaussian.log_likelihood_sym / kl_sym), this lane's action dims of sample c
        float llr = 0.f, kl = 0.f, zz[S::CB4][4];
#pragma unroll
        for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = 16 * cb + 4 * q + r;
                zz[cb][r] = 0.f;
                if (d < S::D4 && ok) {
                    const float omu = in.omu[cb][r], a = in.act[cb][r], ols = in.ols[cb][r];
                    const float eo = expf(-ols);
                    const float z = (a - mu[cb][r]) * inv_std[cb][r], zo = (a - omu) * eo;
                    llr += (ols - ls[cb][r]) + 0.5f * (zo * zo - z * z);
                    zz[cb][r] = z;
                    if (MODE == F3_LOSSKL) {
                        const float s2 = expf(2.f * ls[cb][r]), os2 = expf(2.f * ols), dm = omu - mu[cb][r];
                        kl += (dm * dm + os2 - s2) / (2.f * s2 + KL_EPS) + ls[cb][r] - ols;
                    }
                }
            }
        llr = xsum_q3(llr);                                 // the four q-lanes of sample c hold its action dims between them
        const float la = ok ? expf(llr) * in.adv : 0.f;     // lr * adv
        if (q == 0) acc0 -= la * k.inv_n;                   // surr_loss = -mean(lr * adv), once per sample
        if (MODE == F3_LOSSKL) { acc1 += kl * k.inv_n; continue; }
        const float w = -la * k.inv_n;
        f32x4* __restrict__ uw = k.u + (size_t)tile * S::CB4 * 64 + lane;
#pragma unroll
        for (int cb = 0; cb < S::CB4; ++cb) {
            f32x4 um;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                um[r] = w * zz[cb][r] * inv_std[cb][r];                               // d loss / d mean = w (a - mu) / std^2
                if (16 * cb + 4 * q + r < S::D4) dls[cb][r] += w * (zz[cb][r] * zz[cb][r] - 1.f);      // d loss / d log_std
            }
            uw[cb * 64] = um;
        }
    }
    if (MODE == F3_CACHE) return;
    // ---- per-block sums -> columns [ls_off, ls_off + D4) and P, P+1, P+2 of this block's partial row (fixed order over the waves)
    __syncthreads();
    float* red = lds;                                       // [NW][3 + P4]
    constexpr int RW = 3 + S::P4;
    {
        const float a0 = xsum_c3(xsum_q3(acc0)), a1 = xsum_c3(xsum_q3(acc1));
        if (lane == 0) { red[wave * RW] = a0; red[wave * RW + 1] = a1; red[wave * RW + 2] = 0.f; }
#pragma unroll
        for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float s = xsum_c3(dls[cb][r]); if (c == 0) red[wave * RW + 3 + 16 * cb + 4 * q + r] = s; }
    }
    __syncthreads();
    if (tid < 3 + S::D4) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[w * RW + tid];
        float* row = k.partials + (size_t)blockIdx.x * k.row_stride;
        if (tid < 3) row[k.P + tid] = a;
        else if (MODE == F3_GRAD) row[k.ls_off + tid - 3] = (th[k.ls_off + tid - 3] > LOG_MIN_STD) ? a : 0.f;
    }
}

// =========================================================================================================================================
// tangent chain of the Fisher-vector product on the cached activations
template <class S, int NW>
__global__ void __launch_bounds__(NW * 64) k_f3_jvp(F3K k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, q = lane >> 4;
    float* V0 = lds; float* V1 = V0 + S::F0; float* V2 = V1 + S::F1; float* V3 = V2 + S::F2;
    float* W1 = V3 + S::F3; float* W2 = W1 + S::F1; float* W3 = W2 + S::F2;
    const float* __restrict__ th = k.theta;
    copy_tab<S::LDS_FWD, NW * 64>(V0, k.img + S::IMG_V, tid);
    copy_tab<S::F1 + S::F2 + S::F3, NW * 64>(W1, k.img + S::IMG_F + S::F0, tid);      // the forward tables of layers 1 .. 3 (their bias rows meet the constant unit's zero tangent)
    float fisher_w[S::CB4][4];                              // d2 KL / d mean^2 = 1 / (s^2 + eps/2)
#pragma unroll
    for (int cb = 0; cb < S::CB4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int d = 16 * cb + 4 * q + r;
            fisher_w[cb][r] = (d < S::D4) ? 1.0f / (expf(2.f * fmaxf(th[k.ls_off + d], LOG_MIN_STD)) + 0.5f * KL_EPS) : 0.f;
        }
    const long long ntiles = (k.N + 15) / 16;
    struct In { float xb[S::KS0]; f32x4 h[S::NHB]; int vld, nrem; };
    auto fetch = [&](long long tile_, In& in) {
        const long long tile = (tile_ < ntiles) ? tile_ : ntiles - 1;
        fetch_xb<S>(k, tile, c, q, in.xb, in.nrem);
        const long long nl = tile * 16 + ((c < in.nrem) ? c : in.nrem - 1);
        in.vld = (k.valid == nullptr) ? 1 : (int)k.valid[nl];
        const f32x4* __restrict__ hr = k.hc + (size_t)tile * S::NHB * 64 + lane;
#pragma unroll
        for (int b = 0; b < S::NHB; ++b) in.h[b] = hr[b * 64];
    };
    In nxt;
    const long long tstride = (long long)gridDim.x * NW;
    long long tile = (long long)blockIdx.x * NW + wave;
    fetch(tile, nxt);
    __syncthreads();
    float accw = 0.f;
    for (; tile < ntiles; tile += tstride) {
        In in = nxt;
        fetch(tile + tstride, nxt);
        asm volatile("" ::: "memory");
        fix_xb<S>(in.xb, q);
        const bool ok = (c < in.nrem) && in.vld != 0;
        f32x4 h1[S::CB1], h2[S::CB2], h3[S::CB3], t1[S::CB1], t2[S::CB2], t3[S::CB3], tm[S::CB4];
#pragma unroll
        for (int cb = 0; cb < S::CB1; ++cb) h1[cb] = in.h[cb];
#pragma unroll
        for (int cb = 0; cb < S::CB2; ++cb) h2[cb] = in.h[S::CB1 + cb];
#pragma unroll
        for (int cb = 0; cb < S::CB3; ++cb) h3[cb] = in.h[S::CB1 + S::CB2 + cb];
        zero_acc(t1);
        chain<S::KS0, S::CB1, false>(V0, lane, [&](int kk) { return in.xb[kk]; }, t1);
        zero_acc(t2);
        chain<S::KS1, S::CB2, false>(V1, lane, [&](int kk) { return h1[kk >> 2][kk & 3]; }, t2);      // independent of t1: runs under its tanh' factors
        dtanh_mul(t1, h1);
        chain<S::KS1, S::CB2, false>(W1, lane, [&](int kk) { return t1[kk >> 2][kk & 3]; }, t2);
        zero_acc(t3);
        chain<S::KS2, S::CB3, false>(V2, lane, [&](int kk) { return h2[kk >> 2][kk & 3]; }, t3);
        dtanh_mul(t2, h2);
        chain<S::KS2, S::CB3, false>(W2, lane, [&](int kk) { return t2[kk >> 2][kk & 3]; }, t3);
        zero_acc(tm);
        chain<S::KS3, S::CB4, false>(V3, lane, [&](int kk) { return h3[kk >> 2][kk & 3]; }, tm);
        dtanh_mul(t3, h3);
        chain<S::KS3, S::CB4, false>(W3, lane, [&](int kk) { return t3[kk >> 2][kk & 3]; }, tm);
        f32x4* __restrict__ uw = k.u + (size_t)tile * S::CB4 * 64 + lane;
#pragma unroll
        for (int cb = 0; cb < S::CB4; ++cb) {
            f32x4 um;
#pragma unroll
            for (int r = 0; r < 4; ++r) um[r] = ok ? tm[cb][r] * fisher_w[cb][r] * k.inv_n : 0.f;
            uw[cb * 64] = um;
        }
        if (ok && q == 0) accw += k.inv_n;
    }
    __syncthreads();
    float* red = lds;
    { const float aw = xsum_c3(xsum_q3(accw)); if (lane == 0) red[wave] = aw; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[w];
        float* row = k.partials + (size_t)blockIdx.x * k.row_stride;
        row[k.P] = 0.f; row[k.P + 1] = 0.f; row[k.P + 2] = a;
    }
}

// =========================================================================================================================================
// back-prop of U + the four weight-gradient products
template <class S>
__global__ void __launch_bounds__(S::BWD_WAVES * 64) k_f3_bwd(F3K k) {
    constexpr int NW = S::BWD_WAVES, TS = S::TS, TILE = S::TILE;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, q = lane >> 4;
    float* B3 = lds; float* B2 = B3 + S::B3; float* B1 = B2 + S::B2;
    float* TL = B1 + S::B1 + wave * (S::NTB * TILE);
    float* T_U = TL; float* T_H3 = T_U + S::CB4 * TILE; float* T_H2 = T_H3 + S::CB3 * TILE; float* T_H1 = T_H2 + S::CB2 * TILE;
    float* T_D3 = T_H1 + S::CB1 * TILE; float* T_D2 = T_D3 + S::CB3 * TILE;
    copy_tab<S::B3 + S::B2 + S::B1, NW * 64>(B3, k.img + S::IMG_B, tid);
    const long long ntiles = (k.N + 15) / 16;
    // Registers: 272 gradient accumulators leave ~240 for everything else, so nothing is held twice.  The cached activations and U of tile t+1 are
    // fetched once tile t's copies are dead (behind its delta chain: they fly under the 272 gradient MFMAs, ~3.6 us), the transposed observations of
    // tile t at its start (consumed by its last product).
    struct In { f32x4 h[S::NHB]; f32x4 u[S::CB4]; };
    auto fetch = [&](long long tile_, In& in) {
        const long long tile = (tile_ < ntiles) ? tile_ : ntiles - 1;
        const f32x4* __restrict__ hr = k.hc + (size_t)tile * S::NHB * 64 + lane;
#pragma unroll
        for (int b = 0; b < S::NHB; ++b) in.h[b] = hr[b * 64];
        const f32x4* __restrict__ ur = k.u + (size_t)tile * S::CB4 * 64 + lane;
#pragma unroll
        for (int b = 0; b < S::CB4; ++b) in.u[b] = ur[b * 64];
    };
    auto fetch_xt = [&](long long tile, float (&xt)[4][S::XI], int& nrem) {      // observations transposed: lane (c, q), register s -> feature 16 ci + c of sample 4 q + s
        const long long n0 = tile * 16;
        nrem = (int)((k.N - n0 < 16) ? k.N - n0 : 16);
        const float* __restrict__ ob = k.obs + n0 * S::D0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = 4 * q + s, slc = (sl < nrem) ? sl : nrem - 1;
#pragma unroll
            for (int ci = 0; ci < S::XI; ++ci) { const int f = 16 * ci + c; xt[s][ci] = ob[slc * S::D0 + ((f < S::D0) ? f : S::D0 - 1)]; }
        }
    };
    auto fix_xt = [&](float (&xt)[4][S::XI], int nrem) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ci = 0; ci < S::XI; ++ci) {
                if (16 * ci + 15 >= S::D0) { const int f = 16 * ci + c; if (f == S::D0) xt[s][ci] = 1.0f; else if (f > S::D0) xt[s][ci] = 0.f; }
                if (4 * q + s >= nrem) xt[s][ci] = 0.f;
            }
    };
    In in;
    const long long tstride = (long long)gridDim.x * NW;
    long long tile = (long long)blockIdx.x * NW + wave;
    fetch(tile, in);
    __syncthreads();
    f32x4 G0[S::XI][S::CB1], G1[S::CB1][S::CB2], G2[S::CB2][S::CB3], G3[S::CB3][S::CB4];
#pragma unroll
    for (int a = 0; a < S::XI; ++a) zero_acc(G0[a]);
#pragma unroll
    for (int a = 0; a < S::CB1; ++a) zero_acc(G1[a]);
#pragma unroll
    for (int a = 0; a < S::CB2; ++a) zero_acc(G2[a]);
#pragma unroll
    for (int a = 0; a < S::CB3; ++a) zero_acc(G3[a]);

    auto put_tile = [&](float* T, const f32x4& val) {       // D fragment [unit 4q+r][sample c] -> T[unit][sample]
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(4 * q + r) * TS + c] = val[r];
    };
    auto get_tile = [&](const float* T) -> f32x4 { return *(const f32x4*)&T[c * TS + 4 * q]; };      // lane (unit c, q): samples 4q .. 4q+3

    for (; tile < ntiles; tile += tstride) {
        float xt[4][S::XI]; int nrem;
        fetch_xt(tile, xt, nrem);
        asm volatile("" ::: "memory");
        f32x4 h1[S::CB1], h2[S::CB2], h3[S::CB3], um[S::CB4];
#pragma unroll
        for (int cb = 0; cb < S::CB1; ++cb) { h1[cb] = in.h[cb]; put_tile(T_H1 + cb * TILE, h1[cb]); }
#pragma unroll
        for (int cb = 0; cb < S::CB2; ++cb) { h2[cb] = in.h[S::CB1 + cb]; put_tile(T_H2 + cb * TILE, h2[cb]); }
#pragma unroll
        for (int cb = 0; cb < S::CB3; ++cb) { h3[cb] = in.h[S::CB1 + S::CB2 + cb]; put_tile(T_H3 + cb * TILE, h3[cb]); }
#pragma unroll
        for (int cb = 0; cb < S::CB4; ++cb) { um[cb] = in.u[cb]; put_tile(T_U + cb * TILE, um[cb]); }
        // ---- deltas: D3 = (W3 U) (1 - H3^2), D2 = (W2 D3) (1 - H2^2) in the chain's orientation; D1 with swapped operands -> [sample 4q+r][unit c]
        f32x4 d3[S::CB3], d2[S::CB2], d1n[S::CB1];
        zero_acc(d3);
        chain<S::KS4, S::CB3, false>(B3, lane, [&](int kk) { return um[kk >> 2][kk & 3]; }, d3);
        dtanh_mul(d3, h3);
#pragma unroll
        for (int cb = 0; cb < S::CB3; ++cb) put_tile(T_D3 + cb * TILE, d3[cb]);
        SCHED_FENCE();
        zero_acc(d2);
        chain<S::KS3, S::CB2, false>(B2, lane, [&](int kk) { return d3[kk >> 2][kk & 3]; }, d2);
        dtanh_mul(d2, h2);
#pragma unroll
        for (int cb = 0; cb < S::CB2; ++cb) put_tile(T_D2 + cb * TILE, d2[cb]);
        SCHED_FENCE();
        zero_acc(d1n);
        chain<S::KS2, S::CB1, true>(B1, lane, [&](int kk) { return d2[kk >> 2][kk & 3]; }, d1n);
        wave_sync_lds3();
        fetch(tile + tstride, in);                          // this tile's copies are dead: the next tile's loads fly under the gradient products
        asm volatile("" ::: "memory");
        // ---- weight gradients G_l[i][j] += sum_n a_l[i][n] d_{l+1}[j][n]: k-step s contracts samples 4q + s
        // (scheduling fences between the groups: left alone, the scheduler hoists every group's LDS reads to the top and spills ~120 registers)
        SCHED_FENCE();
        {
            f32x4 a3[S::CB3], bu[S::CB4];
#pragma unroll
            for (int cb = 0; cb < S::CB3; ++cb) a3[cb] = get_tile(T_H3 + cb * TILE);
#pragma unroll
            for (int cb = 0; cb < S::CB4; ++cb) bu[cb] = get_tile(T_U + cb * TILE);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ci = 0; ci < S::CB3; ++ci)
#pragma unroll
                    for (int cj = 0; cj < S::CB4; ++cj) G3[ci][cj] = MFMA16(a3[ci][s], bu[cj][s], G3[ci][cj]);
        }
        SCHED_FENCE();
        {
            f32x4 a2[S::CB2], b3[S::CB3];
#pragma unroll
            for (int cb = 0; cb < S::CB2; ++cb) a2[cb] = get_tile(T_H2 + cb * TILE);
#pragma unroll
            for (int cb = 0; cb < S::CB3; ++cb) b3[cb] = get_tile(T_D3 + cb * TILE);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ci = 0; ci < S::CB2; ++ci)
#pragma unroll
                    for (int cj = 0; cj < S::CB3; ++cj) G2[ci][cj] = MFMA16(a2[ci][s], b3[cj][s], G2[ci][cj]);
        }
        SCHED_FENCE();
        {
            f32x4 a1[S::CB1], b2[S::CB2];
#pragma unroll
            for (int cb = 0; cb < S::CB1; ++cb) a1[cb] = get_tile(T_H1 + cb * TILE);
#pragma unroll
            for (int cb = 0; cb < S::CB2; ++cb) b2[cb] = get_tile(T_D2 + cb * TILE);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ci = 0; ci < S::CB1; ++ci)
#pragma unroll
                    for (int cj = 0; cj < S::CB2; ++cj) G1[ci][cj] = MFMA16(a1[ci][s], b2[cj][s], G1[ci][cj]);
            SCHED_FENCE();
            dtanh_mul(d1n, a1);                             // the layer-1 tanh' factor in the orientation the swapped product came out in
            fix_xt(xt, nrem);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ci = 0; ci < S::XI; ++ci)
#pragma unroll
                    for (int cj = 0; cj < S::CB1; ++cj) G0[ci][cj] = MFMA16(xt[s][ci], d1n[cj][s], G0[ci][cj]);
        }
        wave_sync_lds3();
    }

    // ---- epilogue: waves -> two LDS rows in fragment order ((w0 + w2), (w1 + w3)) -> natural theta layout -> this block's global partial row
    __syncthreads();
    float* R = lds + (wave & 1) * (S::GBN * 256);
    auto frag_rw = [&](bool add) {
        f32x4* r4 = (f32x4*)R;
        // (LDS float atomics without return instead of the read / add / write of waves 2, 3 were measured: ds_add_f32 x 272 took 40 us)
        auto one = [&](int blk, const f32x4& g) { f32x4 t = g; if (add) t += r4[blk * 64 + lane]; r4[blk * 64 + lane] = t; };
#pragma unroll
        for (int ci = 0; ci < S::XI; ++ci)
#pragma unroll
            for (int cj = 0; cj < S::CB1; ++cj) one(S::GB0 + ci * S::CB1 + cj, G0[ci][cj]);
#pragma unroll
        for (int ci = 0; ci < S::CB1; ++ci)
#pragma unroll
            for (int cj = 0; cj < S::CB2; ++cj) one(S::GB1 + ci * S::CB2 + cj, G1[ci][cj]);
#pragma unroll
        for (int ci = 0; ci < S::CB2; ++ci)
#pragma unroll
            for (int cj = 0; cj < S::CB3; ++cj) one(S::GB2 + ci * S::CB3 + cj, G2[ci][cj]);
#pragma unroll
        for (int ci = 0; ci < S::CB3; ++ci)
#pragma unroll
            for (int cj = 0; cj < S::CB4; ++cj) one(S::GB3 + ci * S::CB4 + cj, G3[ci][cj]);
    };
    if (wave < 2) frag_rw(false);
    __syncthreads();
    if (wave >= 2) frag_rw(true);
    __syncthreads();
    // fragment address of G_l[i][j]: block (i / 16) * CJ + j / 16, lane 16 ((i % 16) / 4) + j % 16, register i % 4
    float* row = k.partials + (size_t)blockIdx.x * k.row_stride;
    const float* RA = lds; const float* RB = lds + S::GBN * 256;
    auto frag = [&](int base, int CJ, int i, int j) { return ((base + (i >> 4) * CJ + (j >> 4)) * 64 + 16 * ((i & 15) >> 2) + (j & 15)) * 4 + (i & 3); };
#pragma unroll 4
    for (int p = tid; p < S::NPAR; p += NW * 64) {
        int f;
        if (p < k.w_off[1]) { const int o = p - k.w_off[0]; f = (p < k.b_off[0]) ? frag(S::GB0, S::CB1, o / S::D1, o % S::D1) : frag(S::GB0, S::CB1, S::D0, p - k.b_off[0]); }
        else if (p < k.w_off[2]) { const int o = p - k.w_off[1]; f = (p < k.b_off[1]) ? frag(S::GB1, S::CB2, o / S::D2, o % S::D2) : frag(S::GB1, S::CB2, S::D1, p - k.b_off[1]); }
        else if (p < k.w_off[3]) { const int o = p - k.w_off[2]; f = (p < k.b_off[2]) ? frag(S::GB2, S::CB3, o / S::D3, o % S::D3) : frag(S::GB2, S::CB3, S::D2, p - k.b_off[2]); }
        else { const int o = p - k.w_off[3]; f = (p < k.b_off[3]) ? frag(S::GB3, S::CB4, o / S::D4, o % S::D4) : frag(S::GB3, S::CB4, S::D3, p - k.b_off[3]); }
        row[p] = RA[f] + RB[f];
    }
}

typedef F3Shape<55, 100, 50, 25, 21> ShHumanoid;
constexpr int FWD_NW = 8, JVP_NW = 8;

}  // namespace

// -------------------------------------------------------------------------------------------------------------------------------------------
int policy_f3_select(const ProblemDesc& pd) {
    const NetDesc& n = pd.pol;
    if (n.n_layers != 4) return 0;
    for (int l = 0; l < 3; ++l) if (n.act[l] != METRPO_ACT_TANH) return 0;
    typedef ShHumanoid S;
    if (n.dims[0] != S::D0 || n.dims[1] != S::D1 || n.dims[2] != S::D2 || n.dims[3] != S::D3 || n.dims[4] != S::D4 || pd.na != S::D4) return 0;
    if (n.n_params != S::NPAR || n.w_off[0] != 0) return 0;
    return 1;
}

// workspace: tanh activations of (theta, batch) in the MFMA D layout, 1 KB per (tile, 16-unit block), and the mean-adjoint U likewise
static int f3_ensure(metrpo_ctx* c, long long N) {
    typedef ShHumanoid S;
    const size_t tiles = (size_t)((N + 15) / 16);
    const size_t need = tiles * (S::NHB + S::CB4) * 64 * sizeof(float) * 4 + (size_t)S::IMG_FLOATS * sizeof(float);
    if (need > c->f3_cap) {
        if (c->d_f3) { ws_retire(c, c->d_f3); c->d_f3 = nullptr; c->f3_cap = 0; }
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_f3, need));
        c->f3_cap = need;
        c->f3_rows = -1;
    }
    return METRPO_OK;
}

template <class K> static int f3_attr(metrpo_ctx* c, K kern, size_t sh) {
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    return METRPO_OK;
}

// mode 0 gradient, 1 Fisher-vector product, 2 loss + KL; per-block rows of P + 3 floats land in `partials` (the layout k_finalize reads)
int policy_f3_launch(metrpo_ctx* c, int mode, const metrpo_batch* b, const float* theta, const float* vf, float* partials, int nblocks,
                     hipStream_t st) {
    typedef ShHumanoid S;
    const long long N = b->N;
    int rc = f3_ensure(c, N); if (rc) return rc;
    const size_t tiles = (size_t)((N + 15) / 16);
    F3K k = {};
    k.obs = b->d_obs; k.act = b->d_act; k.adv = b->d_adv; k.old_mean = b->d_old_mean; k.old_ls = b->d_old_log_std; k.ls_stride = b->old_log_std_stride;
    k.valid = b->d_valid; k.N = N; k.inv_n = (float)b->inv_n_global; k.skip = c->ls_skip; k.theta = theta; k.v = vf;
    k.hc = (f32x4*)c->d_f3; k.u = k.hc + tiles * S::NHB * 64;
    float* img = (float*)(k.u + tiles * S::CB4 * 64);
    k.img = img;
    auto build_image = [&](int what) { hipLaunchKernelGGL((k_f3_image<S>), dim3(64), dim3(256), 0, st, k, img, what); };
    k.partials = partials; k.row_stride = c->pd.P + 3; k.P = c->pd.P; k.ls_off = c->pd.pol.n_params;
    for (int l = 0; l < 4; ++l) { k.w_off[l] = c->pd.pol.w_off[l]; k.b_off[l] = c->pd.pol.b_off[l]; }
    const size_t sh_fwd = sizeof(float) * S::LDS_FWD, sh_jvp = sizeof(float) * S::LDS_JVP, sh_bwd = sizeof(float) * S::LDS_BWD;
    const dim3 g(nblocks);
    if (mode == 2) {
        build_image(IMG_WHAT_F); c->f3_img_ok = 0;          // (the trial theta's tables: the image no longer belongs to the cached activations)
        if ((rc = f3_attr(c, k_f3_fwd<S, F3_LOSSKL, FWD_NW>, sh_fwd))) return rc;
        hipLaunchKernelGGL((k_f3_fwd<S, F3_LOSSKL, FWD_NW>), g, dim3(FWD_NW * 64), sh_fwd, st, k);
        HIP_TRY(c, hipGetLastError());
        return METRPO_OK;
    }
    if (mode == 0) {
        build_image(IMG_WHAT_F | IMG_WHAT_B); c->f3_img_ok = 1;
        if ((rc = f3_attr(c, k_f3_fwd<S, F3_GRAD, FWD_NW>, sh_fwd))) return rc;
        hipLaunchKernelGGL((k_f3_fwd<S, F3_GRAD, FWD_NW>), g, dim3(FWD_NW * 64), sh_fwd, st, k);
        // the activations stay valid for the Fisher-vector products of this CG solve (run_trpo_update raises hcache_on around it)
        c->f3_rows = c->hcache_on ? N : -1; c->f3_obs = b->d_obs; c->f3_theta = theta;
    } else {
        const bool have = c->hcache_on && c->f3_img_ok && c->f3_rows == N && c->f3_obs == b->d_obs && c->f3_theta == theta;
        build_image(IMG_WHAT_V | (have ? 0 : (IMG_WHAT_F | IMG_WHAT_B)));
        if (!have) {
            c->f3_img_ok = 0;
            if ((rc = f3_attr(c, k_f3_fwd<S, F3_CACHE, FWD_NW>, sh_fwd))) return rc;
            hipLaunchKernelGGL((k_f3_fwd<S, F3_CACHE, FWD_NW>), g, dim3(FWD_NW * 64), sh_fwd, st, k);
            c->f3_rows = -1;
        }
        if ((rc = f3_attr(c, k_f3_jvp<S, JVP_NW>, sh_jvp))) return rc;
        hipLaunchKernelGGL((k_f3_jvp<S, JVP_NW>), g, dim3(JVP_NW * 64), sh_jvp, st, k);
    }
    if ((rc = f3_attr(c, k_f3_bwd<S>, sh_bwd))) return rc;
    hipLaunchKernelGGL((k_f3_bwd<S>), g, dim3(S::BWD_WAVES * 64), sh_bwd, st, k);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
