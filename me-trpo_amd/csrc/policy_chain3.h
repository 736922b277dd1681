// Transposed MFMA chain of a tanh policy with THREE hidden layers (Humanoid's 100-50-25, params-humanoid.json), shared by the pre-steps of the
// step-wise rollout (rollout_gemm.hip: k_big_pre_mfma3) and of the deterministic sweeps (det_gemm.hip: k_dg_pre_mfma3).  Same scheme as the 2 x 32
// pre-kernels: one wave per 16-row tile, every layer's D fragment is the next layer's B operand, widths padded to whole 16-unit tiles (zero
// weights; tanh(0) = 0 meets zero rows of the next layer).  The fragment image depends on theta only: built once per launch chain by
// k_pre_mfma3_image, copied into LDS with 16-byte loads by every step's kernel.
#pragma once
#include "mfma_common.h"

template <int NS, int NA, int W1, int W2, int W3>
struct P3 {
    static constexpr int NS_KS = cdiv(NS, 4), C1 = cdiv(W1, 16), C2 = cdiv(W2, 16), C3 = cdiv(W3, 16), CO = cdiv(NA, 16);
    static constexpr int O_F0 = 0, O_F1 = O_F0 + NS_KS * C1 * 64, O_F2 = O_F1 + 4 * C1 * C2 * 64, O_F3 = O_F2 + 4 * C2 * C3 * 64, O_B0 = O_F3 + 4 * C3 * CO * 64,
                         O_B1 = O_B0 + 16 * C1, O_B2 = O_B1 + 16 * C2, O_B3 = O_B2 + 16 * C3, IMG = O_B3 + 16 * CO;
    static constexpr int pLS = NS * W1 + W1 + W1 * W2 + W2 + W2 * W3 + W3 + W3 * NA + NA;      // rllab's flat order [W0, b0, ..., Wout, bout, log_std]
    static_assert(IMG % 4 == 0, "image tables are multiples of 16 floats");
    // image: global -> LDS, all 256 threads
    static __device__ __forceinline__ void load_image(float* lds, const float* __restrict__ img, int tid) {
        constexpr int NQ = IMG / 4, NIT = cdiv(NQ, 256);
        float4 w4[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int i = it * 256 + tid; w4[it] = (i < NQ) ? ((const float4*)img)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int it = 0; it < NIT; ++it) { const int i = it * 256 + tid; if (i < NQ) ((float4*)lds)[i] = w4[it]; }
    }
    // mean of the policy for the wave's 16-row tile ST [16][NS] (LDS): mu[cb][r] = action dim 16 cb + 4 q + r of row c
    static __device__ __forceinline__ void forward(const float* lds, const float* ST, int lane, int c, int q, f32x4 (&mu)[CO]) {
        f32x4 p0[C1], p1[C2], p2[C3];
#pragma unroll
        for (int cb = 0; cb < C1; ++cb) p0[cb] = *(const f32x4*)&lds[O_B0 + 16 * cb + 4 * q];
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) {
            const int f = 4 * s_ + q;
            const float x = (f < NS) ? ST[c * NS + f] : 0.0f;
#pragma unroll
            for (int cb = 0; cb < C1; ++cb) p0[cb] = MFMA16(lds[O_F0 + (s_ * C1 + cb) * 64 + lane], x, p0[cb]);
        }
#pragma unroll
        for (int cb = 0; cb < C1; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p0[cb][rr] = tanh_fast(p0[cb][rr]);
#pragma unroll
        for (int cb = 0; cb < C2; ++cb) p1[cb] = *(const f32x4*)&lds[O_B1 + 16 * cb + 4 * q];
#pragma unroll
        for (int kk = 0; kk < 4 * C1; ++kk)
#pragma unroll
            for (int cb = 0; cb < C2; ++cb) p1[cb] = MFMA16(lds[O_F1 + (kk * C2 + cb) * 64 + lane], p0[kk >> 2][kk & 3], p1[cb]);
#pragma unroll
        for (int cb = 0; cb < C2; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p1[cb][rr] = tanh_fast(p1[cb][rr]);
#pragma unroll
        for (int cb = 0; cb < C3; ++cb) p2[cb] = *(const f32x4*)&lds[O_B2 + 16 * cb + 4 * q];
#pragma unroll
        for (int kk = 0; kk < 4 * C2; ++kk)
#pragma unroll
            for (int cb = 0; cb < C3; ++cb) p2[cb] = MFMA16(lds[O_F2 + (kk * C3 + cb) * 64 + lane], p1[kk >> 2][kk & 3], p2[cb]);
#pragma unroll
        for (int cb = 0; cb < C3; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) p2[cb][rr] = tanh_fast(p2[cb][rr]);
#pragma unroll
        for (int cb = 0; cb < CO; ++cb) mu[cb] = *(const f32x4*)&lds[O_B3 + 16 * cb + 4 * q];
#pragma unroll
        for (int kk = 0; kk < 4 * C3; ++kk)
#pragma unroll
            for (int cb = 0; cb < CO; ++cb) mu[cb] = MFMA16(lds[O_F3 + (kk * CO + cb) * 64 + lane], p2[kk >> 2][kk & 3], mu[cb]);
    }
};

// fragment image of a three-hidden-layer tanh policy for k_big_pre_mfma3.  Layer l's k-step kk contracts, in lane (cc, qq), input unit
// 16 (kk >> 2) + 4 qq + (kk & 3): the D-fragment order of the previous layer's output (register kk & 3 of tile kk >> 2), so nothing is transposed
// between layers.  [layer-0 fragments | layer 1 | layer 2 | output layer | b0 | b1 | b2 | b3], widths padded to 16 with zeros.
template <int NS, int NA, int W1, int W2, int W3>
static __global__ void k_pre_mfma3_image(const float* __restrict__ theta, float* __restrict__ img) {
    constexpr int NS_KS = cdiv(NS, 4), C1 = cdiv(W1, 16), C2 = cdiv(W2, 16), C3 = cdiv(W3, 16), CO = cdiv(NA, 16);
    constexpr int pW0 = 0, pb0 = NS * W1, pW1 = pb0 + W1, pb1 = pW1 + W1 * W2, pW2 = pb1 + W2, pb2 = pW2 + W2 * W3, pW3 = pb2 + W3, pb3 = pW3 + W3 * NA;
    constexpr int O_F0 = 0, O_F1 = O_F0 + NS_KS * C1 * 64, O_F2 = O_F1 + 4 * C1 * C2 * 64, O_F3 = O_F2 + 4 * C2 * C3 * 64, O_B0 = O_F3 + 4 * C3 * CO * 64,
                  O_B1 = O_B0 + 16 * C1, O_B2 = O_B1 + 16 * C2, O_B3 = O_B2 + 16 * C3, IMG = O_B3 + 16 * CO;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= IMG) return;
    float w = 0.0f;
    const int ln = i & 63, cc = ln & 15, qq = ln >> 4;
    if (i < O_F1) { const int f = i >> 6, s_ = f / C1, cb = f % C1, in = 4 * s_ + qq, u = 16 * cb + cc; if (in < NS && u < W1) w = theta[pW0 + in * W1 + u]; }
    else if (i < O_F2) { const int f = (i - O_F1) >> 6, kk = f / C2, cb = f % C2, in = 16 * (kk >> 2) + 4 * qq + (kk & 3), u = 16 * cb + cc; if (in < W1 && u < W2) w = theta[pW1 + in * W2 + u]; }
    else if (i < O_F3) { const int f = (i - O_F2) >> 6, kk = f / C3, cb = f % C3, in = 16 * (kk >> 2) + 4 * qq + (kk & 3), u = 16 * cb + cc; if (in < W2 && u < W3) w = theta[pW2 + in * W3 + u]; }
    else if (i < O_B0) { const int f = (i - O_F3) >> 6, kk = f / CO, cb = f % CO, in = 16 * (kk >> 2) + 4 * qq + (kk & 3), u = 16 * cb + cc; if (in < W3 && u < NA) w = theta[pW3 + in * NA + u]; }
    else if (i < O_B1) { const int u = i - O_B0; if (u < W1) w = theta[pb0 + u]; }
    else if (i < O_B2) { const int u = i - O_B1; if (u < W2) w = theta[pb1 + u]; }
    else if (i < O_B3) { const int u = i - O_B2; if (u < W3) w = theta[pb2 + u]; }
    else { const int u = i - O_B3; if (u < NA) w = theta[pb3 + u]; }
    img[i] = w;
}
template <int NS, int NA, int W1, int W2, int W3> constexpr int pre_mfma3_image_floats() {
    return (cdiv(NS, 4) * cdiv(W1, 16) + 4 * cdiv(W1, 16) * cdiv(W2, 16) + 4 * cdiv(W2, 16) * cdiv(W3, 16) + 4 * cdiv(W3, 16) * cdiv(NA, 16)) * 64 +
           16 * (cdiv(W1, 16) + cdiv(W2, 16) + cdiv(W3, 16) + cdiv(NA, 16));
}

