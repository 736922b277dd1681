// Shared definitions of the MFMA rollout kernels (rollout_mfma.hip: head-per-wave; rollout_coop.hip: cooperative heads).
#pragma once
#include "device_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Compile-time-only fence that ties K independent accumulator chains together: every chain's MFMA issued so far is an input of the
// (empty) statement and every later one consumes its output, so no instruction scheduler can run one chain ahead of the others and
// turn K interleaved independent chains into back-to-back DEPENDENT matrix instructions (seen in the single-tile instantiation of
// rollout_coop.hip: 8 dependent MFMAs in a row, each waiting out the full pipeline depth).  Emits no instruction.
template <int K> __device__ __forceinline__ void pin_order(f32x4 (&a)[K]) {
    // "v": rollout_coop.hip is compiled with -amdgpu-mfma-vgpr-form, MFMA results live in the vector half.  (With "a" the layer-1
    // results sat in accumulation registers and every ReLU needed a v_accvgpr_read first -- which queues behind the MFMA in flight:
    // the 20 layer-2 MFMAs took 80 cycles each instead of 32.)
    if constexpr (K == 5) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]));
    else {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("" : "+v"(a[k]));
    }
}

// max(x, 0) as ONE instruction (v_max_i32 on the bit pattern: negative floats, -0 included, are negative integers): fmaxf costs two on
// an MFMA result, a canonicalising v_max x, x first, and every float builtin is folded back into it.  NaNs with a clear sign bit
// propagate (as in the reference's tf.nn.relu); no inline assembly -- the compiler must see the read of the MFMA result to place
// the wait states the hardware does not interlock.
__device__ __forceinline__ float relu1(float x) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

template <int ENV> struct EnvDim;
template <> struct EnvDim<METRPO_ENV_SWIMMER>      { static constexpr int NS = 10, NA = 2, NDROP = 2; };
template <> struct EnvDim<METRPO_ENV_HALF_CHEETAH> { static constexpr int NS = 18, NA = 6, NDROP = 1; };
template <> struct EnvDim<METRPO_ENV_ANT>          { static constexpr int NS = 29, NA = 8, NDROP = 2; };
template <> struct EnvDim<METRPO_ENV_HOPPER>       { static constexpr int NS = 11, NA = 3, NDROP = 0; };
template <> struct EnvDim<METRPO_ENV_SNAKE>        { static constexpr int NS = 14, NA = 4, NDROP = 2; };

#ifndef XOR_SUM_PERMLANE
#define XOR_SUM_PERMLANE 0      // measured in the resident rollout (29 independent sums per post wave and step): +0.3 .. 0.9 % per rollout -- the LDS crossbar pipelines them better; the update kernels (one or two sums on a tile's dependent chain) take the vector-ALU form
#endif
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
// A value DEFINED in an accumulation register stays there: the matrix instructions read their A operand from either register file, while a value
// the allocator merely parks in the accumulation half (more than 256 live registers in a one-wave-per-SIMD kernel) is copied back before every
// use, and each of those copies queues behind the matrix instruction in flight (rollout_resident.hip: 10.2 -> 8.5 ms at 2 x 1024).  Not a
// general win: the cooperative rollout kernel (94 parked registers, 45 copies per step) got SLOWER with its fragments pinned there (0.430 -> 0.452 ms).
__device__ __forceinline__ float in_acc_reg(float v) { float a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }

constexpr int al4(int a) { return (a + 3) & ~3; }

template <int ENV, int DH, int PH>
struct Cfg {
    static constexpr int NS = EnvDim<ENV>::NS, NA = EnvDim<ENV>::NA, NDROP = EnvDim<ENV>::NDROP;
    static constexpr int NIN = NS + NA - NDROP;
    static constexpr int NIN_KS = cdiv(NIN, 4), NS_KS = cdiv(NS, 4);
    static constexpr int DH_CB = cdiv(DH, 16), PH_CB = cdiv(PH, 16), OUT_CB = cdiv(NS, 16);
    static constexpr int NSP = 16 * OUT_CB;                        // padded state row in the exchange buffer
    // resident dynamics layout of one head: W0 b0 W1 b1 W2 b2, every array on a 16-byte boundary (NetDesc, api.hip:build_net)
    static constexpr int dW0 = 0, db0 = al4(NIN * DH), dW1 = al4(db0 + DH), db1 = al4(dW1 + DH * DH), dW2 = al4(db1 + DH),
                         db2 = al4(dW2 + DH * NS), PD = al4(db2 + NS);
    // flat policy layout (rllab order): W0 b0 W1 b1 Wout bout log_std
    static constexpr int pW0 = 0, pb0 = NS * PH, pW1 = pb0 + PH, pb1 = pW1 + PH * PH, pW2 = pb1 + PH,
                         pb2 = pW2 + PH * NA, pLS = pb2 + NA;
    // per-wave LDS (floats): ST | NX | ACT | dyn biases (3 x padded) | policy biases (3 x padded)
    static constexpr int BD = 16 * DH_CB, BP = 16 * PH_CB;
    static constexpr int W_ST = 0, W_NX = W_ST + 16 * NS, W_ACT = W_NX + 16 * NS, W_BD0 = ((W_ACT + 16 * NA + 3) / 4) * 4,
                         W_BD1 = W_BD0 + BD, W_BD2 = W_BD1 + BD, W_BP0 = W_BD2 + NSP, W_BP1 = W_BP0 + BP,
                         W_BP2 = W_BP1 + BP, W_TOTAL = W_BP2 + 16;
};

__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave complete in issue order; this only stops the compiler from moving
    // LDS accesses of different lanes across the point.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float xor_sum(float v) {      // sum over the 4 lanes (e, q=0..3) of one env
#if XOR_SUM_PERMLANE
    // v_permlane16_swap / v_permlane32_swap (gfx950): the same butterfly on the vector ALU, bit for bit (policy_mfma.hip: xsum_q)
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    const u32x2_ a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const u32x2_ b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
#endif
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

