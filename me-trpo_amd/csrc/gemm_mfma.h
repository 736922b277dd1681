// Batched f32 GEMM on the matrix core (v_mfma_f32_32x32x2_f32, exact f32 fmaf chains) with fused epilogues.
// Shared by the step-wise rollout (rollout_gemm.hip) and the ensemble dynamics training (dyn_train.hip).
//   C[h] = epilogue( opA(A[h]) [M x Kd]  *  opB(W[h]) [Kd x N] )        h = blockIdx.z (head / model)
//   opA = A (row-major [M][Kd], lda) or A^T (A stored [Kd][M]);  opB = W (row-major [Kd][N], ldw) or W^T (stored [N][Kd])
// block = 256 threads = 2x2 waves; wave tile (32*TM) x (32*TN); block tile (64*TM) x (64*TN); BK = 16; LDS double buffer
#pragma once
#include <algorithm>
#include "device_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef GEMM_XCD_SWIZZLE
#define GEMM_XCD_SWIZZLE 1
#endif
#ifndef GEMM_A_KMAJOR
#define GEMM_A_KMAJOR 1
#endif

enum { EPI_BIAS_ID = 0, EPI_BIAS_RELU = 1, EPI_BIAS_TANH = 2, EPI_RELU_MASK = 3, EPI_ADAM = 4, EPI_PLAIN = 5, EPI_PARTIAL = 6, EPI_DTANH = 7, EPI_RELU_OUT = 8, EPI_RELU_OUT64 = 9, EPI_RELU_OUT48 = 10 };   // fused skinny layer of <= 32 / <= 64 / <= 48 columns

struct GemmEpi {                 // epilogue operands (unused fields may be null)
    const float* bias; long long strideBias;          // EPI_BIAS_*: bias[col]
    const float* mask; long long strideMask; int ldm; // EPI_RELU_MASK: C = acc * (mask[row][col] > 0);  EPI_DTANH: C = (acc + bias?) * (1 - mask[row][col]^2)
    float* am; float* av; long long strideAdam;       // EPI_ADAM: C is the weight matrix, updated in place; am/av same layout
    float lr_t, beta1, beta2, eps, decay;             //           lr_t = lr*sqrt(1-b2^t)/(1-b1^t); decay = lr*reg_constant (SGD on the regulariser)
    float* bvec; float* bam; float* bav;              // EPI_ADAM: bias of the same layer; its gradient = column sums of opB(W) (= dZ), which the
                                                      //           blockIdx.y == 0 blocks accumulate while streaming B (fixed order) and apply
    // EPI_PARTIAL (split-K for weight-gradient GEMMs with too few output tiles to fill the chip): blockIdx.z = head * splits + split,
    // split s contracts rows [s*kchunk, (s+1)*kchunk) and writes its M x N partial followed by the N partial column sums to
    // part + (s * heads + head) * stridePart; k_adam_apply (dyn_train.hip) adds the splits in index order and applies Adam.
    float* part; long long stridePart; int splits, kchunk;
    // EPI_RELU_OUT (64x64 or 128x128 tiles): the tile relu(acc + bias) is NOT stored; it is contracted on the spot with the next (skinny, <= 64
    // column) layer's weights w2[head][N][no] and the BM x no partial goes to part + (blockIdx.x * heads + head) * stridePart, row stride no --
    // one partial per column block of this layer, added (in block order, plus that layer's bias) by the consumer.  The hidden
    // activation matrix (M x N floats per head) is then never written to or read back from HBM.
    const float* w2; long long strideW2; int no;
};

// PD = prefetch distance of the global loads in k-tiles (register sets in flight).  1: the next tile's loads fly under this tile's MFMAs -- enough when
// eight workgroups per CU hide each other's latency.  4 (64x64 tiles only): for launches of a few workgroups per CU (M = 100 .. 1000 rows: the
// params-file shapes of the training step, the merged-round rollouts and the deterministic sweeps), where a k-tile's eight MFMAs per wave
// (0.2 us) cannot cover an L2 / HBM round trip and the kernel ran at the latency of one workgroup's 32-64 dependent k-tiles.  Same k order: same sums.
template <int TM, int TN, int EPI, bool TA, bool TB, bool AL, int PD = 1>
__global__ void __launch_bounds__(256) k_gemm_mfma(const float* __restrict__ A, long long strideA, int lda,
                                                   const float* __restrict__ W, long long strideW, int ldw,
                                                   float* __restrict__ C, long long strideC, int ldc, int M, int N, int Kd, GemmEpi ep) {
    constexpr int BM = 64 * TM, BN = 64 * TN, BK = 16;     // BK = 32 for the 64x64 tile (half the barriers, half the resident blocks): C3 step 0.204 -> 0.218 ms, not kept
    // A tile in LDS.  Row-major A (the forward layers: !TA) with aligned rows keeps its k-contiguous quads: As[m][k] (row stride BK + 4 floats: the
    // 16 lanes of a b128 pass hit 16 disjoint bank groups), written by ONE ds_write_b128 per staged quad instead of four scattered b32 stores and
    // read by one ds_read_b128 per four MFMA steps -- slot lk of step 4 h + e then contracts k = 8 h + 4 lk + e, and the B operand reads that row.
    // Measured (bench.py --config): 128x128 tiles +1.8 % (C4 5.14 -> 5.05 s); 64x64 tiles -3 % (C3 160 -> 165 ms: eight MFMAs per k-tile and wave do not
    // cover the longer b128 latency), so only the large tile takes it.
    constexpr bool AKM = GEMM_A_KMAJOR && !TA && AL && TM * TN == 4;
    constexpr int AS_ROWS = AKM ? BM : BK, AS_COLS = AKM ? BK + 4 : BM + 4;
    __shared__ __attribute__((aligned(16))) float As[2][AS_ROWS][AS_COLS];   // As[k][m], or As[m][k] (AKM)
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + 4];     // Bs[k][n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (each with its own 4 MB L2), so in launch order the
    // column blocks that share an A row panel land on 8 different L2s and every one of them fetches the panel from beyond L2.  Remapped, each
    // XCD works through a CONTIGUOUS run of the logical tile order (column block fastest, then row block, then head): the panel is fetched
    // once per XCD and the head's weights stay in that L2.  Bijective for any workgroup count; a placement hint only, never a correctness matter.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (GEMM_XCD_SWIZZLE) {
        const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const unsigned L = bx + gx * (by + gy * bz), xcd = L & 7u, q8 = nwg >> 3, r8 = nwg & 7u;
        const unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
        bx = (int)(t % gx); by = (int)((t / gx) % gy); bz = (int)(t / (gx * gy));
    }
    int head = bz, kbeg = 0, kend = Kd;
    if (EPI == EPI_PARTIAL) {
        const int sp = bz % ep.splits;
        head = bz / ep.splits;
        kbeg = sp * ep.kchunk; kend = min(Kd, kbeg + ep.kchunk);
        C = ep.part + ((size_t)sp * (gridDim.z / ep.splits) + head) * ep.stridePart;
    } else {
        C += (size_t)head * strideC;
    }
    const int m0 = by * BM, n0 = bx * BN;
    A += (size_t)head * strideA; W += (size_t)head * strideW;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // "strided" staging (source is [X][k] with k contiguous: float4 along k, scattered into S[k][x]) for A when !TA and for B when TB;
    // "direct" staging (source is [k][X] with X contiguous: float4 along X, stored as float4) for A when TA and for B when !TB.
    constexpr int SA_P = BM / 64, SB_P = BN / 64;                          // strided staging: 64 rows x 4 quads per pass
    constexpr int DA_R = 256 / (BM / 4), DA_P = BK / DA_R;                 // direct A: rows per pass, passes
    constexpr int DB_R = 256 / (BN / 4), DB_P = BK / DB_R;                 // direct B
    constexpr int NA = TA ? DA_P : SA_P, NB = TB ? SB_P : DB_P;
    float4 ra[PD][NA], rb[PD][NB];

    // AL (host-checked: every base pointer, leading dimension and head stride is a multiple of 4 floats and so are the extents along the
    // contiguous axes): a quad is either fully inside or fully outside -> one predicated 16-byte load, no alignment test, no scalar tail.
    auto ld4 = [](const float* src, int have) -> float4 {               // up to 4 valid floats starting at src
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AL) { if (have > 0) v = *(const float4*)src; return v; }
        if (have >= 4 && (((uintptr_t)src) & 15) == 0) return *(const float4*)src;
        if (have > 0) v.x = src[0];
        if (have > 1) v.y = src[1];
        if (have > 2) v.z = src[2];
        if (have > 3) v.w = src[3];
        return v;
    };
    auto load_tiles = [&](int k0, int set) {                       // `set` is a constant wherever this is inlined (unrolled callers)
        if (!TA) {
#pragma unroll
            for (int p = 0; p < SA_P; ++p) {                       // 4 consecutive lanes read the 4 quads (64 B) of one row
                const int kq = (tid & 3) * 4, m = m0 + (tid >> 2) + p * 64;
                ra[set][p] = (m < M) ? ld4(A + (size_t)m * lda + k0 + kq, kend - (k0 + kq)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int p = 0; p < DA_P; ++p) {
                const int k = k0 + tid / (BM / 4) + p * DA_R, m = m0 + (tid % (BM / 4)) * 4;
                ra[set][p] = (k < kend) ? ld4(A + (size_t)k * lda + m, M - m) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (!TB) {
#pragma unroll
            for (int p = 0; p < DB_P; ++p) {
                const int k = k0 + tid / (BN / 4) + p * DB_R, n = n0 + (tid % (BN / 4)) * 4;
                rb[set][p] = (k < kend) ? ld4(W + (size_t)k * ldw + n, N - n) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int p = 0; p < SB_P; ++p) {
                const int kq = (tid & 3) * 4, n = n0 + (tid >> 2) + p * 64;
                rb[set][p] = (n < N) ? ld4(W + (size_t)n * ldw + k0 + kq, kend - (k0 + kq)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tiles = [&](int buf, int set) {
        if (!TA) {
#pragma unroll
            for (int p = 0; p < SA_P; ++p) {
                const int kq = (tid & 3) * 4, m = (tid >> 2) + p * 64;
                if constexpr (AKM) *(float4*)&As[buf][m][kq] = ra[set][p];
                else { As[buf][kq + 0][m] = ra[set][p].x; As[buf][kq + 1][m] = ra[set][p].y; As[buf][kq + 2][m] = ra[set][p].z; As[buf][kq + 3][m] = ra[set][p].w; }
            }
        } else {
#pragma unroll
            for (int p = 0; p < DA_P; ++p) *(float4*)&As[buf][tid / (BM / 4) + p * DA_R][(tid % (BM / 4)) * 4] = ra[set][p];
        }
        if (!TB) {
#pragma unroll
            for (int p = 0; p < DB_P; ++p) *(float4*)&Bs[buf][tid / (BN / 4) + p * DB_R][(tid % (BN / 4)) * 4] = rb[set][p];
        } else {
#pragma unroll
            for (int p = 0; p < SB_P; ++p) {
                const int kq = (tid & 3) * 4, n = (tid >> 2) + p * 64;
                Bs[buf][kq + 0][n] = rb[set][p].x; Bs[buf][kq + 1][n] = rb[set][p].y; Bs[buf][kq + 2][n] = rb[set][p].z; Bs[buf][kq + 3][n] = rb[set][p].w;
            }
        }
    };

    const int nk = (kend - kbeg + BK - 1) / BK;
    const bool colsum = (EPI == EPI_PARTIAL && by == 0) || (EPI == EPI_ADAM && by == 0 && ep.bvec != nullptr);
    float csum = 0.0f;                                           // thread (tid % BN, tid / BN): column sum over its k slice
    constexpr int CS_S = 256 / BN, CS_K = BK / CS_S;             // k slices per tile, rows per slice
    // register set s holds tile kt with kt % PD == s from its load (issued PD iterations before its use) to its LDS store
    load_tiles(kbeg, 0);
    store_tiles(0, 0);
#pragma unroll
    for (int u = 1; u < PD; ++u) if (u < nk) load_tiles(kbeg + u * BK, u);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        const int kt = kt0 + u;
        if (kt >= nk) continue;
        const int buf = kt & 1;
        if (kt + PD < nk) load_tiles(kbeg + (kt + PD) * BK, u);           // set u = kt % PD is free: tile kt went to LDS an iteration ago; these loads fly under PD tiles' MFMAs
        const int li = lane & 31, lk = lane >> 5;
        if constexpr (AKM) {
#pragma unroll
            for (int h8 = 0; h8 < BK / 8; ++h8) {
                float4 a4[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) a4[i] = *(const float4*)&As[buf][wm * 32 * TM + i * 32 + li][8 * h8 + 4 * lk];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bf[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][8 * h8 + 4 * lk + e][wn * 32 * TN + j * 32 + li];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float af = (e == 0) ? a4[i].x : (e == 1) ? a4[i].y : (e == 2) ? a4[i].z : a4[i].w;
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = As[buf][kk + lk][wm * 32 * TM + i * 32 + li];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Bs[buf][kk + lk][wn * 32 * TN + j * 32 + li];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        }
        if (colsum) {
#pragma unroll
            for (int kk = 0; kk < CS_K; ++kk) csum += Bs[buf][(tid / BN) * CS_K + kk][tid % BN];
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1, (u + 1) % PD);
        __syncthreads();
    }
    }
    if (colsum) {                                                // bias gradient + Adam (tf.train.AdamOptimizer) for columns n0..n0+BN
        float* red = &As[0][0][0];                               // all MFMA reads of As are behind the last barrier
        red[tid] = csum;
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float g = 0.0f;
#pragma unroll
            for (int sl = 0; sl < CS_S; ++sl) g += red[sl * BN + tid];
            if (EPI == EPI_PARTIAL) { C[(size_t)M * N + n0 + tid] = g; }
            else {
            const size_t bi = (size_t)head * ep.strideAdam + n0 + tid;
            const float m1 = ep.beta1 * ep.bam[bi] + (1.0f - ep.beta1) * g;
            const float v1 = ep.beta2 * ep.bav[bi] + (1.0f - ep.beta2) * g * g;
            ep.bam[bi] = m1; ep.bav[bi] = v1;
            const float w = ep.bvec[bi];
            ep.bvec[bi] = w - ep.lr_t * m1 / (sqrtf(v1) + ep.eps) - ep.decay * w;
            }
        }
    }
    if constexpr (EPI == EPI_RELU_OUT || EPI == EPI_RELU_OUT64 || EPI == EPI_RELU_OUT48) {
        static_assert(TM == TN, "fused skinny output layer: square block tiles");
        // The block's BN columns are handled in TN chunks of 64.  Per chunk: stage 1, H = relu(acc + bias) -> LDS, k-major for the second product
        // (k = the chunk's 64 columns: rows 0..31 in As, 32..63 in Bs; every read of the main loop is behind its last barrier); stage 2, wave w ->
        // rows 16 TM w .. 16 TM (w + 1) - 1, NT = ceil(no / 16) column tiles, 16 k-steps of v_mfma_f32_16x16x4_f32 accumulating over the chunks.
        constexpr int NT = (EPI == EPI_RELU_OUT) ? 2 : (EPI == EPI_RELU_OUT48) ? 3 : 4, NOP = 16 * NT, W2U = 64 * NOP / 256;     // column tiles, padded columns, W2 elements per thread and chunk
        __shared__ float W2s[64][NOP + 1];
        typedef float f32x4_ __attribute__((ext_vector_type(4)));
        float* Hlo = &As[0][0][0]; float* Hhi = &Bs[0][0][0];                      // each 32 x (BM + 4) floats
        static_assert(2 * AS_ROWS * AS_COLS >= 32 * (BM + 4), "A tile too small for the fused output layer's staging");
        const int c16 = lane & 15, q4 = lane >> 4;
        f32x4_ o[TM][NT];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) o[i][ct] = f32x4_{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < TN; ++ch) {
            float w2r[W2U];                                                        // the chunk's slice of the output weights: all loads in flight at once,
#pragma unroll
            for (int u = 0; u < W2U; ++u) {                                        // behind the LDS traffic of stage 1
                const int e = tid + 256 * u, k = e / NOP, jj = e % NOP;
                w2r[u] = (jj < ep.no) ? ep.w2[(size_t)head * ep.strideW2 + (size_t)(n0 + ch * 64 + k) * ep.no + jj] : 0.0f;
            }
            if (ch > 0) __syncthreads();                                           // stage 2 of the previous chunk has read Hs / W2s
            if (TN == 1 || wn == ch) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = (TN == 1 ? wn * 32 : j * 32) + (lane & 31);
                    const float bv = ep.bias[(size_t)head * ep.strideBias + n0 + ch * 64 + nl];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float* dst = (nl < 32 ? Hlo + nl * (BM + 4) : Hhi + (nl - 32) * (BM + 4)) + wm * 32 * TM + i * 32 + 4 * (lane >> 5);
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2)] = fmaxf(acc[i][j][r] + bv, 0.0f);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < W2U; ++u) { const int e = tid + 256 * u; W2s[e / NOP][e % NOP] = w2r[u]; }
            __syncthreads();
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int k = 4 * s2 + q4;
                const float* hrow = (k < 32 ? Hlo + k * (BM + 4) : Hhi + (k - 32) * (BM + 4)) + wave * 16 * TM + c16;
                float av[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = hrow[16 * i];
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    const float bw = W2s[k][16 * ct + c16];
#pragma unroll
                    for (int i = 0; i < TM; ++i) o[i][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bw, o[i][ct], 0, 0, 0);
                }
            }
        }
        float* P = ep.part + ((size_t)bx * gridDim.z + head) * ep.stridePart;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                            // D layout: row 4 q + r, col c
                const int row = m0 + wave * 16 * TM + 16 * i + 4 * q4 + r;
                if (row < M) {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) if (16 * ct + c16 < ep.no) P[(size_t)row * ep.no + 16 * ct + c16] = o[i][ct][r];
                }
            }
        return;
    }
    // epilogue: C/D layout of 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * 32 * TN + j * 32 + (lane & 31);
            float bv = 0.0f;
            if (EPI <= EPI_BIAS_TANH || (EPI == EPI_DTANH && ep.bias != nullptr)) bv = (col < N) ? ep.bias[(size_t)head * ep.strideBias + col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < M && col < N) {
                    float v = acc[i][j][r];
                    const size_t ci = (size_t)row * ldc + col;
                    if (EPI == EPI_BIAS_ID) v += bv;
                    else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bv, 0.0f);
                    else if (EPI == EPI_BIAS_TANH) v = tanh_fast(v + bv);
                    else if (EPI == EPI_RELU_MASK) v = (ep.mask[(size_t)head * ep.strideMask + (size_t)row * ep.ldm + col] > 0.0f) ? v : 0.0f;
                    else if (EPI == EPI_DTANH) { const float hm = ep.mask[(size_t)head * ep.strideMask + (size_t)row * ep.ldm + col]; v = (v + bv) * fmaf(-hm, hm, 1.0f); }
                    if (EPI == EPI_ADAM) {                       // tf.train.AdamOptimizer update of one weight, gradient = v
                        const size_t ai = (size_t)head * ep.strideAdam + ci;
                        const float m1 = ep.beta1 * ep.am[ai] + (1.0f - ep.beta1) * v;
                        const float v1 = ep.beta2 * ep.av[ai] + (1.0f - ep.beta2) * v * v;
                        ep.am[ai] = m1; ep.av[ai] = v1;
                        const float w = C[ci];
                        C[ci] = w - ep.lr_t * m1 / (sqrtf(v1) + ep.eps) - ep.decay * w;
                    } else {
                        C[ci] = v;
                    }
                }
            }
        }
}

template <int TM, int TN, int EPI, bool TA, bool TB>
static inline void gemm_mfma_launch(const float* A, long long sA, int lda, const float* W, long long sW, int ldw, float* C, long long sC,
                                    int ldc, int M, int N, int Kd, int heads, const GemmEpi& ep, hipStream_t st) {
    dim3 grid((N + 64 * TN - 1) / (64 * TN), (M + 64 * TM - 1) / (64 * TM), heads * (EPI == EPI_PARTIAL ? ep.splits : 1));
    auto m4 = [](long long v) { return (v & 3) == 0; };
    // contiguous axes: A rows run along Kd (or along M when TA), W rows along N (or along Kd when TB)
    const bool al = m4((long long)(uintptr_t)A >> 2) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)W & 15) == 0) && m4(sA) && m4(sW) && m4(lda) && m4(ldw) &&
                    m4(TA ? M : Kd) && m4(TB ? Kd : N) && (EPI != EPI_PARTIAL || m4(ep.kchunk));
    // few workgroups per CU and a long contraction: deeper prefetch (see k_gemm_mfma's PD)
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    if constexpr (TM * TN == 1) {
        if (nblocks <= 1280 && Kd >= 256) {
            if (al) hipLaunchKernelGGL((k_gemm_mfma<TM, TN, EPI, TA, TB, true, 4>), grid, dim3(256), 0, st, A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, ep);
            else hipLaunchKernelGGL((k_gemm_mfma<TM, TN, EPI, TA, TB, false, 4>), grid, dim3(256), 0, st, A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, ep);
            return;
        }
    }
    if (al) hipLaunchKernelGGL((k_gemm_mfma<TM, TN, EPI, TA, TB, true>), grid, dim3(256), 0, st, A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, ep);
    else hipLaunchKernelGGL((k_gemm_mfma<TM, TN, EPI, TA, TB, false>), grid, dim3(256), 0, st, A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, ep);
}

// Tile choice shared by every caller: 128x128 block tiles when that already yields enough workgroups to fill the 256 CUs a few
// times over, otherwise halve the tile in M and then in N (more, smaller workgroups beat idle CUs on the small-batch shapes:
// M = 500 rows x N = 512 x 5 heads is 80 tiles of 128x128 but 320 of 64x64).
template <int EPI, bool TA, bool TB>
static inline void gemm_auto(const float* A, long long sA, int lda, const float* W, long long sW, int ldw, float* C, long long sC, int ldc, int M,
                             int N, int Kd, int heads, const GemmEpi& ep, hipStream_t st) {
    auto blocks = [&](int tm, int tn) { return (long long)((M + 64 * tm - 1) / (64 * tm)) * ((N + 64 * tn - 1) / (64 * tn)) * heads; };
    int tm = (M > 64) ? 2 : 1, tn = (N > 64) ? 2 : 1;
    const long long want = 2048;            // ~8 workgroups per CU (measured: 64x64 tiles beat 128x128 by 15 % at 800 tiles, 128x128 wins from ~8k tiles)
    if (tm == 2 && blocks(tm, tn) < want) tm = 1;
    if (tn == 2 && blocks(tm, tn) < want) tn = 1;
    if (tm == 2 && tn == 2) gemm_mfma_launch<2, 2, EPI, TA, TB>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
    else if (tm == 2) gemm_mfma_launch<2, 1, EPI, TA, TB>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
    else if (tn == 2) gemm_mfma_launch<1, 2, EPI, TA, TB>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
    else gemm_mfma_launch<1, 1, EPI, TA, TB>(A, sA, lda, W, sW, ldw, C, sC, ldc, M, N, Kd, heads, ep, st);
}

// Hidden layer + skinny (<= 64 column) output layer in one launch (EPI_RELU_OUT), on the square tile shape gemm_auto would pick for the hidden
// layer: 64x64 (small batches: C0-params-file, C2, C3 shapes) or 128x128 (C4).  Returns the tile factor (1 or 2), 0 if not applicable.
static inline int gemm_fused_out_tile(int M, int N, int heads, int no) {
    auto blocks = [&](int tm, int tn) { return (long long)((M + 64 * tm - 1) / (64 * tm)) * ((N + 64 * tn - 1) / (64 * tn)) * heads; };
    if (no > 64) return 0;
    if (M > 64 && N > 64 && (N % 128) == 0 && blocks(2, 2) >= 2048) return 2;                 // same rule as gemm_auto -> <2, 2>
    if ((N % 64) == 0 && blocks(2, 1) < 2048 && blocks(1, 2) < 2048) return 1;                 // -> <1, 1>
    return 0;
}
static inline size_t gemm_fused_out_part_floats(int M, int N, int heads, int no, int tile) { return (size_t)(N / (64 * tile)) * heads * ((((size_t)M * no) + 3) & ~(size_t)3); }
// part receives N / (64 tile) partials per head ({splits, stridePart} as reported by gemm_skinny_bias's defer)
static inline void gemm_relu_fused_out(int tile, const float* A, long long sA, int lda, const float* W, long long sW, int ldw, const float* bias, long long sBias,
                                       const float* W2, long long sW2, int no, int M, int N, int Kd, int heads, float* part, hipStream_t st, int* splits,
                                       long long* stridePart) {
    GemmEpi ep = {};
    ep.bias = bias; ep.strideBias = sBias; ep.w2 = W2; ep.strideW2 = sW2; ep.no = no;
    ep.part = part; ep.stridePart = (((long long)M * no) + 3) & ~3LL;
#define FUSED_OUT_LAUNCH(T_, E_) gemm_mfma_launch<T_, T_, E_, false, false>(A, sA, lda, W, sW, ldw, nullptr, 0, N, M, N, Kd, heads, ep, st)
    if (tile == 2) { if (no <= 32) FUSED_OUT_LAUNCH(2, EPI_RELU_OUT); else if (no <= 48) FUSED_OUT_LAUNCH(2, EPI_RELU_OUT48); else FUSED_OUT_LAUNCH(2, EPI_RELU_OUT64); }
    else { if (no <= 32) FUSED_OUT_LAUNCH(1, EPI_RELU_OUT); else if (no <= 48) FUSED_OUT_LAUNCH(1, EPI_RELU_OUT48); else FUSED_OUT_LAUNCH(1, EPI_RELU_OUT64); }
#undef FUSED_OUT_LAUNCH
    *splits = N / (64 * tile); *stridePart = ep.stridePart;
}

// ---- skinny output layers (N <= 64 columns, long contraction): C[h] = bias[h] + A[h] . W[h] as split-K partials + an ordered reduce ----
// A 64-column tile per (64 rows, head) leaves most CUs idle and every workgroup walks the whole K axis; splitting K multiplies the
// workgroups and shortens each walk (the layer is bound by reading A).  Partials are summed in split order: deterministic.
static __global__ void k_splitk_bias_reduce(int splits, int heads, long long stridePart, const float* __restrict__ part, int M, int N,
                                     const float* __restrict__ bias, long long strideBias, float* __restrict__ C, long long strideC, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int head = blockIdx.y;
    if (i >= (long long)M * N) return;
    float v = bias ? bias[(size_t)head * strideBias + (int)(i % N)] : 0.0f;
    for (int s = 0; s < splits; ++s) v += part[((size_t)s * heads + head) * stridePart + i];
    C[(size_t)head * strideC + i] = (act == 1) ? fmaxf(v, 0.0f) : (act == 2) ? tanh_fast(v) : v;
}

// split-K pays when the launch has few output tiles and a long contraction: skinny output layers (N <= 64) and small-batch hidden layers
static inline int skinny_splits(int M, int N, int Kd, int heads) {
    if (Kd < 256) return 1;
    const long long blocks = (long long)((M + 63) / 64) * ((N + 63) / 64) * heads;
    if (blocks >= 512) return 1;
    int want = (int)std::min<long long>(8, std::max<long long>(1, 1024 / std::max<long long>(blocks, 1)));
    want = std::min(want, Kd / 128);
    return std::max(want, 1);
}
static inline size_t skinny_part_floats(int M, int N, int Kd, int heads) {
    const int s = skinny_splits(M, N, Kd, heads);
    return s > 1 ? (size_t)s * heads * ((((size_t)M * N + N) + 3) & ~(size_t)3) : 0;
}
// part: workspace of >= skinny_part_floats(...) floats (may be NULL when that is 0).  act: 0 identity, 1 relu, 2 tanh (METRPO_ACT_*).
// defer != NULL: when the layer runs as split-K partials, skip the reduce launch and report {splits, stridePart} instead -- the
// caller's next kernel adds the partials and the bias itself (rollout_gemm.hip: k_big_post); {0, 0} when C was written directly.
struct SkinnyDefer { int splits; long long stridePart; };
static inline void gemm_skinny_bias(const float* A, long long sA, int lda, const float* W, long long sW, int ldw, const float* bias, long long sBias,
                                    float* C, long long sC, int M, int N, int Kd, int heads, float* part, hipStream_t st, int act = 0,
                                    SkinnyDefer* defer = nullptr) {
    if (defer) { defer->splits = 0; defer->stridePart = 0; }
    const int S = skinny_splits(M, N, Kd, heads);
    if (S <= 1 || part == nullptr) {
        GemmEpi ep = {}; ep.bias = bias; ep.strideBias = sBias;
        if (act == 1) gemm_auto<EPI_BIAS_RELU, false, false>(A, sA, lda, W, sW, ldw, C, sC, N, M, N, Kd, heads, ep, st);
        else if (act == 2) gemm_auto<EPI_BIAS_TANH, false, false>(A, sA, lda, W, sW, ldw, C, sC, N, M, N, Kd, heads, ep, st);
        else gemm_auto<EPI_BIAS_ID, false, false>(A, sA, lda, W, sW, ldw, C, sC, N, M, N, Kd, heads, ep, st);
        return;
    }
    GemmEpi ep = {};
    ep.part = part; ep.stridePart = (((long long)M * N + N) + 3) & ~3LL; ep.splits = S; ep.kchunk = (((Kd + S - 1) / S) + 15) & ~15;
    ep.splits = (Kd + ep.kchunk - 1) / ep.kchunk;
    gemm_mfma_launch<1, 1, EPI_PARTIAL, false, false>(A, sA, lda, W, sW, ldw, nullptr, 0, N, M, N, Kd, heads, ep, st);
    if (defer && act == 0) { defer->splits = ep.splits; defer->stridePart = ep.stridePart; return; }
    const long long tot = (long long)M * N;
    hipLaunchKernelGGL(k_splitk_bias_reduce, dim3((unsigned)((tot + 255) / 256), heads), dim3(256), 0, st, ep.splits, heads, ep.stridePart, part, M, N,
                       bias, sBias, C, sC, act);
}
