// Persistent, evenly split ("stream-K") f32 MFMA kernel for the wide layers of the dynamics ensemble (training.py:171-214: h <- relu(h W + b) per
// hidden layer, identity output layer; all K heads of model_based_rl.py:91-97 in one launch).  Replaces, for layers of >= 256 units, the
// tile-per-workgroup GEMM of gemm_mfma.h on the step-wise rollout path (rollout_gemm.hip):
//
//   * a WAVE owns 16 rows (envs) x 256 columns of the layer's output for the whole contraction: its 16 accumulator tiles of
//     v_mfma_f32_16x16x4_f32 are computed TRANSPOSED (D[n][m] = sum_k W[k][n] H[m][k]: srcA = weight fragment, srcB = activation), so a lane
//     holds row m = lane % 16 in every tile and the D fragment of one layer is the srcB operand of the next without leaving registers:
//         x (global, 16 rows per wave) -> layer 0 (producer: 2 x S0 matrix instructions per 32-unit chunk, bias as one more input row)
//           -> relu -> main layer (128 matrix instructions per chunk) -> bias + relu -> output layer (64 x OT) -> [rows][<= 64] partial
//     Only WEIGHTS go through LDS; the activations of a 2-hidden-layer net never touch HBM.
//   * a workgroup = 8 waves = 128 rows sharing the weight chunks: W1[32 k][256 n] (+ the layer-0 slice of those 32 units) per chunk, copied by
//     the LDS DMA (global_load_lds_dwordx4: no register round trip, no ds_write) into a ring of FOUR stages, three chunks ahead.  Chunk q + 1 is
//     complete behind the barrier that ended chunk q - 1, so the first operand reads of a chunk are issued ahead of the barrier in front of it.
//     Operand reads: one ds_read_b128 per four matrix instructions (a lane's four consecutive columns belong to four different accumulator
//     tiles), conflict-free without padding, issued one step (16 matrix instructions) ahead of their use.
//   * the two waves of a SIMD take turns on its matrix pipe (waves 0-3 run at raised priority): per chunk, a wave has 128 matrix instructions
//     and ~100 instructions of bookkeeping (next copies, next activation rows, the producer).  The PRIORITY wave does its bookkeeping BEHIND its
//     matrix instructions, the other one IN FRONT of them -- each in the shadow of its partner's matrix phase; with the same order in both, the
//     two bookkeeping phases coincide behind the barrier and the pipe idles through them (first version of this kernel: 70 % busy).
//   * the grid is one workgroup per CU and every workgroup gets the same number of chunks, whatever the tile count (200 tiles on 256 CUs at the
//     C3 share ran 78 % of the chip; 7.8 tiles per CU at C4 lost 2 % to the last round): a workgroup's range of the (tile, chunk) sequence
//     starts and ends inside tiles.  The piece that BEGINS a tile is computed first and its accumulators are handed to the next workgroup
//     through HBM (write-through stores + a per-wave flag); the piece that ENDS a tile is computed last, starting from the accumulators the
//     previous workgroup exported long before -- so every output is still ONE k-ordered fmaf chain, bit for bit what an unsplit tile computes,
//     and nobody waits for a workgroup that was dispatched after it (no co-residency assumption: a producer never waits before it exports
//     unless its whole range lies inside one tile, and then only for lower-numbered workgroups).
//
//   * FEWER tiles than CUs (a.late; small batches: the Humanoid params file has 80 tiles): cut that way a tile would be three or four pieces in a row, each
//     waiting for the one before it.  Instead every piece starts from ZERO, so the pieces of a tile run at the same time on different CUs; every piece but
//     the tile's last exports ITS OWN sums at the end of its range, and the last piece imports them all at its end and adds them to its own, first piece
//     first (round 5: until then each piece added the running sum of its predecessor and handed it on -- a chain of 2-3 hand-overs of ~9 us behind the
//     last matrix instruction, 71.7 us against the tile GEMM's 69.9 us at the Humanoid params file's shape; now ONE hand-over).  An output is then
//     ((p_last + p0) + p1) + ... of k-ordered chains: still a fixed order for a given grid, no longer independent of the grid.
//     Waits are again only for lower-numbered workgroups, and only at the end of a range.
//
// Summation order of one output: chunks of 32 k in order; inside a chunk the steps (j, e) = (0,0) .. (1,3), step (j, e) adding
// k = 16 j + e + {0, 4, 8, 12} in that order (the four lane groups of the matrix instruction) -- fixed, independent of grid and split.
#pragma once
#ifndef SK_BAR2
// 1: ONE workgroup barrier per TWO chunks.  A chunk's copies then go two entries ahead (three with a barrier per chunk) -- the stage they overwrite was last
// read two chunks ago and a barrier lies in between whichever parity --, the barrier stands behind the ODD chunks, and an odd chunk does not read the next
// chunk's first operands ahead of it (that entry was copied during the even chunk before it: complete for everybody only behind the barrier).
#define SK_BAR2 1
#endif
#ifndef SK_COPY_TOP
#define SK_COPY_TOP 0
#endif
#ifndef SK_PRIO
#define SK_PRIO 0
#endif
#ifndef SK_FETCH_ONLY
#define SK_FETCH_ONLY 0
#endif
#include <type_traits>
#include <utility>
#include <algorithm>
#include "mfma_common.h"

enum { SK_A_GLOBAL = 0, SK_A_PRODUCER = 1 };      // main layer's input: activations [M][K1] in HBM | computed from x by layer 0 on the fly
enum { SK_EPI_STORE = 0, SK_EPI_OUT = 1 };        // relu(acc + b1) stored row-major | contracted with the (<= 64 column) output layer, partial per column block
enum { SKF_ZERO = 1, SKF_IMPORT = 2, SKF_EXPORT = 4, SKF_LAST = 8, SKF_EPI = 16, SKF_EPILAST = 32, SKF_NONE = 64, SKF_NEWTILE = 128, SKF_LATE = 256,
       SKF_NOPROD = 512, SKF_ARRIVE = 1024 };   // mlp_persist.h: the entry reuses the layer-0 activations of the entry before it | last entry of a tile

struct SkRec;
struct SkArgs {
    int M, heads, K1, N;                       // rows per head; main layer W1[K1][N], K1 % 32 == 0, N % 256 == 0
    int tm;                                    // rows per tile (SkForm<NW>::TM; set by sk_plan)
    const float* A; long long strideA; int lda;   // SK_A_GLOBAL: A[head][M][lda]; SK_A_PRODUCER: x[M][lda], x[n_in] = 1 (bias slot), zeros up to 4 S0
    const float* W0; long long strideW0;       // SK_A_PRODUCER: layer-0 weights [>= 4 S0 rows][K1] row-major, row n_in = its bias (the resident layout)
    const float* W1; long long strideW1;
    const float* b1; long long strideB1;       // SK_EPI_STORE
    float* C; long long strideC; int ldc;      // SK_EPI_STORE
    const float* epi;                          // SK_EPI_OUT: images [head][CB][E][SK epi floats] (k_sk_epi_image)
    float* part; long long stridePart; int ldp;   // SK_EPI_OUT: part[(cb * heads + head) * stridePart + m * ldp + o], ldp = 16 OT
    float* xacc; unsigned* xflag; unsigned epoch; double* err;   // accumulator hand-over: [grid][8 waves][16 tiles][64 lanes] float4, one flag per (workgroup, wave)
#ifdef SK_DEBUG                                // tools/ubench/mlp_sk_bench.hip
    int skip;                                  // leave-one-out timing (results invalid): 1 no LDS-DMA copies, 2 no activation loads, 4 no barrier, 8 no schedule fetch
    unsigned long long* dbg;                   // non-NULL: per-workgroup {shader cycles, 100 MHz ticks, entries} of the main loop
    unsigned long long* dbg2;                  // non-NULL: phase stamps (shader clock) of chunks 8 .. 11 of workgroup 0, waves 0 and 4: [wave / 4][chunk - 8][6]
#endif
    const int* hdr; const SkRec* recs;         // schedule: entries per workgroup | [grid][sched_cap] records (k_sk_sched)
    int RB, CB, NCk, L, tiles, sched_cap;      // row blocks of tm rows, column blocks of 256, chunks per tile, units per tile (chunks + epilogue weight)
    int xcd;                                   // > 1: workgroup b is taken to run on XCD b % xcd and gets range (b % xcd) * (grid / xcd) + b / xcd of the unit sequence (below)
    int team;                                  // XCD teams (below): the workgroups of one XCD walk the tiles in lock step, sharing weight slices AND activation rows in their L2
    int late;                                  // FEWER tiles than workgroups: the pieces of a tile run side by side from zero and are ADDED at their ends (below)
    long long units;
};

template <int OT> struct SkEpi {
    static constexpr int E = (OT == 4) ? 2 : 1, UPC = 4 / E;                 // EPI chunks per tile; 64-column groups u per EPI chunk
    static constexpr int FLOATS = 256 + UPC * 1024 * OT;                      // [b1 part, 256 floats][W2 part]
};
// Form of the workgroup.  NW = 8: ONE workgroup of 8 waves per CU, a tile of 128 rows, a ring of four chunk stages (copies two or three chunks ahead).
// NW = 4 (round 6): TWO workgroups of 4 waves per CU, a tile of 64 rows each, a ring of TWO stages each (copies one chunk ahead: 2 x <= 37 KB per workgroup,
// two workgroups in the CU's 160 KB).  The two waves of a SIMD then belong to DIFFERENT workgroup barriers: whatever one of them waits for -- the barrier,
// its copies' drain, the first operand reads behind the barrier, a schedule fetch -- the other one's matrix instructions fill.  Same per-wave code, same
// k order of every sum; a weight chunk feeds 64 rows instead of 128 (twice the L2 -> LDS copy traffic per FLOP; operand reads unchanged).
template <int NW> struct SkForm {
    static_assert(NW == 8 || NW == 4, "8 waves x 1 workgroup per CU, or 4 waves x 2");
    static constexpr int TM = 16 * NW, NST = (NW == 8) ? 4 : 2, WPC = 8 / NW, NPW = 32 / NW;      // rows per tile | ring stages | workgroups per CU | W1 rows (1 KB pieces) a wave copies per chunk
    static constexpr bool BAR2 = (NW == 8) && (SK_BAR2 != 0);
    static constexpr int AHEAD = (NW == 8) ? (BAR2 ? 2 : 3) : 1;
};
template <int N, class F, int... I> __device__ __forceinline__ void sk_static_for_(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void sk_static_for(F&& f) { sk_static_for_<N>(f, std::make_integer_sequence<int, N>{}); }
template <int AMODE, int EPI, int S0, int OT> struct SkGeom {
    static constexpr int NI0 = (S0 + 1) / 2;                                  // 1 KB pieces of the layer-0 slice of a chunk
    static constexpr int W0F = (AMODE == SK_A_PRODUCER) ? NI0 * 256 : 0;
    static constexpr int B1O = 8192 + W0F;                                    // SK_EPI_STORE: the tile's 256 bias values ride in its last chunk's stage
    static constexpr int STAGE = (EPI == SK_EPI_STORE) ? B1O + 256 : (W0F > 256 ? 8192 + W0F : 8192 + 256);
};

__device__ __forceinline__ void sk_glds16(const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// Image of the epilogue operands of one (head, column block, EPI chunk): b1 in the lane order of the accumulators, the output layer's weights as
// srcA fragments read by ds_read_b128 (four steps v per read).  One launch per weight version (rollout_gemm.hip: once per launch chain).
template <int OT>
__global__ void k_sk_epi_image(const float* __restrict__ b1, long long strideB1, const float* __restrict__ W2, long long strideW2, int no, int N, int heads,
                               float* __restrict__ img) {
    using EP = SkEpi<OT>;
    const int CB = N / 256;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, tot = (long long)heads * CB * EP::E * EP::FLOATS;
    if (idx >= tot) return;
    const int f = (int)(idx % EP::FLOATS), e = (int)((idx / EP::FLOATS) % EP::E), cb = (int)((idx / ((long long)EP::FLOATS * EP::E)) % CB),
              head = (int)(idx / ((long long)EP::FLOATS * EP::E * CB));
    float v = 0.0f;
    if (f < 256) {
        const int ul = f >> 6, g = (f >> 4) & 3, r = (f >> 2) & 3, vv = f & 3;
        if (ul < EP::UPC) v = b1[(size_t)head * strideB1 + cb * 256 + 64 * (e * EP::UPC + ul) + 16 * g + 4 * r + vv];
    } else {
        int p = f - 256;
        const int vv = p & 3; p >>= 2;
        const int i = p & 15; p >>= 4;
        const int ot = p % OT; p /= OT;
        const int g = p & 3; p >>= 2;
        const int r = p & 3, ul = p >> 2;
        const int n = cb * 256 + 64 * (e * EP::UPC + ul) + 16 * g + 4 * r + vv, o = 16 * ot + i;
        if (o < no) v = W2[(size_t)head * strideW2 + (size_t)n * no + o];
    }
    img[idx] = v;
}

// One entry of a workgroup's schedule (32 bytes, read with ONE s_load_dwordx8): everything the per-chunk bookkeeping would otherwise derive with
// integer divisions.  Built by k_sk_sched once per shape into global memory, so that the main kernel reads it through the SCALAR cache: on a SIMD
// whose older wave is issuing matrix instructions back to back, the younger wave's VECTOR instructions do not issue at all (measured: its ~100
// instructions of bookkeeping took exactly as long as the partner's 128 MFMAs), while scalar, LDS and memory instructions do.
struct SkRec { int w; int m0; unsigned offA, offC;          // first half: what the chunk itself and the look-ahead need.  w = chunk | flags << 16; element offsets
               int w2; unsigned offW1, offW0; int tile; };   // second half: what the LDS-DMA copies need (w2 = w)
typedef int sk_i32x4 __attribute__((ext_vector_type(4)));

template <int AMODE, int EPI, int S0, int OT>
__global__ void __launch_bounds__(512) k_sk_sched(const SkArgs a, int* __restrict__ hdr, SkRec* __restrict__ recs) {
    using EP = SkEpi<OT>;
    constexpr int E = EP::E;
    constexpr bool PROD = (AMODE == SK_A_PRODUCER), OUT = (EPI == SK_EPI_OUT);
    // this workgroup's range of the unit sequence, cut into pieces; entry t of the schedule is computed by thread t
    const int tid = threadIdx.x;
    const long long U = a.units;
    const int G = gridDim.x, b = blockIdx.x, NCk = a.NCk;
    // XCD-aware ranges (a.xcd = 8 on MI355X; the dispatcher deals workgroups to the XCDs round-robin): the workgroups of ONE XCD get CONSECUTIVE ranges, so
    // the tiles that share a weight slice (same head and column block: consecutive tiles) are worked on by CUs behind the same L2 and the slice is
    // fetched from HBM once instead of once per XCD.  The unit sequence is cut into a.xcd parts at TILE boundaries (nothing is handed over between
    // XCD groups), each part evenly over its grid / xcd workgroups; inside a part range i belongs to workgroup i * xcd + x, so a piece is still handed
    // to a HIGHER-numbered workgroup (blockIdx + xcd): nobody waits for a workgroup dispatched after it.
    // XCD TEAMS (a.team; activations read from memory: SK_A_GLOBAL): with ranges of consecutive tiles per workgroup, the 32 workgroups behind one L2 sit at 32
    // different places of their tiles -- a weight slice or a 128-row block of activations fetched by one of them has left the 4 MB L2 (48 MB stream through it per
    // tile time) before the next one wants it, and the C4 share moved 3.1 GB per launch for 0.6 GB of operands.  Here the unit of the split is a SUPER-TILE = the
    // GP = grid / 8 tiles {PP = GP / CB consecutive (head, row block) pairs} x {CB column blocks}; slot i = blockIdx / 8 of every XCD owns tile (pair i / CB,
    // column block i % CB) of each super-tile, the XCDs split the super-tile sequence evenly (ranges start and end inside super-tiles), and all slots of an XCD
    // have the SAME range: they step through the same chunks at the same time, so a weight chunk is fetched once for the PP workgroups that share the column
    // block and an activation chunk once for the CB that share the row block.  Pieces are handed to the SAME slot of the NEXT XCD (blockIdx + 1: still a
    // higher-numbered workgroup; through memory, as before); every output remains one k-ordered chain.
    const int tph = a.CB * a.RB;
    bool team_idle_tail = false; int team_pairs = 0, team_PP = 1, team_slot = 0;
    long long u0, u1;
    if (a.team) {
        const int GP = G / 8, x = b % 8;
        team_slot = b / 8; team_PP = GP / a.CB; team_pairs = a.heads * a.RB;
        const long long n_st = (team_pairs + team_PP - 1) / team_PP, SU = n_st * a.L;
        u0 = SU * x / 8; u1 = SU * (x + 1) / 8;
        team_idle_tail = true;
    } else if (a.xcd > 1) {
        const int GP = G / a.xcd, x = b % a.xcd, i = b / a.xcd;
        auto part = [&](int xx) -> long long { return xx >= a.xcd ? U : ((U * xx / a.xcd + a.L / 2) / a.L) * a.L; };
        const long long s0 = part(x), s1 = part(x + 1);
        u0 = s0 + (s1 - s0) * i / GP; u1 = s0 + (s1 - s0) * (i + 1) / GP;
    } else { u0 = U * b / G; u1 = U * (b + 1) / G; }
    auto cut = [&](long long u, int& t, int& o) { t = (int)(u / a.L); o = (int)(u % a.L); if (o >= NCk) { ++t; o = 0; } };
    int ts, cs, te, ce;
    cut(u0, ts, cs); cut(u1, te, ce);
    const int Le = NCk + (OUT ? E : 0);
    int n_h = 0, nf = 0, tf = ts, n_t = 0;
    if (ts == te) n_h = ce > cs ? ce - cs : 0;                                 // one piece inside one tile: chunks [cs, ce)
    else { n_h = ce; tf = ts + (cs > 0 ? 1 : 0); nf = te - tf; n_t = cs > 0 ? (NCk - cs) + (OUT ? E : 0) : 0; }
    const int nq = n_h + nf * Le + n_t;
    if (tid == 0) hdr[b] = nq;
    for (int t = tid; t < nq + 4; t += 512) {
        SkRec r = {};
        if (t >= nq) { r.w = r.w2 = (SKF_NONE | SKF_EPI) << 16; r.tile = -1; recs[(size_t)b * a.sched_cap + t] = r; continue; }     // four sentinels behind the last entry
        int tile, c, fl = 0;                                                   // c: chunk, or EPI chunk index when fl & SKF_EPI
        if (t < n_h) {                                                         // the piece that begins a tile: first, exported at once
            tile = te; c = (ts == te ? cs : 0) + t;
            if (t == 0) fl |= (c == 0 || a.late) ? SKF_ZERO : SKF_IMPORT;
            if (t == n_h - 1) fl |= SKF_EXPORT;                                                        // (a.late: a middle piece exports ITS OWN sums, the tile's last piece adds them all)
        } else if (t < n_h + nf * Le) {
            const int tt = t - n_h; tile = tf + tt / Le; c = tt % Le;
            if (c == 0) fl |= SKF_ZERO;
            if (c == NCk - 1) fl |= SKF_LAST;
            if (c >= NCk) { c -= NCk; fl = SKF_EPI | (c == E - 1 ? SKF_EPILAST : 0); }
        } else {                                                               // the piece that ends a tile: last, continues the previous workgroup's sums
            const int tt = t - n_h - nf * Le; tile = ts; c = cs + tt;
            if (tt == 0) fl |= a.late ? SKF_ZERO : SKF_IMPORT;
            if (c == NCk - 1) fl |= SKF_LAST | (a.late ? SKF_LATE : 0);
            if (c >= NCk) { c -= NCk; fl = SKF_EPI | (c == E - 1 ? SKF_EPILAST : 0); }
        }
        int head = tile / tph, cb = (tile / a.RB) % a.CB, rb = tile % a.RB;
        if (a.team) {                                                          // `tile` is a super-tile: this slot's tile of it (none beyond the last pair: an idle entry)
            const int pair = tile * team_PP + team_slot / a.CB;
            cb = team_slot % a.CB; head = pair / a.RB; rb = pair % a.RB;
            if (pair >= team_pairs) { r.w = r.w2 = (SKF_NONE | SKF_EPI) << 16; r.tile = -1; recs[(size_t)b * a.sched_cap + t] = r; continue; }
            tile = (head * a.CB + cb) * a.RB + rb;
        }
        if (t == 0 || t == n_h || (t >= n_h && t < n_h + nf * Le && (t - n_h) % Le == 0) || t == n_h + nf * Le) fl |= SKF_NEWTILE;   // first entry of a piece
        r.w = r.w2 = c | (fl << 16); r.m0 = rb * a.tm;
        r.offA = (unsigned)(head * a.strideA) + ((!PROD && !(fl & SKF_EPI)) ? 32u * c : 0u);
        r.offC = OUT ? (unsigned)(((long long)cb * a.heads + head) * a.stridePart) : (unsigned)(head * a.strideC + cb * 256);
        if (fl & SKF_EPI) { r.offW1 = (unsigned)(((head * a.CB + cb) * E + c) * EP::FLOATS); r.offW0 = 0; }
        else {
            r.offW1 = (unsigned)(head * a.strideW1 + (long long)(32 * c) * a.N + cb * 256);
            r.offW0 = PROD ? (unsigned)(head * a.strideW0 + 32 * c) : (unsigned)(head * a.strideB1 + cb * 256);
        }
        r.tile = tile;
        recs[(size_t)b * a.sched_cap + t] = r;
    }
}

// one 1 KB LDS-DMA piece: lane's 16 bytes at sbase + voff -> LDS lds_addr + 16 lane.  Scalar base + 32-bit lane offset: no vector address
// arithmetic per piece (hipcc folds a per-lane offset into a 64-bit VGPR base with one v_lshl_add_u64 per piece).  Invisible to hipcc's vmcnt
// counting: drained by the explicit waits of the chunk loop.  M0 is saved and restored inside the statement.
#ifdef SK_DEBUG
__device__ int sk_dbg_noldswrite = 0;                                         // experiment: the copies fetch their 16 bytes into a dead register instead of LDS (results invalid)
#endif
__device__ __forceinline__ void sk_glds16_s(unsigned voff, const void* sbase, unsigned lds_addr) {
#if defined(SK_DEBUG) && SK_FETCH_ONLY
    { f32x4 dead; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dead) : "v"(voff), "s"(sbase) : "memory"); return; }
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int AMODE, int EPI, int S0, int OT, int NW = 8>
__global__ void __launch_bounds__(64 * NW, 8 / NW) k_mlp_sk(const SkArgs a) {      // (two waves per SIMD either way: without the second bound the 4-wave form is given 512 registers and keeps its accumulators in the accumulation half, a move in and out around every matrix instruction)
    using EP = SkEpi<OT>;
    using GE = SkGeom<AMODE, EPI, S0, OT>;
    using FM = SkForm<NW>;
    constexpr int E = EP::E, UPC = EP::UPC, NI0 = GE::NI0, B1O = GE::B1O, STAGE = GE::STAGE, NST = FM::NST, NPW = FM::NPW, AHEAD = FM::AHEAD;
    constexpr bool BAR2 = FM::BAR2;
    constexpr bool PROD = (AMODE == SK_A_PRODUCER), OUT = (EPI == SK_EPI_OUT);
    static_assert(EP::FLOATS <= STAGE || !OUT, "EPI image larger than a ring stage");
    static_assert(NI0 <= 8, "layer-0 slice: at most 8 one-KB pieces (one or two per wave)");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const ring = lds;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = a.hdr[blockIdx.x];
    if (nq == 0) return;
#ifdef SK_DEBUG
    if (a.dbg != nullptr && tid == 0) a.dbg[8 * blockIdx.x + 3] = wall_clock64();                  // kernel entry
#endif
    struct RecL { int kc, fl, m0; unsigned offA, offC; };                      // the chunk itself / the look-ahead
    struct RecI { int fl; unsigned offW1, offW0; };                            // its LDS-DMA copies
    const SkRec* const my_recs = a.recs + (size_t)blockIdx.x * a.sched_cap;
    auto decL = [](const sk_i32x4& v) { RecL r; r.kc = v[0] & 0xFFFF; r.fl = v[0] >> 16; r.m0 = v[1]; r.offA = (unsigned)v[2]; r.offC = (unsigned)v[3]; return r; };
    auto decI = [](const sk_i32x4& v) { RecI r; r.fl = v[0] >> 16; r.offW1 = (unsigned)v[1]; r.offW0 = (unsigned)v[2]; return r; };
    // Schedule entries by scalar loads, load + wait in ONE statement (q <= nq + 3: sentinels behind the last entry).  Left to hipcc the loads sink to
    // their first use or become vector loads + readfirstlane; issued in one statement and waited for in another, their destination registers
    // are fair game for a spill BEFORE the data lands (seen with 34 spilled SGPRs: garbage offsets, memory faults).  The ~200 cycles this
    // statement blocks its wave are the partner wave's to fill: the two halves of the workgroup fetch at different points of the matrix phase.
    auto fetch2 = [&](int ql, int qi, RecL& rl, RecI& ri) {
        sk_i32x4 vl, vi;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(vl), "=&s"(vi) : "s"(my_recs + ql), "s"(my_recs + qi));
        rl = decL(vl); ri = decI(vi);
    };

    // per-thread constants of the LDS-DMA copies (byte offsets from a scalar base)
    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ring;
    const unsigned w1_voff = (unsigned)(wave * a.N + 4 * lane) * 4u;           // row (wave + NW t) of the chunk, this lane's 16 bytes
    unsigned w0_voff = 0;                                                      // PROD: piece ii = wave (+ NW: 32 more rows of W0) of the layer-0 slice image [s][jt][g][i']
    if constexpr (PROD) { const int p = (wave % (NI0 < NW ? NI0 : NW)) * 64 + lane, i4 = p & 3, gg = (p >> 2) & 3, jt = (p >> 4) & 1, s = p >> 5; w0_voff = (unsigned)((4 * s + gg) * a.K1 + 16 * jt + 4 * i4) * 4u; }
    const unsigned lane16 = (unsigned)lane * 16u;
    // LDS-DMA copies of entry q into ring stage q % NST, as NPW + 1 PIECES per wave (spread over the matrix phase of the chunk that issues them)
    auto issue_piece = [&](const RecI& r, int q, auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t > NPW) return;
        if (r.fl & SKF_NONE) return;
#ifdef SK_DEBUG
        if (a.skip & 1) return;
        if ((a.skip & 32) && (t & 1)) return;                                  // half the copy traffic (the stage keeps older, equally random rows)
#endif
        const unsigned st = ring_lds + (unsigned)((q & (NST - 1)) * STAGE) * 4u;
        if (!(r.fl & SKF_EPI)) {
            if constexpr (t < NPW) sk_glds16_s(w1_voff, (const char*)(a.W1 + r.offW1) + (size_t)t * ((size_t)a.N * (4 * NW)), st + (unsigned)(wave + NW * t) * 1024u);   // row wave + NW t
            else {
                if constexpr (!OUT) { if ((r.fl & SKF_LAST) && wave == NW - 1) sk_glds16_s(lane16, a.b1 + r.offW0, st + B1O * 4u); }
                if constexpr (PROD) {
                    if (wave < NI0) sk_glds16_s(w0_voff, a.W0 + r.offW0, st + (8192u + (unsigned)wave * 256u) * 4u);
                    if constexpr (NI0 > NW) { if (wave + NW < NI0) sk_glds16_s(w0_voff, a.W0 + r.offW0 + (size_t)(8 * NW) * a.K1, st + (8192u + (unsigned)(wave + NW) * 256u) * 4u); }   // piece ii + NW: W0 rows 8 NW further on
                }
            }
        } else if constexpr (OUT) {
            constexpr int NIE = EP::FLOATS / 256;
            static_assert(NIE <= NW * (NPW + 1), "EPI image: at most NPW + 1 pieces per wave");
            const int ii = wave + NW * t;
            if (ii < NIE) sk_glds16_s(lane16, (const char*)(a.epi + r.offW1) + (size_t)ii * 1024, st + (unsigned)ii * 1024u);
        }
    };
    auto issue = [&](const RecI& r, int q) { sk_static_for<NPW + 1>([&](auto tc) { issue_piece(r, q, tc); }); };
#ifdef SK_DEBUG
    auto drain_vm = [&]() { if (a.skip & 64) return; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); };   // (64: leave-one-out, no wait for the copies)
#else
    auto drain_vm = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); };   // the asm copies | the loads hipcc counts
#endif

    f32x4 acc[4][4];                                                           // [u][v]: columns 64 u + 16 g + 4 r + v of the block, row i
    f32x4 oacc[OT];
    // srcB of a chunk's 8 steps: set q & 1 feeds chunk q; SK_A_GLOBAL fills the other set for chunk q + 1 meanwhile.  (With one set + a copy per chunk the
    // copy's eight v_mov were, on the SIMD's younger wave, eight waits for a gap in the older wave's matrix stream.)
    f32x4 hs[2][2];
    float xr[PROD ? S0 : 1];
    (void)oacc; (void)xr;

    auto load_x = [&](const RecL& r) {
        const int m = min(r.m0 + wave * 16 + i, a.M - 1);
        const float* xp = a.A + r.offA + (size_t)m * a.lda + g;
#pragma unroll
        for (int s = 0; s < S0; ++s) xr[s] = xp[4 * s];
    };
    unsigned a_voff = 0;                                                       // SK_A_GLOBAL: byte offset of this lane's row quad inside the head's activations, per tile
    auto set_a_voff = [&](const RecL& r) { a_voff = (unsigned)(min(r.m0 + wave * 16 + i, a.M - 1) * a.lda + 4 * g) * 4u; };
    auto load_a = [&](const RecL& r, f32x4 (&dst)[2]) {                        // two 16-byte loads, scalar base + lane offset: no vector address arithmetic; waited for by drain_a
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:64" : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(a_voff), "s"(a.A + r.offA) : "memory");
    };
    auto drain_a = [&](f32x4 (&dst)[2]) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst[0]), "+v"(dst[1]) :: "memory"); };
    auto produce = [&](int q, f32x4 (&dst)[2]) {                               // layer 0 for the 32 units of chunk q, relu, in srcB layout
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        const float* w0 = ring + (q & (NST - 1)) * STAGE + 8192 + lane;
#pragma unroll
        for (int s = 0; s < S0; ++s) {
            d0 = MFMA16(w0[(2 * s) * 64], xr[s], d0);
            d1 = MFMA16(w0[(2 * s + 1) * 64], xr[s], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { dst[0][r] = relu1(d0[r]); dst[1][r] = relu1(d1[r]); }
    };

    // srcA operands: one ds_read_b128 = four accumulator tiles of one step.  A chunk is 32 GROUPS of {wait for this group's operand, request the one
    // four groups ahead, four MFMAs}: two non-matrix instructions in the shadow of four matrix instructions, no vector ALU work at all.  Inline
    // assembly, because hipcc sinks compiler-visible reads down to their first use (read, wait the whole LDS latency, four MFMAs, read ...); it
    // therefore does not count them either: the waits are written out (in order, four reads in flight: lgkmcnt(3) = the oldest has landed; scalar
    // loads in flight can only make that wait stricter).  w[k & 7] feeds group k.
    f32x4 w[8];
    const unsigned lane_off = (unsigned)((4 * g) * 256 + 4 * i) * 4u;
    auto stage_addr = [&](int q) { return ring_lds + (unsigned)((q & (NST - 1)) * STAGE) * 4u + lane_off; };
#define SK_WOFF(k) (((16 * ((k) >> 4) + (((k) >> 2) & 3)) * 256 + 64 * ((k) & 3)) * 4)
    auto first4 = [&](unsigned addr) {                                         // groups 0 .. 3 of a chunk, complete on return
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                     : "v"(addr), "i"(SK_WOFF(0)), "i"(SK_WOFF(1)), "i"(SK_WOFF(2)), "i"(SK_WOFF(3)));
    };

    // ---- prologue: entries 0 .. 2 under way, the first rows loaded ----
    RecL r0, r1; RecI r3;
    {   // entries 0 .. 3 by one statement (one wait instead of four scalar-load round trips: the prologue is 3-4 % of a 130 us launch)
        sk_i32x4 l0, l1, i0, i1, i2, i3;
        asm volatile("s_load_dwordx4 %0, %6, 0x0\n\ts_load_dwordx4 %1, %6, 0x20\n\ts_load_dwordx4 %2, %6, 0x10\n\ts_load_dwordx4 %3, %6, 0x30\n\t"
                     "s_load_dwordx4 %4, %6, 0x50\n\ts_load_dwordx4 %5, %6, 0x70\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(l0), "=&s"(l1), "=&s"(i0), "=&s"(i1), "=&s"(i2), "=&s"(i3) : "s"(my_recs));
        r0 = decL(l0); r1 = decL(l1); r3 = AHEAD == 1 ? decI(i1) : (AHEAD == 2 ? decI(i2) : decI(i3));      // the first AHEAD entries' copies under way; r3 = the record of the next one
        issue(decI(i0), 0); if (AHEAD >= 2) issue(decI(i1), 1); if (AHEAD >= 3) issue(decI(i2), 2);
    }
    if constexpr (PROD) load_x(r0);
    else { set_a_voff(r0); load_a(r0, hs[0]); drain_a(hs[0]); }
    drain_vm();
    __syncthreads();
    first4(stage_addr(0));
#ifdef SK_DEBUG
    unsigned long long dbg_c0 = 0, dbg_w0 = 0;
    if (a.dbg != nullptr) { dbg_c0 = __builtin_readcyclecounter(); dbg_w0 = wall_clock64(); }
#endif

    bool export_issued = false;
    const int prev_d = a.team ? 1 : (a.xcd > 1 ? a.xcd : 1);
    constexpr int XW = NW;                                                     // hand-over slots and flags per workgroup                   // the workgroup that holds the range in front of this one (teams: the same slot of the previous XCD)
    auto wait_prev = [&](int dist = 0) {                                       // the export flag of this launch of the previous workgroup (or the one `dist` in front; bounded: report, do not hang)
        const unsigned* fp = a.xflag + (size_t)(blockIdx.x - (dist > 0 ? dist : prev_d)) * XW + wave;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000ull) { if (lane == 0) *a.err = 1.0; break; }      // 2 s at 100 MHz
        }
    };
    auto body = [&](auto par_, const int q) {
        constexpr int PAR = decltype(par_)::value;
        // PRE: the next chunk's first operands are read AHEAD of the barrier that ends this chunk (its stage is complete since an earlier barrier).  Not behind
        // the odd chunks of the barrier-per-two-chunks form, and never with the two-stage ring (the next chunk's stage is being filled during this chunk).
        constexpr bool PRE = (NW == 8) && !(BAR2 && PAR == 1);
        f32x4 (&h)[2] = hs[PAR];
        f32x4 (&hn)[2] = hs[PAR ^ 1];
        const bool nmain = !(r1.fl & SKF_EPI);
        const float* st = ring + (q & (NST - 1)) * STAGE;
#ifdef SK_DEBUG
        const bool stamp = a.dbg2 != nullptr && blockIdx.x == 0 && (wave & 3) == 0 && q >= 8 && q < 12;
        unsigned long long* sp_ = a.dbg2 + ((wave >> 2) * 4 + (q - 8)) * 6;
#define SK_STAMP(k) do { if (stamp) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) sp_[k] = t_; } } while (0)
#else
#define SK_STAMP(k) do { } while (0)
#endif
        SK_STAMP(0);
#if SK_PRIO                                                                    // experiment: the two workgroups of a CU take turns at raised priority, chunk by chunk
        if (SK_PRIO == 1 || blockIdx.x < gridDim.x / 2) __builtin_amdgcn_s_setprio(PAR ? 1 : 0); else __builtin_amdgcn_s_setprio(PAR ? 0 : 1);
#endif
#if SK_COPY_TOP                                                                // experiment: 1 = the SIMD's younger waves (4 .. 7) issue the whole chunk's copies HERE, 2 = all waves
        const bool top_copy = (SK_COPY_TOP == 2 || wave >= 4) && !(r0.fl & SKF_EPI);
        if (top_copy) issue(r3, q + AHEAD);
#else
        constexpr bool top_copy = false;
#endif
        // bookkeeping: the vector part here; the schedule entries q + 3 (to copy) and q + 2 (the next look-ahead) are fetched and the copies issued INSIDE the
        // matrix phase below (scalar + memory instructions: two or three per group of four MFMAs cost nothing)
        if constexpr (PROD) {
            if (!(r0.fl & SKF_EPI)) produce(q, h);                             // xr: loaded a chunk ago, drained at that chunk's end
            if (nmain && (r1.fl & SKF_NEWTILE)) load_x(r1);
        } else {
            if (nmain) { if (r1.fl & SKF_NEWTILE) set_a_voff(r1); load_a(r1, hn); }
        }
        RecI r4; RecL rn;                                                      // fetched inside the matrix phase: copies of entry q + 4 (issued NEXT chunk), look-ahead q + 2
        SK_STAMP(1);

        if (!(r0.fl & SKF_EPI)) {
            if (r0.fl & SKF_ZERO) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (r0.fl & SKF_IMPORT) {                                          // the previous workgroup's partial sums of this tile (exported at its start)
                wait_prev();
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xacc + ((size_t)(blockIdx.x - prev_d) * XW + wave) * 4096), 0, 16384, 0x00020000);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rs, ((u * 4 + v) * 64 + lane) * 16, 0, 16);   // sc1: served beyond this CU's L1
                        acc[u][v] = __builtin_bit_cast(f32x4, raw);
                    }
                __builtin_amdgcn_s_waitcnt(0x0F70);                            // inside the branch: at the merge point hipcc would otherwise drain the queue on EVERY chunk
            }
            {
                const unsigned cur_a = stage_addr(q), nxt_a = stage_addr(q + 1);
                auto group = [&](auto kc_) {                                   // operand of group k + 4 (the NEXT chunk's first groups from k = 28 on: its stage is
                    constexpr int k = decltype(kc_)::value;                    // complete since the barrier that ended chunk q - 1) requested in front of group k's MFMAs
                    constexpr int j = k >> 4, e = (k >> 2) & 3, u = k & 3;
                    if constexpr (k < 28)
                        asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %0, %2 offset:%3" : "=&v"(w[(k + 4) & 7]), "+v"(w[k & 7]) : "v"(cur_a), "i"(SK_WOFF(k + 4)));
                    else if (nmain && PRE)
                        asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %0, %2 offset:%3" : "=&v"(w[(k + 4) & 7]), "+v"(w[k & 7]) : "v"(nxt_a), "i"(SK_WOFF(k - 28)));
                    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[k & 7]) : "i"(31 - k));
                    __builtin_amdgcn_sched_barrier(0);                         // the MFMAs below stay BEHIND the statement (hipcc hoists register-only instructions past asm)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] = MFMA16(w[k & 7][v], h[j][e], acc[u][v]);
                    __builtin_amdgcn_sched_barrier(0);
                };
#define SK_G4(b) group(std::integral_constant<int, (b)>{}); group(std::integral_constant<int, (b) + 1>{}); group(std::integral_constant<int, (b) + 2>{}); group(std::integral_constant<int, (b) + 3>{});
                // the copies of entry q + 3 go out EARLY in the phase (they are drained at its end: issued in its last groups they were waited for, 300-1000
                // cycles per chunk); the schedule fetch (it blocks its wave ~200 cycles) late, at different points for the two waves of a SIMD
                // (NW = 4: nine pieces per wave, two per slot)
#define SK_P(sl) if (!top_copy) { issue_piece(r3, q + AHEAD, std::integral_constant<int, (8 / NW) * (sl)>{}); if constexpr (NW == 4) issue_piece(r3, q + AHEAD, std::integral_constant<int, (8 / NW) * (sl) + 1>{}); }
                SK_G4(0)  SK_P(0);
                SK_G4(4)  SK_P(1);
                SK_G4(8)  SK_P(2);
                SK_G4(12) SK_P(3);
                SK_G4(16) SK_P(4);
                SK_G4(20) if (wave < NW / 2) fetch2(q + 2, q + AHEAD + 1, rn, r4);
                SK_G4(24) if (wave >= NW / 2) fetch2(q + 2, q + AHEAD + 1, rn, r4);
                SK_G4(28)
                // the next chunk's first operands must be COMPLETE before the loop's back edge: hipcc takes an asm read's destination as written when the
                // statement ends and may copy those registers where control flow merges (seen: half-landed copies behind an EPI chunk, 1 % errors)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            }
            SK_STAMP(2);
            if (r0.fl & SKF_LATE) {                                            // a.late, the piece that ENDS a tile: the sums of ALL earlier pieces of the tile, added to its own
                // The earlier pieces sit in the workgroups b - n_prev .. b - 1 (consecutive ranges: no XCD remap with a.late), each exported at the end of
                // its range -- all at about the same time, so the tile closes ONE hand-over behind its last matrix instruction instead of a chain of
                // n_prev (each link: store + drain + flag + four load passes, ~9 us).  Order of the adds: own, then the pieces from the tile's first on.
                int n_prev = 1;
                {
                    const int G = (int)gridDim.x, b = (int)blockIdx.x;
                    const long long U = a.units, u_me = U * b / G;
                    int ts_ = (int)(u_me / a.L); if ((int)(u_me % a.L) >= a.NCk) ++ts_;
                    for (; n_prev < b; ++n_prev) {                             // b - n_prev holds the piece that BEGINS the tile: its range starts in an earlier tile or on chunk 0
                        const long long u = U * (b - n_prev) / G;
                        int t_ = (int)(u / a.L), o_ = (int)(u % a.L);
                        if (o_ >= a.NCk) { ++t_; o_ = 0; }
                        if (t_ < ts_ || o_ == 0) break;
                    }
                }
                for (int d = n_prev; d >= 1; --d) {
                    wait_prev(d);
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xacc + ((size_t)(blockIdx.x - d) * XW + wave) * 4096), 0, 16384, 0x00020000);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                              // 16 registers at a time
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        u32x4 raw[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) raw[v] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((u * 4 + v) * 64 + lane) * 16, 0, 16);
                        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[u][v] += __builtin_bit_cast(f32x4, raw[v]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (r0.fl & SKF_EXPORT) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.xacc + ((size_t)blockIdx.x * XW + wave) * 4096), 0, 16384, 0x00020000);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[u][v]), rs, ((u * 4 + v) * 64 + lane) * 16, 0, 16);   // sc1: write-through
                    }
                export_issued = true;                                          // the flag follows behind this chunk's drain (below): no separate wait for the stores
            }
            if constexpr (!OUT) {
                if (r0.fl & SKF_LAST) {
                    const int m = r0.m0 + wave * 16 + i;
                    const float* bp = st + B1O + 16 * g;
                    float* cp = a.C + r0.offC + (size_t)m * a.ldc + 16 * g;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 bq = *(const f32x4*)(bp + 64 * u + 4 * r);
                            f32x4 o;
#pragma unroll
                            for (int v = 0; v < 4; ++v) o[v] = relu1(acc[u][v][r] + bq[v]);
                            if (m < a.M) *(f32x4*)(cp + 64 * u + 4 * r) = o;
                        }
                }
            }
        } else if constexpr (OUT) {
            issue(r3, q + AHEAD);
            fetch2(q + 2, q + AHEAD + 1, rn, r4);
            if (!(r0.fl & SKF_NONE)) {
                auto epi_chunk = [&](auto ee) {
                    constexpr int EE = decltype(ee)::value;
                    if (EE == 0) {
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) oacc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int ul = 0; ul < UPC; ++ul) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 bq = *(const f32x4*)(st + (ul * 4 + g) * 16 + 4 * r);
                            f32x4 wf[OT];
#pragma unroll
                            for (int ot = 0; ot < OT; ++ot) wf[ot] = *(const f32x4*)(st + 256 + ((((ul * 4 + r) * 4 + g) * OT + ot) * 16 + i) * 4);
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float hb = relu1(acc[EE * UPC + ul][v][r] + bq[v]);
#pragma unroll
                                for (int ot = 0; ot < OT; ++ot) oacc[ot] = MFMA16(wf[ot][v], hb, oacc[ot]);
                            }
                        }
                    }
                };
                if (E == 1 || r0.kc == 0) epi_chunk(std::integral_constant<int, 0>{});
                else epi_chunk(std::integral_constant<int, E - 1>{});
                if (r0.fl & SKF_EPILAST) {
                    const int m = r0.m0 + wave * 16 + i;
                    float* pp = a.part + r0.offC + (size_t)m * a.ldp + 4 * g;
                    if (m < a.M) {
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) *(f32x4*)(pp + 16 * ot) = oacc[ot];
                    }
                }
                if (nmain && PRE) first4(stage_addr(q + 1));
            }
        }
        SK_STAMP(3);
        if constexpr (!PROD) drain_a(hn);                                      // (nothing to wait for behind the line below; names the registers the asm loads wrote)
        drain_vm();                                                            // this wave's copies of entry q + 3 (issued early in the chunk) and its rows: landed long ago
        if (export_issued) {                                                   // ... and its accumulator stores: every storing wave has drained before its flag.  (A counted
            // wait that left the 16 stores in flight -- vmcnt(16) -- let the barrier pass with LDS-DMA copies still under way: stores and loads do not
            // retire in one order; seen as a rollout that was not bitwise repeatable at the C4 share.)
            if (lane == 0) __hip_atomic_store(a.xflag + (size_t)blockIdx.x * XW + wave, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            export_issued = false;
        }
        SK_STAMP(4);
#ifdef SK_DEBUG
        if (!(a.skip & 4))
#endif
        if (!BAR2 || PAR == 1) __builtin_amdgcn_s_barrier();                   // entry q + AHEAD (BAR2: q + 1, q + 2) complete in LDS for everybody; everybody is done reading stage q % NST
        asm volatile("" ::: "memory");
        if (!PRE && nmain) first4(stage_addr(q + 1));
        SK_STAMP(5);
        r0 = r1; r1 = rn; r3 = r4;
    };
    for (int q = 0; q < nq; q += 2) {                                          // unrolled by two: the register set of the srcB operands alternates
        body(std::integral_constant<int, 0>{}, q);
        if (q + 1 < nq) body(std::integral_constant<int, 1>{}, q + 1);
    }

#ifdef SK_DEBUG
    if (a.dbg != nullptr && lane == 0) {                                       // shader cycles and 100 MHz ticks of this workgroup's main loop, its chunk count
        if (tid == 0) { a.dbg[8 * blockIdx.x] = __builtin_readcyclecounter() - dbg_c0; a.dbg[8 * blockIdx.x + 1] = wall_clock64() - dbg_w0; a.dbg[8 * blockIdx.x + 2] = (unsigned long long)nq;
                        a.dbg[8 * blockIdx.x + 4] = dbg_w0; }
        atomicMax(&a.dbg[8 * blockIdx.x + 5], wall_clock64());                 // exit of the last wave
    }
#endif
}

// ---- host side ----
struct SkPlan { int grid; size_t lds_bytes; size_t xacc_floats; int nflags; size_t sched_bytes; };
template <int AMODE, int EPI, int S0, int OT, int NW = 8>
static inline SkPlan sk_plan(SkArgs& a, int n_sm, int grid_override = 0) {
    using GE = SkGeom<AMODE, EPI, S0, OT>;
    using FM = SkForm<NW>;
    n_sm *= FM::WPC;                                                           // workgroups the chip holds at a time
    a.tm = FM::TM;
    a.RB = (a.M + FM::TM - 1) / FM::TM; a.CB = a.N / 256; a.NCk = a.K1 / 32;
    a.L = a.NCk + ((EPI == SK_EPI_OUT) ? SkEpi<OT>::E : 1);
    a.tiles = a.heads * a.CB * a.RB;
    a.units = (long long)a.tiles * a.L;
    SkPlan p;
    p.grid = grid_override > 0 ? grid_override : ((a.tiles < n_sm && !a.late) ? a.tiles : n_sm);
    if (a.xcd > 1 && (a.late || p.grid % a.xcd != 0 || a.tiles < p.grid)) a.xcd = 0;      // whole XCD groups, at least a tile per workgroup
    // teams: 8 XCDs x GP slots, GP a multiple of the column blocks, at least ~2 super-tiles per XCD
    if (a.team && (AMODE != SK_A_GLOBAL || a.late || a.xcd != 8 || p.grid % 8 != 0 || (p.grid / 8) % a.CB != 0 || a.tiles < 2LL * p.grid)) a.team = 0;
    long long cap = a.units / p.grid + 2LL * a.L + 8;
    if (a.team) { const long long PP = (p.grid / 8) / a.CB, n_st = ((long long)a.heads * a.RB + PP - 1) / PP; cap = n_st * a.L / 8 + 2LL * a.L + 8; }
    a.sched_cap = (int)cap;
    p.lds_bytes = (size_t)FM::NST * GE::STAGE * sizeof(float);
    p.xacc_floats = (size_t)p.grid * NW * 4096;
    p.nflags = p.grid * NW;
    p.sched_bytes = (size_t)p.grid * cap * sizeof(SkRec) + (((size_t)p.grid * sizeof(int) + 255) & ~(size_t)255);
    return p;
}
// schedule of this shape into `mem` (>= sched_bytes, 256-byte aligned): [hdr (grid ints, padded)][records]; sets a.hdr / a.recs.  Once per shape.
template <int AMODE, int EPI, int S0, int OT>
static inline hipError_t sk_build_sched(SkArgs& a, const SkPlan& p, void* mem, hipStream_t st) {
    int* hdr = (int*)mem;
    SkRec* recs = (SkRec*)((char*)mem + (((size_t)p.grid * sizeof(int) + 255) & ~(size_t)255));
    hipLaunchKernelGGL((k_sk_sched<AMODE, EPI, S0, OT>), dim3(p.grid), dim3(512), 0, st, a, hdr, recs);
    a.hdr = hdr; a.recs = recs;
    return hipGetLastError();
}
template <int AMODE, int EPI, int S0, int OT, int NW = 8>
static inline hipError_t sk_launch(const SkArgs& a, const SkPlan& p, hipStream_t st) {
    auto kern = k_mlp_sk<AMODE, EPI, S0, OT, NW>;
    {   // every launch (the attribute belongs to the CURRENT device and the call is cheap: a process-wide cache broke a second context on another GPU)
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(64 * NW), p.lds_bytes, st, a);
    return hipGetLastError();
}

// ---- layer 0 of a net whose first hidden layer is STORED (three hidden layers; two behind an input too wide for the producer) -----------------------
// relu(x W0 + b0) for all heads: a contraction of only n_in + 1 <= 4 S0 values per output, so the tile GEMM spends its time in prologues and epilogues
// (C4 share: 259 us per step for 20 GFLOP).  Here a workgroup keeps its 256-column slice of W0 (bias as one more input row, x[n_in] = 1) in LDS as srcA
// fragments for the whole launch and walks its share of the rows: a wave = 16 rows, x in S0 registers (the next 16 rows' loads under way), 64 S0 matrix
// instructions per 16 rows x 256 columns fed by S0 registers of operands that are re-read a whole column group ahead (one ds_read_b128 feeds four
// instructions), the tile's sixteen 16-byte stores issued together behind the wait for the next rows.  Same transposed form and k order as the producer
// of k_mlp_sk.  C4 share (20 heads x 6250 rows, 76 + 1 inputs -> 1024): 205 us (tile GEMM: 258 us).
struct L0Args { int M, heads, N, ldx, rows_per_wg, nsplit; const float* x; const float* W0; long long strideW0; float* C; long long strideC; };
template <int S0>
__global__ void __launch_bounds__(512) k_l0_rows(const L0Args a) {
    extern __shared__ __attribute__((aligned(16))) float img[];               // [s][jq][lane][u]: W0[4 s + g][col0 + 16 (4 jq + u) + i], lane = 16 g + i
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int CQ = a.N / 256;
    const int sp = blockIdx.x % a.nsplit, cq = (blockIdx.x / a.nsplit) % CQ, head = blockIdx.x / (a.nsplit * CQ);      // (other orders of the grid measured the same)
    const float* __restrict__ W = a.W0 + (size_t)head * a.strideW0 + cq * 256;
    for (int idx = tid; idx < S0 * 1024; idx += 512) {
        const int u = idx & 3, ln = (idx >> 2) & 63, jq = (idx >> 8) & 3, s = idx >> 10;
        img[idx] = W[(size_t)(4 * s + (ln >> 4)) * a.N + 16 * (4 * jq + u) + (ln & 15)];
    }
    __syncthreads();
    const int r1 = min(a.M, (sp + 1) * a.rows_per_wg);
    float* __restrict__ Ch = a.C + (size_t)head * a.strideC + cq * 256 + 4 * g;
    float xr[S0], xn[S0];
    auto load_x = [&](int m0_, float (&dst)[S0]) {
        const float* __restrict__ xp = a.x + (size_t)min(m0_ + i, a.M - 1) * a.ldx + g;
#pragma unroll
        for (int s = 0; s < S0; ++s) dst[s] = xp[4 * s];
    };
    int m0 = sp * a.rows_per_wg + wave * 16;
    if (m0 < r1) load_x(m0, xr);
    f32x4 w[S0];                                                               // srcA operands of the column group at hand (see the loop)
#pragma unroll
    for (int s = 0; s < S0; ++s) w[s] = *(const f32x4*)&img[((s * 4) * 64 + lane) * 4];
    __builtin_amdgcn_s_waitcnt(0x0F70);                                        // nothing pending at the loop head: otherwise every iteration's matrix run waits for "the loads of xr" -- in fact for the previous tile's stores
    for (; m0 < r1; m0 += 128) {
        const bool more = m0 + 128 < r1;
        if (more) load_x(m0 + 128, xn);                                        // the next 16 rows are under way while these are multiplied
        asm volatile("" ::: "memory");                                         // ... issued HERE (left alone, hipcc sinks the loads to their use in the next iteration)
        const int row = min(m0 + i, a.M - 1);
        f32x4 o[16];
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            // w[s] feeds the four matrix instructions of step s and is re-read for the NEXT column group (the next tile's first one after the last:
            // the image never changes) the moment they are issued: every operand read is a whole column group ahead of its use.  sched_barrier keeps
            // the order (hipcc otherwise sinks each read to its use: read, wait the LDS latency, four instructions, read ...).
            f32x4 d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S0; ++s) {
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = MFMA16(w[s][u], xr[s], d[u]);
                __builtin_amdgcn_sched_barrier(0);
                w[s] = *(const f32x4*)&img[((s * 4 + ((jq + 1) & 3)) * 64 + lane) * 4];
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[4 * jq + u][r] = relu1(d[u][r]);
        }
        // The tile's 16 stores go out only HERE, behind the wait for the next rows' loads: loads and stores share one counter that does not retire in
        // order, so the compiler waits with vmcnt(0) -- placed at the loop head, with the stores of each column group issued as they were ready, that wait
        // sat behind 16 fresh stores in every iteration (267 us per C4 step; 184 us with the stores left out).  Now it finds only loads issued a tile ago.
        if (more) {
#pragma unroll
            for (int s = 0; s < S0; ++s) xr[s] = xn[s];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (m0 + i < a.M) {
#pragma unroll
            for (int k = 0; k < 16; ++k) *(f32x4*)(Ch + (size_t)row * a.N + 16 * k) = o[k];
        }
    }
}
template <int S0> static inline hipError_t l0_rows_launch(L0Args a, int n_cu, hipStream_t st) {
    const int pairs = a.heads * (a.N / 256), rb = (a.M + 127) / 128;
    a.nsplit = std::max(1, std::min(rb, n_cu / std::max(1, pairs)));           // whole (head, column slice) pairs per CU round; at least 128 rows per workgroup
    a.rows_per_wg = ((rb + a.nsplit - 1) / a.nsplit) * 128;
    const size_t lds = (size_t)S0 * 1024 * sizeof(float);
    auto kern = k_l0_rows<S0>;
    if (lds > 64 * 1024) { const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(kern, dim3(pairs * a.nsplit), dim3(512), lds, st, a);
    return hipGetLastError();
}
