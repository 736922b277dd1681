// Resident rollout for SMALL batches of 2 x 512 and 2 x 1024 dynamics ensembles (the reference's own params files: K = 5, B = 100 envs,
// 50 000 samples per iteration = three rounds of 200 steps for Swimmer / Snake, five of 100 for HalfCheetah / Hopper, >= 500 steps in chunks
// for Ant): the whole time loop in ONE launch, weights resident in registers, the steps chained through 8-byte packets.
// Same reference path as the other rollout kernels (samplers/vectorized_sampler.py:45-116, env_helpers.py:597-635, training.py:218-269).
//
// Why: at B = 100 a step of the step-wise GEMM path (rollout_gemm.hip) is four dependent launches of 4-19 us that each re-read the
// weights from L2 and leave most CUs idle; 600 steps take 9 ms whatever the launch mechanism.  Here
//   * a COMPUTE workgroup owns (round, model k, slice of WS hidden-1 units).  Its 4 PRODUCER waves (one per SIMD) keep a quarter each of W0
//     and of the slice's W1 columns as MFMA fragments in registers for the whole rollout and walk the round's env tiles (16 envs each): per
//     tile they recompute their quarter of hidden layer 0 (cheaper than fetching 512 activations per env from another CU every step) and
//     contract it with the slice -- one uninterrupted matrix-instruction stream per SIMD; its 4 FINISHER waves add the quarters, apply
//     bias + ReLU and emit the slice's contribution to the output layer: [ns x 16 envs] partial sums (resident_compute);
//   * a POST wave owns (round, env tile): policy forward (the MFMA chain of k_big_pre_mfma), action noise, normalised input -> X
//     packets; then it adds the DH/WS partials of each env's selected model in slice order, applies the residual, reward, done,
//     reset, and writes the trajectory rows.  R x ceil(B/16) post waves on the CUs the compute grid leaves free (resident_post);
//   * hand-over in both directions by {32 data bits | 32-bit step stamp} packets in an uncached exchange region (agent-scope relaxed
//     8-byte stores / loads: no fence, no flag, no grid barrier; xchg_device.h uses the same idea between GPUs).  Nobody waits for
//     more than ITS tile: a tile's round trip (partial sums out, next input in: ~6 us) passes while the CU works on the round's other tiles.
// All K heads are evaluated every step (as the reference's graph does); only simple sampling modes (step_rand / eps_rand / one_model).
// 2 x 1024 nets (every params file but Swimmer's) take the 4-wave form of the compute role, resident_compute_wide below.  The same form serves
// metrpo_validation_cost (the per-model validation rollouts of build_policy_graph, model_based_rl.py:106-151) with its own post role:
// resident_post_det / k_validation_resident / launch_validation_resident at the end of this file.
// Everything else (B > 128, other widths, policies other than 2 x 32, model_mean_std / model_med) stays on rollout_gemm.hip.  The grid must be resident as a whole:
// every wait is bounded (2 s), a launch that gives up is reported by the next metrpo_trpo_update / metrpo_comm_check and retires the kernel
// for its context (metrpo_internal.h: rollout_error_seen).
#include "mfma_common.h"
#include <string.h>
// test hook RESIDENT_PLAN: comma-separated words -- "norotate" (fixed deal of tiles to columns), "nosentinel" (post wave polls without the sentinel read)
static inline bool resident_plan_has(const metrpo_ctx* c, const char* word) { const char* v = ctx_opt(c, OPT_RESIDENT_PLAN); return v != nullptr && strstr(v, word) != nullptr; }

// Developer instrumentation (SRC=rollout_resident.hip tools/build_variant.sh restiming -DRES_TIMING): shader-clock sums per phase of the
// waves of compute workgroup 0 and of the first post workgroup, read back with metrpo_debug_resident_phases (tools/resident_phases.py).
#ifdef RES_TIMING
__device__ unsigned long long g_res_phase[2][8][8];
__device__ unsigned long long g_res_wall[4][256];     // wall clock (100 MHz) of tile (round 0, tile 0): X pushed | X seen by workgroup 0 | P pushed by workgroup 0 | P complete at the post wave
#define RT_WALL(i, cond, tau_) { if ((cond) && lane == 0 && (tau_) < 256) g_res_wall[i][tau_] = wall_clock64(); }
__device__ unsigned long long g_res_wmax[2][256];     // latest over ALL workgroups serving tile 0: X seen | P pushed
#define RT_WMAX(i, cond, tau_) { if ((cond) && lane == 0 && (tau_) < 256) atomicMax(&g_res_wmax[i][tau_], wall_clock64()); }
extern "C" int32_t metrpo_debug_resident_wmax(unsigned long long* out) { const int rc = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_wmax), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : -1; static unsigned long long zero[512]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_res_wmax), zero, sizeof(zero)); return rc; }
extern "C" int32_t metrpo_debug_resident_wall(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_wall), sizeof(unsigned long long) * 1024) == hipSuccess ? 0 : -1; }
#define RT_DECL unsigned long long rt_t = __builtin_readcyclecounter(); unsigned long long rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); rt_acc[i] += n_ - rt_t; rt_t = n_; }
#define RT_DUMP(role, first) { if (lane == 0 && (int)blockIdx.x == (first)) for (int i_ = 0; i_ < 8; ++i_) g_res_phase[role][wave][i_] = rt_acc[i_]; }
extern "C" int32_t metrpo_debug_resident_phases(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_phase), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1; }
#else
#define RT_DECL
#define RT_MARK(i)
#define RT_DUMP(role, first)
#define RT_WALL(i, cond, tau_)
#define RT_WMAX(i, cond, tau_)
#endif

struct ResidentK {
    int R, round0, rounds_total, NT, NSL, U, PW, steps;   // rounds of this launch, first round, env tiles, slices, compute blocks, post waves per block, steps per round
    int part_stride, part_off, Btot;                      // validation-cost mode in batch chunks: det_part[m * part_stride + part_off + tile], costs are means over Btot envs
    int det, NTM; float gamma; const float* s0; double* det_part;   // validation-cost mode (metrpo_validation_cost): tiles per model, discount, start states [B][ns], per-(model, tile) cost sums
    int NTC;                                              // 4-wave form: env tiles per workgroup COLUMN (the R * NT tiles of the launch are dealt to U / (K NSL) columns); else = NT
    int sentinel;                                         // post waves wait on one packet per slice before they read a tile's partial sums (resident_post)
    int rot;                                              // 4-wave form: tile g belongs to column (g + step) mod columns instead of a fixed run (uneven deals of few tiles: Ant's 7 tiles on 3 columns)
    unsigned int seq0;                                    // packets of local step tau carry seq0 + tau + 1
    int skip_block;                                       // test hook (METRPO_RESIDENT_TEST_SKIP): this workgroup behaves as if it had never been scheduled; -1 otherwise
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
// true once a packet stamped for step `seq` has landed -- or a LATER one.  Later stamps only ever meet a reader nobody waits for: the post wave
// releases the input of step tau + 1 of a tile when the partial sums of the heads its 16 envs SELECTED at step tau are in, so a workgroup of a head
// no env of the tile selected may still be on step tau when its input slot (or its producers' LDS partials) already carries tau + 1.  It then
// computes on what it finds -- a result nobody reads -- and catches up; an exact-match test would leave it waiting for a stamp that is gone.
__device__ __forceinline__ bool res_fresh(unsigned long long pk, unsigned int seq) { return (int)((unsigned int)(pk >> 32) - seq) >= 0; }

// A compute workgroup = 4 PRODUCER waves (one per SIMD) + 4 FINISHER waves.  Producer kappa owns a quarter of the hidden-0 units
// (16 j-tiles / 4) and keeps the matching MFMA fragments of W0 and of the workgroup's W1 slice in REGISTERS for the whole rollout; it
// walks the env tiles of the round in order: input packets of tile t (prefetched during tile t-1) -> J/4 x (NIN_KS + 4 MT) MFMAs in eight
// independent accumulator chains -> its partial pre-activations of the slice's hidden-1 units for tile t into LDS, stamped.  No barrier:
// the four SIMDs run one uninterrupted matrix-instruction stream each, and a tile's hand-over latency (partial sums out, next input
// in: ~5 us) passes while the CU works on the round's other tiles.  Finisher i serves tiles t = i (mod 4): waits for the four stamps,
// adds the quarters in order, bias + ReLU, output-layer MFMAs with the slice's W2 rows, packets out.
template <int NS, int NIN, int DH, int WS>
__device__ __forceinline__ void resident_compute(const ProblemDesc& pd, const ResidentK& z, const float* __restrict__ dyn, float* lds) {
    constexpr int NIN_KS = cdiv(NIN + 1, 4), J = DH / 16, JQ = J / 4, MT = WS / 16, OUT_CB = cdiv(NS, 16), NSP = 16 * OUT_CB;      // NIN + 1: the bias rides as one more input
    constexpr int O_STAMP = 0, O_PART = O_STAMP + 32;          // floats: stamps [8 tiles][4 producers] | partials [tile][producer][mt][lane] f32x4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    const int K = pd.K, NSL = z.NSL, NT = z.NT;
    const int u = blockIdx.x, rho = u / (K * NSL), k = (u / NSL) % K, sl = u % NSL, col0 = sl * WS;
    const float* __restrict__ W = dyn + (size_t)k * pd.dyn.n_params;
    const float* __restrict__ W0 = W + pd.dyn.w_off[0];
    const float* __restrict__ W1 = W + pd.dyn.w_off[1];
    const float* __restrict__ W2 = W + pd.dyn.w_off[2];
    if (tid < 32) ((unsigned int*)lds)[O_STAMP + tid] = z.seq0;                  // stamps of this launch start at seq0 + 1
    __syncthreads();
    // MFMA16(a, b, acc): a = A[m = lane & 15][k = lane >> 4], b = B[k = lane >> 4][n = lane & 15], acc[r] = D[4 (lane >> 4) + r][n]: output register r
    // of lane (c, q) is unit 4q + r of its 16-unit tile = the B operand of k-slot q of the next layer's MFMA number r, so the next
    // layer's A fragments are gathered in that order and nothing is ever transposed.
    if (wave < 4) {
        const int kap = wave;
#ifdef RES_TIMING
        unsigned long long rt_stale = 0;
#endif
        float w0f[JQ][NIN_KS], w1f[MT][JQ][4];
#pragma unroll
        for (int jj = 0; jj < JQ; ++jj) {
            const int j = kap * JQ + jj;
#pragma unroll
            for (int kk = 0; kk < NIN_KS; ++kk) {                      // input slot NIN carries the constant 1 (post wave): its weight row is the bias
                const int in = 4 * kk + q;
                w0f[jj][kk] = (in < NIN) ? W0[(size_t)in * DH + 16 * j + c] : (in == NIN ? W[pd.dyn.b_off[0] + 16 * j + c] : 0.0f);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) w1f[mt][jj][r] = W1[(size_t)(16 * j + 4 * q + r) * DH + col0 + 16 * mt + c];
        }
        const unsigned long long* xbase = z.X + ((size_t)(rho * NT) * (4 * NIN_KS)) * 16 + c;
        unsigned long long pk[NIN_KS];
        auto fetch = [&](int t) {
#pragma unroll
            for (int kk = 0; kk < NIN_KS; ++kk) pk[kk] = res_ld(xbase + ((size_t)t * (4 * NIN_KS) + 4 * kk + q) * 16);
        };
        // hidden-0 tiles of this producer for the env tile whose input is in pk (waits for it): JQ independent chains of NIN_KS MFMAs
        auto layer0 = [&](f32x4 (&h)[JQ], int t, unsigned int seq, auto&& flush) -> bool {
            float x[NIN_KS];
            ResSpin sp;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < NIN_KS; ++kk) { ok = ok && res_fresh(pk[kk], seq); x[kk] = __uint_as_float((unsigned int)pk[kk]); }
                if (__all(ok)) break;
                flush();                                                // the input is late: it may be waiting for the very tile whose hand-over is still pending here
                if (sp.give_up(z)) return false;
                fetch(t);
            }
#ifdef RES_TIMING
            if (sp.spins > 0) rt_stale += 1;
#endif
            fetch((t + 1 < NT) ? t + 1 : 0);                            // the next tile's input (after the last tile: stamped for the next step) is in flight during this tile's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < JQ; ++jj) h[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NIN_KS; ++kk)
#pragma unroll
                for (int jj = 0; jj < JQ; ++jj) h[jj] = MFMA16(w0f[jj][kk], x[kk], h[jj]);
            return true;
        };
        // this producer's quarter of the slice's hidden-1 pre-activations: 4 MT independent chains over its JQ x 4 k-steps
        auto layer1 = [&](f32x4 (&a2)[MT][4], f32x4 (&h)[JQ]) {
#pragma unroll
            for (int jj = 0; jj < JQ; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hr = relu1(h[jj][r]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a2[mt][r] = (jj == 0) ? MFMA16(w1f[mt][jj][r], hr, (f32x4{0.f, 0.f, 0.f, 0.f})) : MFMA16(w1f[mt][jj][r], hr, a2[mt][r]);
                }
        };
        // partial sums of env tile t out to the finisher (LDS operations of one wave complete in issue order: the stamp lands after the data)
        auto hand_over = [&](f32x4 (&a2)[MT][4], int t, unsigned int seq) {
            f32x4* part = (f32x4*)(lds + O_PART) + ((size_t)(t * 4 + kap) * MT) * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) part[mt * 64] = (a2[mt][0] + a2[mt][1]) + (a2[mt][2] + a2[mt][3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_store((unsigned int*)lds + O_STAMP + t * 4 + kap, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        fetch(0);
        RT_DECL
        // flattened (step, tile) sequence, two tiles per trip on alternating accumulator sets: the tail of a tile (chain drain, sums, LDS
        // write) is issued after the NEXT tile's hidden-0 MFMAs, so the matrix pipe does not idle between tiles
        const int total = z.steps * NT;
        f32x4 accA[MT][4], accB[MT][4], h[JQ];
        int tA = 0, tB = 0; unsigned int sA = z.seq0 + 1u, sB = sA;
        bool pendA = false, pendB = false;
        int t = 0; unsigned int seq = z.seq0 + 1u;
        auto advance = [&]() { if (++t == NT) { t = 0; ++seq; } };
        auto flushA = [&]() { if (pendA) { hand_over(accA, tA, sA); pendA = false; } };
        auto flushB = [&]() { if (pendB) { hand_over(accB, tB, sB); pendB = false; } };
        for (int it = 0; it < total; it += 2) {
            if (!layer0(h, t, seq, flushB)) return;
            RT_MARK(0)
            RT_WALL(1, blockIdx.x == 0 && wave == 0 && t == 0, (int)(seq - z.seq0) - 1)
            flushB();
            layer1(accA, h);
            tA = t; sA = seq; pendA = true; advance();
            if (it + 1 < total) {
                if (!layer0(h, t, seq, flushA)) return;
                flushA();
                layer1(accB, h);
                tB = t; sB = seq; pendB = true; advance();
            }
            RT_MARK(1)
        }
        flushA(); flushB();
#ifdef RES_TIMING
        rt_acc[2] = rt_stale;
#endif
        RT_DUMP(0, 0)
        return;
    }
    // ---- finishers
    const int fi = wave - 4;
    float w2f[OUT_CB][MT][4];
    f32x4 b1f[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b1f[mt][r] = W[pd.dyn.b_off[1] + col0 + 16 * mt + 4 * q + r];
#pragma unroll
            for (int ocb = 0; ocb < OUT_CB; ++ocb) { const int dim = 16 * ocb + c; w2f[ocb][mt][r] = (dim < NS) ? W2[(size_t)(col0 + 16 * mt + 4 * q + r) * NS + dim] : 0.0f; }
        }
    }
    RT_DECL
    for (int tau = 0; tau < z.steps; ++tau) {
        const unsigned int seq = z.seq0 + (unsigned int)tau + 1u;
        for (int t = fi; t < NT; t += 4) {
            {
                ResSpin sp;
                for (;;) {
                    const unsigned int st = __hip_atomic_load((const unsigned int*)lds + O_STAMP + t * 4 + (lane & 3), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__all((int)(st - seq) >= 0)) break;
                    if (sp.give_up(z)) return;
                }
            }
            RT_MARK(0)
            const f32x4* part = (const f32x4*)(lds + O_PART) + ((size_t)(t * 4) * MT) * 64 + lane;
            f32x4 o[OUT_CB];
#pragma unroll
            for (int ocb = 0; ocb < OUT_CB; ++ocb) o[ocb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 h2 = b1f[mt];
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) h2 += part[(kp * MT + mt) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[r] = relu1(h2[r]);
#pragma unroll
                for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[ocb] = MFMA16(w2f[ocb][mt][r], h2[r], o[ocb]);
            }
            unsigned long long* pp = z.P + ((((size_t)(rho * NT + t) * K + k) * NSL + sl) * NSP) * 16 + c;
#pragma unroll
            for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int dim = 16 * ocb + 4 * q + r; if (dim < NS) res_st(pp + dim * 16, seq, o[ocb][r]); }
            RT_MARK(1)
            RT_WALL(2, blockIdx.x == 0 && t == 0, tau)
        }
    }
    RT_DUMP(0, 0)
}

// ---- compute role, 2 x 1024 ensembles (the reference's params-half-cheetah / -hopper / -snake.json) --------------------------------------
// A slice of WS = 64 hidden-1 units of a 1024-wide layer is 256 fragment registers per lane and the quarter of W0 another 16 NIN_KS: only a
// wave that owns its SIMD's whole 512-register file can hold them, so the workgroup is the 4 PRODUCER waves alone (__launch_bounds__(256, 1)) and
// each of them is also the finisher of the env tiles t = its index (mod 4).  At this width a tile is ~11 000 matrix-pipe cycles per SIMD: the
// drain of the accumulator chains at the end of a tile (2 % of it) needs no second accumulator set, and the finish of a tile (32 MFMAs, W2
// fragments and b1 from an LDS image) rides behind the producer's next tile.  Hidden layer 0 is evaluated four 16-unit tiles at a time.
// Packets, stamps, slot layout and the rules about stale stamps are those of resident_compute.
template <int NS, int NIN, int DH, int WS, bool ROT = false>
__device__ __forceinline__ void resident_compute_wide(const ProblemDesc& pd, const ResidentK& z, const float* __restrict__ dyn, float* lds) {
    constexpr int NIN_KS = cdiv(NIN + 1, 4), J = DH / 16, JQ = J / 4, JB = 4, MT = WS / 16, OUT_CB = cdiv(NS, 16), NSP = 16 * OUT_CB;
    // Wide inputs (Ant: 35 + 1 inputs = 9 k-steps): the W0 fragments of the k-steps beyond the sixth live in an LDS image instead of 16 more
    // registers each -- with the W1 slice filling the accumulation half, everything else has to fit the 256 architectural registers, and what
    // does not is spilled to scratch memory inside the tile loop (31 loads per tile, 10 instead of 6 us).  The image takes half of the partial-sum slots' space.
    constexpr int W0L = (NIN_KS > 6) ? NIN_KS - 6 : 0, W0R = NIN_KS - W0L;
    constexpr int NSLOT = W0L ? 4 : 8;                         // LDS slots for the tiles' partial sums: tile t of a column uses slot t % NSLOT
    constexpr int O_STAMP = 0, O_DONE = 32, O_W2 = 48, O_B1 = O_W2 + OUT_CB * MT * 256, O_W0 = O_B1 + MT * 256, O_PART = O_W0 + 4 * JQ * W0L * 64;   // floats: stamps [slot][producer] | consumed [slot] | W2 fragments [ocb][mt][lane][r] | b1 [mt][lane][r] | partials [slot][producer][mt][lane]
    static_assert(JQ % JB == 0, "hidden-0 tiles are taken four at a time");
    const int tid = threadIdx.x, lane = tid & 63, kap = tid >> 6, c = lane & 15, q = lane >> 4;
    const int K = pd.K, NSL = z.NSL;
    const int u = blockIdx.x, col = u / (K * NSL), k = (u / NSL) % K, sl = u % NSL, col0 = sl * WS;
    // The R * NT env tiles of the launch (tile g = round * NT + tile of the round: what the post waves and the packet slots are indexed by) are
    // dealt to the workgroup columns in runs of NTC: five rounds of seven tiles are three columns of 12 / 12 / 11, one launch, every CU busy --
    // instead of 3 + 2 rounds in two launches.  NT below = the tiles of THIS column.
    // Validation-cost mode (z.det): tile g = model * NTM + tile of the batch belongs to ONE model -- the workgroups of model k deal model k's NTM tiles
    // to their columns and never see the others'.
    const int g0 = z.det ? k * z.NTM + col * z.NTC : col * z.NTC;
    // ROTATING deal (z.rot; launches of few tiles that do not divide by the columns -- Ant's one round of 7 tiles on 3 columns is 3 / 2 / 2 and the rollout
    // runs at the pace of the column with 3): tile g is served by column (g + step) mod NCOL, so every column has 3, 2, 2, 3, ... tiles and all of
    // them the same 7 per three steps.  Every column holds the same weight slices and the packets are indexed by the global tile, so nothing else
    // moves; local tile t of step tau is global tile first(tau) + t NCOL.
    constexpr bool rot = ROT;                                    // an instantiation of its own: the fixed deal keeps its constant tile count and addresses
    const int NCOL = z.U / (K * NSL), G = z.R * z.NT;
    auto first_at = [&](int tau_) -> int { return ((col - tau_) % NCOL + NCOL) % NCOL; };
    auto cnt_at = [&](int tau_) -> int { return rot ? (G - first_at(tau_) + NCOL - 1) / NCOL : (z.det ? max(0, min(z.NTC, z.NTM - col * z.NTC)) : max(0, min(z.NTC, G - g0))); };
    auto g_at = [&](int tau_, int t_) -> int { return rot ? first_at(tau_) + t_ * NCOL : g0 + t_; };
    int NT = cnt_at(0);                                          // tiles of this column in the current step (constant unless rot)
    if (NT == 0) return;
    const float* __restrict__ W = dyn + (size_t)k * pd.dyn.n_params;
    const float* __restrict__ W0 = W + pd.dyn.w_off[0];
    const float* __restrict__ W1 = W + pd.dyn.w_off[1];
    const float* __restrict__ W2 = W + pd.dyn.w_off[2];
    if (tid < 48) ((unsigned int*)lds)[tid] = 0u;               // stamps and consumed marks count the uses of a slot within this launch (1, 2, ...)
    for (int i = tid; i < OUT_CB * MT * 256; i += 256) {
        const int r = i & 3, ln = (i >> 2) & 63, mt = (i >> 8) % MT, ocb = i / (256 * MT), dim = 16 * ocb + (ln & 15);
        lds[O_W2 + i] = (dim < NS) ? W2[(size_t)(col0 + 16 * mt + 4 * (ln >> 4) + r) * NS + dim] : 0.0f;
    }
    for (int i = tid; i < MT * 256; i += 256) {
        const int r = i & 3, ln = (i >> 2) & 63, mt = i >> 8;
        lds[O_B1 + i] = W[pd.dyn.b_off[1] + col0 + 16 * mt + 4 * (ln >> 4) + r];
    }
    float w0f[JQ][W0R], w1f[MT][JQ][4];
    float* w0l = lds + O_W0 + (size_t)kap * JQ * W0L * 64 + lane;      // this wave's image: [jj][kk - W0R][lane]
#pragma unroll
    for (int jj = 0; jj < JQ; ++jj) {
        const int j = kap * JQ + jj;
#pragma unroll
        for (int kk = 0; kk < NIN_KS; ++kk) {
            const int in = 4 * kk + q;
            const float w = (in < NIN) ? W0[(size_t)in * DH + 16 * j + c] : (in == NIN ? W[pd.dyn.b_off[0] + 16 * j + c] : 0.0f);
            if (kk < W0R) w0f[jj][kk < W0R ? kk : 0] = w; else w0l[(jj * W0L + (kk - W0R)) * 64] = w;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // the slice's W1 fragments live in ACCUMULATION registers (the matrix instruction reads its A operand from either file): defined
                // there, they stay there -- left to itself the allocator parks what does not fit into the 256 architectural registers in the other
                // half and copies it back before every use, and each of those copies queues behind the matrix instruction in flight
                w1f[mt][jj][r] = in_acc_reg(W1[(size_t)(16 * j + 4 * q + r) * DH + col0 + 16 * mt + c]);
            }
    }
    __syncthreads();                                            // stamps, W2 / b1 / W0 images
    const unsigned long long* xbase = z.X + c;
    unsigned long long pk[NIN_KS];
    auto fetch = [&](int g) {                                    // g: global tile
#pragma unroll
        for (int kk = 0; kk < NIN_KS; ++kk) pk[kk] = res_ld(xbase + ((size_t)g * (4 * NIN_KS) + 4 * kk + q) * 16);
    };
    // A slot serves the tiles t, t + 8, ... of the column, step after step: use number uid(tau, t) = tau * (uses of the slot per step) + t / 8.
    // Producers stamp a slot with uid + 1 once their quarter is in it, the finisher marks it consumed (uid + 1) once it has read the four
    // quarters, and a producer writes use n only over a consumed use n - 1: nobody can lap the finisher (every wait below keeps finishing).
    // Rotating deal (at most NSLOT tiles per step, slot = local tile): the uses of slot t before step tau = the steps tau' < tau with more than t tiles,
    // i.e. first(tau') < G - t NCOL; first(.) runs through all residues once per NCOL steps.
    auto uid_of = [&](int tau, int t) -> unsigned int {
        const int s_ = t & (NSLOT - 1);
        if (!rot) return (unsigned int)(tau * ((NT - s_ + NSLOT - 1) / NSLOT) + t / NSLOT);
        const int lim = G - t * NCOL, full = tau / NCOL, rem = tau - full * NCOL;
        int n = full * max(0, min(NCOL, lim));
        for (int j = 0; j < rem; ++j) n += (first_at(j) < lim) ? 1 : 0;
        return (unsigned int)n;
    };
    // ---- finisher duties of this wave: tiles kap, kap + 4, ... of the column, every step, in order
    int dt = kap, dtau = 0;
    auto duty_norm = [&]() { while (dtau < z.steps && dt >= cnt_at(dtau)) { dt = kap; ++dtau; } };      // next (step, local tile = kap mod 4) this wave finishes
    auto try_finish = [&]() -> bool {
        duty_norm();
        if (dtau >= z.steps) return false;
        const int s_ = dt & (NSLOT - 1);
        const unsigned int want = uid_of(dtau, dt) + 1u;
        const unsigned int st = __hip_atomic_load((const unsigned int*)lds + O_STAMP + s_ * 4 + (lane & 3), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!__all((int)(st - want) >= 0)) return false;
        const f32x4* part = (const f32x4*)(lds + O_PART) + ((size_t)(s_ * 4) * MT) * 64 + lane;
        f32x4 h2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            h2[mt] = *(const f32x4*)&lds[O_B1 + (mt * 64 + lane) * 4];
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) h2[mt] += part[(kp * MT + mt) * 64];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the quarters are in registers: the slot may be overwritten
        if (lane == 0) __hip_atomic_store((unsigned int*)lds + O_DONE + s_, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        f32x4 o[OUT_CB];
#pragma unroll
        for (int ocb = 0; ocb < OUT_CB; ++ocb) o[ocb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = relu1(h2[mt][r]);
#pragma unroll
            for (int ocb = 0; ocb < OUT_CB; ++ocb) {
                const f32x4 w2 = *(const f32x4*)&lds[O_W2 + ((ocb * MT + mt) * 64 + lane) * 4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[ocb] = MFMA16(w2[r], h2[mt][r], o[ocb]);
            }
        }
        const unsigned int dseq = z.seq0 + (unsigned int)dtau + 1u;
        // rotating deal: a region per column, so that every packet address keeps ONE writer (a workgroup of a head nobody selected may run a step late:
        // its packet of step tau must not be able to land on top of another column's packet of step tau + 1)
        unsigned long long* pp = z.P + ((((size_t)((rot ? col * G : 0) + g_at(dtau, dt)) * K + k) * NSL + sl) * NSP) * 16 + c;
#pragma unroll
        for (int ocb = 0; ocb < OUT_CB; ++ocb)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int dim = 16 * ocb + 4 * q + r; if (dim < NS) res_st(pp + dim * 16, dseq, o[ocb][r]); }
        RT_WALL(2, k == 0 && sl == 0 && g_at(dtau, dt) == 0, dtau)
        RT_WMAX(1, g_at(dtau, dt) == 0, dtau)
        dt += 4;
        duty_norm();
        return true;
    };
    fetch(g_at(0, 0));
    int t = 0, tau = 0; unsigned int seq = z.seq0 + 1u;
#ifdef RES_TIMING
    const int wave = kap;
#endif
    RT_DECL
    while (tau < z.steps) {
        float x[NIN_KS];
        RT_MARK(3)
        {
            ResSpin sp;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < NIN_KS; ++kk) { ok = ok && res_fresh(pk[kk], seq); x[kk] = __uint_as_float((unsigned int)pk[kk]); }
                if (__all(ok)) break;
                (void)try_finish();                                     // the input may be waiting for the very tile this wave still has to finish
                if (sp.give_up(z)) return;
                fetch(g_at(tau, t));
            }
        }
        fetch((t + 1 < NT) ? g_at(tau, t + 1) : g_at(tau + 1, 0));        // (past the last step: a packet nobody waits for)
        RT_MARK(0)
        RT_WALL(1, k == 0 && sl == 0 && kap == 0 && g_at(tau, t) == 0, tau)
        RT_WMAX(0, g_at(tau, t) == 0, tau)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < JQ; jb += JB) {                           // (running layer 0 a batch ahead of its consumers measured no faster: 8.64 vs 8.50 ms)
            f32x4 h[JB];
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) h[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NIN_KS; ++kk)
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) h[jj] = MFMA16((kk < W0R) ? w0f[jb + jj][kk < W0R ? kk : 0] : w0l[((jb + jj) * W0L + (kk - W0R)) * 64], x[kk], h[jj]);
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hr = relu1(h[jj][r]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a2[mt] = MFMA16(w1f[mt][jb + jj][r], hr, a2[mt]);
                }
        }
        RT_MARK(1)
        {   // partial sums of env tile t out (LDS operations of one wave complete in issue order: the stamp lands after the data)
            const int s_ = t & (NSLOT - 1);
            const unsigned int uid = uid_of(tau, t);
            if (uid > 0) {                                              // the slot's previous occupant must have been read by its finisher
                ResSpin sp;
                while ((int)(__hip_atomic_load((const unsigned int*)lds + O_DONE + s_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) - uid) < 0) {
                    if (!try_finish() && sp.give_up(z)) return;
                }
            }
            f32x4* part = (f32x4*)(lds + O_PART) + ((size_t)(s_ * 4 + kap) * MT) * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) part[mt * 64] = a2[mt];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_store((unsigned int*)lds + O_STAMP + s_ * 4 + kap, uid + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (NT <= 4 && (t & 3) == kap) {
            // a column of few tiles (one round of 7 tiles over three columns: Ant's chunks, single rounds) comes back to a tile before a finish deferred
            // behind the next tile's MFMAs has made its way through the post wave: its finisher waits for the other three quarters (the producers run
            // the same tile at the same time) and finishes at once -- earlier duties first, so nobody it waits for can be waiting for it
            ResSpin sp;
            while (dtau < z.steps && (dtau < tau || (dtau == tau && dt <= t))) { if (!try_finish() && sp.give_up(z)) return; }
        } else (void)try_finish();
        RT_MARK(2)
        if (++t == NT) { t = 0; ++seq; ++tau; if (rot) NT = cnt_at(tau); }
    }
    RT_DUMP(0, 0)
    {   // what is left of this wave's duties (the other producers' last hand-overs may still be under way)
        ResSpin sp;
        for (;;) { duty_norm(); if (dtau >= z.steps) break; if (!try_finish() && sp.give_up(z)) return; }
    }
}

// ---- post role ------------------------------------------------------------------------------------------------------------------------
// value held by lane q of an env's four lanes out of (a0, a1, a2, a3)
__device__ __forceinline__ float sel4(int q, float a0, float a1, float a2, float a3) { return (q & 2) ? ((q & 1) ? a3 : a2) : ((q & 1) ? a1 : a0); }

// The four lanes (c, q = 0..3) of env c hold the env's whole state in registers (the same values): nothing on the path from "last partial
// sum arrived" to "next input pushed" goes through memory except the 16 x NA clipped / normalised actions (LDS).  Everything that does not
// feed the next input -- Philox blocks and Box-Muller of the NEXT step, the reset row, head choice, output-layer bias, trajectory stores --
// is issued while the compute workgroups are busy with the step.
template <int ENV>
__device__ __forceinline__ void resident_post(const ProblemDesc& pd, const RolloutK& r, const ResidentK& z, const float* __restrict__ dyn,
                                              const float* __restrict__ theta, const float* __restrict__ norm, float* lds) {
    static_assert(ENV != METRPO_ENV_HUMANOID, "2 x 32 policies only");
    using C = Cfg<ENV, 64, 32>;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NIN = C::NIN, PH = 32, NS_KS = C::NS_KS, NIN_KS = cdiv(NIN + 1, 4), NSP = C::NSP;   // X element NIN = 1 (bias input of layer 0)
    constexpr int O_PF1 = NS_KS * 2 * 64, O_PF2 = O_PF1 + 16 * 64, O_B0 = O_PF2 + 8 * 64, O_B1 = O_B0 + 32, O_B2 = O_B1 + 32, IMG = ((O_B2 + 16 + 3) / 4) * 4;
    constexpr int PW_LDS = 2 * 16 * NA;
    constexpr int DCH = (NS <= 12) ? NS : (NS + 1) / 2;               // output dims per batch of partial-sum loads (4 slices x DCH packets in flight per lane)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    for (int i = tid; i < IMG; i += (int)blockDim.x) {                // policy fragment image (layout of k_big_pre_mfma, rollout_gemm.hip)
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
    float* UA = lds + IMG + wave * PW_LDS; float* XA = UA + 16 * NA;
    const int b0 = w * 16, b = b0 + c;
    const bool active = b < B;
    const int bc = active ? b : 0;
    const uint64_t genv = r.stream_offset + (uint64_t)bc;
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    const float* diff_mean = norm + 2 * (NS + NA); const float* diff_std = diff_mean + NS;
    const float* __restrict__ log_std = theta + C::pLS;
    unsigned long long* xp = z.X + ((size_t)(rho * z.NT + w) * (4 * NIN_KS)) * 16 + c;
    const unsigned long long* pbase = z.P + ((size_t)(rho * z.NT + w) * K) * NSL * NSP * 16 + c;
    // per-lane constants: normaliser of X element 4 kk + q, sigma and normaliser of action dims 4q .. 4q+3, de-normaliser of every state dim
    float xm[NIN_KS], xr[NIN_KS], sig[4], am[4], ar[4], dmn[NS], dsd[NS];
#pragma unroll
    for (int kk = 0; kk < NIN_KS; ++kk) {
        const int f = 4 * kk + q, src = (f < NS - NDROP) ? f + NDROP : NS + (f - (NS - NDROP));
        xm[kk] = (f < NIN) ? in_mean[src] : 0.0f; xr[kk] = (f < NIN) ? 1.0f / in_std[src] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = 4 * q + j;
        sig[j] = (d < NA) ? __expf(fmaxf(log_std[d], LOG_MIN_STD)) : 0.0f;
        am[j] = (d < NA) ? in_mean[NS + d] : 0.0f; ar[j] = (d < NA) ? 1.0f / in_std[NS + d] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) { dmn[i] = diff_mean[i]; dsd[i] = diff_std[i]; }
    // ---- state at the first step of this round
    float s[NS];
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
#pragma unroll
        for (int i = 0; i < NS; ++i) s[i] = !active ? 0.0f : (row < 0 ? r.init_obs[(size_t)bc * NS + i] : r.pool[(size_t)row * NS + i]);
    }
    // draws of a step: the step's Philox block (head choice, reset row / model) and the policy noise of this lane's action dims
    uint4 dstep; float zn[4];
    auto draw = [&](int t_loc) {
        const size_t tb = (size_t)t_loc * B + bc;
        dstep = rng_draw(r.seed, genv, r.t0 + t_loc, RNG_STEP, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d0 = 4 * q + 2 * h;
            zn[2 * h] = 0.0f; zn[2 * h + 1] = 0.0f;
            if (d0 >= NA || r.determ) continue;
            if (r.eps != nullptr) { zn[2 * h] = r.eps[tb * NA + d0]; if (d0 + 1 < NA) zn[2 * h + 1] = r.eps[tb * NA + d0 + 1]; }
            else {
                const uint4 blk = (d0 == 0) ? dstep : rng_draw(r.seed, genv, r.t0 + t_loc, RNG_STEP, d0 >> 1);
                normal2(blk.x, blk.y, zn[2 * h], zn[2 * h + 1]);
            }
        }
    };
    draw(round * z.steps);
    RT_DECL
    for (int tau = 0; tau < z.steps; ++tau) {
        const unsigned int seq = z.seq0 + (unsigned int)tau + 1u;
        const int t_loc = round * z.steps + tau;                      // row of the trajectory tensors; draws are keyed by r.t0 + t_loc
        const size_t tb = (size_t)t_loc * B + bc;
        // ---- policy.get_actions (MFMA chain of k_big_pre_mfma; input k-slot q of step s_ = state dim 4 s_ + q)
        f32x4 p0[2], p1[2];
        p0[0] = *(const f32x4*)&lds[O_B0 + 4 * q]; p0[1] = *(const f32x4*)&lds[O_B0 + 16 + 4 * q];
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) {
            const float xs = sel4(q, s[4 * s_], (4 * s_ + 1 < NS) ? s[(4 * s_ + 1 < NS) ? 4 * s_ + 1 : 0] : 0.0f, (4 * s_ + 2 < NS) ? s[(4 * s_ + 2 < NS) ? 4 * s_ + 2 : 0] : 0.0f,
                                  (4 * s_ + 3 < NS) ? s[(4 * s_ + 3 < NS) ? 4 * s_ + 3 : 0] : 0.0f);
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
        // ---- actions of dims 4q .. 4q+3 (env_helpers.py:599 clip; training.py:228,146-151 normalisation)
        float av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = 4 * q + j;
            if (d >= NA) continue;
            av[j] = r.determ ? mu[j] : fmaf(zn[j], sig[j], mu[j]);
            const float ac = fminf(fmaxf(av[j], -1.0f), 1.0f);
            UA[c * NA + d] = ac;
            XA[c * NA + d] = (ac - am[j]) * ar[j];
        }
        wave_lds_sync();
#pragma unroll
        for (int kk = 0; kk < NIN_KS; ++kk) {
            const int f = 4 * kk + q;                                 // X element of this lane: state dim f + NDROP, then the action dims
            constexpr int NSD = NS - NDROP;
            auto sd = [&](int e) { const int i = 4 * kk + e + NDROP; return (4 * kk + e < NSD) ? s[(i < NS) ? i : 0] : 0.0f; };
            float v = sel4(q, sd(0), sd(1), sd(2), sd(3));
            if (f >= NSD && f < NIN) v = XA[c * NA + f - NSD]; else v = (v - xm[kk]) * xr[kk];
            if (!active || f > NIN) v = 0.0f;
            else if (f == NIN) v = 1.0f;
            res_st(xp + f * 16, seq, v);
        }
        RT_MARK(0)
        RT_WALL(0, g == 0, tau)
        // ---- off the critical path (the compute workgroups are busy with the step now)
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int d = 4 * q + j; if (d < NA) { r.act[tb * NA + d] = av[j]; r.mean[tb * NA + d] = mu[j]; } }
#pragma unroll
            for (int i = 0; i < NS; ++i) if ((i & 3) == q) r.obs[tb * NS + i] = s[i];
        }
        int sel = cur_model;                                          // which head this env follows this step (env_helpers.py:617-634)
        if (r.sam_mode == METRPO_SAM_STEP_RAND) sel = (r.model_idx != nullptr) ? r.model_idx[tb] : rng_index(dstep.z, K);
        if (r.sam_mode == METRPO_SAM_ONE_MODEL) sel = 0;
        const int ts_new = ts + 1;
        const bool dn_h = ts_new >= r.H;                              // env_helpers.py:603-604
        // Ant also ends an episode on the state it reaches (com_ant_env.py is_done): known only when the partial sums are in, so its reset row is
        // fetched every step, off the critical path like everything here
        constexpr bool STATE_DONE = (ENV == METRPO_ENV_ANT);
        float prow[NS];
        int reset_model = cur_model;
        if (dn_h || STATE_DONE) {                                     // reset (env_helpers.py:585-595): row and model from this step's block
            const size_t rb = (size_t)(t_loc + 1) * B + bc;
            const int row = (r.reset_idx != nullptr) ? r.reset_idx[rb] : rng_index(dstep.w, r.n_pool);
            reset_model = (r.reset_model != nullptr) ? r.reset_model[rb] : rng_index16(dstep.z, K);
#pragma unroll
            for (int i = 0; i < NS; ++i) prow[i] = r.pool[(size_t)row * NS + i];
        }
        float su2 = 0.0f;
#pragma unroll
        for (int d = 0; d < NA; ++d) { const float a = UA[c * NA + d]; su2 = fmaf(a, a, su2); }
        const float* __restrict__ b2 = dyn + (size_t)sel * pd.dyn.n_params + pd.dyn.b_off[2];
        float bias[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) bias[i] = b2[i];
        if (tau + 1 < z.steps) draw(t_loc + 1);                        // next step's block and noise
        // ---- output layer of head `sel`: lane q adds slices q, q + 4, ... in slice order, then the four lanes' sums are added (fixed tree)
        const unsigned long long* pq = pbase + (size_t)sel * NSL * NSP * 16;
        if (z.rot) pq += (size_t)((g + tau) % (z.U / (K * NSL))) * (size_t)(z.R * z.NT) * K * NSL * NSP * 16;      // the region of the column that serves this tile in this step
        float out[NS];
        RT_MARK(1)
#ifdef RES_NO_SENT      // the RES_TIMING build of the Ant instantiation (scratch + 141 spilled scalars) faults with this loop compiled in: its phase runs use -DRES_NO_SENT
        if (false) {
#else
        if (z.sentinel) {
#endif
            // Wait on ONE packet per slice first -- the dim a finisher stores in its last instruction -- and read the rest only when those are in: a
            // polling round over all NSL x NS packets of an env takes 2 - 4 us (two dependent batches of 60 loads per lane through the fabric), and the
            // round that finds everything started, on average, half a round before the last packet landed.  Every packet still carries its own stamp:
            // the full read below checks them all and repeats if one is behind.
            constexpr int SENT = 16 * (C::OUT_CB - 1) + ((NS - 16 * (C::OUT_CB - 1) - 1 < 3) ? NS - 16 * (C::OUT_CB - 1) - 1 : 3);
            ResSpin sp;
            for (;;) {
                bool ok = true;
                if (active) {
                    for (int s0 = 0; s0 < NSL; s0 += 16) {
                        unsigned long long pk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = res_ld(pq + ((size_t)(s0 + 4 * j + q) * NSP + SENT) * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) ok = ok && res_fresh(pk[j], seq);
                    }
                }
                if (__all(ok)) break;
                if (sp.give_up(z)) return;
            }
        }
        {
            ResSpin sp;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < NS; ++i) out[i] = 0.0f;
                if (active) {
                    for (int s0 = 0; s0 < NSL; s0 += 16) {
#pragma unroll
                        for (int d0 = 0; d0 < NS; d0 += DCH) {
                            unsigned long long pk[4][DCH];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int i = 0; i < DCH; ++i) if (d0 + i < NS) pk[j][i] = res_ld(pq + ((size_t)(s0 + 4 * j + q) * NSP + d0 + i) * 16);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int i = 0; i < DCH; ++i)
                                    if (d0 + i < NS) { ok = ok && res_fresh(pk[j][i], seq); out[d0 + i] += __uint_as_float((unsigned int)pk[j][i]); }
                        }
                    }
                }
                if (__all(ok)) break;
                if (sp.give_up(z)) return;
            }
        }
        RT_MARK(2)
        RT_WALL(3, g == 0, tau)
        // ---- de-normalise + residual (training.py:257), reward (env_helpers.py:601)
        float v[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) v[i] = fmaf(dsd[i], xor_sum(out[i]) + bias[i], dmn[i]) + s[i];
        float cost = 0.0f;
        if constexpr (ENV == METRPO_ENV_SWIMMER) cost = -(v[5] - 1e-2f * (su2 / (float)NA));
        else if constexpr (ENV == METRPO_ENV_HALF_CHEETAH) cost = -fminf(fmaxf(v[9] - 1e-1f * 0.5f * su2, -10.0f), 10.0f);
        else if constexpr (ENV == METRPO_ENV_HOPPER) {
            float pen = 0.0f;
#pragma unroll
            for (int j = 2; j < NS; ++j) pen += fmaxf(fabsf(v[j]) - 100.0f, 0.0f);
            cost = -(v[5] - 0.01f * 0.5f * su2 - 10.0f * fmaxf(0.45f - v[0], 0.0f) - 10.0f * fmaxf(fabsf(v[1]) - 0.2f, 0.0f) - pen);
        } else if constexpr (ENV == METRPO_ENV_SNAKE) cost = -(v[7] - 1e-2f * 0.5f * su2);
        else if constexpr (ENV == METRPO_ENV_ANT) cost = -(v[15] - 1e-2f * 0.5f * su2 + 0.05f);
        bool dn = dn_h;
        if constexpr (STATE_DONE) {                                   // not (0.2 <= z <= 1.0 and every state dim finite), as k_big_post
            bool fin = true;
#pragma unroll
            for (int i = 0; i < NS; ++i) fin = fin && isfinite(v[i]);
            dn = dn_h || !((v[2] >= 0.2f) && (v[2] <= 1.0f) && fin);
        }
        const int next_model = dn ? reset_model : cur_model;
        if (active && q == 0) { r.rew[tb] = -cost; r.done[tb] = dn ? 1 : 0; r.tpath[tb] = ts_new - 1; }
#pragma unroll
        for (int i = 0; i < NS; ++i) s[i] = !active ? 0.0f : (dn ? prow[i] : v[i]);
        ts = dn ? 0 : ts_new;
        cur_model = next_model;
        if (round == z.rounds_total - 1 && tau == z.steps - 1 && active) {
            if (r.last_obs != nullptr) {
#pragma unroll
                for (int i = 0; i < NS; ++i) if ((i & 3) == q) r.last_obs[(size_t)b * NS + i] = s[i];
            }
            if (q == 0) { if (r.last_ts != nullptr) r.last_ts[b] = ts; if (r.last_model != nullptr) r.last_model[b] = cur_model; }
        }
        RT_MARK(3)
    }
    RT_DUMP(1, z.U)
}

// ---- post role of the validation-cost mode (build_policy_graph's forward, model_based_rl.py:106-151: what k_validation / k_det_mfma compute) ---------
// Wave = (model m, env tile): deterministic policy (action = clipped mean), next state from head m only, cost of the state reached, Ant's sticky
// dones mask (cost x (1 - dones), then dones |= is_done), sum_t gamma^t cost per env; no draws, no resets, no trajectory rows.  The tile's
// sum over its envs / B goes to det_part[m * NTM + tile]; k_det_cost_reduce adds a model's tiles in order.
template <int ENV, int NTW>
__device__ __forceinline__ void resident_post_det(const ProblemDesc& pd, const ResidentK& z, int B, const float* __restrict__ dyn,
                                                  const float* __restrict__ theta, const float* __restrict__ norm, float* lds) {
    using C = Cfg<ENV, 64, 32>;
    constexpr int NS = C::NS, NA = C::NA, NDROP = C::NDROP, NIN = C::NIN, PH = 32, NS_KS = C::NS_KS, NIN_KS = cdiv(NIN + 1, 4), NSP = C::NSP;
    constexpr int O_PF1 = NS_KS * 2 * 64, O_PF2 = O_PF1 + 16 * 64, O_B0 = O_PF2 + 8 * 64, O_B1 = O_B0 + 32, O_B2 = O_B1 + 32, IMG = ((O_B2 + 16 + 3) / 4) * 4;
    constexpr int PW_LDS = 2 * 16 * NA;
    constexpr int DCH = (NS <= 12) ? NS : (NS + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, q = lane >> 4;
    for (int i = tid; i < IMG; i += (int)blockDim.x) {                // policy fragment image (layout of k_big_pre_mfma, rollout_gemm.hip)
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
    // a wave serves NTW tiles (fewer post workgroups leave room for another column of compute workgroups): g_j = (wave index) * NTW + j
    const int wv = ((int)blockIdx.x - z.U) * z.PW + wave;
    const int K = pd.K, NSL = z.NSL, G = K * z.NTM;
    if (wave >= z.PW || wv * NTW >= G) return;
    const float* in_mean = norm; const float* in_std = norm + (NS + NA);
    const float* diff_mean = norm + 2 * (NS + NA); const float* diff_std = diff_mean + NS;
    float xm[NIN_KS], xr[NIN_KS], am[4], ar[4], dmn[NS], dsd[NS];
#pragma unroll
    for (int kk = 0; kk < NIN_KS; ++kk) {
        const int f = 4 * kk + q, src = (f < NS - NDROP) ? f + NDROP : NS + (f - (NS - NDROP));
        xm[kk] = (f < NIN) ? in_mean[src] : 0.0f; xr[kk] = (f < NIN) ? 1.0f / in_std[src] : 0.0f;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { const int d = 4 * q + jj; am[jj] = (d < NA) ? in_mean[NS + d] : 0.0f; ar[jj] = (d < NA) ? 1.0f / in_std[NS + d] : 0.0f; }
#pragma unroll
    for (int i = 0; i < NS; ++i) { dmn[i] = diff_mean[i]; dsd[i] = diff_std[i]; }
    int gt[NTW], mt_[NTW]; bool live[NTW], active[NTW];
    float s[NTW][NS], su2[NTW], dones[NTW];
    double acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        gt[j] = wv * NTW + j; live[j] = gt[j] < G;
        const int gc = live[j] ? gt[j] : 0;
        mt_[j] = gc / z.NTM;
        const int b = (gc % z.NTM) * 16 + c;
        active[j] = live[j] && b < B;
#pragma unroll
        for (int i = 0; i < NS; ++i) s[j][i] = active[j] ? z.s0[(size_t)b * NS + i] : 0.0f;
        acc[j] = 0.0; dones[j] = 0.0f; su2[j] = 0.0f;
    }
    float* UAw = lds + IMG + wave * NTW * PW_LDS;
    double gpow = 1.0;
    for (int tau = 0; tau < z.steps; ++tau) {
        const unsigned int seq = z.seq0 + (unsigned int)tau + 1u;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
        if (!live[j]) continue;
        float* UA = UAw + j * PW_LDS; float* XA = UA + 16 * NA;
        unsigned long long* xp = z.X + ((size_t)gt[j] * (4 * NIN_KS)) * 16 + c;
        f32x4 p0[2], p1[2];
        p0[0] = *(const f32x4*)&lds[O_B0 + 4 * q]; p0[1] = *(const f32x4*)&lds[O_B0 + 16 + 4 * q];
#pragma unroll
        for (int s_ = 0; s_ < NS_KS; ++s_) {
            const float xs = sel4(q, s[j][4 * s_], (4 * s_ + 1 < NS) ? s[j][(4 * s_ + 1 < NS) ? 4 * s_ + 1 : 0] : 0.0f, (4 * s_ + 2 < NS) ? s[j][(4 * s_ + 2 < NS) ? 4 * s_ + 2 : 0] : 0.0f,
                                  (4 * s_ + 3 < NS) ? s[j][(4 * s_ + 3 < NS) ? 4 * s_ + 3 : 0] : 0.0f);
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
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int d = 4 * q + jj;
            if (d >= NA) continue;
            const float ac = fminf(fmaxf(mu[jj], -1.0f), 1.0f);       // model_based_rl.py:128
            UA[c * NA + d] = ac;
            XA[c * NA + d] = (ac - am[jj]) * ar[jj];
        }
        wave_lds_sync();
#pragma unroll
        for (int kk = 0; kk < NIN_KS; ++kk) {
            const int f = 4 * kk + q;
            constexpr int NSD = NS - NDROP;
            auto sd = [&](int e) { const int i = 4 * kk + e + NDROP; return (4 * kk + e < NSD) ? s[j][(i < NS) ? i : 0] : 0.0f; };
            float v = sel4(q, sd(0), sd(1), sd(2), sd(3));
            if (f >= NSD && f < NIN) v = XA[c * NA + f - NSD]; else v = (v - xm[kk]) * xr[kk];
            if (!active[j] || f > NIN) v = 0.0f;
            else if (f == NIN) v = 1.0f;
            res_st(xp + f * 16, seq, v);
        }
        float su = 0.0f;
#pragma unroll
        for (int d = 0; d < NA; ++d) { const float a = UA[c * NA + d]; su = fmaf(a, a, su); }
        su2[j] = su;
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
        if (!live[j]) continue;
        const int m = mt_[j];
        const unsigned long long* pq = z.P + (((size_t)gt[j] * K + m) * NSL) * NSP * 16 + c;
        const float* __restrict__ b2 = dyn + (size_t)m * pd.dyn.n_params + pd.dyn.b_off[2];
        float out[NS];
        {
            ResSpin sp;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < NS; ++i) out[i] = 0.0f;
                if (active[j]) {
                    for (int s0_ = 0; s0_ < NSL; s0_ += 16) {
#pragma unroll
                        for (int d0 = 0; d0 < NS; d0 += DCH) {
                            unsigned long long pk[4][DCH];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int i = 0; i < DCH; ++i) if (d0 + i < NS) pk[jj][i] = res_ld(pq + ((size_t)(s0_ + 4 * jj + q) * NSP + d0 + i) * 16);
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                for (int i = 0; i < DCH; ++i)
                                    if (d0 + i < NS) { ok = ok && res_fresh(pk[jj][i], seq); out[d0 + i] += __uint_as_float((unsigned int)pk[jj][i]); }
                        }
                    }
                }
                if (__all(ok)) break;
                if (sp.give_up(z)) return;
            }
        }
        float v[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) v[i] = fmaf(dsd[i], xor_sum(out[i]) + b2[i], dmn[i]) + s[j][i];
        const float su = su2[j];
        float cost = 0.0f;
        if constexpr (ENV == METRPO_ENV_SWIMMER) cost = -(v[5] - 1e-2f * (su / (float)NA));
        else if constexpr (ENV == METRPO_ENV_HALF_CHEETAH) cost = -fminf(fmaxf(v[9] - 1e-1f * 0.5f * su, -10.0f), 10.0f);
        else if constexpr (ENV == METRPO_ENV_HOPPER) {
            float pen = 0.0f;
#pragma unroll
            for (int jj = 2; jj < NS; ++jj) pen += fmaxf(fabsf(v[jj]) - 100.0f, 0.0f);
            cost = -(v[5] - 0.01f * 0.5f * su - 10.0f * fmaxf(0.45f - v[0], 0.0f) - 10.0f * fmaxf(fabsf(v[1]) - 0.2f, 0.0f) - pen);
        } else if constexpr (ENV == METRPO_ENV_SNAKE) cost = -(v[7] - 1e-2f * 0.5f * su);
        else if constexpr (ENV == METRPO_ENV_ANT) {                   // cost_tf(..., dones) then the dones update (model_based_rl.py:134-137)
            cost = -(v[15] - 1e-2f * 0.5f * su + 0.05f) * (1.0f - dones[j]);
            bool fin = true;
#pragma unroll
            for (int i = 0; i < NS; ++i) fin = fin && isfinite(v[i]);
            dones[j] = fmaxf(dones[j], ((v[2] >= 0.2f) && (v[2] <= 1.0f) && fin) ? 0.0f : 1.0f);
        }
        if (active[j]) acc[j] += gpow * (double)cost;
#pragma unroll
        for (int i = 0; i < NS; ++i) s[j][i] = active[j] ? v[i] : 0.0f;
        }
        gpow *= (double)z.gamma;
    }
    // a tile's sum over its envs (lanes q = 0 hold one env each; fixed butterfly over the 16 c-lanes), / B
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        double t = (q == 0 && active[j]) ? acc[j] : 0.0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0 && live[j]) z.det_part[(size_t)mt_[j] * z.part_stride + z.part_off + gt[j] % z.NTM] = t / (double)z.Btot;
    }
}

template <int ENV, int DH, int WS, int NTW>
__global__ void __launch_bounds__(256, 1) k_validation_resident(ProblemDesc pd, int B, ResidentK z, const float* __restrict__ dyn,
                                                                const float* __restrict__ theta, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using C = Cfg<ENV, 64, 32>;
    if ((int)blockIdx.x < z.U) resident_compute_wide<C::NS, C::NIN, DH, WS>(pd, z, dyn, lds);
    else resident_post_det<ENV, NTW>(pd, z, B, dyn, theta, norm, lds);
}

template <int ENV, int DH, int WS>
__global__ void __launch_bounds__(512) k_rollout_resident(ProblemDesc pd, RolloutK r, ResidentK z, const float* __restrict__ dyn,
                                                          const float* __restrict__ theta, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using C = Cfg<ENV, 64, 32>;
    if (r.stop != nullptr && *r.stop != 0) return;                   // the sampling loop already ended (metrpo_sampler_progress); uniform over the grid
    if ((int)blockIdx.x == z.skip_block) return;
    if ((int)blockIdx.x < z.U) resident_compute<C::NS, C::NIN, DH, WS>(pd, z, dyn, lds);
    else resident_post<ENV>(pd, r, z, dyn, theta, norm, lds);
}

template <int ENV, int DH, int WS, bool ROT = false>
__global__ void __launch_bounds__(256, 1) k_rollout_resident_wide(ProblemDesc pd, RolloutK r, ResidentK z, const float* __restrict__ dyn,
                                                                  const float* __restrict__ theta, const float* __restrict__ norm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using C = Cfg<ENV, 64, 32>;
    if (r.stop != nullptr && *r.stop != 0) return;
    if ((int)blockIdx.x == z.skip_block) return;
    if ((int)blockIdx.x < z.U) resident_compute_wide<C::NS, C::NIN, DH, WS, ROT>(pd, z, dyn, lds);
    else resident_post<ENV>(pd, r, z, dyn, theta, norm, lds);
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------
template <int ENV, int DH, int WS> static size_t resident_lds_bytes() {
    using C = Cfg<ENV, 64, 32>;
    constexpr int MT = WS / 16;
    const size_t comp = (size_t)(32 + 8 * 4 * MT * 256) * sizeof(float);
    const size_t post = (size_t)((C::NS_KS * 2 + 24) * 64 + 84 + 8 * (2 * 16 * C::NA)) * sizeof(float);
    return std::max(comp, post);
}
template <int ENV, int DH, int WS> static size_t resident_lds_bytes_wide() {
    using C = Cfg<ENV, 64, 32>;
    constexpr int MT = WS / 16, OUT_CB = (C::NS + 15) / 16, NIN_KS = (C::NIN + 1 + 3) / 4, W0L = (NIN_KS > 6) ? NIN_KS - 6 : 0, NSLOT = W0L ? 4 : 8;
    const size_t comp = (size_t)(48 + OUT_CB * MT * 256 + MT * 256 + 4 * (DH / 64) * W0L * 64 + NSLOT * 4 * MT * 256) * sizeof(float);
    const size_t post = (size_t)((C::NS_KS * 2 + 24) * 64 + 84 + 4 * (2 * 16 * C::NA)) * sizeof(float);
    return std::max(comp, post);
}
typedef void (*resident_kernel_t)(ProblemDesc, RolloutK, ResidentK, const float*, const float*, const float*);
struct ResidentEntry { int env, ns, na, n_drop, dh, ws, threads; resident_kernel_t fn; size_t lds; resident_kernel_t fn_rot; };     // threads: 512 = 4 producer + 4 finisher waves, 256 = 4 waves doing both (2 x 1024)
#define RES_ENTRY(ENV, DH, WS) {ENV, EnvDim<ENV>::NS, EnvDim<ENV>::NA, EnvDim<ENV>::NDROP, DH, WS, 512, k_rollout_resident<ENV, DH, WS>, resident_lds_bytes<ENV, DH, WS>(), nullptr}
#define RES_ENTRY_WIDE(ENV, DH, WS) {ENV, EnvDim<ENV>::NS, EnvDim<ENV>::NA, EnvDim<ENV>::NDROP, DH, WS, 256, k_rollout_resident_wide<ENV, DH, WS>, resident_lds_bytes_wide<ENV, DH, WS>(), k_rollout_resident_wide<ENV, DH, WS, true>}
static const ResidentEntry* resident_table(int* n) {
    static const ResidentEntry tab[] = {                                // per (env, width): narrowest slice first
        RES_ENTRY(METRPO_ENV_SWIMMER, 512, 16), RES_ENTRY(METRPO_ENV_SWIMMER, 512, 32),
        RES_ENTRY(METRPO_ENV_HOPPER, 512, 32), RES_ENTRY(METRPO_ENV_SNAKE, 512, 32), RES_ENTRY(METRPO_ENV_HALF_CHEETAH, 512, 32),
        RES_ENTRY_WIDE(METRPO_ENV_HOPPER, 1024, 64), RES_ENTRY_WIDE(METRPO_ENV_SNAKE, 1024, 64), RES_ENTRY_WIDE(METRPO_ENV_HALF_CHEETAH, 1024, 64),
        RES_ENTRY_WIDE(METRPO_ENV_SWIMMER, 1024, 64), RES_ENTRY_WIDE(METRPO_ENV_ANT, 1024, 64),
    };
    *n = (int)(sizeof(tab) / sizeof(tab[0]));
    return tab;
}

// METRPO_EUNSUPPORTED: this shape / call stays on the step-wise path (rollout_gemm.hip)
int launch_rollout_resident(metrpo_ctx* c, const metrpo_rollout_args* a, hipStream_t st) {
    const ProblemDesc& pd = c->pd;
    if (c->rollout_variant == 1 || c->res_failed || !ctx_exclusive(c)) return METRPO_EUNSUPPORTED;
    if (pd.dyn.n_layers != 3 || pd.dyn.dims[1] != pd.dyn.dims[2] || pd.dyn.act[0] != METRPO_ACT_RELU || pd.dyn.act[1] != METRPO_ACT_RELU ||
        pd.dyn.act[2] != METRPO_ACT_IDENTITY) return METRPO_EUNSUPPORTED;
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH || pd.pol.act[1] != METRPO_ACT_TANH)
        return METRPO_EUNSUPPORTED;
    if (!(a->sam_mode == METRPO_SAM_STEP_RAND || a->sam_mode == METRPO_SAM_EPS_RAND || a->sam_mode == METRPO_SAM_ONE_MODEL)) return METRPO_EUNSUPPORTED;
    if (a->B > 128) return METRPO_EUNSUPPORTED;
    // The compute and post workgroups of a launch wait for each other's packets: the WHOLE grid has to be on the chip at once.  The grid is sized by
    // the CUs that really schedule this process's waves (a census, not the device property: CU masks, partitions, reserved CUs) and checked
    // against the runtime's occupancy answer for the kernel picked below -- a launch that would not be co-resident is never issued (it used to burn
    // its 2 s hand-over bound first and report invalid trajectories); the bounded wait stays as the backstop against another tenant.
    const int n_cu = sched_cus(c, st);
    const int B = a->B, K = pd.K, H = a->H, DH = pd.dyn.dims[1], NT = (B + 15) / 16;
    // rounds of a horizon-terminated rollout are independent given the counter-based draws (see launch_rollout_gemm): they run side by side
    int R = 1;
    if (H > 0 && a->T % H == 0 && a->T / H >= 2 && pd.env != METRPO_ENV_ANT && a->t0 == 0 && a->d_init_obs == nullptr && a->d_stop == nullptr &&
        ctx_opt(c, OPT_SEQ_ROUNDS) == nullptr) R = a->T / H;
    const int steps = a->T / R;
    int n = 0;
    const ResidentEntry* tab = resident_table(&n);
    // narrowest slice (most CUs, least work per step) whose grid -- all rounds side by side -- still fits the chip; if even the widest
    // slice does not fit, as many rounds at a time as do (the last group may be smaller), every group a launch of its own
    const ResidentEntry* pick = nullptr; int Rg = 0, PW = 0;
    auto fits = [&](const ResidentEntry* e, int rg, int* pw_out) {
        if (e->threads == 256) {                                        // 4-wave form: the tiles of the rg rounds are dealt to as many workgroup columns as fit -- one is enough
            *pw_out = 4;
            return K * (DH / e->ws) + (rg * NT + 3) / 4 <= n_cu;
        }
        for (int pw = 1; pw <= e->threads / 64; pw *= 2)
            if (rg * K * (DH / e->ws) + (rg * NT + pw - 1) / pw <= n_cu) { *pw_out = pw; return true; }
        return false;
    };
    // (the post wave adds the slices in batches of 16: DH / ws must be a multiple of 16)
    auto matches = [&](const ResidentEntry& e) { return e.env == pd.env && e.ns == pd.ns && e.na == pd.na && e.n_drop == pd.n_drop && e.dh == DH && (DH / e.ws) % 16 == 0; };
    const char* ws_env = ctx_opt(c, OPT_RESIDENT_WS);                // test hook: pin the slice width (results are bit-identical only at equal widths)
    const ResidentEntry* widest = nullptr;
    for (int i = 0; i < n && !pick; ++i) {
        if (!matches(tab[i]) || (ws_env != nullptr && atoi(ws_env) != tab[i].ws)) continue;
        widest = &tab[i];
        if (fits(&tab[i], R, &PW)) { pick = &tab[i]; Rg = R; }
    }
    for (int rg = R - 1; widest && rg >= 1 && !pick; --rg)
        if (fits(widest, rg, &PW)) { pick = widest; Rg = rg; }
    if (!pick) return METRPO_EUNSUPPORTED;
    if (pick->lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)pick->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick->lds));
    if (!grid_is_coresident(c, (const void*)pick->fn, pick->threads, pick->lds, 1, st)) return METRPO_EUNSUPPORTED;      // >= one workgroup per CU, exclusive device
    const int NSL = DH / pick->ws, OUT_CB = (pd.ns + 15) / 16, NIN_KS = (pd.nin + 1 + 3) / 4;
    // rotating deal (ResidentK::rot): launches of the 4-wave form whose few tiles do not divide by the columns; one partial-sum region per column
    const bool may_rot = pick->threads == 256 && pick->fn_rot != nullptr && !resident_plan_has(c, "norotate");
    const int max_cols = std::max(1, n_cu / (K * (DH / pick->ws)));
    const size_t nX = (size_t)Rg * NT * 4 * NIN_KS * 16, nP = (size_t)Rg * NT * K * NSL * 16 * OUT_CB * 16 * (may_rot ? max_cols : 1);
    const size_t need = (nX + nP + 32) * sizeof(unsigned long long);
    if (need > c->res_cap) {
        ws_retire(c, c->d_res);
        c->d_res = nullptr; c->res_cap = 0;
        // ORDINARY device memory.  The packets only ever move through agent-scope atomics (L2-served, never L1), so the memory type buys nothing:
        // uncached, fine-grained and ordinary memory all measure 2.73 ms at the params-file shape.  It is ordinary memory because a region that was
        // allocated hipDeviceMallocUncached and later hipFree'd can come back from hipMalloc as somebody's ordinary buffer with lines of its old
        // life still sitting in one XCD's L2: seen as a step-wise rollout (rollout_gemm.hip) of a LATER engine reading two stale cache lines of
        // its fresh workspace, gone after evicting the L2s (tests/test_gpu_resident.py::test_stepwise_workspace_after_freed_resident_regions).
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_res, need));
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, need, st));
        c->res_cap = need; c->res_seq = 0;
    }
    if ((unsigned long long)c->res_seq + (unsigned long long)((R + Rg - 1) / Rg) * (steps + 1) >= 0xfffffff0ull) {   // stamps would wrap: start over on a clean region
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, c->res_cap, st));
        c->res_seq = 0;
    }
    if (pick->lds > 64 * 1024) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)pick->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick->lds));
        if (pick->fn_rot) HIP_TRY(c, hipFuncSetAttribute((const void*)pick->fn_rot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick->lds));
    }
    RolloutK rk = make_rollout_k(a);
    for (int round0 = 0; round0 < R; round0 += Rg) {
        const int rg = std::min(Rg, R - round0);
        ResidentK z;
        z.R = rg; z.round0 = round0; z.rounds_total = R; z.NT = NT; z.NSL = NSL; z.U = rg * K * NSL; z.PW = PW; z.steps = steps; z.NTC = NT; z.rot = 0;
        z.det = 0; z.NTM = 0; z.gamma = 1.0f; z.s0 = nullptr; z.det_part = nullptr; z.part_stride = 0; z.part_off = 0; z.Btot = 0;
        if (pick->threads == 256) {
            const int G = rg * NT, post_blocks = (G + PW - 1) / PW;
            const int ncol = std::max(1, std::min(G, (n_cu - post_blocks) / (K * NSL)));
            z.NTC = (G + ncol - 1) / ncol;
            z.U = ((G + z.NTC - 1) / z.NTC) * K * NSL;
            const int cols = z.U / (K * NSL);
            z.rot = (may_rot && cols > 1 && z.NTC <= 4 && G % cols != 0 && G >= cols) ? 1 : 0;
        }
        z.seq0 = c->res_seq; c->res_seq += (unsigned int)steps + 1u;
        z.skip_block = -1;
        // few tiles per column (Ant's chunks, single rounds): the step is a latency chain and the post wave's polling rounds are on it; with many tiles per
        // column the partial sums are there before the post wave looks, and the extra round trip of the sentinel read costs 2 - 4 % (half-cheetah, 5 rounds)
        z.sentinel = (pick->threads == 256 && z.NTC <= 4 && !resident_plan_has(c, "nosentinel")) ? 1 : 0;
        if (const char* sk = ctx_opt(c, OPT_RESIDENT_TEST_SKIP)) z.skip_block = atoi(sk);
        z.abort_cell = (unsigned int*)c->d_res;
        z.X = (unsigned long long*)c->d_res + 32; z.P = z.X + nX;
        z.err = comm_err_cell(c) + 1;                               // scal[S_ROLLERR]
        const int grid = z.U + (rg * NT + PW - 1) / PW;
        if (grid > n_cu) return set_err(c, METRPO_EHIP, "resident rollout: grid larger than the schedulable CUs (launch rule out of step with the census)");
        hipLaunchKernelGGL(z.rot ? pick->fn_rot : pick->fn, dim3(grid), dim3(pick->threads), pick->lds, st, pd, rk, z, c->d_dyn, c->d_theta, c->d_norm);
    }
    HIP_TRY(c, hipGetLastError());
    c->last_rollout_kernel = 4;
    return METRPO_OK;
}

// ---- validation costs (metrpo_validation_cost) on the resident kernel's 4-wave form ------------------------------------------------------------
typedef void (*resident_val_kernel_t)(ProblemDesc, int, ResidentK, const float*, const float*, const float*);
struct ResidentValEntry { int env, ns, na, n_drop, dh, ws; resident_val_kernel_t fn[3]; size_t lds; };      // fn[i]: a post wave serves 1 << i tiles
template <int ENV, int DH, int WS> static size_t resident_val_lds_bytes() {                                 // compute role | post role with four tiles per wave
    using C = Cfg<ENV, 64, 32>;
    return std::max(resident_lds_bytes_wide<ENV, DH, WS>(), (size_t)((C::NS_KS * 2 + 24) * 64 + 84 + 4 * 4 * (2 * 16 * C::NA)) * sizeof(float));
}
#define RES_VAL_ENTRY(ENV, DH, WS) {ENV, EnvDim<ENV>::NS, EnvDim<ENV>::NA, EnvDim<ENV>::NDROP, DH, WS, \
    {k_validation_resident<ENV, DH, WS, 1>, k_validation_resident<ENV, DH, WS, 2>, k_validation_resident<ENV, DH, WS, 4>}, resident_val_lds_bytes<ENV, DH, WS>()}
// METRPO_EUNSUPPORTED: this shape stays on the step-wise sweep (det_gemm.hip)
int launch_validation_resident(metrpo_ctx* c, const float* s0, int Bv, int T, double gamma, double* costs, hipStream_t st) {
    static const ResidentValEntry tab[] = {
        RES_VAL_ENTRY(METRPO_ENV_SWIMMER, 512, 32), RES_VAL_ENTRY(METRPO_ENV_HOPPER, 512, 32), RES_VAL_ENTRY(METRPO_ENV_SNAKE, 512, 32), RES_VAL_ENTRY(METRPO_ENV_HALF_CHEETAH, 512, 32),
        RES_VAL_ENTRY(METRPO_ENV_SWIMMER, 1024, 64), RES_VAL_ENTRY(METRPO_ENV_HOPPER, 1024, 64), RES_VAL_ENTRY(METRPO_ENV_SNAKE, 1024, 64), RES_VAL_ENTRY(METRPO_ENV_HALF_CHEETAH, 1024, 64),
        RES_VAL_ENTRY(METRPO_ENV_ANT, 1024, 64),
    };
    const ProblemDesc& pd = c->pd;
    if (c->res_failed || !ctx_exclusive(c) || T <= 0) return METRPO_EUNSUPPORTED;
    if (pd.dyn.n_layers != 3 || pd.dyn.dims[1] != pd.dyn.dims[2] || pd.dyn.act[0] != METRPO_ACT_RELU || pd.dyn.act[1] != METRPO_ACT_RELU ||
        pd.dyn.act[2] != METRPO_ACT_IDENTITY) return METRPO_EUNSUPPORTED;
    if (pd.pol.n_layers != 3 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32 || pd.pol.act[0] != METRPO_ACT_TANH || pd.pol.act[1] != METRPO_ACT_TANH)
        return METRPO_EUNSUPPORTED;
    const int K = pd.K, DH = pd.dyn.dims[1];
    const ResidentValEntry* e = nullptr;
    for (const ResidentValEntry& x : tab)
        if (x.env == pd.env && x.ns == pd.ns && x.na == pd.na && x.n_drop == pd.n_drop && x.dh == DH) { e = &x; break; }
    if (!e) return METRPO_EUNSUPPORTED;
    // model k's workgroups (K x DH / ws of them per column) deal the batch's env tiles to as many columns as fit next to the post workgroups (one wave
    // per (model, tile), four per workgroup).  The batch goes in nb chunks, one launch each: fewer tiles per launch need fewer post workgroups and
    // leave room for a third column -- 500 envs: one launch = 2 columns of 16 tiles per step, three launches = 3 x (3 columns of 4).
    const int NSL = DH / e->ws, PW = 4;
    const int n_cu = sched_cus(c, st);                                      // as in launch_rollout_resident: the grid must be co-resident
    // Cost of a step (us): a column's tiles at the measured tile time (2.0 at 2 x 512 / 32-unit slices, 6.1 at 2 x 1024 / 64), but never less than a
    // tile's round trip through its post wave (8, plus 3 for every further tile the wave serves first), per chunk; plus the launch's prologue
    // (~60 us: weight fragments, LDS images) spread over the T steps.  A post wave serving several tiles frees CUs for another column.
    int nb = 0, NTM = 0, NTC = 0, cols = 0, ntw_i = 0;
    double best = 1e30;
    const double t_tile = (DH >= 1024) ? 6.1 : 2.0, t_trip = 8.0, t_post = 3.0, t_launch = 60.0;
    // test hook VAL_PLAN = "<tiles per post wave: 1 | 2 | 4 | 0 = the model's pick>[,<batch chunks (launches): 1 .. 8>]"
    int ntw_pin = 0, chunks_pin = 0;
    if (const char* vp = ctx_opt(c, OPT_VAL_PLAN)) { ntw_pin = atoi(vp); if (const char* cm = strchr(vp, ',')) chunks_pin = atoi(cm + 1); }
    for (int wi = 0; wi < 3; ++wi) {
        const int ntw = 1 << wi;
        if (ntw_pin != 0 && ntw_pin != ntw) continue;
        for (int n = 1; n <= 8; ++n) {
            if (chunks_pin != 0 && chunks_pin != n) continue;
            const int bc = (Bv + n - 1) / n, ntm = (bc + 15) / 16, post = ((K * ntm + ntw - 1) / ntw + PW - 1) / PW;
            const int ncol = std::min(ntm, (n_cu - post) / (K * NSL));
            if (ncol < 1) continue;
            const int ntc = (ntm + ncol - 1) / ncol;
            const double cost = n * (std::max(std::max(ntc * t_tile, t_trip + (ntw - 1) * t_post), ntw * t_post * 1.5) + t_launch / T);
            if (cost < best - 1e-9) { best = cost; nb = n; NTM = ntm; NTC = ntc; cols = (ntm + ntc - 1) / ntc; ntw_i = wi; }
        }
    }
    if (nb == 0) return METRPO_EUNSUPPORTED;
    const int NTW = 1 << ntw_i;
    const int Bc = (Bv + nb - 1) / nb, G = K * NTM, post_blocks = ((G + NTW - 1) / NTW + PW - 1) / PW;
    const int OUT_CB = (pd.ns + 15) / 16, NIN_KS = (pd.nin + 1 + 3) / 4;
    const size_t nX = (size_t)G * 4 * NIN_KS * 16, nP = (size_t)G * K * NSL * 16 * OUT_CB * 16;
    const size_t need = (nX + nP + 32) * sizeof(unsigned long long);
    if (need > c->res_cap) {
        ws_retire(c, c->d_res);
        c->d_res = nullptr; c->res_cap = 0;
        HIP_TRY(c, ws_alloc(c, (void**)&c->d_res, need));                      // ordinary device memory: see launch_rollout_resident
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, need, st));
        c->res_cap = need; c->res_seq = 0;
    }
    if ((unsigned long long)c->res_seq + (unsigned long long)nb * (T + 1) >= 0xfffffff0ull) {
        HIP_TRY(c, hipMemsetAsync(c->d_res, 0, c->res_cap, st));
        c->res_seq = 0;
    }
    { const int rc = ensure_detpart_n(c, (size_t)K * nb * NTM); if (rc) return rc; }
    if (e->lds > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)e->fn[ntw_i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds));
    if (!grid_is_coresident(c, (const void*)e->fn[ntw_i], 256, e->lds, (long long)cols * K * NSL + post_blocks, st)) return METRPO_EUNSUPPORTED;
    HIP_TRY(c, hipMemsetAsync(val_err_cell(c), 0, sizeof(double), st));       // this launch chain's own time-out cell (read by k_det_cost_reduce below)
    for (int ch = 0; ch < nb; ++ch) {
        const int b_lo = ch * Bc, bn = std::min(Bc, Bv - b_lo);
        ResidentK z;
        z.R = 1; z.round0 = 0; z.rounds_total = 1; z.NT = NTM; z.NSL = NSL; z.U = cols * K * NSL; z.PW = PW; z.steps = T; z.NTC = NTC; z.rot = 0;
        z.det = 1; z.NTM = NTM; z.gamma = (float)gamma; z.s0 = s0 + (size_t)b_lo * pd.ns; z.det_part = c->d_detpart;
        z.part_stride = nb * NTM; z.part_off = ch * NTM; z.Btot = Bv;
        z.seq0 = c->res_seq; c->res_seq += (unsigned int)T + 1u;
        z.skip_block = -1; z.sentinel = 0;
        z.abort_cell = (unsigned int*)c->d_res;
        z.X = (unsigned long long*)c->d_res + 32; z.P = z.X + nX;
        z.err = val_err_cell(c);
        hipLaunchKernelGGL(e->fn[ntw_i], dim3(z.U + post_blocks), dim3(256), e->lds, st, pd, std::max(bn, 0), z, c->d_dyn, c->d_theta, c->d_norm);
    }
    HIP_TRY(c, hipGetLastError());
    return launch_det_cost_reduce(c, nb * NTM, c->d_detpart, costs, st, val_err_cell(c));
}
