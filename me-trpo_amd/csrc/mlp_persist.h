// Persistent stream-K rollout: ALL steps of a rollout chunk of a two-hidden-layer ensemble (hidden >= 256: BASELINE's C2 / C3 shapes) in ONE launch.
// The launch-per-step path (rollout_gemm.hip + mlp_streamk.h) pays, per 146 us step at the C3 share: a pre / post launch (12 us), two kernel
// boundaries, the stream-K launch's prologue and drain, the accumulator hand-over of the even split (1.56 tiles per workgroup: 64 MB per step), and
// the tile quantisation the even split exists to avoid.  None of that is needed once the time loop is inside the launch:
//
//   * the unit of work is a WHOLE tile (128 envs x 256 columns of one head: the k_mlp_sk tile, same transposed MFMA chain, same LDS-DMA ring, same operand
//     reads -- mlp_streamk.h), the tiles of ALL steps form one sequence, and workgroup (x, i) -- XCD list x = blockIdx % 8, i = blockIdx / 8 -- takes
//     positions i, i + G/8, ... of list x.  Over hundreds of steps every workgroup gets the same number of tiles to within one: no split tiles, no hand-over;
//   * step t + 1 of a 128-env row block needs only step t of THAT row block (all heads and column blocks: NSL tiles).  Every finished tile adds one to
//     the row block's arrival counter (a fire-and-forget atomic behind the barrier that ends its last chunk).  A FEW workgroups of the launch (NCLOSE = 8, one
//     per XCD; the other 248 compute) do nothing but close steps: closer c watches the counters of row blocks c, c + 8, ..., and as soon as a row block's step
//     is complete it de-normalises + adds the residual, selects over the heads, forms reward / done / reset, evaluates the policy and writes the normalised
//     input row of step t + 1 for those 128 envs -- the wave functions of big_prepost.h, the very code the launch-per-step path runs as
//     k_big_pre_mfma<ENV, true> -- and raises the row block's ready flag.  A tile waits for its row block's flag before it loads its input rows (probed a
//     chunk ahead, so the common case costs nothing).  The tiles of one step are ordered row block first: a row block's tiles finish together and, once the
//     workgroups have staggered themselves, about half a tile time before the next step's tiles of that row block come up;
//     (first version: the workgroup whose arrival completed a step closed it itself, through a call inside the chunk loop.  The returning atomic, the votes
//     that keep a workgroup from blocking while it owes a closing, and above all the call -- every loop-carried register live across it -- cost 73 scalar and
//     28 vector spills INSIDE the matrix phase: 6.5 us per chunk against k_mlp_sk's 4.7.  Compute workgroups now owe nothing, may block freely, and their
//     loop is k_mlp_sk's plus a scalar cursor.)
//   * each XCD list holds the tiles of 1/8 of the (head, column block) weight slices (2.5 slices of 20 at the C2 / C3 shapes), row block first: the
//     workgroups behind one L2 stream the same 2-3 slices, which therefore stay in that L2 for the whole launch.
//
// Inter-workgroup visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   tile -> closer:   output partials by 16-byte sc1 (write-through) stores, every wave drains (vmcnt 0), workgroup barrier, ONE agent-scope atomic add on the row
//                     block's counter.  The closer: relaxed agent-scope poll, barrier, sc1 loads of the partials (SKP_NO_FENCES = 0: a one-lane agent acquire first);
//   closer -> tiles:  X rows by sc1 stores; S / U / ts / cur_model (read by the same closer one step later) by plain stores; every wave drains, barrier, one-lane
//                     relaxed agent store of the flag (SKP_NO_FENCES = 0: an agent release + drain in front of it).  A tile's waves poll the flag and read their X rows with agent-scope loads.
// Waits are bounded (2 s) and report through the sticky rollout error cell; the launch needs its whole grid on the chip (one workgroup per CU:
// grid_is_coresident, probe.hip) -- otherwise, and for every shape outside (two hidden layers, producer-sized input, 2 x 32 policy), the launch-per-step
// path runs.  Summation order of every output: the k-ordered chain of an unsplit k_mlp_sk tile -- bit for bit the launch-per-step path's results.
#pragma once
#include <vector>
#include "mlp_streamk.h"
#include "big_prepost.h"

struct SkpPost {                                   // what the closing part needs; lives in device memory (written by k_skp_post_args in front of the launch)
    ProblemDesc pd; RolloutK r; BigState st; const float* theta; const float* norm;
};
struct SkpArgs {
    SkArgs a;                                      // shapes and pointers of the fused launch (x rows, W0, W1, epilogue images, output partials): mlp_streamk.h
    const SkRec* tab;                              // [8][Jmax][L] chunk records of ONE step's tiles, per XCD list, row block first; [8 Jmax L] = the sentinel record
    int Jx[8];                                     // tiles per step in list x
    int nclose;                                    // closing workgroups (1 .. 8): blockIdx < nclose; the compute workgroups of list x are the other blockIdx = x (mod 8)
    int Jmax, L, G8, T, NSL;                       // entries per tile (chunks + epilogue chunks); workgroups per list; steps of this launch; tiles per row-block step
    int* xflag; unsigned* arrive;                  // [RB] each, zeroed in front of the launch: x of step t is ready when xflag[rb] >= t; arrivals so far
    const int32_t* stop;                           // metrpo_sampler_progress's flag (constant during the launch)
    const SkpPost* post;
    // in-launch stop rule (metrpo_rollout_args::stop_batch; 0 = off): cnt[T] samples of the paths completed at each step | closed[T] row blocks closed per step |
    // misc: [0] lock, [1] next step to fold, [2] stop step (SKP_NOHALT until known) | cum: samples folded so far (without *stop_cum0)
    long long stop_batch; const double* stop_cum0; unsigned long long* cnt; unsigned* closed; unsigned* misc; unsigned long long* cum;
    int nowait;                                    // developer timing (option PERSIST_STATS=2; results INVALID): no tile waits for its row block's flag
    unsigned long long* stats;                     // developer statistics (option PERSIST_STATS; NULL otherwise): per workgroup {100 MHz ticks of the launch, ticks blocked (compute: on a flag,
                                                   // longest wave; closer: on arrival counters), 0, steps closed, ticks spent closing, tiles, ticks in the prologue, 1 = closing role}
};

__global__ void k_skp_post_args(SkpPost v, SkpPost* dst) { if (threadIdx.x == 0) *dst = v; }

enum { SKP_HALT = 0x40000000, SKP_NOHALT = 0x7fffffff };      // ready-flag value "nothing behind the stop step is computed any more" | misc[2] before the stop step is known
// 1 (round 5): the closing workgroups issue no cache-wide release / acquire.  What crosses workgroups inside the launch already travels by sc1 stores and loads (the
// tiles' partials, the X rows); S / U / ts / cur_model are read back by the closer that wrote them; the trajectory tensors are read behind the launch.  Closing 17.5 -> 17.2 us
// at C3, rollouts 110.5 -> 110.2 / 47.2 -> 47.1 ms (C3 / C2); bitwise tests and the 400-rollout soak unchanged.
#ifndef SKP_NO_FENCES
#define SKP_NO_FENCES 1
#endif
#if SKP_NO_FENCES && !defined(__gfx942__) && !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "SKP_NO_FENCES relies on gfx942 / gfx950 sc1 write-through stores: build other targets with -DSKP_NO_FENCES=0"
#endif
enum { SKP_NCLOSE = 8 };                           // most closing workgroups of a launch (SkpArgs::nclose of them: blockIdx 0 .. nclose - 1, on different XCDs)

// Step t closed and step t + 1 prepared for the 128 envs of row block rb, by all 8 waves of a closing workgroup: the wave functions of the launch-per-step
// pre-kernel (big_prepost.h).  scratch: [8][16][NS] floats of LDS state tiles; img: the policy image (PreImg<ENV>).
template <int ENV>
__device__ __forceinline__ void skp_close_and_prepare(const SkpPost* __restrict__ pp, int t, int rb, const float* img, float* scratch, int* xflag,
                                                      unsigned long long* cnt_t = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b0 = rb * 128 + wave * 16;
    int len = 0;
    if (b0 < pp->r.B) {
        PreLane<ENV> pl;
        float* ST = scratch + wave * 16 * Cfg<ENV, 64, 32>::NS;
        big_pre_head<ENV, true, true>(pp->pd, pp->r, t + 1, pp->theta, pp->norm, pp->st, ST, b0, lane, pl, cnt_t != nullptr ? &len : nullptr);
        big_pre_tail<ENV, true>(pp->r, t + 1, pp->st, img, ST, b0, lane, pl);
    }
    if (cnt_t != nullptr) {                                                            // this wave's completed-path samples of step t (integers: any order of the adds)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);
        if (lane == 0 && len != 0) (void)__hip_atomic_fetch_add(cnt_t, (unsigned long long)len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // every writing wave drains
    __syncthreads();
    if (threadIdx.x == 0) {
#if !SKP_NO_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // (the compiler may drop the wait behind buffer_wbl2: restated where it cannot)
#endif
        __hip_atomic_store(xflag + rb, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The closing role: workgroup c of SKP_NCLOSE watches row blocks c, c + NCLOSE, ...; for every step in order, every one of its row blocks in order.  It waits
// for nothing but arrival counters (whose tiles need only flags this closer raised a step earlier): no cycle, whatever the dispatch order of the compute workgroups.
template <int ENV>
__device__ __forceinline__ void skp_closer_role(const SkpArgs& p, float* lds) {
    using IM = PreImg<ENV>;
    float* const img = lds;
    float* const scratch = lds + IM::IMG;
    int* const sh = (int*)(scratch + 8 * 16 * Cfg<ENV, 64, 32>::NS);
    const int tid = threadIdx.x;
    for (int k = tid; k < IM::IMG; k += 512) img[k] = IM::entry(p.post->theta, k);
    __syncthreads();
    const int RB = (p.a.M + 127) / 128;
    const bool stopping = p.stop_batch > 0;
    unsigned long long st_wait = 0, st_work = 0, st_n = 0;
    const unsigned long long st_t0 = wall_clock64();
    bool halted = false;
    for (int t = 0; t + 1 < p.T && !halted; ++t)
        for (int rb = blockIdx.x; rb < RB; rb += p.nclose) {
            const unsigned long long w0 = wall_clock64();
            if (tid == 0) {
                const unsigned want = (unsigned)(t + 1) * (unsigned)p.NSL;
                int go = 1;
                if (stopping && t > (int)__hip_atomic_load(p.misc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) go = 0;
                while (go && __hip_atomic_load(p.arrive + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (stopping && t > (int)__hip_atomic_load(p.misc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) go = 0;      // the tiles behind the stop step may never arrive
                    if (wall_clock64() - w0 > 200000000ull) { __hip_atomic_store(p.a.err, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // 2 s at 100 MHz
                    if (__hip_atomic_load(p.a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) break;
                }
#if !SKP_NO_FENCES
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // the tiles' partials; (S / U / ts / cur_model are this workgroup's own stores of a step ago)
#endif
                sh[0] = go;
            }
            __syncthreads();
            if (!sh[0]) { halted = true; break; }
            const unsigned long long w1 = wall_clock64();
            skp_close_and_prepare<ENV>(p.post, t, rb, img, scratch, p.xflag, stopping ? p.cnt + t : nullptr);
            if (stopping && tid == 0) {
                // Step t is complete once all RB row blocks are closed; complete steps are folded into the running total IN STEP ORDER by whoever completes
                // one (a short critical section), and the first step at which the total reaches stop_batch becomes the stop step: every step up to it is
                // closed by then (the sampler's loop condition, tested once per step: vectorized_sampler.py:60,104).
                if (__hip_atomic_fetch_add(p.closed + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)RB) {
                    while (atomicCAS(p.misc, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
                    unsigned nx = __hip_atomic_load(p.misc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned long long cum = __hip_atomic_load(p.cum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const double cum0 = *p.stop_cum0, need = (double)p.stop_batch;
                    while ((int)nx + 1 < p.T && __hip_atomic_load(p.closed + nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)RB) {
                        cum += __hip_atomic_load(p.cnt + nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (cum0 + (double)cum >= need && __hip_atomic_load(p.misc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)SKP_NOHALT)
                            __hip_atomic_store(p.misc + 2, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ++nx;
                    }
                    __hip_atomic_store(p.cum, cum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.misc + 1, nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_store(p.misc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();                                                    // the state tiles are rewritten by the next row block
            st_wait += w1 - w0; st_work += wall_clock64() - w1; ++st_n;
        }
    if (halted && tid == 0)                                                     // nothing behind the stop step is closed: release every tile that waits on this closer's row blocks
        for (int rb = blockIdx.x; rb < RB; rb += p.nclose) __hip_atomic_store(p.xflag + rb, (int)SKP_HALT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.stats != nullptr && tid == 0) { unsigned long long* o = p.stats + 8 * (size_t)blockIdx.x; o[0] = wall_clock64() - st_t0; o[1] = st_wait; o[3] = st_n; o[4] = st_work; o[7] = 1; }
}

// WIDE: a workgroup's tile is TWO adjacent column blocks of one (head, row block) -- 128 x 512 -- whose chunks alternate in the entry sequence and share the
// layer-0 producer (each 32-unit slice of layer 0 is computed once per 512 columns instead of once per 256: -6 % matrix instructions at 2 x 512, -4 % at
// 2 x 1024).  Every column block keeps its own accumulators, its own epilogue and its own output partial: the sums are those of two separate tiles.
template <int ENV, int S0, int OT, bool WIDE>
__global__ void __launch_bounds__(512) k_sk_persist(const SkpArgs p) {
    using EP = SkEpi<OT>;
    using GE = SkGeom<SK_A_PRODUCER, SK_EPI_OUT, S0, OT>;
    constexpr int E = EP::E, UPC = EP::UPC, NI0 = GE::NI0, STAGE = GE::STAGE, NH = WIDE ? 2 : 1;
    static_assert(EP::FLOATS <= STAGE, "EPI image larger than a ring stage");
    static_assert(NI0 <= 8, "layer-0 slice: at most 8 one-KB pieces (one per wave)");
    static_assert(PreImg<ENV>::IMG + 8 * 16 * Cfg<ENV, 64, 32>::NS + 4 <= 4 * STAGE, "the closing role's image and state tiles fit the ring's LDS");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (p.stop != nullptr && *p.stop != 0) return;                             // the sampling loop already ended
    if ((int)blockIdx.x < p.nclose) { skp_closer_role<ENV>(p, lds); return; }
    float* const ring = lds;
    const SkArgs& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = blockIdx.x & 7;                                             // XCD list = the XCD the dispatcher puts this workgroup on (blockIdx % 8)
    const int wi = (int)(blockIdx.x >> 3) - (xl < p.nclose ? 1 : 0);           // first position: the list's workgroups in blockIdx order, its closer (if any) left out
    const int Jx = p.Jx[xl], L = p.L, G8 = p.G8 - (xl < p.nclose ? 1 : 0), T = p.T;      // p.G8 = gridDim / 8
    const long long npos = (long long)T * Jx;
    const int n_tiles = (wi < npos) ? (int)((npos - wi + G8 - 1) / G8) : 0;
    const int nq = n_tiles * L;
    if (nq == 0 || G8 <= 0) return;
    const unsigned long long st_t0 = wall_clock64();
    const SkRec* const tabx = p.tab + (size_t)xl * p.Jmax * L;
    const SkRec* const sentinel = p.tab + (size_t)8 * p.Jmax * L;

    // position of an entry of this workgroup's sequence: tile jj of list xl at step t, chunk c.  Scalar bookkeeping only.
    struct Cur { int jj, t, c; };
    auto adv = [&](Cur& k) { if (++k.c == L) { k.c = 0; k.jj += G8; while (k.jj >= Jx) { k.jj -= Jx; ++k.t; } } };
    auto rec_of = [&](const Cur& k) -> const SkRec* { return (k.t < T) ? tabx + ((size_t)k.jj * L + k.c) : sentinel; };

    struct RecL { int kc, fl, m0, t; unsigned offC; };                         // the chunk itself / the look-ahead (t: its tile's step)
    struct RecI { int fl; unsigned offW1, offW0; };                            // its LDS-DMA copies
    auto decL = [](const sk_i32x4& v, int t) { RecL r; r.kc = v[0] & 0xFFFF; r.fl = v[0] >> 16; r.m0 = v[1]; r.offC = (unsigned)v[3]; r.t = t; return r; };
    auto decI = [](const sk_i32x4& v) { RecI r; r.fl = v[0] >> 16; r.offW1 = (unsigned)v[1]; r.offW0 = (unsigned)v[2]; return r; };
    auto fetch2 = [&](const SkRec* pl_, const SkRec* pi_, int tl, RecL& rl, RecI& ri) {       // load + wait in ONE statement (mlp_streamk.h: why)
        sk_i32x4 vl, vi;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(vl), "=&s"(vi) : "s"(pl_), "s"(pi_));
        rl = decL(vl, tl); ri = decI(vi);
    };

    const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ring;
    const unsigned w1_voff = (unsigned)(wave * a.N + 4 * lane) * 4u;
    unsigned w0_voff;
    { const int pz = (wave % NI0) * 64 + lane, i4 = pz & 3, gg = (pz >> 2) & 3, jt = (pz >> 4) & 1, s = pz >> 5; w0_voff = (unsigned)((4 * s + gg) * a.K1 + 16 * jt + 4 * i4) * 4u; }
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_piece = [&](const RecI& r, int q, auto tc) {
        constexpr int t = decltype(tc)::value;
        if (r.fl & SKF_NONE) return;
        const unsigned st = ring_lds + (unsigned)((q & 3) * STAGE) * 4u;
        if (!(r.fl & SKF_EPI)) {
            if constexpr (t < 4) {
                // (row stride made opaque per use: left visible, the four row-block bases W1 + t * 32 N of every entry are hoisted into eight more live scalar
                //  registers -- this kernel's scalar file is full, and a spilled scalar comes back through v_readlane, a VECTOR instruction inside the matrix phase)
                unsigned rs = (unsigned)a.N * 32u;                         // bytes between row blocks t and t + 1: 8 rows x N floats
                asm volatile("" : "+s"(rs));
                sk_glds16_s(w1_voff, (const char*)(a.W1 + r.offW1) + (size_t)t * rs, st + (unsigned)(wave + 8 * t) * 1024u);
            }
            else { if (wave < NI0 && !(r.fl & SKF_NOPROD)) sk_glds16_s(w0_voff, a.W0 + r.offW0, st + (8192u + (unsigned)wave * 256u) * 4u); }
        } else {
            constexpr int NIE = EP::FLOATS / 256;
            const int ii = wave + 8 * t;
            if (ii < NIE) sk_glds16_s(lane16, (const char*)(a.epi + r.offW1) + (size_t)ii * 1024, st + (unsigned)ii * 1024u);
        }
    };
    auto issue = [&](const RecI& r, int q) {
        issue_piece(r, q, std::integral_constant<int, 0>{}); issue_piece(r, q, std::integral_constant<int, 1>{}); issue_piece(r, q, std::integral_constant<int, 2>{});
        issue_piece(r, q, std::integral_constant<int, 3>{}); issue_piece(r, q, std::integral_constant<int, 4>{});
    };
    auto drain_vm = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); };

    f32x4 acc[NH][4][4];
    f32x4 oacc[NH][OT];
    f32x4 hs[2];                                                               // layer-0 activations of the chunk's 32 units, srcB layout (WIDE: shared by the two column blocks' entries)
    float xr[S0];

    // ready flag of a tile's row block: one agent-scope load (every lane the same word), issued a chunk ahead of the wait
    auto probe_flag = [&](const RecL& r) -> int { return __hip_atomic_load(p.xflag + (r.m0 >> 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    unsigned long long st_blocked_w = 0;
    bool my_halt = false;                                                      // this wave saw the halt value on a ready flag (the stop step is behind us)
    int* const lds_halt = (int*)(lds + 4 * STAGE);
    auto wait_x = [&](const RecL& r, int seen) {                               // bounded: report, do not hang
        if (seen >= r.t || p.nowait) { my_halt = my_halt || seen >= (int)SKP_HALT; return; }
        const unsigned long long t0 = wall_clock64();
        int fv;
        while ((fv = __hip_atomic_load(p.xflag + (r.m0 >> 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < r.t) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 200000000ull) { if (lane == 0) __hip_atomic_store(a.err, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // 2 s at 100 MHz
            if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0) break;      // somebody gave up: the launch's results are invalid, leave quickly
        }
        my_halt = my_halt || fv >= (int)SKP_HALT;
        st_blocked_w += wall_clock64() - t0;
    };
    auto load_x = [&](const RecL& r) {                                         // agent-scope loads: the rows were written by another workgroup of this launch
        const int m = min(r.m0 + wave * 16 + i, a.M - 1);
        float* xp = const_cast<float*>(a.A) + (size_t)m * a.lda + g;
#pragma unroll
        for (int s = 0; s < S0; ++s) xr[s] = __hip_atomic_load(xp + 4 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto produce = [&](int q, f32x4 (&dst)[2]) {
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        const float* w0 = ring + (q & 3) * STAGE + 8192 + lane;
#pragma unroll
        for (int s = 0; s < S0; ++s) {
            d0 = MFMA16(w0[(2 * s) * 64], xr[s], d0);
            d1 = MFMA16(w0[(2 * s + 1) * 64], xr[s], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { dst[0][r] = relu1(d0[r]); dst[1][r] = relu1(d1[r]); }
    };

    f32x4 w[8];
    const unsigned lane_off = (unsigned)((4 * g) * 256 + 4 * i) * 4u;
    auto stage_addr = [&](int q) { return ring_lds + (unsigned)((q & 3) * STAGE) * 4u + lane_off; };
    auto first4 = [&](unsigned addr) {
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                     : "v"(addr), "i"(SK_WOFF(0)), "i"(SK_WOFF(1)), "i"(SK_WOFF(2)), "i"(SK_WOFF(3)));
    };

    // ---- prologue: entries 0 .. 2 under way, the first rows loaded ----
    Cur cA, cB;                                                                // cA: the entry the next look-ahead fetch reads (q + 2); cB: the next copy record (q + 4)
    RecL r0, r1; RecI r3;
    {
        Cur k; k.jj = wi % Jx; k.t = wi / Jx; k.c = 0;
        const SkRec* e0 = rec_of(k); const int t0_ = k.t; adv(k);
        const SkRec* e1 = rec_of(k); const int t1_ = k.t; adv(k);
        cA = k;
        const SkRec* e2 = rec_of(k); adv(k);
        if (SK_BAR2) cB = k;                                                       // (SK_BAR2, mlp_streamk.h: copies two entries ahead -- the next copy record is q + 3)
        const SkRec* e3 = rec_of(k); adv(k);
        if (!SK_BAR2) cB = k;
        sk_i32x4 l0, l1, i0, i1, i2, i3;
        asm volatile("s_load_dwordx4 %0, %6, 0x0\n\ts_load_dwordx4 %1, %7, 0x0\n\ts_load_dwordx4 %2, %6, 0x10\n\ts_load_dwordx4 %3, %7, 0x10\n\t"
                     "s_load_dwordx4 %4, %8, 0x10\n\ts_load_dwordx4 %5, %9, 0x10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(l0), "=&s"(l1), "=&s"(i0), "=&s"(i1), "=&s"(i2), "=&s"(i3) : "s"(e0), "s"(e1), "s"(e2), "s"(e3));
        r0 = decL(l0, t0_); r1 = decL(l1, t1_); r3 = SK_BAR2 ? decI(i2) : decI(i3);
        issue(decI(i0), 0); issue(decI(i1), 1); if (!SK_BAR2) issue(decI(i2), 2);
    }
    if (tid == 0) *lds_halt = 0;
    wait_x(r0, -1);
    load_x(r0);
    drain_vm();
    __syncthreads();
    first4(stage_addr(0));
    const unsigned long long st_pro = wall_clock64() - st_t0;

    int seen = 0;                                                              // flag value probed for the next tile
    bool halted = false;                                                       // the stop step is behind this workgroup's next tile: leave
    __amdgpu_buffer_rsrc_t part_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.part, 0, 0xFFFFFFFFu, 0x00020000);
    auto body = [&](auto par_, const int q) {
        constexpr int PAR = decltype(par_)::value;
        constexpr int H = PAR % NH;                          // which of the tile's column blocks this entry belongs to (entries alternate)
        f32x4 (&h)[2] = hs;
        const bool nmain = !(r1.fl & SKF_EPI);
        const float* st = ring + (q & 3) * STAGE;
        if (!(r0.fl & SKF_EPI) && H == 0) produce(q, h);
        // The next tile's rows: loaded HERE if its row block's flag (probed a chunk ago) was already raised -- normally long since.  If not, the wave must not block
        // yet: its workgroup still owes the arrival of the tile in progress, and the awaited step may (in small problems: does) depend on exactly that tile.  It
        // blocks behind the arrival, at the end of this chunk.
        const bool boundary = nmain && (r1.fl & SKF_NEWTILE);
        const bool have_x = !boundary || seen >= r1.t || p.nowait;
        if (boundary && seen >= (int)SKP_HALT) my_halt = true;
        if (boundary && have_x) load_x(r1);
        RecI r4; RecL rn;
        const SkRec* const fa = rec_of(cA); const SkRec* const fb = rec_of(cB); const int fat = cA.t;
        adv(cA); adv(cB);

        if (!(r0.fl & SKF_EPI)) {
            if (r0.fl & SKF_ZERO) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[H][u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                const unsigned cur_a = stage_addr(q), nxt_a = stage_addr(q + 1);
                auto group = [&](auto kc_) {
                    constexpr int k = decltype(kc_)::value;
                    constexpr int j = k >> 4, e = (k >> 2) & 3, u = k & 3;
                    if constexpr (k < 28)
                        asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %0, %2 offset:%3" : "=&v"(w[(k + 4) & 7]), "+v"(w[k & 7]) : "v"(cur_a), "i"(SK_WOFF(k + 4)));
                    else if (nmain && !(SK_BAR2 && PAR == 1))
                        asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %0, %2 offset:%3" : "=&v"(w[(k + 4) & 7]), "+v"(w[k & 7]) : "v"(nxt_a), "i"(SK_WOFF(k - 28)));
                    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[k & 7]) : "i"(31 - k));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[H][u][v] = MFMA16(w[k & 7][v], h[j][e], acc[H][u][v]);
                    __builtin_amdgcn_sched_barrier(0);
                };
#define SKP_G4(b) group(std::integral_constant<int, (b)>{}); group(std::integral_constant<int, (b) + 1>{}); group(std::integral_constant<int, (b) + 2>{}); group(std::integral_constant<int, (b) + 3>{});
#define SKP_P(t) issue_piece(r3, q + (SK_BAR2 ? 2 : 3), std::integral_constant<int, (t)>{})
                SKP_G4(0)  SKP_P(0);
                SKP_G4(4)  SKP_P(1);
                SKP_G4(8)  SKP_P(2);
                SKP_G4(12) SKP_P(3);
                SKP_G4(16) SKP_P(4);
                SKP_G4(20) if (wave < 4) { fetch2(fa, fb, fat, rn, r4); if (rn.fl & SKF_NEWTILE) seen = probe_flag(rn); }
                SKP_G4(24) if (wave >= 4) { fetch2(fa, fb, fat, rn, r4); if (rn.fl & SKF_NEWTILE) seen = probe_flag(rn); }
                SKP_G4(28)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            }
        } else {
            issue(r3, q + (SK_BAR2 ? 2 : 3));
            fetch2(fa, fb, fat, rn, r4);
            if (rn.fl & SKF_NEWTILE) seen = probe_flag(rn);                    // (two epilogue chunks: the next tile comes into view here; sentinels carry no NEWTILE)
            if (!(r0.fl & SKF_NONE)) {
                auto epi_chunk = [&](auto ee) {
                    constexpr int EE = decltype(ee)::value;
                    if (EE == 0) {
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) oacc[H][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                                const float hb = relu1(acc[H][EE * UPC + ul][v][r] + bq[v]);
#pragma unroll
                                for (int ot = 0; ot < OT; ++ot) oacc[H][ot] = MFMA16(wf[ot][v], hb, oacc[H][ot]);
                            }
                        }
                    }
                };
                if (E == 1 || r0.kc == 0) epi_chunk(std::integral_constant<int, 0>{});
                else epi_chunk(std::integral_constant<int, E - 1>{});
                if (r0.fl & SKF_EPILAST) {
                    const int m = r0.m0 + wave * 16 + i;
                    if (m < a.M) {                                            // write-through (sc1): read by a closing workgroup, possibly behind another L2
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const unsigned off = (r0.offC + (unsigned)m * (unsigned)a.ldp + 4u * g) * 4u;
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, oacc[H][ot]), part_rs, off + 64u * ot, 0, 16);
                    }
                }
                if (nmain && !(SK_BAR2 && PAR == 1)) first4(stage_addr(q + 1));
            }
        }
        drain_vm();                                                            // copies of entry q + 3, the rows, the output partials (and the probe)
        if (my_halt && tid == 0) *lds_halt = 1;                                // (wave 0's view decides for the workgroup: the loop must end for all waves at the same entry)
        // SK_BAR2 (mlp_streamk.h): the barrier stands behind the ODD chunks -- and behind every entry that needs the whole workgroup at its end: a tile's arrival
        // (all waves' partials drained), the halt vote at a tile boundary
        const bool bar_here = !SK_BAR2 || PAR == 1 || (r0.fl & SKF_ARRIVE) || (p.stop_batch > 0 && (r0.fl & SKF_NEWTILE));
        if (!(p.nowait & 2) && bar_here) __builtin_amdgcn_s_barrier();         // (nowait & 2: developer timing without the chunk barrier -- results invalid)
        asm volatile("" ::: "memory");
        if (SK_BAR2 && PAR == 1 && nmain) first4(stage_addr(q + 1));
        if (p.stop_batch > 0 && (r0.fl & SKF_NEWTILE) && __builtin_amdgcn_readfirstlane(*(volatile int*)lds_halt)) { halted = true; return; }      // uniform: r0 and the LDS word are the same for every wave
        // every wave's partials are complete behind the barrier: the tile arrives (fire and forget; the last step is closed by the host's k_big_post)
        if ((r0.fl & SKF_ARRIVE) && r0.t + 1 < T && tid == 0) (void)__hip_atomic_fetch_add(p.arrive + (r0.m0 >> 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!have_x) { wait_x(r1, -1); load_x(r1); }                            // (a compute workgroup that has arrived for all it finished owes nobody anything: it may block)
        r0 = r1; r1 = rn; r3 = r4;
    };
    for (int q = 0; q < nq && !halted; q += 2) {
        body(std::integral_constant<int, 0>{}, q);
        if (q + 1 < nq && !halted) body(std::integral_constant<int, 1>{}, q + 1);
    }
    if (p.stats != nullptr && lane == 0) {
        unsigned long long* o = p.stats + 8 * (size_t)blockIdx.x;
        atomicMax(&o[1], st_blocked_w);                                        // the wave that waited longest
        if (tid == 0) { o[0] = wall_clock64() - st_t0; o[5] = (unsigned long long)n_tiles; o[6] = st_pro; }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------------------
// The chunk records of ONE step's tiles, per XCD list.  A tile = NH adjacent column blocks of one (head, row block); a "slice" = one (head, tile column) pair.
// List x holds the (slice, row block) pairs u = slice * RB + rb of [x TPS / 8, (x + 1) TPS / 8) -- 1/8 of the weight slices, so that the workgroups behind one
// L2 stream the same 2-3 slices -- ordered row block first (a row block's tiles finish together, early).  Entries of a tile: the chunks of its column blocks
// alternating (c = 0: block 0, block 1; c = 1: ...), then their epilogue chunks alternating; SKF_NOPROD on the entries that reuse the other block's layer-0
// activations, SKF_ARRIVE on the tile's last entry.
template <int OT>
static inline void skp_build_tab(const SkArgs& a, bool wide, std::vector<SkRec>& tab, int (&Jx)[8], int& Jmax, int& L_out, int& NSL_out) {
    using EP = SkEpi<OT>;
    constexpr int E = EP::E;
    const int NH = wide ? 2 : 1;
    const int RB = (a.M + 127) / 128, CB = a.N / 256, CBW = CB / NH, NCk = a.K1 / 32, NSL = a.heads * CBW, TPS = NSL * RB, L = NH * (NCk + E);
    std::vector<std::vector<std::pair<int, int>>> lists(8);
    for (int x = 0; x < 8; ++x) {
        const int u0 = (int)((long long)TPS * x / 8), u1 = (int)((long long)TPS * (x + 1) / 8);
        for (int u = u0; u < u1; ++u) lists[x].push_back({u % RB, u / RB});       // (rb, slice)
        std::sort(lists[x].begin(), lists[x].end());
    }
    Jmax = 0;
    for (int x = 0; x < 8; ++x) { Jx[x] = (int)lists[x].size(); Jmax = std::max(Jmax, Jx[x]); }
    L_out = L; NSL_out = NSL;
    tab.assign((size_t)8 * Jmax * L + 1, SkRec{});
    for (int x = 0; x < 8; ++x)
        for (int jj = 0; jj < Jx[x]; ++jj) {
            const int rb = lists[x][jj].first, sl = lists[x][jj].second, head = sl / CBW, cbw = sl % CBW;
            for (int e = 0; e < L; ++e) {
                const int hf = e % NH, c = e / NH, cb = cbw * NH + hf;          // column block of this entry | chunk (or NCk + epilogue chunk)
                SkRec r = {};
                int fl = 0, kc = c;
                if (c < NCk) { if (c == 0) fl |= SKF_ZERO; if (e == 0) fl |= SKF_NEWTILE; if (c == NCk - 1) fl |= SKF_LAST; if (hf != 0) fl |= SKF_NOPROD; }
                else { kc = c - NCk; fl = SKF_EPI | (kc == E - 1 ? SKF_EPILAST : 0); }
                if (e == L - 1) fl |= SKF_ARRIVE;
                r.w = r.w2 = kc | (fl << 16); r.m0 = rb * 128;
                r.offA = (unsigned)(head * a.strideA);
                r.offC = (unsigned)(((long long)cb * a.heads + head) * a.stridePart);
                if (fl & SKF_EPI) { r.offW1 = (unsigned)(((head * CB + cb) * E + kc) * EP::FLOATS); r.offW0 = 0; }
                else { r.offW1 = (unsigned)(head * a.strideW1 + (long long)(32 * c) * a.N + cb * 256); r.offW0 = (unsigned)(head * a.strideW0 + 32 * c); }
                r.tile = (head * CB + cb) * RB + rb;
                tab[((size_t)x * Jmax + jj) * L + e] = r;
            }
        }
    SkRec sn = {};
    sn.w = sn.w2 = (SKF_NONE | SKF_EPI) << 16; sn.tile = -1;
    tab[(size_t)8 * Jmax * L] = sn;
}
