// One-shot direct all-reduce of small float64 vectors over peer-mapped receive regions (comm.hip owns the mapping).
//
// The exchanges of a TRPO iteration are 8 B .. 100 KB sums whose cost is pure latency.  xGMI is a fully connected point-to-point
// fabric, so instead of a ring (2(G-1) dependent hops) every rank WRITES its vector straight into a slot of every peer's
// receive region and then adds the G slots of its own region in rank order:
//   * one fabric traversal per all-reduce, no intermediate hop, no host involvement, no second kernel;
//   * rank-ordered float64 sums -> every rank obtains bit-identical results (identical CG / line-search trajectories);
//   * no separate flag write and no fence: a float64 travels as two 8-byte packets {32 data bits | 32-bit sequence number}
//     (the "LL" idea: an aligned 8-byte store is indivisible on the fabric, a packet whose stamp equals the current sequence
//     number is complete).  Receivers poll the packets themselves.
// Slot reuse: two parities.  A rank starts exchange s+2 (same parity as s) only after it finished s+1, which needed every peer's
// s+1 packets, which a peer sends after it finished reading s -- so nobody overwrites packets that are still being read.
// All accesses are system-scope relaxed atomics (sc0 sc1 on gfx950): they bypass the non-coherent L2 for peer-written lines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XCHG_MAX_WORLD 8

struct XchgK {
    int world;                                  // 0 / 1: no exchange
    int rank;
    unsigned int seq;                           // sequence number of THIS exchange (>= 1; the regions start zeroed)
    int cap;                                    // float64 slots per (parity, source)
    unsigned long long* peer[XCHG_MAX_WORLD];   // receive region of rank q as mapped into this process (peer[rank]: the local one)
    double* err;                                // device cell set to 1.0 when a wait ran into the time limit
    unsigned long long timeout_ticks;           // wall_clock64() ticks (100 MHz)
};

__device__ __forceinline__ size_t xchg_slot(const XchgK& x, int src, int idx) {
    return (((size_t)(x.seq & 1u) * x.world + src) * (size_t)x.cap + idx) * 2;
}

// this rank's element idx -> slot [rank] of every rank's region (its own included)
__device__ __forceinline__ void xchg_push(const XchgK& x, int idx, double val) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(val);
    const unsigned long long p0 = ((unsigned long long)x.seq << 32) | (b & 0xffffffffull);
    const unsigned long long p1 = ((unsigned long long)x.seq << 32) | (b >> 32);
    const size_t o = xchg_slot(x, x.rank, idx);
#pragma unroll
    for (int q = 0; q < XCHG_MAX_WORLD; ++q) {
        if (q < x.world) {
            __hip_atomic_store(x.peer[q] + o, p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(x.peer[q] + o + 1, p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// sum over the ranks, in rank order, of element idx; waits until every rank's packets of this sequence number have landed
__device__ __forceinline__ double xchg_pull_sum(const XchgK& x, int idx) {
    double vals[XCHG_MAX_WORLD];
    unsigned int need = (x.world >= 32) ? 0xffffffffu : ((1u << x.world) - 1u);
    const unsigned long long* base = x.peer[x.rank];
    unsigned long long t0 = 0;
    int spins = 0;
    while (need) {
#pragma unroll
        for (int g = 0; g < XCHG_MAX_WORLD; ++g) {
            if (g < x.world && (need >> g & 1u)) {
                const size_t o = xchg_slot(x, g, idx);
                const unsigned long long a = __hip_atomic_load(base + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long b = __hip_atomic_load(base + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned int)(a >> 32) == x.seq && (unsigned int)(b >> 32) == x.seq) {
                    vals[g] = __longlong_as_double((long long)((b << 32) | (a & 0xffffffffull)));
                    need &= ~(1u << g);
                }
            }
        }
        if (need) {
            if (++spins == 64) t0 = wall_clock64();
            if (spins > 64) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > x.timeout_ticks) {            // a peer never arrived: surface it instead of hanging the GPU
                    *x.err = 1.0;
#pragma unroll
                    for (int g = 0; g < XCHG_MAX_WORLD; ++g) if (g < x.world && (need >> g & 1u)) vals[g] = __longlong_as_double(0x7ff8000000000000ll);
                    need = 0;
                }
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < XCHG_MAX_WORLD; ++g) if (g < x.world) s += vals[g];
    return s;
}
