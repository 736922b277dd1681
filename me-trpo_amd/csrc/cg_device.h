// Device-side krylov.cg vector algebra ([rllab] rllab.misc.krylov.cg / ConjugateGradientOptimizer.optimize), float64,
// one thread block.  Shared by the stand-alone kernels in api.hip and by k_finalize's fused tail (policy_update.hip):
// when no all-reduce sits between the FVP reduction and the CG step, the LAST block of k_finalize (ticket counter,
// agent-scope release/acquire as in the guide's inter-workgroup recipe) runs the step itself and saves a launch.
#pragma once
#include "metrpo_internal.h"
#ifndef CG_MARK
#define CG_MARK(i)
#endif

enum { S_RDOTR = 0, S_DONE = 1, S_BETA = 2, S_XHX = 3, S_ITERS = 4, S_LOSS0 = 5, S_COMMERR = 6, S_ROLLERR = 7 };   // S_COMMERR: sticky time-out cell of the one-shot exchanges (xchg_device.h)
//   // S_LOSS0: surrogate loss at theta (copy of gout[0]: one read-back fetches scal | lk)

struct CgTail {
    int op;                 // 0 none, 1 = CG iteration, 2 = step-size finish from an explicit H.d, 3 = CG initialisation from the gradient,
                            // 4 = accept test of one line-search trial (ls_decide)
    int P, last;            // last: final CG iteration; with `implicit_hd` it also finishes (step = beta * x) from the CG recurrence
    int implicit_hd;
    double reg, tol, max_kl;
    double *x, *r, *p, *z, *step, *scal;
    const double* gout;     // [1 + P]: loss, gradient (b of the solve)
    float* pf;
    const int* vpos; float* imgval;   // non-NULL: every pf element with an image position (policy_mfma.hip tangent tables) is also stored there
    unsigned int* ticket;   // zero at allocation; the last block of every reduction resets it
    // op 4 (metrpo_trpo_update_begin: speculative line-search trials decided on the device)
    double* ls;             // [4]: index of the trial the search stopped at (-1: none yet) | its loss | its KL | 1 = its theta was taken
    const double* lk;       // [2]: loss, KL of this trial (the reduction's own output)
    float* th; const float* th_try;
    int trial, accept_violation;
    double* pub_dst;        // non-NULL on the LAST speculated trial: its reduction also publishes scal[8] | lk[2] | ls[4] to pinned host memory,
    unsigned long long pub_stamp;   // followed by this stamp at pub_dst[16] (ls_publish) -- what k_ls_publish does as a launch of its own
    // The NEXT line-search trial's theta, built where its inputs appear instead of by a k_try_theta launch of its own (api.hip): behind the step-size
    // finish (ops 1 / 2: trial 0, which also opens the search: nx_ls <- "not stopped yet") and behind a trial's accept test when the search goes on (op 4).
    float* nx_try = nullptr; const float* nx_prev = nullptr; double nx_ratio = 1.0; double* nx_ls = nullptr;
};

// float copy of the next FVP input, element i; mirrored into the weight-fragment image the cached-activation FVP copies (see CgTail::vpos)
struct PfOut {
    float* pf; const int* vpos; float* imgval;
    __device__ __forceinline__ void put(int i, float v) const { pf[i] = v; if (vpos != nullptr) { const int q = vpos[i]; if (q >= 0) imgval[q] = v; } }
};
// r, p <- g; x <- 0; pf <- (float) g   (krylov.cg prologue)
__device__ __forceinline__ void cg_init_body(int P, const double* gout, double* x, double* r, double* p, PfOut pf, double* scal, double* sh);

// wave sum on the DPP path (quad swaps, half-row mirror, row mirror, then the four row totals through readlane, in row order): the
// __shfl_down ladder is 12 dependent ds_bpermute round trips per float64 sum, ~1 us of every fused CG step
template <int CTRL> __device__ __forceinline__ double dpp_add_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v = dpp_add_f64<0xB1>(v); v = dpp_add_f64<0x4E>(v); v = dpp_add_f64<0x141>(v); v = dpp_add_f64<0x140>(v);
    const long long b = __double_as_longlong(v);
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane((int)b, 16 * r), hi = __builtin_amdgcn_readlane((int)(b >> 32), 16 * r);
        t += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return t;
}
__device__ __forceinline__ double blk_sum(double v, double* sh) {
    v = wave_sum_f64(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += sh[i];     // every thread sums the same values in the same order
    return r;
}

// d . (H + reg I) d without another Fisher-vector product.  krylov.cg keeps r = b - A x as a recurrence (r -= v * A p with the
// very z = A p the FVP kernels returned), so A x = g - r holds to float64 rounding and
//     initial_step_size = sqrt(2 max_kl / (x . A x + 1e-8)),  x . A x = x . (g - r)
// is the quantity [rllab] ConjugateGradientOptimizer.optimize obtains from an 11th f_Hx(descent_direction) call.  Both carry the
// float32 rounding of the FVPs (relative 1e-6); tests/test_gpu_engine.py compares the two routes.
__device__ __forceinline__ void cg_finish_implicit(int P, double max_kl, const double* x, const double* r, const double* g,
                                                   double* step, double* scal, double* sh) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) acc += x[i] * (g[i] - r[i]);
    const double xhx = blk_sum(acc, sh);
    double beta = sqrt(2.0 * max_kl * (1.0 / (xhx + 1e-8)));
    if (isnan(beta)) beta = 1.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) step[i] = beta * x[i];
    if (threadIdx.x == 0) { scal[S_BETA] = beta; scal[S_XHX] = xhx; }
}

__device__ __forceinline__ void cg_init_body(int P, const double* gout, double* x, double* r, double* p, PfOut pf, double* scal, double* sh) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const double g = gout[1 + i];
        x[i] = 0.0; r[i] = g; p[i] = g; pf.put(i, (float)g);
        acc += g * g;
    }
    const double rdotr = blk_sum(acc, sh);
    if (threadIdx.x == 0) { scal[S_RDOTR] = rdotr; scal[S_DONE] = 0.0; scal[S_ITERS] = 0.0; scal[S_LOSS0] = gout[0]; }
}

// What a CG iteration reads that the Fisher-vector product it follows does NOT write (p, r, x and the two scalars): k_finalize loads these
// before it waits on the arrival ticket, so the last block's CG step has one global round trip (z) left instead of two dependent ones.
constexpr int CG_R = 2;              // register-resident fast path: P <= 2 * blockDim (1024 threads)
constexpr int CG_RBIG = 13;          // ... and P <= 13 * blockDim without the prefetch
struct CgPre { double pv[CG_R], rv[CG_R], xv[CG_R], done, rdotr; bool have; };
__device__ __forceinline__ void cg_prefetch(const CgTail& t, CgPre& pre) {
    pre.have = (t.op == 1 && t.P <= CG_R * (int)blockDim.x);
    if (!pre.have) return;
    pre.done = t.scal[S_DONE]; pre.rdotr = t.scal[S_RDOTR];
#pragma unroll
    for (int j = 0; j < CG_R; ++j) {
        const int i = threadIdx.x + j * blockDim.x;
        pre.pv[j] = pre.rv[j] = pre.xv[j] = 0.0;
        if (i < t.P) { pre.pv[j] = t.p[i]; pre.rv[j] = t.r[i]; pre.xv[j] = t.x[i]; }
    }
}

// every vector element is read ONCE (one global round trip) and kept in registers across the two reductions: element i = tid + j*blockDim, R per thread.
// Per thread the same elements in the same order as the loop form below (t, t + blockDim, ..): the same sums bit for bit.  R = CG_R takes the vectors the
// caller prefetched (CgPre); R = CG_RBIG serves the 100-50-25 policy's 12 492 parameters (the loop form's dependent global passes were 40 us of every CG step).
template <int R> __device__ __forceinline__ void cg_step_regs(int P, double reg, double tol, int last, double* x, double* r, double* p, double* z,
                                                              PfOut pf, double* scal, double* sh, double rdotr, const CgPre* pre) {
    const bool pf_ = (R == CG_R && pre != nullptr && pre->have);

        double pv[R], zv[R], rv[R], xv[R];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = threadIdx.x + j * blockDim.x;
            pv[j] = zv[j] = rv[j] = xv[j] = 0.0;
            // (every load on a clamped address, unconditionally: inside the `i < P` branch the R x 4 loads of a thread were R dependent round trips --
            //  6 us of the 100-50-25 policy's CG step, tools/fin_phases.py)
            const int ic = (i < P) ? i : P - 1;
            const double pl = pf_ ? 0.0 : p[ic], rl = pf_ ? 0.0 : r[ic], xl = pf_ ? 0.0 : x[ic], zl = z[ic];
            if (i < P) {
                if (pf_) { pv[j] = pre->pv[j < CG_R ? j : 0]; rv[j] = pre->rv[j < CG_R ? j : 0]; xv[j] = pre->xv[j < CG_R ? j : 0]; } else { pv[j] = pl; rv[j] = rl; xv[j] = xl; }
                zv[j] = zl + reg * pv[j]; acc += pv[j] * zv[j];
            }
        }
        CG_MARK(5)
        const double pz = blk_sum(acc, sh);
        CG_MARK(6)
        const double v = rdotr / pz;
        acc = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) { xv[j] += v * pv[j]; rv[j] -= v * zv[j]; acc += rv[j] * rv[j]; }
        const double newrdotr = blk_sum(acc, sh);
        CG_MARK(7)
        const double mu = newrdotr / rdotr;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int i = threadIdx.x + j * blockDim.x;
            if (i < P) {
                const double pn = rv[j] + mu * pv[j];
                z[i] = zv[j]; x[i] = xv[j]; r[i] = rv[j]; p[i] = pn;
                pf.put(i, last ? (float)xv[j] : (float)pn);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            scal[S_RDOTR] = newrdotr;
            scal[S_ITERS] += 1.0;
            if (newrdotr < tol) scal[S_DONE] = 1.0;
        }
        }

// one krylov.cg iteration after z = f_Ax(p) has been formed (z lacks the reg term: added here)
__device__ __forceinline__ void cg_step_body(int P, double reg, double tol, int last, double* x, double* r, double* p, double* z,
                                             PfOut pf, double* scal, double* sh, const CgPre* pre = nullptr) {
    const bool pf_ = (pre != nullptr && pre->have);
    if ((pf_ ? pre->done : scal[S_DONE]) != 0.0) {
        if (last) for (int i = threadIdx.x; i < P; i += blockDim.x) pf.put(i, (float)x[i]);   // next FVP input is x (step scale)
        return;
    }
    const double rdotr = pf_ ? pre->rdotr : scal[S_RDOTR];
    if (P <= CG_R * (int)blockDim.x) { cg_step_regs<CG_R>(P, reg, tol, last, x, r, p, z, pf, scal, sh, rdotr, pre); return; }
    if (P <= 4 * (int)blockDim.x) { cg_step_regs<4>(P, reg, tol, last, x, r, p, z, pf, scal, sh, rdotr, nullptr); return; }      // (Ant's 2 x 32 policy: 2 288)
    if (P <= CG_RBIG * (int)blockDim.x) { cg_step_regs<CG_RBIG>(P, reg, tol, last, x, r, p, z, pf, scal, sh, rdotr, nullptr); return; }
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) { const double zi = z[i] + reg * p[i]; z[i] = zi; acc += p[i] * zi; }
    const double pz = blk_sum(acc, sh);
    const double v = rdotr / pz;
    acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        x[i] += v * p[i];
        const double ri = r[i] - v * z[i];
        r[i] = ri;
        acc += ri * ri;
    }
    const double newrdotr = blk_sum(acc, sh);
    const double mu = newrdotr / rdotr;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const double pn = r[i] + mu * p[i];
        p[i] = pn;
        pf.put(i, last ? (float)x[i] : (float)pn);       // float copy of the next FVP input
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        scal[S_RDOTR] = newrdotr;
        scal[S_ITERS] += 1.0;
        if (newrdotr < tol) scal[S_DONE] = 1.0;
    }
}

// initial_step_size = sqrt(2 * max_kl / (d . Hx(d) + 1e-8)); nan -> 1; step = beta * d
__device__ __forceinline__ void cg_finish_body(int P, double reg, double max_kl, const double* x, double* z, double* step,
                                               double* scal, double* sh) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) acc += x[i] * (z[i] + reg * x[i]);
    const double xhx = blk_sum(acc, sh);
    double beta = sqrt(2.0 * max_kl * (1.0 / (xhx + 1e-8)));
    if (isnan(beta)) beta = 1.0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) step[i] = beta * x[i];
    if (threadIdx.x == 0) { scal[S_BETA] = beta; scal[S_XHX] = xhx; }
}

// One pass of ConjugateGradientOptimizer.optimize's backtracking loop, decided where the numbers are: the loop's break test
// (loss < loss_before and kl <= max_kl) and, when it breaks, the acceptance rule that follows the loop (rejects a NaN, loss >= loss_before or
// kl >= max_kl unless accept_violation) -- an accepted trial's theta replaces the policy at once.  Later speculative trials see ls[0] >= 0
// and leave without doing anything.  One block, all threads; lk was written by this block (a barrier lies in between).
__device__ __forceinline__ void ls_decide(const CgTail& t) {
    const double loss = t.lk[0], kl = t.lk[1], lb = t.scal[S_LOSS0];
    if (loss < lb && kl <= t.max_kl) {
        const bool acc = !(isnan(loss) || isnan(kl) || loss >= lb || kl >= t.max_kl) || t.accept_violation != 0;
        if (acc) for (int i = threadIdx.x; i < t.P; i += blockDim.x) t.th[i] = t.th_try[i];
        if (threadIdx.x == 0) { t.ls[0] = (double)t.trial; t.ls[1] = loss; t.ls[2] = kl; t.ls[3] = acc ? 1.0 : 0.0; }
    } else if (threadIdx.x == 0) { t.ls[1] = loss; t.ls[2] = kl; }
}

// cur_param = prev_param - ratio * flat_descent_step (ConjugateGradientOptimizer.optimize's trial point; k_try_theta's arithmetic).  One block, all threads,
// behind a barrier that follows the writes of t.step.
__device__ __forceinline__ void ls_next_theta(const CgTail& t) {
    for (int i = threadIdx.x; i < t.P; i += blockDim.x) t.nx_try[i] = (float)((double)t.nx_prev[i] - t.nx_ratio * t.step[i]);
}
__device__ __forceinline__ void ls_open_and_first_theta(const CgTail& t) {
    if (t.nx_try == nullptr) return;
    __syncthreads();
    if (t.nx_ls != nullptr && threadIdx.x == 0) { t.nx_ls[0] = -1.0; t.nx_ls[1] = NAN; t.nx_ls[2] = NAN; t.nx_ls[3] = 0.0; }
    ls_next_theta(t);
}

// The update's outcome (scal[8] | lk[2] | ls[4], contiguous at t.scal) into pinned host memory, then the stamp the host is polling for.  One wave:
// program order + vmcnt.  System-scope stores bypass L2; once the wave's own stores are acknowledged the stamp may follow -- a system-scope
// FENCE here would write back the whole L2 (15 us that the next kernel on the stream waits for).  Call behind a __syncthreads().
__device__ __forceinline__ void ls_publish(const double* __restrict__ src, double* dst, unsigned long long stamp) {
    if (threadIdx.x >= 64) return;
    if (threadIdx.x < 14) __hip_atomic_store(dst + threadIdx.x, src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store((unsigned long long*)(dst + 16), stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// CG tail dispatch shared by k_finalize and the in-kernel reduction of policy_mfma.hip (one block, all threads)
__device__ __forceinline__ void cg_tail_run(const CgTail& t, double* sh, const CgPre* pre = nullptr) {
    if (t.op == 1) {
        cg_step_body(t.P, t.reg, t.tol, t.last, t.x, t.r, t.p, t.z, PfOut{t.pf, t.vpos, t.imgval}, t.scal, sh, pre);
        if (t.last && t.implicit_hd) {
            __syncthreads();
            cg_finish_implicit(t.P, t.max_kl, t.x, t.r, t.gout + 1, t.step, t.scal, sh);
            ls_open_and_first_theta(t);
        }
    } else if (t.op == 2) { cg_finish_body(t.P, t.reg, t.max_kl, t.x, t.z, t.step, t.scal, sh); ls_open_and_first_theta(t); }
    else if (t.op == 3) cg_init_body(t.P, t.gout, t.x, t.r, t.p, PfOut{t.pf, t.vpos, t.imgval}, t.scal, sh);
    else if (t.op == 4) {
        ls_decide(t);
        if (t.nx_try != nullptr && !(t.lk[0] < t.scal[S_LOSS0] && t.lk[1] <= t.max_kl)) ls_next_theta(t);      // the search goes on (ls_decide's test, uniform over the block)
        if (t.pub_dst != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); ls_publish(t.scal, t.pub_dst, t.pub_stamp); }
    }
}
