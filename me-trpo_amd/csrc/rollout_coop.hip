// Cooperative-heads MFMA rollout: dispatch table + launch rule (kernel: rollout_coop_kernel.h)
#define COOP_MAIN_TU 1
#include "rollout_coop_kernel.h"

// -------------------------------------------------------------------------------------------------
// K = 5 is every params file's n_models; 1 ... 4 heads run the same kernel (everything in it is a function of K: LDS map, fragment counts, head loops) instead of
// falling to the head-per-wave kernel (2.9x at B = 5000).  The launch rule's measured co-residency constants are those of K = 5.
// K = 6 ... 10 (round 6): a wave holds K x 23 weight fragments in registers -- with the whole register file of its SIMD (one workgroup per CU, 512 registers) the
// kernel still holds them (a few dozen spilled at K = 10); two workgroups per CU do not exist for these.  Tables in rollout_coop_k<K>.hip.
#define CENTRY_ENV(ENVID) CENTRY(ENVID, 5), CENTRY(ENVID, 4), CENTRY(ENVID, 3), CENTRY(ENVID, 2), CENTRY(ENVID, 1)
static const CoopEntry kCoop[] = {
    CENTRY_ENV(METRPO_ENV_SWIMMER), CENTRY_ENV(METRPO_ENV_HALF_CHEETAH), CENTRY_ENV(METRPO_ENV_HOPPER),
    CENTRY_ENV(METRPO_ENV_SNAKE), CENTRY_ENV(METRPO_ENV_ANT),
};
extern const CoopEntry kCoopK6[5], kCoopK7[5], kCoopK8[5], kCoopK9[5], kCoopK10[5];
constexpr int N_COOP = (int)(sizeof(kCoop) / sizeof(kCoop[0]));
// config index: [0, N_COOP) -> kCoop; N_COOP + 5 (K - 6) + env slot -> the K = 6 ... 10 tables
static const CoopEntry& coop_entry(int idx) {
    if (idx < N_COOP) return kCoop[idx];
    const CoopEntry* wide[5] = {kCoopK6, kCoopK7, kCoopK8, kCoopK9, kCoopK10};
    return wide[(idx - N_COOP) / 5][(idx - N_COOP) % 5];
}

// config index or -1; the shape class (env dims, widths, activations) is the head-per-wave table's (mfma_shape_config: whatever K)
int coop_select_config(metrpo_ctx* c) {
    const ProblemDesc& pd = c->pd;
    if (mfma_shape_config(c) < 0 || pd.dyn.dims[1] != 64 || pd.dyn.dims[2] != 64 || pd.pol.dims[1] != 32 || pd.pol.dims[2] != 32) return -1;
    for (int i = 0; i < N_COOP + 25; ++i)
        if (coop_entry(i).env == pd.env && coop_entry(i).K == pd.K && coop_entry(i).kern[1][0] != nullptr) return i;
    return -1;
}

// Launch rule.  tiles <= n_sm: one workgroup per tile, each alone on its CU (spill-free instantiation with the whole register file).
// More tiles: either ONE workgroup per CU with the tiles x T tile-steps dealt out evenly (tiles migrate between workgroups once, see the
// kernel), or two co-resident workgroups per CU working through the tiles in rounds of 2 n_sm.  Measured (tools/mig_sweep*.py): a tile-step
// costs x1 alone on a CU and `pair` x1 per PAIR-step when two share it (1.50 swimmer, 1.58 snake, 1.65 hopper: the two-per-CU instantiations of
// the latter spill a little), so the first takes tiles / n_sm units of T x1 and the second `pair` per full round plus, for <= n_sm left-over
// tiles, 1 (swimmer) up to `pair` (the others: the left-overs start next to still-running neighbours); the cheaper one is launched.  That is
// the migrating schedule for the shipped B = 5000 (313 tiles on 256 CUs: 1.22 vs 1.50 -- co-residency would leave 57 CUs with two tiles
// setting the wall time while 199 idle half of it), co-residency at 2 or 4 tiles per CU, the migrating schedule again at 2.4.  Envs with more
// than 16 state dims (half-cheetah, Ant) always run one workgroup per CU: their two-per-CU instantiation spills 100+ registers and is
// 1.6-2 x slower per tile-step.
// Zero-padded copy of a narrow ensemble in the 64 x 64 layout (metrpo_internal.h: coop_pad_cfg).  One thread per element of the padded layout.
__global__ void __launch_bounds__(256) k_pad_dyn(NetDesc src, NetDesc dst, int K, const float* __restrict__ s, float* __restrict__ d) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)K * dst.n_params) return;
    const int k = (int)(idx / dst.n_params), o = (int)(idx % dst.n_params);
    const float* sk = s + (size_t)k * src.n_params;
    float v = 0.0f;
    for (int l = 0; l < dst.n_layers; ++l) {
        const int din = dst.dims[l], dout = dst.dims[l + 1];
        if (o >= dst.w_off[l] && o < dst.w_off[l] + din * dout) {
            const int i = (o - dst.w_off[l]) / dout, j = (o - dst.w_off[l]) % dout;
            if (i < src.dims[l] && j < src.dims[l + 1]) v = sk[src.w_off[l] + i * src.dims[l + 1] + j];
        } else if (o >= dst.b_off[l] && o < dst.b_off[l] + dout) {
            const int j = o - dst.b_off[l];
            if (j < src.dims[l + 1]) v = sk[src.b_off[l] + j];
        }
    }
    d[idx] = v;
}

int launch_pad_dyn(metrpo_ctx* c, hipStream_t st) {
    const long long n = (long long)c->pd.K * c->dyn_pad.n_params;
    hipLaunchKernelGGL(k_pad_dyn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c->pd.dyn, c->dyn_pad, c->pd.K, c->d_dyn, c->d_dyn_pad);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

int launch_rollout_coop(metrpo_ctx* c, int idx, const RolloutK& r_in, hipStream_t st, bool padded) {
    const CoopEntry& en = coop_entry(idx);
    const size_t sh = sizeof(float) * (size_t)en.lds_floats;
    const int tiles = (r_in.B + 15) / 16;
    RolloutK r = r_in;
    // CUs that really schedule this process's waves (census, probe.hip): under a CU mask the even tile-step deal is made over THOSE.  The migrating
    // schedule does not need its grid co-resident to be correct (a tile's producer is a lower-numbered workgroup that waits for nobody), only to be fast.
    const int n_cu = (tiles > c->n_sm / 2) ? sched_cus(c, st) : c->n_sm;
    bool one = tiles <= n_cu || en.always_one;
    if (!one) {
        const long long slots = 2LL * n_cu, rounds = tiles / slots, rest = tiles % slots;
        const double t_one = (double)tiles / n_cu;
        const double t_two = en.pair * rounds + (rest == 0 ? 0.0 : rest <= n_cu ? 1.0 + en.rem_slope * (en.pair - 1.0) * rest / n_cu : en.pair);
        one = t_one < t_two;
    }
    if (c->rollout_variant == 2) one = false;
    // a GPU shared with other compute processes (METRPO_NO_RESIDENT=1, the switch that also keeps rollout_resident.hip out): the migrating schedule's
    // consumers wait for producer workgroups of their own grid, which other processes' workgroups can keep off the chip until the bounded wait gives up
    if (one && tiles > n_cu && !ctx_exclusive(c)) one = false;
    if (!one && en.kern[0][0] == nullptr) return METRPO_EUNSUPPORTED;      // K > 5: only the one-workgroup-per-CU instantiation exists (caller: head-per-wave kernel / tile GEMMs)
    const int grid = one ? (tiles < n_cu ? tiles : n_cu) : tiles;
    if (one && tiles > grid) {                                           // hand-over slots: flag[tiles] | ts[16 tiles] | model[16 tiles] | obs[16 tiles][ns]
        const int ns = c->pd.ns;
        if (c->mig_cap < tiles) {
            if (c->d_mig) { ws_retire(c, c->d_mig); c->d_mig = nullptr; c->mig_cap = 0; }
            const size_t bytes = sizeof(int32_t) * (size_t)tiles * (1 + 32 + 16 * ns);
            HIP_TRY(c, ws_alloc(c, (void**)&c->d_mig, bytes));
            HIP_TRY(c, hipMemsetAsync(c->d_mig, 0, bytes, st));
            c->mig_cap = tiles; c->mig_epoch = 0;
        }
        int32_t* base = (int32_t*)c->d_mig;
        r.mig_flag = base; r.mig_ts = base + c->mig_cap; r.mig_model = base + 17 * (size_t)c->mig_cap;
        r.mig_obs = (float*)(base + 33 * (size_t)c->mig_cap);
        r.mig_epoch = ++c->mig_epoch;
        r.mig_err = comm_err_cell(c) + 1;                                // scal[S_ROLLERR]
    }
    const bool draws = r.eps || r.model_idx || r.sel_noise || r.reset_idx || r.reset_model;
    const coop_kernel_t kern = en.kern[one ? 1 : 0][draws ? 1 : 0];
    if (sh > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    const float* dyn = c->d_dyn;
    if (padded) { const int rc = launch_pad_dyn(c, st); if (rc) return rc; dyn = c->d_dyn_pad; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), sh, st, r, dyn, c->d_theta, c->d_norm);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}
