// C-ABI entry points of libmetrpo.so (include/metrpo.h): context management, argument checking,
// dispatch to the kernels, and the host driver of one TRPO update.
#include "metrpo_internal.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include "cg_device.h"
#include "trace.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>

// serialised: the concurrent rounds of rollout_gemm.hip enqueue from several host threads and may fail together
static std::mutex g_err_mutex;
int set_err(metrpo_ctx* c, int code, const std::string& msg) {
    if (c) { std::lock_guard<std::mutex> lock(g_err_mutex); c->err = msg; }
    return code;
}

extern "C" int32_t metrpo_abi_version(void) { return METRPO_ABI_VERSION; }

extern "C" const char* metrpo_status_string(int32_t s) {
    switch (s) {
    case METRPO_OK: return "ok";
    case METRPO_EINVAL: return "invalid argument";
    case METRPO_ENULL: return "null pointer";
    case METRPO_EHIP: return "HIP runtime error";
    case METRPO_EUNSUPPORTED: return "unsupported configuration";
    case METRPO_ESTATE: return "dynamics/policy not set";
    }
    return "unknown status";
}

static bool build_net(NetDesc* n, int n_in, const int32_t* hidden, const int32_t* acts, int n_hidden, int n_out,
                      int default_act, bool align16) {
    if (n_hidden < 0 || n_hidden > METRPO_MAX_LAYERS || n_in <= 0 || n_out <= 0) return false;
    n->n_layers = n_hidden + 1;
    n->dims[0] = n_in;
    for (int l = 0; l < n_hidden; ++l) {
        if (hidden[l] <= 0) return false;
        n->dims[l + 1] = hidden[l];
        n->act[l] = acts ? acts[l] : default_act;
        if (n->act[l] < METRPO_ACT_IDENTITY || n->act[l] > METRPO_ACT_TANH) return false;
    }
    n->dims[n_hidden + 1] = n_out;
    n->act[n_hidden] = METRPO_ACT_IDENTITY;
    int off = 0, aoff = 0, mw = 0;
    auto up = [&](int v) { return align16 ? ((v + 3) & ~3) : v; };
    for (int l = 0; l < n->n_layers; ++l) {
        n->w_off[l] = off; off = up(off + n->dims[l] * n->dims[l + 1]);
        n->b_off[l] = off; off = up(off + n->dims[l + 1]);
        n->api_w_off[l] = aoff; aoff += n->dims[l] * n->dims[l + 1];
        n->api_b_off[l] = aoff; aoff += n->dims[l + 1];
    }
    for (int l = 0; l <= n->n_layers; ++l) mw = std::max(mw, n->dims[l]);
    n->n_params = off;
    n->api_n_params = aoff;
    n->max_width = mw;
    return true;
}

// CG workspace layout (doubles): gout[1+P] | x[P] | r[P] | p[P] | z[P] | step[P] | scal[8] | lk[2] | ls[4]
struct CgView { double *gout, *x, *r, *p, *z, *step, *scal, *lk, *ls; };
static CgView cg_view(metrpo_ctx* c) {
    const int P = c->pd.P;
    CgView v;
    v.gout = c->d_cg; v.x = v.gout + 1 + P; v.r = v.x + P; v.p = v.r + P; v.z = v.p + P; v.step = v.z + P;
    v.scal = v.step + P; v.lk = v.scal + 8;      // lk directly behind scal[8], ls behind lk: run_trpo_update reads scal | lk | ls back as ONE 14-double copy
    v.ls = v.lk + 2;                             // device-side line-search state (cg_device.h: CgTail::ls)
    return v;
}

// Dense caller-visible dynamics layout <-> 16-byte aligned resident layout (NetDesc).  grid = (blocks, n_models).
struct RepackDesc { int n_seg; int api_off[2 * MAXL], dev_off[2 * MAXL], len[2 * MAXL]; int api_stride, dev_stride; };
__global__ void k_repack_dyn(RepackDesc d, const float* __restrict__ src, float* __restrict__ dst, int to_device) {
    const int k = blockIdx.y;
    for (int s = 0; s < d.n_seg; ++s) {
        const float* a = src + (size_t)k * (to_device ? d.api_stride : d.dev_stride) + (to_device ? d.api_off[s] : d.dev_off[s]);
        float* b = dst + (size_t)k * (to_device ? d.dev_stride : d.api_stride) + (to_device ? d.dev_off[s] : d.api_off[s]);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.len[s]; i += gridDim.x * blockDim.x) b[i] = a[i];
    }
}
static void repack_dyn(const metrpo_ctx* c, const float* src, float* dst, int n_models, bool to_device, hipStream_t st) {
    const NetDesc& n = c->pd.dyn;
    RepackDesc d;
    d.n_seg = 2 * n.n_layers; d.api_stride = n.api_n_params; d.dev_stride = n.n_params;
    for (int l = 0; l < n.n_layers; ++l) {
        d.api_off[2 * l] = n.api_w_off[l]; d.dev_off[2 * l] = n.w_off[l]; d.len[2 * l] = n.dims[l] * n.dims[l + 1];
        d.api_off[2 * l + 1] = n.api_b_off[l]; d.dev_off[2 * l + 1] = n.b_off[l]; d.len[2 * l + 1] = n.dims[l + 1];
    }
    hipLaunchKernelGGL(k_repack_dyn, dim3(64, n_models), dim3(256), 0, st, d, src, dst, to_device ? 1 : 0);
}


// Diagnostics hook (not part of include/metrpo.h; bench.py's roofline.update.fvp): with option TIME_FVP set, launch_fvp_tail brackets the Fisher-vector-product
// KERNEL of every CG iteration (not its reduction) with HIP events on the update's stream.  Returns their mean in microseconds, the count in *n.
extern "C" int32_t metrpo_debug_fvp_us(metrpo_ctx* c, double* mean_us, int32_t* n) {
    if (!c || !mean_us || !n) return METRPO_ENULL;
    *mean_us = 0.0; *n = 0;
    const int pairs = c->fvp_ev_n / 2;
    if (pairs == 0) return METRPO_OK;
    HIP_TRY(c, hipEventSynchronize(c->fvp_ev[2 * pairs - 1]));
    double sum = 0.0;
    for (int i = 0; i < pairs; ++i) { float ms = 0.0f; HIP_TRY(c, hipEventElapsedTime(&ms, c->fvp_ev[2 * i], c->fvp_ev[2 * i + 1])); sum += ms; }
    *mean_us = sum / pairs * 1e3; *n = pairs; c->fvp_ev_n = 0;
    return METRPO_OK;
}


// Diagnostics hook (tests/test_gpu_api.py): outgrown workspaces this context holds back instead of freeing them inside a launch entry point (metrpo_internal.h: ws_retire);
// sweep != 0 frees them now (a synchronising call, like the sweep the library runs by itself past WS_RETIRED_MAX).  Returns the count in front of the sweep.
extern "C" int32_t metrpo_debug_ws_retired(metrpo_ctx* c, unsigned long long* bytes, int32_t sweep) {
    if (!c) return METRPO_ENULL;
    const int32_t n = (int32_t)c->ws_retired.size();
    if (bytes) *bytes = (unsigned long long)c->ws_retired_bytes;
    if (sweep) ws_sweep(c);
    return n;
}
// Diagnostics hook (tools/persist_stats.py): per-workgroup statistics of the last persistent stream-K launch made with option PERSIST_STATS set
// (mlp_persist.h: SkpArgs::stats), 8 values per workgroup; returns the number of workgroups (0: none recorded).
extern "C" int32_t metrpo_debug_persist_stats(metrpo_ctx* c, unsigned long long* out, int32_t cap_wgs, void* stream) {
    if (!c || !out) return METRPO_ENULL;
    if (!c->d_skp_stats || c->skp_stats_n == 0) return 0;
    const int n = std::min(cap_wgs, c->skp_stats_n);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_skp_stats, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));
    return n;
}

// ---- option table (metrpo_internal.h: METRPO_OPT_LIST) ----
static const char* const OPT_NAMES[OPT_COUNT] = {
#define X(n) #n,
    METRPO_OPT_LIST(X)
#undef X
};
const char* metrpo_opt_name(int id) { return (id >= 0 && id < OPT_COUNT) ? OPT_NAMES[id] : nullptr; }
int metrpo_opt_id(const char* key) {
    if (!key) return -1;
    std::string k(key);
    for (auto& ch : k) ch = (char)toupper((unsigned char)ch);
    if (k.rfind("METRPO_", 0) == 0) k = k.substr(7);
    for (int i = 0; i < OPT_COUNT; ++i) if (k == OPT_NAMES[i]) return i;
    return -1;
}
// fields of the context derived from an option (id < 0: all of them)
static void opt_apply(metrpo_ctx* c, int id) {
    if (id < 0) c->upd_tiles_per_wave = 1;
    // (NO_RESIDENT does not touch c->exclusive -- the caller's metrpo_set_exclusive value: ctx_exclusive() combines the two at every use.)
    if (id == OPT_XCHG_TIMEOUT_MS && ctx_opt(c, OPT_XCHG_TIMEOUT_MS)) { const long long v = atoll(ctx_opt(c, OPT_XCHG_TIMEOUT_MS)); if (v > 0) c->xg_timeout = (unsigned long long)v * 100000ull; }
}
// value == NULL unsets the key.  Read by the NEXT launch; options that size a workspace or select a kernel table entry at set_dynamics / set_policy time
// (none today) would say so here.
extern "C" int32_t metrpo_set_option(metrpo_ctx* c, const char* key, const char* value) {
    if (!c || !key) return METRPO_ENULL;
    const int id = metrpo_opt_id(key);
    if (id < 0) return set_err(c, METRPO_EINVAL, std::string("set_option: unknown key '") + key + "'");
    c->opt_set[id] = (value != nullptr); c->opt_val[id] = value ? value : "";
    opt_apply(c, id);
    return METRPO_OK;
}
// returns the value's length (copied into buf, NUL-terminated, truncated to cap - 1), METRPO_UNSET when the key is unset, METRPO_EINVAL for an unknown key
extern "C" int32_t metrpo_get_option(metrpo_ctx* c, const char* key, char* buf, int32_t cap) {
    if (!c || !key) return METRPO_ENULL;
    const int id = metrpo_opt_id(key);
    if (id < 0) return set_err(c, METRPO_EINVAL, std::string("get_option: unknown key '") + key + "'");
    if (!c->opt_set[id]) return METRPO_UNSET;
    if (buf && cap > 0) { const size_t n = std::min<size_t>(c->opt_val[id].size(), (size_t)cap - 1); memcpy(buf, c->opt_val[id].data(), n); buf[n] = 0; }
    return (int32_t)c->opt_val[id].size();
}
// key of option i (NULL beyond the table): lets a binding enumerate the switches
extern "C" const char* metrpo_option_name(int32_t i) { return metrpo_opt_name(i); }

extern "C" int32_t metrpo_create(metrpo_ctx** out, int32_t device, const metrpo_dims* d) {
    if (!out || !d) return METRPO_ENULL;
    *out = nullptr;
    if (d->ns <= 0 || d->na <= 0 || d->n_models <= 0 || d->n_drop < 0 || d->n_drop >= d->ns) return METRPO_EINVAL;
    if (d->env < METRPO_ENV_SWIMMER || d->env > METRPO_ENV_SNAKE) return METRPO_EINVAL;
    // the analytic rewards index fixed state columns (see metrpo_env)
    const int min_ns[] = {6, 10, 16, 1, 6, 8};
    if (d->ns < min_ns[d->env]) return METRPO_EINVAL;
    metrpo_ctx* c = new (std::nothrow) metrpo_ctx();
    if (!c) return METRPO_EINVAL;
    c->device = device; c->dims = *d;
    c->d_dyn = c->d_norm = c->d_theta = nullptr; c->have_dyn = c->have_pol = false;
    c->d_dyn_img = c->d_pol_img = nullptr; c->pol_img_idx = -1; c->d_pol_imgval = nullptr; c->d_pol_vpos = nullptr; c->img_live = 0;
    c->d_bptt = nullptr; c->bptt_cap = 0; c->det_cfg = -1; c->d_detpart = nullptr; c->detpart_cap = 0; c->det_gemm = 0; c->d_dg = nullptr; c->dg_cap = 0; c->vjp_gm = nullptr; c->ls_skip = nullptr; c->d_pol_adam = nullptr; c->pol_adam_t = 0; c->mfma_cfg = -1; c->pol_mfma = -1; c->coop_cfg = -1; c->rollout_variant = 0;
    c->d_partials = nullptr; c->partials_cap = 0; c->d_cg = nullptr; c->d_vf = nullptr; c->d_theta_try = nullptr;
    c->d_valbuf = nullptr; c->h_pinned = nullptr; c->n_sm = 256; c->n_cu_sched = 0; c->fallback_logged = 0; c->fvp_ev_n = 0; c->fvp_ev_made = 0; c->d_skp_tab = nullptr; c->d_skp_stats = nullptr; c->skp_stats_n = 0; c->skp_tab_cap = 0; c->persist_failed = 0; for (int i = 0; i < 8; ++i) c->skp_key[i] = -1;
    for (int i = 0; i < OPT_COUNT; ++i) {                     // the ONLY place the library reads the environment for kernel selection: defaults of the option table
        const std::string ev = std::string("METRPO_") + metrpo_opt_name(i);
        const char* e = getenv(ev.c_str());
        c->opt_set[i] = (e != nullptr); c->opt_val[i] = e ? e : "";
    }
    c->exclusive = 1;                                         // until metrpo_set_exclusive(ctx, 0) says otherwise (option NO_RESIDENT is combined with it in ctx_exclusive())
    opt_apply(c, -1);
    c->d_vbuf = nullptr; c->vbuf_cap = 0; c->d_gae_part = nullptr; c->gae_part_cap = 0; c->d_gram_part = nullptr; c->gram_cap = 0; c->d_big = nullptr; c->big_cap = 0; c->d_res = nullptr; c->res_cap = 0; c->res_seq = 0; c->res_failed = 0; c->last_rollout_kernel = -1; c->upd_pending = 0; c->upd_spec = 0; c->upd_changed_in_end = 0; c->h_upd = nullptr; c->upd_stamp = 0; c->side_ready = 0; c->d_ticket = nullptr; c->d_hcache = nullptr; c->hcache_cap = 0; c->hcache_on = 0; c->d_mig = nullptr; c->mig_cap = 0; c->mig_epoch = 0; c->nccl_comm = nullptr; c->comm_world = 0; c->comm_rank = 0; c->pol_path = 1; c->d_pg = nullptr; c->pg_cap = 0; c->pg_fwd_rows = -1; c->pg_fwd_obs = nullptr; c->pol_f3 = 0; c->d_f3 = nullptr; c->f3_cap = 0; c->f3_rows = -1; c->f3_obs = nullptr; c->f3_theta = nullptr; c->f3_img_ok = 0; c->d_adam = nullptr; c->adam_t = 0; c->d_train = nullptr; c->train_cap = 0; c->d_train_part = nullptr; c->train_part_cap = 0;
    ProblemDesc& pd = c->pd;
    pd.env = d->env; pd.ns = d->ns; pd.na = d->na; pd.K = d->n_models; pd.n_drop = d->n_drop;
    pd.nin = d->ns + d->na - d->n_drop;
    if (!build_net(&pd.dyn, pd.nin, d->dyn_hidden, d->dyn_act, d->dyn_n_hidden, d->ns, METRPO_ACT_RELU, true) ||
        !build_net(&pd.pol, d->ns, d->pol_hidden, nullptr, d->pol_n_hidden, d->na, METRPO_ACT_TANH, false)) {
        delete c;
        return METRPO_EINVAL;
    }
    pd.P = pd.pol.n_params + d->na;
    *out = c;
    if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return METRPO_EHIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_sm = prop.multiProcessorCount;
    const size_t ncg = (size_t)(1 + pd.P) + 5 * (size_t)pd.P + 8 + 2 + 4 + 1;      // ... | scal[8] | lk[2] | ls[4] | validation time-out cell (val_err_cell)
    if (hipMalloc(&c->d_dyn, sizeof(float) * (size_t)pd.K * pd.dyn.n_params) != hipSuccess ||
        hipMalloc(&c->d_norm, sizeof(float) * (2 * (pd.ns + pd.na) + 2 * pd.ns)) != hipSuccess ||
        hipMalloc(&c->d_theta, sizeof(float) * pd.P) != hipSuccess ||
        hipMalloc(&c->d_vf, sizeof(float) * pd.P) != hipSuccess ||
        hipMalloc(&c->d_theta_try, sizeof(float) * pd.P) != hipSuccess ||
        hipMalloc(&c->d_cg, sizeof(double) * ncg) != hipSuccess ||
        hipMalloc(&c->d_valbuf, sizeof(double) * pd.K) != hipSuccess ||
        hipMalloc(&c->d_ticket, sizeof(unsigned int)) != hipSuccess ||
        hipHostMalloc(&c->h_pinned, sizeof(double) * 16) != hipSuccess) {
        c->err = "device allocation failed";
        return METRPO_EHIP;
    }
    if (hipMemset(c->d_dyn, 0, sizeof(float) * (size_t)pd.K * pd.dyn.n_params) != hipSuccess) { c->err = "hipMemset failed"; return METRPO_EHIP; }
    if (hipMemset(c->d_cg, 0, sizeof(double) * ncg) != hipSuccess) { c->err = "hipMemset failed"; return METRPO_EHIP; }   // scal[S_COMMERR] starts clear
    if (hipMemset(c->d_ticket, 0, sizeof(unsigned int)) != hipSuccess) { c->err = "hipMemset failed"; return METRPO_EHIP; }   // the reductions' arrival counter resets itself
    c->mfma_cfg = mfma_select_config(c);
    c->pol_mfma = policy_mfma_select(pd);
    c->pol_f3 = policy_f3_select(pd);
    c->coop_cfg = coop_select_config(c);
    c->coop_pad_cfg = -1; c->d_dyn_pad = nullptr;
    if (c->coop_cfg < 0 && pd.dyn.n_layers == 3 && pd.dyn.dims[1] <= 64 && pd.dyn.dims[2] <= 64) {      // narrow nets (either width below 64): zero-padded to the fused kernel's 64 x 64
        const int32_t hid64[2] = {64, 64}, acts[2] = {pd.dyn.act[0], pd.dyn.act[1]};
        if (build_net(&c->dyn_pad, pd.nin, hid64, acts, 2, pd.ns, METRPO_ACT_RELU, true)) {
            const NetDesc real = pd.dyn;
            pd.dyn = c->dyn_pad;
            const int cfg = coop_select_config(c);                         // the table's view of the padded shape (env dims, activations, policy, K, LDS)
            pd.dyn = real;
            if (cfg >= 0) {
                const size_t bytes = sizeof(float) * (size_t)pd.K * c->dyn_pad.n_params;
                if (hipMalloc(&c->d_dyn_pad, bytes) != hipSuccess || hipMemset(c->d_dyn_pad, 0, bytes) != hipSuccess) { c->err = "hipMalloc failed (padded dynamics)"; return METRPO_EHIP; }
                c->coop_pad_cfg = cfg;
            }
        }
    }
    c->det_cfg = det_mfma_select(c);
    c->det_padded = (c->coop_cfg < 0 && c->coop_pad_cfg >= 0) ? 1 : 0;
    c->det_gemm = det_gemm_applicable(c) ? 1 : 0;
    c->rollout_variant = 0;
    (void)sched_cus(c, nullptr);                              // CU census here, not inside the first resident / cooperative launch (probe.hip)
    return METRPO_OK;
}

extern "C" int32_t metrpo_destroy(metrpo_ctx* c) {
    if (!c) return METRPO_ENULL;
    if (c->nccl_comm) (void)metrpo_comm_destroy(c);
    (void)metrpo_comm_ipc_detach(c);
    if (c->xg_region) (void)hipFree(c->xg_region);
    void* bufs[] = {c->d_dyn, c->d_norm, c->d_theta, c->d_vf, c->d_theta_try, c->d_cg, c->d_valbuf, c->d_partials,
                    c->d_dyn_img, c->d_pol_img, c->d_pol_imgval, c->d_pol_vpos, c->d_vbuf, c->d_gae_part, c->d_train_part, c->d_gram_part, c->d_big, c->d_res, c->d_ticket, c->d_hcache, c->d_mig, c->d_pg, c->d_f3, c->d_adam, c->d_train, c->d_bptt, c->d_pol_adam, c->d_detpart, c->d_dg};
    for (void* p : bufs) if (p) (void)hipFree(p);
    if (c->side_ready) {
        for (int i = 0; i < METRPO_MAX_PAR_ROUNDS - 1; ++i) { (void)hipStreamDestroy(c->side_stream[i]); (void)hipEventDestroy(c->ev_join[i]); }
        (void)hipEventDestroy(c->ev_fork);
    }
    for (int i = 0; i < c->fvp_ev_made; ++i) (void)hipEventDestroy(c->fvp_ev[i]);
    ws_sweep(c);
    if (c->d_dyn_pad) (void)hipFree(c->d_dyn_pad);
    if (c->d_skp_tab) (void)hipFree(c->d_skp_tab);
    if (c->d_skp_stats) (void)hipFree(c->d_skp_stats);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_upd) (void)hipHostFree(c->h_upd);
    delete c;
    return METRPO_OK;
}

extern "C" const char* metrpo_last_error(const metrpo_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
extern "C" int32_t metrpo_dyn_param_count(const metrpo_ctx* c) { return c ? c->pd.dyn.api_n_params : METRPO_ENULL; }
extern "C" int32_t metrpo_policy_param_count(const metrpo_ctx* c) { return c ? c->pd.P : METRPO_ENULL; }

extern "C" int32_t metrpo_set_dynamics(metrpo_ctx* c, const float* p, const float* in_mean, const float* in_std,
                                       const float* diff_mean, const float* diff_std, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!p || !in_mean || !in_std || !diff_mean || !diff_std) return set_err(c, METRPO_ENULL, "set_dynamics: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const ProblemDesc& pd = c->pd;
    const int nx = pd.ns + pd.na;
    repack_dyn(c, p, c->d_dyn, pd.K, true, st);
    HIP_TRY(c, hipMemcpyAsync(c->d_norm, in_mean, sizeof(float) * nx, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + nx, in_std, sizeof(float) * nx, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + 2 * nx, diff_mean, sizeof(float) * pd.ns, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + 2 * nx + pd.ns, diff_std, sizeof(float) * pd.ns, hipMemcpyDeviceToDevice, st));
    c->have_dyn = true;
    if (c->mfma_cfg >= 0) return mfma_prepare_dynamics(c, st);
    return METRPO_OK;
}

extern "C" int32_t metrpo_set_policy(metrpo_ctx* c, const float* theta, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!theta) return set_err(c, METRPO_ENULL, "set_policy: NULL pointer");
    if (c->upd_pending) return set_err(c, METRPO_ESTATE, "set_policy: an update begun with metrpo_trpo_update_begin is still open (metrpo_trpo_update_end decides which theta stands)");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, theta, sizeof(float) * c->pd.P, hipMemcpyDeviceToDevice, st));
    c->have_pol = true;
    if (c->mfma_cfg >= 0) return mfma_prepare_policy(c, st);
    return METRPO_OK;
}

extern "C" int32_t metrpo_get_policy(metrpo_ctx* c, float* out, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!out) return set_err(c, METRPO_ENULL, "get_policy: NULL pointer");
    if (!c->have_pol) return set_err(c, METRPO_ESTATE, "policy not set");
    HIP_TRY(c, hipMemcpyAsync(out, c->d_theta, sizeof(float) * c->pd.P, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return METRPO_OK;
}

#define NEED_POL(c) if (!(c)->have_pol) return set_err((c), METRPO_ESTATE, "metrpo_set_policy has not been called")
#define NEED_DYN(c) if (!(c)->have_dyn) return set_err((c), METRPO_ESTATE, "metrpo_set_dynamics has not been called")

extern "C" int32_t metrpo_policy_actions(metrpo_ctx* c, const float* obs, const float* eps, int32_t B, float* actions,
                                         float* mean, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (B < 0) return set_err(c, METRPO_EINVAL, "policy_actions: B < 0");
    if (B == 0) return METRPO_OK;
    if (!obs || !actions || !mean) return set_err(c, METRPO_ENULL, "policy_actions: NULL pointer");
    return launch_policy_actions(c, obs, eps, B, actions, mean, (hipStream_t)stream);
}

static bool sam_ok(int m) { return m >= METRPO_SAM_STEP_RAND && m <= METRPO_SAM_ONE_MODEL; }

extern "C" int32_t metrpo_step(metrpo_ctx* c, const float* s, const float* a, int32_t B, int32_t sam_mode,
                               const int32_t* model_idx, const float* noise, float* s_next, float* reward,
                               uint8_t* done, float* next_all, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_DYN(c);
    if (!sam_ok(sam_mode)) return set_err(c, METRPO_EINVAL, "sam mode is not defined");      // env_helpers.py:634
    if (B == 0) return METRPO_OK;
    if (!s || !a || !s_next || !reward || !done) return set_err(c, METRPO_ENULL, "step: NULL pointer");
    if ((sam_mode == METRPO_SAM_STEP_RAND || sam_mode == METRPO_SAM_EPS_RAND) && !model_idx)
        return set_err(c, METRPO_ENULL, "step: model_idx required for step_rand/eps_rand");
    if (sam_mode == METRPO_SAM_MODEL_MEAN_STD && !noise) return set_err(c, METRPO_ENULL, "step: noise required for model_mean_std");
    if (B < 0) return set_err(c, METRPO_EINVAL, "step: B < 0");
    if (B == 0) return METRPO_OK;
    return launch_step(c, s, a, B, sam_mode, model_idx, noise, s_next, reward, done, next_all, (hipStream_t)stream);
}

// a shape that the fast kernels' tables do not hold: remembered for metrpo_rollout_note and reported ONCE per context on stderr (option QUIET silences it)
static void note_off_table(metrpo_ctx* c, const std::string& why) {
    c->rollout_note = why;
    if (!c->fallback_logged && ctx_opt(c, OPT_QUIET) == nullptr) fprintf(stderr, "metrpo: rollout off the fast path: %s\n", why.c_str());
    c->fallback_logged = 1;
}
extern "C" int32_t metrpo_rollout(metrpo_ctx* c, const metrpo_rollout_args* a, void* stream) {
    TraceRange trace_("metrpo:rollout (obtain_samples: policy + env)");
    if (!c) return METRPO_ENULL;
    if (!a) return set_err(c, METRPO_ENULL, "rollout: args NULL");
    NEED_DYN(c); NEED_POL(c);
    if (!sam_ok(a->sam_mode)) return set_err(c, METRPO_EINVAL, "sam mode is not defined");
    if (a->B < 0 || a->T < 0 || a->H <= 0 || a->n_pool <= 0) return set_err(c, METRPO_EINVAL, "rollout: bad B/T/H/n_pool");
    if (!a->d_pool || !a->d_obs || !a->d_act || !a->d_rew || !a->d_mean || !a->d_done || !a->d_tpath)
        return set_err(c, METRPO_ENULL, "rollout: required pointer is NULL");
    if (a->t0 < 0) return set_err(c, METRPO_EINVAL, "rollout: t0 < 0");
    if (a->stop_batch < 0 || (a->stop_batch > 0 && a->d_stop_cum == nullptr)) return set_err(c, METRPO_EINVAL, "rollout: stop_batch needs d_stop_cum (and must not be negative)");
    if ((a->d_init_obs != nullptr) != (a->d_init_ts != nullptr) || (a->d_init_obs != nullptr) != (a->d_init_model != nullptr))
        return set_err(c, METRPO_EINVAL, "rollout: d_init_obs, d_init_ts and d_init_model must be given together");
    if (a->B == 0 || a->T == 0) return METRPO_OK;
    c->rollout_note.clear();
    if (c->mfma_cfg >= 0 || c->coop_cfg >= 0 || c->coop_pad_cfg >= 0) {
        int coop = 0;
        const int rc = launch_rollout_mfma(c, a, (hipStream_t)stream, &coop);
        if (rc != METRPO_EUNSUPPORTED) {
            c->last_rollout_kernel = coop ? 2 : 1;
            if (!coop && c->rollout_variant == 0)
                note_off_table(c, c->coop_cfg < 0 ? "the cooperative fused kernel (rollout_coop.hip) holds 2 x 64 dynamics + 2 x 32 policy with K = 1 ... 10 heads (Ant: 8, half-cheetah: 9 -- LDS); this shape (K = " +
                                  std::to_string(c->pd.K) + ") runs on the head-per-wave fused kernel (rollout_mfma.hip), ~2.5x the cooperative kernel's time per head"
                                  : "the cooperative fused kernel holds K = " + std::to_string(c->pd.K) + " > 5 heads only with a CU to itself per workgroup; more 16-env tiles than CUs on a device that "
                                    "is not exclusive (metrpo_set_exclusive / NO_RESIDENT): head-per-wave fused kernel (rollout_mfma.hip)");
            return rc;
        }
        if (c->coop_cfg >= 0 && c->rollout_variant == 0)
            note_off_table(c, "the cooperative fused kernel holds K = " + std::to_string(c->pd.K) + " > 8 heads only with a CU to itself per workgroup; more 16-env tiles than CUs on a device that "
                              "is not exclusive: step-wise tile GEMMs (rollout_gemm.hip), ~8x the fused kernel's time");
    }
    if (gemm_path_applicable(c)) {                                                           // large dynamics nets
        const int rc = launch_rollout_resident(c, a, (hipStream_t)stream);                   // ... at small batch: the whole time loop in one launch
        if (rc != METRPO_EUNSUPPORTED) return rc;
        c->last_rollout_kernel = 3;
        if (c->coop_cfg < 0 && c->mfma_cfg < 0 && mfma_shape_config(c) >= 0 && c->rollout_variant == 0)
            note_off_table(c, "K = " + std::to_string(c->pd.K) + " heads of this 2 x 64-class shape are beyond the fused kernels (cooperative: K <= 10, half-cheetah 9, Ant 8 -- "
                              "LDS; head-per-wave: K <= 8): step-wise tile GEMMs (rollout_gemm.hip), ~3x the fused kernels' time per head (profiles/r06_coop_heads.txt)");
        {
            bool m256 = true; std::string w;
            for (int l = 1; l < c->pd.dyn.n_layers; ++l) { m256 = m256 && (c->pd.dyn.dims[l] % 256 == 0); w += (l > 1 ? "x" : "") + std::to_string(c->pd.dyn.dims[l]); }
            if (!m256 && c->rollout_note.empty() && c->rollout_variant == 0)
                note_off_table(c, "dynamics hidden widths " + w + " are neither the fused kernels' 64x64 nor multiples of 256 (stream-K / persistent / resident kernels): step-wise tile "
                                  "GEMMs (rollout_gemm.hip), 3-4 launches per step -- at B = 5000 hidden 96 takes 3.8 ms where 64 takes 0.43 (INTEGRATION.md section 9)");
        }
        return launch_rollout_gemm(c, a, (hipStream_t)stream);
    }
    c->last_rollout_kernel = 0;
    {
        std::string w;
        for (int l = 1; l < c->pd.dyn.n_layers; ++l) w += (l > 1 ? "x" : "") + std::to_string(c->pd.dyn.dims[l]);
        note_off_table(c, "dynamics hidden widths " + w + " have no matrix-core rollout kernel (fused kernels: two hidden layers of 64; GEMM / stream-K / resident paths: every "
                          "hidden layer >= 16, ns <= 64): thread-per-env kernel (rollout_generic.hip), ~80x the fused kernels' time per env step");
    }
    return launch_rollout_generic(c, a, (hipStream_t)stream);
}
// why the last metrpo_rollout of this context ran outside the fast dispatch table ("" when it did not)
extern "C" const char* metrpo_rollout_note(const metrpo_ctx* c) { return c ? c->rollout_note.c_str() : ""; }
// which kernel family the last metrpo_rollout of this context ran on (-1: none yet): 0 generic, 1 head-per-wave MFMA, 2 cooperative MFMA,
// 3 step-wise GEMM, 4 resident (rollout_resident.hip), 5 step-wise with the stream-K fused ensemble kernel (mlp_streamk.h)
extern "C" int32_t metrpo_last_rollout_kernel(const metrpo_ctx* c) { return c ? c->last_rollout_kernel : METRPO_ENULL; }

extern "C" int32_t metrpo_sampler_progress(metrpo_ctx* c, const uint8_t* done, const int32_t* tpath, int32_t T, int32_t B, int32_t t0,
                                           int64_t batch_size, double* counts, double* state, int32_t* stop, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!done || !tpath || !counts || !state || !stop) return set_err(c, METRPO_ENULL, "sampler_progress: NULL pointer");
    if (T < 0 || B < 0 || t0 < 0 || batch_size < 0) return set_err(c, METRPO_EINVAL, "sampler_progress: bad T/B/t0/batch_size");
    if (T == 0 || B == 0) return METRPO_OK;
    return launch_sampler_progress(c, done, tpath, T, B, t0, batch_size, counts, state, stop, (hipStream_t)stream);
}

// test/diagnostic hook: force the generic kernel regardless of the MFMA table
extern "C" int32_t metrpo_rollout_generic(metrpo_ctx* c, const metrpo_rollout_args* a, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!a) return set_err(c, METRPO_ENULL, "rollout: args NULL");
    NEED_DYN(c); NEED_POL(c);
    if (!sam_ok(a->sam_mode)) return set_err(c, METRPO_EINVAL, "sam mode is not defined");
    if (a->B <= 0 || a->T <= 0 || a->H <= 0 || a->n_pool <= 0) return set_err(c, METRPO_EINVAL, "rollout: bad B/T/H/n_pool");
    if (!a->d_pool || !a->d_obs || !a->d_act || !a->d_rew || !a->d_mean || !a->d_done || !a->d_tpath)
        return set_err(c, METRPO_ENULL, "rollout: required pointer is NULL");
    return launch_rollout_generic(c, a, (hipStream_t)stream);
}
extern "C" int32_t metrpo_has_mfma_path(const metrpo_ctx* c) { return (c && c->mfma_cfg >= 0) ? 1 : 0; }
// test hook: 0 = fastest rollout kernel available, 1 = head-per-wave MFMA kernel even where the cooperative one exists,
// 2 = cooperative kernel in its two-workgroups-per-CU instantiation whatever the batch (rollout_coop.hip picks by tile count otherwise)
extern "C" int32_t metrpo_set_rollout_variant(metrpo_ctx* c, int32_t v) {
    if (!c) return METRPO_ENULL;
    c->rollout_variant = v;
    return (v != 1 && (c->coop_cfg >= 0 || (v == 0 && c->coop_pad_cfg >= 0))) ? 2 : (c->mfma_cfg >= 0 ? 1 : (gemm_path_applicable(c) ? 3 : 0));
}
// which policy-update kernels a batch of N samples would run on: 1 fused MFMA (policy_mfma.hip), 2 GEMM path (policy_gemm.hip), 0 generic
extern "C" int32_t metrpo_update_path(const metrpo_ctx* c, int64_t N) {
    if (!c) return METRPO_ENULL;
    return policy_gemm_applicable(c, N) ? 2 : ((c->pol_mfma >= 0 || f3_active(c)) ? 1 : 0);
}
// test hook: 0 forces the generic (VALU) update kernels, 1 restores the MFMA ones when available
extern "C" int32_t metrpo_set_update_path(metrpo_ctx* c, int32_t use_mfma) {
    if (!c) return METRPO_ENULL;
    // 0: generic (VALU) kernels, 1: fastest path of the shape (fused MFMA kernels, else the GEMM path for large N), 2: GEMM path forced
    c->pol_path = (use_mfma == 2) ? 2 : (use_mfma ? 1 : 0);
    c->pol_mfma = (use_mfma == 1) ? policy_mfma_select(c->pd) : -1;
    c->pol_f3 = (use_mfma == 1) ? policy_f3_select(c->pd) : 0;
    c->pg_fwd_rows = -1; c->f3_rows = -1;
    return (c->pol_mfma >= 0 || f3_active(c)) ? 1 : (c->pol_path == 2 ? 2 : 0);
}

extern "C" int32_t metrpo_validation_cost(metrpo_ctx* c, const float* s0, int32_t Bv, int32_t T, double gamma,
                                          double* costs, void* stream) {
    TraceRange trace_("metrpo:validation_cost");
    if (!c) return METRPO_ENULL;
    NEED_DYN(c); NEED_POL(c);
    if (!s0 || !costs) return set_err(c, METRPO_ENULL, "validation_cost: NULL pointer");
    if (Bv <= 0 || T < 0) return set_err(c, METRPO_EINVAL, "validation_cost: bad Bv/T");
    return launch_validation_cost(c, s0, Bv, T, gamma, costs, (hipStream_t)stream);
}

extern "C" int32_t metrpo_gae(metrpo_ctx* c, const float* obs, const float* rew, const uint8_t* done,
                              const int32_t* tpath, int32_t T, int32_t B, const double* coeffs, double gamma, double lam,
                              float* adv, float* ret, uint8_t* valid, double* stats, void* stream) {
    TraceRange trace_("metrpo:process_samples:gae");
    if (!c) return METRPO_ENULL;
    if (!obs || !rew || !done || !tpath || !adv || !ret || !valid || !stats) return set_err(c, METRPO_ENULL, "gae: NULL pointer");
    if (T < 0 || B < 0) return set_err(c, METRPO_EINVAL, "gae: bad T/B");
    if (T == 0 || B == 0) return METRPO_OK;
    return launch_gae(c, obs, rew, done, tpath, T, B, coeffs, gamma, lam, adv, ret, valid, stats, (hipStream_t)stream);
}

extern "C" int32_t metrpo_process_begin(metrpo_ctx* c, float* old_log_std, double* acc, int64_t n_acc, void* stream) {
    TraceRange trace_("metrpo:process_samples:begin");
    if (!c) return METRPO_ENULL;
    if (n_acc < 0 || n_acc > (1LL << 30) || (n_acc > 0 && !acc)) return set_err(c, METRPO_EINVAL, "process_begin: bad accumulator range");
    if (!old_log_std && n_acc == 0) return METRPO_OK;
    return launch_process_begin(c, old_log_std, n_acc > 0 ? acc : nullptr, n_acc, (hipStream_t)stream);
}

extern "C" int32_t metrpo_center_advantages(metrpo_ctx* c, float* adv, const uint8_t* valid, int64_t N,
                                            const double* stats, void* stream) {
    TraceRange trace_("metrpo:process_samples:center");
    if (!c) return METRPO_ENULL;
    if (!adv || !stats) return set_err(c, METRPO_ENULL, "center: NULL pointer");
    if (N <= 0) return METRPO_OK;
    return launch_center(c, adv, valid, N, stats, (hipStream_t)stream);
}

extern "C" int32_t metrpo_baseline_gram(metrpo_ctx* c, const float* obs, const float* ret, const int32_t* tpath,
                                        const uint8_t* valid, int64_t N, double* AtA, double* Aty, void* stream) {
    TraceRange trace_("metrpo:process_samples:baseline_gram");
    if (!c) return METRPO_ENULL;
    if (!obs || !ret || !tpath || !AtA || !Aty) return set_err(c, METRPO_ENULL, "gram: NULL pointer");
    if (N <= 0) return METRPO_OK;
    return launch_gram(c, obs, ret, tpath, valid, N, AtA, Aty, (hipStream_t)stream);
}

extern "C" int32_t metrpo_baseline_solve(metrpo_ctx* c, const double* AtA, const double* Aty, double reg_coeff, double* coeffs, void* stream) {
    TraceRange trace_("metrpo:process_samples:baseline_solve");
    if (!c) return METRPO_ENULL;
    if (!AtA || !Aty || !coeffs) return set_err(c, METRPO_ENULL, "baseline_solve: NULL pointer");
    if (!(reg_coeff >= 0.0)) return set_err(c, METRPO_EINVAL, "baseline_solve: reg_coeff must be >= 0");
    return launch_baseline_solve(c, AtA, Aty, reg_coeff, coeffs, (hipStream_t)stream);
}

extern "C" int32_t metrpo_loss_grad(metrpo_ctx* c, const metrpo_batch* b, double* out, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (!out) return set_err(c, METRPO_ENULL, "loss_grad: out NULL");
    return launch_loss_grad(c, b, out, (hipStream_t)stream);
}
extern "C" int32_t metrpo_fvp(metrpo_ctx* c, const metrpo_batch* b, const double* v, double* hv, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    return launch_fvp(c, b, v, hv, (hipStream_t)stream);
}
extern "C" int32_t metrpo_loss_kl(metrpo_ctx* c, const metrpo_batch* b, const float* theta, double* out, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (!out) return set_err(c, METRPO_ENULL, "loss_kl: out NULL");
    return launch_loss_kl(c, b, theta, out, (hipStream_t)stream);
}

// ---- ensemble dynamics training ----------------------------------------------------------
extern "C" int32_t metrpo_dyn_train_reset(metrpo_ctx* c, void* stream) {
    if (!c) return METRPO_ENULL;
    c->adam_t = 0;
    if (c->d_adam) {
        const size_t nP = (((size_t)c->pd.K * c->pd.dyn.n_params) + 3) & ~(size_t)3;
        HIP_TRY(c, hipMemsetAsync(c->d_adam, 0, 2 * nP * sizeof(float), (hipStream_t)stream));
    }
    return METRPO_OK;
}

extern "C" int32_t metrpo_dyn_train_step(metrpo_ctx* c, const float* x, const float* y, const metrpo_train_params* tp, double* loss_out,
                                         void* stream) {
    TraceRange trace_("metrpo:dyn_train_step");
    if (!c) return METRPO_ENULL;
    NEED_DYN(c);
    if (!x || !y || !tp) return set_err(c, METRPO_ENULL, "dyn_train_step: NULL pointer");
    if (tp->batch_size <= 0 || !(tp->lr >= 0.0)) return set_err(c, METRPO_EINVAL, "dyn_train_step: bad batch_size / lr");
    return launch_dyn_train_step(c, x, y, tp, loss_out, (hipStream_t)stream);
}

extern "C" int32_t metrpo_dyn_eval_losses(metrpo_ctx* c, const float* x, const float* y, int64_t n, double reg_constant, double* losses,
                                          void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_DYN(c);
    if (!x || !y || !losses) return set_err(c, METRPO_ENULL, "dyn_eval_losses: NULL pointer");
    if (n <= 0) return set_err(c, METRPO_EINVAL, "dyn_eval_losses: n must be positive");
    return launch_dyn_eval_losses(c, x, y, n, reg_constant, losses, (hipStream_t)stream);
}

extern "C" int32_t metrpo_get_dynamics(metrpo_ctx* c, float* out, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_DYN(c);
    if (!out) return set_err(c, METRPO_ENULL, "get_dynamics: NULL pointer");
    repack_dyn(c, c->d_dyn, out, c->pd.K, false, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

extern "C" int32_t metrpo_set_dynamics_model(metrpo_ctx* c, int32_t model, const float* p, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_DYN(c);
    if (!p) return set_err(c, METRPO_ENULL, "set_dynamics_model: NULL pointer");
    if (model < 0 || model >= c->pd.K) return set_err(c, METRPO_EINVAL, "set_dynamics_model: model index out of range");
    repack_dyn(c, p, c->d_dyn + (size_t)model * c->pd.dyn.n_params, 1, true, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return METRPO_OK;
}

extern "C" int32_t metrpo_set_normalizers(metrpo_ctx* c, const float* in_mean, const float* in_std, const float* diff_mean,
                                          const float* diff_std, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!in_mean || !in_std || !diff_mean || !diff_std) return set_err(c, METRPO_ENULL, "set_normalizers: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nx = c->pd.ns + c->pd.na, ns = c->pd.ns;
    HIP_TRY(c, hipMemcpyAsync(c->d_norm, in_mean, sizeof(float) * nx, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + nx, in_std, sizeof(float) * nx, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + 2 * nx, diff_mean, sizeof(float) * ns, hipMemcpyDeviceToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_norm + 2 * nx + ns, diff_std, sizeof(float) * ns, hipMemcpyDeviceToDevice, st));
    return METRPO_OK;
}

extern "C" int32_t metrpo_rms_accumulate(metrpo_ctx* c, const float* x, int64_t n, int32_t dim, double* rsum, double* rsumsq, void* stream) {
    if (!c) return METRPO_ENULL;
    if (!x || !rsum || !rsumsq) return set_err(c, METRPO_ENULL, "rms_accumulate: NULL pointer");
    if (n < 0 || dim <= 0) return set_err(c, METRPO_EINVAL, "rms_accumulate: bad n / dim");
    if (n == 0) return METRPO_OK;
    return launch_rms_accumulate(c, x, n, dim, rsum, rsumsq, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// TRPO update driver: [rllab] ConjugateGradientOptimizer.optimize + krylov.cg, vectors resident on
// the device in float64 (the reference keeps them in host NumPy float64), one block per vector op.
// ------------------------------------------------------------------------------------------------
__global__ void k_cg_init(int P, const double* __restrict__ gout, double* x, double* r, double* p, float* pf, double* scal) {
    __shared__ double sh[16];
    cg_init_body(P, gout, x, r, p, PfOut{pf, nullptr, nullptr}, scal, sh);
}

__global__ void k_cg_step(CgTail t, int) {
    __shared__ double sh[16];
    cg_tail_run(t, sh);
}

__global__ void k_cg_finish_implicit(int P, double max_kl, const double* x, const double* r, const double* gout, double* step, double* scal) {
    __shared__ double sh[16];
    cg_finish_implicit(P, max_kl, x, r, gout + 1, step, scal, sh);
}

__global__ void k_cg_finish(int P, double reg, double max_kl, const double* x, double* z, double* step, double* scal) {
    __shared__ double sh[16];
    cg_finish_body(P, reg, max_kl, x, z, step, scal, sh);
}

// the update's outcome into pinned host memory as a launch of its own (no speculated trial whose reduction could carry it: cg_device.h ls_publish)
__global__ void k_ls_publish(const double* __restrict__ src, double* dst, unsigned long long stamp) { ls_publish(src, dst, stamp); }
__global__ void k_ls_reset(double* ls) { if (threadIdx.x == 0) { ls[0] = -1.0; ls[1] = NAN; ls[2] = NAN; ls[3] = 0.0; } }
__global__ void k_zero_f(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

__global__ void k_try_theta(int P, double ratio, const float* __restrict__ prev, const double* __restrict__ step, float* cur, const double* __restrict__ ls,
                            double* ls_reset = nullptr) {
    // trial 0 of a device-decided search also opens it (ls <- "not stopped yet": what k_ls_reset does as a launch of its own); it is never skipped itself
    if (ls_reset != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { ls_reset[0] = -1.0; ls_reset[1] = NAN; ls_reset[2] = NAN; ls_reset[3] = 0.0; }
    if (ls != nullptr && ls[0] >= 0.0) return;                      // speculative trial after the search stopped: cur must keep the accepted theta's source
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) cur[i] = (float)((double)prev[i] - ratio * step[i]);     // cur_param = prev_param - ratio * flat_descent_step
}

// phase: 0 = the whole update with the line search decided on the host (one synchronisation per trial, as the reference does);
//        1 = metrpo_trpo_update_begin: solve + the first `spec` line-search trials enqueued with the accept test on the DEVICE (ls_decide), no
//            synchronisation; 2 = metrpo_trpo_update_end: fetch the outcome, continue on the host from trial `spec` if the search has not stopped
static int run_trpo_update_impl(metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr, metrpo_trpo_diag* diag,
                                double* g_out, double* dir_out, hipStream_t st, int phase, int spec);
// The arrival counter of the fused tails (d_ticket) resets itself in the last block of every reduction, so a completed update leaves it
// at zero.  An update that FAILED half-way (a launch error, a time-out) may not: it is cleared on the error path, where the cost of a
// 4-byte memset does not matter -- a stale count would silently disable every later CG tail.
int run_trpo_update(metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr, metrpo_trpo_diag* diag,
                    double* g_out, double* dir_out, hipStream_t st, int phase, int spec) {
    const int rc = run_trpo_update_impl(c, b, pr, diag, g_out, dir_out, st, phase, spec);
    if (rc != METRPO_OK) {
        (void)hipGetLastError(); (void)hipMemsetAsync(c->d_ticket, 0, sizeof(unsigned int), st); c->upd_pending = 0;
    }
    return rc;
}
// Can this update's line search be decided on the device?  Needs the single-launch-sequence update (no host callback / stand-alone
// all-reduce between a reduction and its consumer) on the fused MFMA or generic kernels; the GEMM path has its own reductions.
static bool device_line_search_ok(const metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr) {
    const bool xg = (pr->allreduce == nullptr && c->xg_world > 1);
    const bool fused = (pr->allreduce == nullptr && ((c->nccl_comm == nullptr && !xg) || (xg && !policy_gemm_applicable(c, b->N) && c->pd.P + 1 <= c->xg_cap)));
    return fused && !policy_gemm_applicable(c, b->N);
}
static int run_trpo_update_impl(metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr, metrpo_trpo_diag* diag,
                                double* g_out, double* dir_out, hipStream_t st, int phase, int spec) {
    const int P = c->pd.P;
    CgView v = cg_view(c);
    int rc;
    // sum over ranks: the caller's callback if given, else the RCCL communicator attached to the ctx (comm.hip), else single rank
#define AR(buf, n) do { if (pr->allreduce) { if ((rc = pr->allreduce(pr->allreduce_user, (buf), (n), (void*)st)) != 0) \
                                                 return set_err(c, METRPO_EINVAL, "allreduce callback failed"); } \
                        else if (xg_fused) { /* summed in the tail of the reduction that produced buf */ } \
                        else if (c->nccl_comm || xg) { if ((rc = comm_allreduce_f64(c, (buf), (n), st)) != 0) return rc; } } while (0)
    // krylov.cg with every vector step fused into the tail of the kernel that produced its input (no all-reduce in between) or
    // as stand-alone one-block kernels after the caller's all-reduce.  The step scale needs d.(H d): by default it is taken
    // from the CG recurrence (cg_device.h: A x = g - r), explicit_final_hvp = 1 spends the extra FVP rllab spends.
    // Sharded over a peer-mapped transport (comm.hip, one-shot exchange): the reductions of the update kernels add the ranks' shares in
    // their own tail (xg_fuse), so the update keeps its single-rank launch sequence -- CG vector steps included.  The GEMM path
    // (policy_gemm.hip) has its own reduction kernels: there the exchange is the stand-alone one-shot kernel between them and the CG step.
    const bool xg = (pr->allreduce == nullptr && c->xg_world > 1);
    // in-tail exchange: per-element packets into ONE slot per source (k_finalize does not split); longer vectors take the stand-alone, chunked exchange
    const bool xg_fused = xg && !policy_gemm_applicable(c, b->N) && P + 1 <= c->xg_cap;
    const bool fused = (pr->allreduce == nullptr && ((c->nccl_comm == nullptr && !xg) || xg_fused));
    c->xg_fuse = xg_fused ? 1 : 0;
    struct FuseOff { metrpo_ctx* c; ~FuseOff() { c->xg_fuse = 0; } } fuse_off{c};
    const int implicit_hd = pr->explicit_final_hvp ? 0 : 1;
    CgTail tl; tl.pub_dst = nullptr; tl.pub_stamp = 0; tl.P = P; tl.last = 0; tl.implicit_hd = implicit_hd; tl.reg = pr->reg_coeff; tl.tol = pr->residual_tol; tl.max_kl = pr->max_kl;
    tl.x = v.x; tl.r = v.r; tl.p = v.p; tl.z = v.z; tl.step = v.step; tl.scal = v.scal; tl.gout = v.gout; tl.pf = c->d_vf; tl.ticket = c->d_ticket;
    tl.vpos = nullptr; tl.imgval = nullptr; tl.ls = nullptr; tl.lk = nullptr; tl.th = nullptr; tl.th_try = nullptr; tl.trial = 0; tl.accept_violation = 0;
    const int nspec = std::min(spec, (int)pr->max_backtracks);
    // Device-decided line search (phase 1): theta of trial 0 is built by the tail that finishes the step size, theta of trial n + 1 by trial n's accept test
    // when the search goes on (cg_device.h: CgTail::nx_*) -- no k_try_theta launches in front of the evaluations.  Where the step does not come out of a
    // cg_tail_run (one-launch solve, cg_iters = 0, explicit H.d without a fused tail) the launches stay.
    bool try0_built = false;
    if (phase != 2) {
    tl.op = 3;
    c->hcache_on = 1;                    // the gradient kernel publishes tanh activations, the CG products of this solve reuse them
    // ... and, when every CG vector comes out of a fused tail, its weight-fragment image; the tails add the tangent entries (policy_mfma.hip)
    c->img_live = (fused && c->pol_mfma >= 0 && !policy_gemm_applicable(c, b->N)) ? 1 : 0;
    struct CacheOff { metrpo_ctx* c; ~CacheOff() { c->hcache_on = 0; c->img_live = 0; } } cache_off{c};
    if (c->img_live) {
        if ((rc = policy_mfma_image_buffers(c))) return rc;
        tl.vpos = c->d_pol_vpos; tl.imgval = c->d_pol_imgval;
    }
    if ((rc = launch_loss_grad(c, b, v.gout, st, fused ? &tl : nullptr))) return rc;
    if (!fused) {
        AR(v.gout, 1 + P);
        hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(1024), 0, st, P, v.gout, v.x, v.r, v.p, c->d_vf, v.scal);   // same block shape as the fused tail: identical summation order
    }
    if (pr->cg_iters == 0) {
        hipLaunchKernelGGL(k_zero_f, dim3((P + 255) / 256), dim3(256), 0, st, c->d_vf, P);
        hipLaunchKernelGGL(k_cg_finish_implicit, dim3(1), dim3(1024), 0, st, P, pr->max_kl, v.x, v.r, v.gout, v.step, v.scal);
    }
    const bool fold_try = (phase == 1 && nspec >= 1);
    auto arm_try0 = [&]() { tl.nx_try = c->d_theta_try; tl.nx_prev = c->d_theta; tl.nx_ratio = 1.0; tl.nx_ls = v.ls; try0_built = true; };      // backtrack_ratio ^ 0
    for (int i = 0; i < pr->cg_iters; ++i) {
        tl.op = 1; tl.last = (i == pr->cg_iters - 1) ? 1 : 0;
        if (fold_try && tl.last && implicit_hd) arm_try0();
        if (fused) { if ((rc = launch_fvp_tail(c, b, c->d_vf, v.p, v.z, &tl, st))) return rc; continue; }
        if ((rc = launch_fvp_f32(c, b, c->d_vf, v.p, v.z, st))) return rc;
        AR(v.z, P);
        hipLaunchKernelGGL(k_cg_step, dim3(1), dim3(1024), 0, st, tl, pr->cg_iters);
    }
    if (!implicit_hd && pr->cg_iters > 0) {                      // rllab's literal route: one more f_Hx on the descent direction
        tl.op = 2; tl.last = 0;
        if (fused && fold_try) arm_try0();
        if (fused) { if ((rc = launch_fvp_tail(c, b, c->d_vf, v.x, v.z, &tl, st))) return rc; }
        else {
            if ((rc = launch_fvp_f32(c, b, c->d_vf, v.x, v.z, st))) return rc;
            AR(v.z, P);
            hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(1024), 0, st, P, pr->reg_coeff, pr->max_kl, v.x, v.z, v.step, v.scal);
        }
    }
    tl.vpos = nullptr; tl.imgval = nullptr; tl.nx_try = nullptr; tl.nx_prev = nullptr; tl.nx_ls = nullptr;
    if (g_out) HIP_TRY(c, hipMemcpyAsync(g_out, v.gout + 1, sizeof(double) * P, hipMemcpyDeviceToDevice, st));
    if (dir_out) HIP_TRY(c, hipMemcpyAsync(dir_out, v.x, sizeof(double) * P, hipMemcpyDeviceToDevice, st));
    }
    if (phase == 1) {
        // ---- the first `nspec` trials of the backtracking line search, each one's accept test in the tail of its own reduction (ls_decide): an
        //      accepted trial's theta is in place when the caller's next launch reads it; trials after the one that stopped the search leave at once
        if (!c->h_upd) { HIP_TRY(c, hipHostMalloc((void**)&c->h_upd, sizeof(double) * 17)); memset(c->h_upd, 0, sizeof(double) * 17); }
        c->upd_stamp += 1;                                          // _end waits for THIS stamp, not for what the caller enqueues after _begin
        if (nspec == 0) hipLaunchKernelGGL(k_ls_reset, dim3(1), dim3(64), 0, st, v.ls);
        for (int n = 0; n < nspec; ++n) {
            const double ratio = std::pow(pr->backtrack_ratio, (double)n);
            // trial 0 opens the search (ls reset rides in its k_try_theta), the last trial's reduction publishes the outcome: no launches of their own
            if (n == 0 && !try0_built)
                hipLaunchKernelGGL(k_try_theta, dim3((P + 255) / 256), dim3(256), 0, st, P, ratio, c->d_theta, v.step, c->d_theta_try, (const double*)nullptr, v.ls);
            CgTail dt = tl;
            if (n + 1 < nspec) { dt.nx_try = c->d_theta_try; dt.nx_prev = c->d_theta; dt.nx_ratio = std::pow(pr->backtrack_ratio, (double)(n + 1)); dt.nx_ls = nullptr; }
            dt.op = 4; dt.ls = v.ls; dt.lk = v.lk; dt.th = c->d_theta; dt.th_try = c->d_theta_try; dt.trial = n; dt.accept_violation = pr->accept_violation;
            if (n == nspec - 1) { dt.pub_dst = c->h_upd; dt.pub_stamp = c->upd_stamp; }
            if ((rc = launch_loss_kl(c, b, c->d_theta_try, v.lk, st, &dt))) return rc;
        }
        if (c->mfma_cfg >= 0 && (rc = mfma_prepare_policy(c, st))) return rc;      // of whatever theta the trials left in place
        if (nspec == 0) hipLaunchKernelGGL(k_ls_publish, dim3(1), dim3(64), 0, st, (const double*)v.scal, c->h_upd, c->upd_stamp);
        HIP_TRY(c, hipGetLastError());
        c->upd_pending = 1; c->upd_spec = nspec; c->upd_batch = *b; c->upd_params = *pr;
        return METRPO_OK;
    }
    // read-backs: ONE copy per line-search trial fetches scal[8] | lk[2] | ls[4] (loss at theta, beta, CG iterations, trial loss and KL)
    double loss = NAN, kl = NAN, loss_before = NAN;
    int n_iter = 0, n_start = 0;
    bool first = true, stopped = false, taken = false;
    if (phase == 2) {
        {   // the outcome _begin's last kernel published (a busy wait: this thread has nothing else to do, and a blocking wait costs 10-20 us to wake)
            volatile unsigned long long* stamp = (volatile unsigned long long*)(c->h_upd + 16);
            const auto t0 = std::chrono::steady_clock::now();
            long spins = 0;
            while (*stamp != c->upd_stamp) {
                if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
                    HIP_TRY(c, hipStreamSynchronize(st));         // surfaces a device fault; a healthy stream has published by now
                    if (*stamp != c->upd_stamp) return set_err(c, METRPO_EHIP, "trpo_update_end: the update's outcome never arrived");
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            for (int i = 0; i < 14; ++i) c->h_pinned[i] = c->h_upd[i];
        }
        if (c->xg_world > 1 && c->h_pinned[S_COMMERR] != 0.0) return set_err(c, METRPO_EHIP, "trpo_update: one-shot all-reduce timed out (a rank did not arrive)");
        if (c->h_pinned[S_ROLLERR] != 0.0) return rollout_error_seen(c, st);
        loss_before = c->h_pinned[S_LOSS0]; first = false;
        loss = c->h_pinned[11]; kl = c->h_pinned[12];
        n_start = nspec; n_iter = nspec > 0 ? nspec - 1 : 0;
        if (c->h_pinned[10] >= 0.0) { stopped = true; n_iter = (int)c->h_pinned[10]; taken = c->h_pinned[13] != 0.0; }
    }
    for (int n = n_start; n < pr->max_backtracks && !stopped; ++n) {
        n_iter = n;
        const double ratio = std::pow(pr->backtrack_ratio, (double)n);
        hipLaunchKernelGGL(k_try_theta, dim3((P + 255) / 256), dim3(256), 0, st, P, ratio, c->d_theta, v.step, c->d_theta_try, (const double*)nullptr, (double*)nullptr);
        if ((rc = launch_loss_kl(c, b, c->d_theta_try, v.lk, st))) return rc;
        AR(v.lk, 2);
        HIP_TRY(c, hipMemcpyAsync(c->h_pinned, v.scal, sizeof(double) * 10, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (c->xg_world > 1 && c->h_pinned[S_COMMERR] != 0.0) return set_err(c, METRPO_EHIP, "trpo_update: one-shot all-reduce timed out (a rank did not arrive)");
        if (c->h_pinned[S_ROLLERR] != 0.0) return rollout_error_seen(c, st);
        if (first) { loss_before = c->h_pinned[S_LOSS0]; first = false; }
        loss = c->h_pinned[8]; kl = c->h_pinned[9];
        if (loss < loss_before && kl <= pr->max_kl) break;
    }
    if (first) {
        HIP_TRY(c, hipMemcpyAsync(c->h_pinned, v.scal, sizeof(double) * 10, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st)); loss_before = c->h_pinned[S_LOSS0];
    }
    bool accepted = true;
    if ((std::isnan(loss) || std::isnan(kl) || loss >= loss_before || kl >= pr->max_kl) && !pr->accept_violation) accepted = false;
    if (stopped) accepted = taken;                               // decided on the device: an accepted trial's theta is already the policy
    else if (accepted) {
        std::swap(c->d_theta, c->d_theta_try);               // both ctx-owned, every launch takes c->d_theta afresh: no copy
        if (c->mfma_cfg >= 0 && (rc = mfma_prepare_policy(c, st))) return rc;
        c->upd_changed_in_end = (phase == 2) ? 1 : 0;        // theta changed on the HOST side of a two-half update: work enqueued after _begin used theta_prev
    }
    if (diag) {
        diag->loss_before = loss_before; diag->loss = loss; diag->kl = kl;
        diag->beta = c->h_pinned[S_BETA]; diag->n_backtrack = n_iter; diag->accepted = accepted ? 1 : 0;
        diag->cg_iters_run = (int)c->h_pinned[S_ITERS];
    }
    HIP_TRY(c, hipGetLastError());
#undef AR
    return METRPO_OK;
}

extern "C" int32_t metrpo_trpo_update(metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr,
                                      metrpo_trpo_diag* diag, double* g_out, double* dir_out, void* stream) {
    TraceRange trace_("metrpo:trpo_update (optimize_policy)");
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (!b || !pr) return set_err(c, METRPO_ENULL, "trpo_update: NULL pointer");
    if (pr->cg_iters < 0 || pr->max_backtracks < 1) return set_err(c, METRPO_EINVAL, "trpo_update: bad cg_iters/max_backtracks");
    if (c->upd_pending) return set_err(c, METRPO_ESTATE, "trpo_update: an update begun with metrpo_trpo_update_begin is still open (call metrpo_trpo_update_end)");
    return run_trpo_update(c, b, pr, diag, g_out, dir_out, (hipStream_t)stream, 0, 0);
}

// The same update in two halves, so that the host can go on enqueuing work (the next rollout) while the line search is being decided: _begin
// launches gradient, CG solve and the first `spec_trials` trials of the backtracking search with the accept test on the device and returns
// WITHOUT synchronising; launches that follow it on the stream see the accepted trial's theta (or theta_prev).  _end synchronises, reports the
// diagnostics and -- only if none of the speculative trials stopped the search -- runs the remaining trials the ordinary way.  If the returned
// *late_out is 1, the policy changed inside _end: work enqueued in between used theta_prev and must be redone.
// Where the accept test cannot run on the device (host all-reduce callback, RCCL transport, GEMM update path) _begin does the whole update
// synchronously and _end only hands out the stored diagnostics.  d_g_out / d_dir_out as in metrpo_trpo_update.
extern "C" int32_t metrpo_trpo_update_begin(metrpo_ctx* c, const metrpo_batch* b, const metrpo_trpo_params* pr, int32_t spec_trials,
                                            double* g_out, double* dir_out, void* stream) {
    TraceRange trace_("metrpo:trpo_update_begin (optimize_policy, line search decided on the device)");
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (!b || !pr) return set_err(c, METRPO_ENULL, "trpo_update_begin: NULL pointer");
    if (pr->cg_iters < 0 || pr->max_backtracks < 1 || spec_trials < 1) return set_err(c, METRPO_EINVAL, "trpo_update_begin: bad cg_iters / max_backtracks / spec_trials");
    if (c->upd_pending) return set_err(c, METRPO_ESTATE, "trpo_update_begin: the previous update is still open (call metrpo_trpo_update_end)");
    if (!device_line_search_ok(c, b, pr)) {
        const int rc = run_trpo_update(c, b, pr, &c->upd_diag, g_out, dir_out, (hipStream_t)stream, 0, 0);
        if (rc == METRPO_OK) { c->upd_pending = 2; c->upd_spec = pr->max_backtracks; }      // complete: _end hands out upd_diag
        return rc;
    }
    return run_trpo_update(c, b, pr, nullptr, g_out, dir_out, (hipStream_t)stream, 1, spec_trials);
}
extern "C" int32_t metrpo_trpo_update_end(metrpo_ctx* c, metrpo_trpo_diag* diag, int32_t* late_out, void* stream) {
    TraceRange trace_("metrpo:trpo_update_end");
    if (!c) return METRPO_ENULL;
    if (late_out) *late_out = 0;
    if (!c->upd_pending) return set_err(c, METRPO_ESTATE, "trpo_update_end: no update is open");
    if (c->upd_pending == 2) { if (diag) *diag = c->upd_diag; c->upd_pending = 0; return METRPO_OK; }
    metrpo_trpo_diag d = {};
    c->upd_changed_in_end = 0;
    const int rc = run_trpo_update(c, &c->upd_batch, &c->upd_params, &d, nullptr, nullptr, (hipStream_t)stream, 2, c->upd_spec);
    c->upd_pending = 0;
    if (rc == METRPO_OK) {
        if (diag) *diag = d;
        // not inferred from n_backtrack: with accept_violation and spec_trials >= max_backtracks no trial stops the search, _end still installs the last
        // trial's theta, and n_backtrack (= spec_trials - 1) is below upd_spec
        if (late_out) *late_out = c->upd_changed_in_end;
    }
    return rc;
}

// ---- BPTT policy update -----------------------------------------------------------------------
extern "C" int32_t metrpo_bptt_grad(metrpo_ctx* c, const float* init, int32_t B, int32_t T, double gamma, double* costs, double* grad,
                                    void* stream) {
    TraceRange trace_("metrpo:bptt_grad");
    if (!c) return METRPO_ENULL;
    NEED_DYN(c); NEED_POL(c);
    if (B <= 0 || T <= 0) return set_err(c, METRPO_EINVAL, "bptt_grad: B and T must be positive");
    if (!init) return set_err(c, METRPO_ENULL, "bptt_grad: NULL pointer");
    return launch_bptt_grad(c, init, B, T, gamma, costs, grad, (hipStream_t)stream);
}

extern "C" int32_t metrpo_policy_adam_reset(metrpo_ctx* c, void* stream) {
    if (!c) return METRPO_ENULL;
    return launch_policy_adam(c, nullptr, 0.0, 0.9, 0.999, 1e-8, 0.0, true, (hipStream_t)stream);
}

extern "C" int32_t metrpo_policy_adam_step(metrpo_ctx* c, const double* grad, double lr, double beta1, double beta2, double eps,
                                           double clip_val, void* stream) {
    if (!c) return METRPO_ENULL;
    NEED_POL(c);
    if (!grad) return set_err(c, METRPO_ENULL, "policy_adam_step: NULL pointer");
    if (!(lr >= 0.0)) return set_err(c, METRPO_EINVAL, "policy_adam_step: bad learning rate");
    return launch_policy_adam(c, grad, lr, beta1, beta2, eps, clip_val, false, (hipStream_t)stream);
}

// test hook (not in metrpo.h): 0 = generic sweep / validation kernels, 1 = fastest path the shape has.  Returns the path in use:
// 1 fused MFMA sweeps (bptt_mfma.hip), 2 GEMM-path sweeps (det_gemm.hip), 0 generic.
extern "C" int32_t metrpo_set_det_path(metrpo_ctx* c, int32_t use_mfma) {
    if (!c) return METRPO_ENULL;
    c->det_cfg = use_mfma ? det_mfma_select(c) : -1;
    c->det_gemm = (use_mfma && det_gemm_applicable(c)) ? 1 : 0;
    return c->det_cfg >= 0 ? 1 : (c->det_gemm ? 2 : 0);
}
