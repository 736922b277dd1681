// Internal definitions shared by the HIP translation units of libmetrpo.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <atomic>
#include <vector>
#include "metrpo.h"
#include "xchg_device.h"

#define MAXL (METRPO_MAX_LAYERS + 1)   // weight layers per MLP (hidden + output)
#define WAVE 64

// One MLP as offsets into a flat float vector: [W0 (n_in x n_out row-major), b0, W1, b1, ...].
struct NetDesc {
    int n_layers;            // weight layers
    int dims[MAXL + 1];      // dims[0] = inputs ... dims[n_layers] = outputs
    int act[MAXL];           // metrpo_act applied after layer l (last: identity)
    int w_off[MAXL];         // offsets in the DEVICE-RESIDENT layout.  Dynamics: every W/b array and every model starts on a
    int b_off[MAXL];         // 16-byte boundary (pads are zero) so the GEMM/MFMA kernels can use float4 loads for all heads;
    int n_params;            // policy: identical to the API layout.  n_params = floats of one network (policy: WITHOUT log_std)
    int api_w_off[MAXL];     // offsets in the dense caller-visible layout of include/metrpo.h
    int api_b_off[MAXL];
    int api_n_params;
    int max_width;           // max over dims
};

// Normaliser block layout in ctx->d_norm: in_mean[ns+na] | in_std[ns+na] | diff_mean[ns] | diff_std[ns]
struct ProblemDesc {
    int env, ns, na, K, n_drop, nin;   // nin = ns + na - n_drop
    NetDesc dyn, pol;
    int P;                             // policy params incl. log_std
};

#define METRPO_MAX_PAR_ROUNDS 8
// ---- variant / tuning switches of a context (metrpo_set_option / metrpo_get_option, include/metrpo.h) ----------------------------------------------
// One table per context, read by the launch paths through ctx_opt(); metrpo_create fills the defaults ONCE from the environment (METRPO_<KEY>), nothing
// else in the library reads the environment for kernel selection.  A key is the upper-case name below (the ABI also takes lower case and a METRPO_ prefix).
#define METRPO_OPT_LIST(X) X(NO_FUSED_OUT) X(NO_L0_ROWS) X(NO_MERGED_ROUNDS) X(NO_RESIDENT) X(NO_RESIDENT_VALIDATION) X(NO_STREAMK) X(PRE_GEMM) X(RESIDENT_PLAN) X(RESIDENT_TEST_SKIP) X(RESIDENT_WS) X(SEQ_ROUNDS) X(STEP_MERGE) X(STREAMK) X(STREAMK_LATE) X(STREAMK_PLACE) X(VAL_PLAN) X(XCHG_TIMEOUT_MS) X(NO_PERSIST) X(QUIET) X(TIME_FVP) X(PERSIST_STATS) X(PERSIST_WIDE) X(PERSIST_NCLOSE) X(NO_POL_FUSED3) X(NO_PRE_SPLIT)
enum MetrpoOpt {
#define X(n) OPT_##n,
    METRPO_OPT_LIST(X)
#undef X
    OPT_COUNT
};

struct metrpo_ctx {
    int device;
    metrpo_dims dims;
    ProblemDesc pd;
    float* d_dyn;        // [K][dyn.n_params]
    float* d_norm;       // 2*(ns+na) + 2*ns
    float* d_theta;      // [P]
    bool have_dyn, have_pol;
    // --- MFMA fast path (rollout_mfma.hip): pre-permuted weight images, built by set_* ---
    float* d_dyn_img;    // per-model register image, see rollout_mfma.hip
    float* d_pol_img;    // (int32 payload) gather map of the policy weight-fragment image, see policy_mfma.hip
    int pol_img_idx;     // table index the map was built for (-1: none)
    // image VALUES of one CG solve (policy_mfma.hip): [weight entries of theta, written by the gradient kernel's block 0 | tangent entries of the
    // current CG vector, written by the fused CG tails through d_pol_vpos (theta index -> image position, -1: none)].  img_live is raised by
    // run_trpo_update while both writers are on the launch sequence; the cached-activation FVP then copies the image instead of gathering it.
    float* d_pol_imgval; int* d_pol_vpos; int img_live;
    // --- BPTT (bptt.hip) ---
    void* d_bptt; size_t bptt_cap;      // XS | WT | GM | gout | costs
    const float* vjp_gm;                // set around the VJP launch of the gradient kernels
    const double* ls_skip;              // set around a speculative line-search evaluation (PolK::skip of the fused MFMA kernels)
    void* d_pol_adam; int pol_adam_t;   // Adam moments of the policy parameters + segment table
    int det_cfg;                        // bptt_mfma.hip table index (-1: generic sweeps / generic validation kernel)
    double* d_detpart; size_t detpart_cap;   // per-tile cost partials of the MFMA forward sweep
    int det_gemm;                       // 1: GEMM-path sweeps (det_gemm.hip) for large dynamics nets
    void* d_dg; size_t dg_cap;          // workspace of the GEMM-path sweeps
    int mfma_cfg;        // index into the instantiation table, -1 = generic path only
    int pol_mfma;        // index into policy_mfma.hip's table, -1 = generic update kernels
    int coop_cfg;        // index into rollout_coop.hip's table, -1 = head-per-wave kernel (rollout_mfma.hip)
    // two hidden layers of at most 64 units each, not both 64 (round 6): the cooperative kernel on a zero-padded copy of the weights in the 64 x 64 layout (padded units: zero weights and
    // bias -> relu(0) = 0 -> they add exact zeros); the copy is rebuilt from d_dyn in front of every rollout launch (one small kernel: no tracking of who wrote d_dyn)
    int coop_pad_cfg; float* d_dyn_pad; NetDesc dyn_pad;
    int det_padded;      // the validation-cost / BPTT sweeps of bptt_mfma.hip run on the same padded copy
    int rollout_variant; // test hook: 0 = fastest available, 1 = head-per-wave MFMA kernel
    // --- workspaces for the update path (lazily sized) ---
    float* d_partials;   // [n_blocks][P+2] per-block partial sums
    size_t partials_cap;
    double* d_cg;        // CG vectors + scalars, see trpo_update.hip
    float* d_vf;         // [P] float copy of the FVP input
    float* d_theta_try;  // [P] line-search candidate
    double* d_valbuf;    // validation-cost accumulators
    double* d_vbuf;      // [N] baseline predictions for the GAE scan
    double* d_gae_part; size_t gae_part_cap;   // k_gae: arrival ticket + one (sum adv, sum adv^2, count) triple per workgroup, added in workgroup order
    size_t vbuf_cap;
    double* d_gram_part; // per-block Gram partials (process.hip)
    size_t gram_cap;
    unsigned int* d_ticket; // arrival counter of k_finalize's fused CG tail
    float* d_hcache; size_t hcache_cap; int hcache_on;   // activation cache of one CG solve (policy_mfma.hip MODE_FVPC)
    void* d_mig; int mig_cap, mig_epoch;            // rollout_coop.hip: hand-over slots of migrating tiles (flag | ts | model | obs per tile)
    void* nccl_comm; int comm_world, comm_rank;   // comm.hip: RCCL communicator attached by metrpo_comm_init (NULL: single rank)
    // comm.hip: one-shot direct all-reduce (xchg_device.h).  xg_region = this rank's receive region (IPC-exported), xg_peer[q] = rank q's
    // region as mapped here; xg_seq counts the exchanges issued so far (identical on every rank: SPMD); xg_fuse is raised by
    // run_trpo_update while the reductions of the update kernels carry the exchange in their own tail
    void* xg_region; void* xg_peer[XCHG_MAX_WORLD]; int xg_world, xg_rank, xg_cap, xg_fuse; unsigned int xg_seq; unsigned long long xg_timeout;
    int pol_path;        // 1 auto (fused MFMA kernels where the shape has them, GEMM path for large N otherwise), 0 generic forced, 2 GEMM path forced
    void* d_pg; size_t pg_cap; long long pg_fwd_rows; const float* pg_fwd_obs;   // policy_gemm.hip workspace + validity of its cached forward pass
    int pol_f3;          // 1: fused MFMA update kernels for three-hidden-layer policies (policy_fused3.hip) serve this shape
    void* d_f3; size_t f3_cap; long long f3_rows; const float* f3_obs; const float* f3_theta; int f3_img_ok;   // policy_fused3.hip: activation cache + mean-adjoint of one (theta, batch) and its validity
    void* d_adam;        // Adam moments [2][K][Pd] + loss accumulators (dyn_train.hip)
    long long adam_t;    // Adam step count
    void* d_train;       // training activation workspace
    double* d_train_part; size_t train_part_cap;   // k_train_out: per-model arrival tickets + per-workgroup loss sums (added in workgroup order)
    size_t train_cap;
    void* d_big;         // workspace of the GEMM step-wise rollout (rollout_gemm.hip)
    size_t big_cap;
    void* d_res; size_t res_cap; unsigned int res_seq;   // rollout_resident.hip: uncached exchange region (abort cell | X packets | P packets) and the step stamps issued so far
    unsigned long long* d_skp_stats; int skp_stats_n;   // option PERSIST_STATS: per-workgroup statistics of the last persistent launch (metrpo_debug_persist_stats)
    std::vector<int> skp_tab_host;                    // host copy of the table below, as raw 32-bit words (source of its asynchronous upload; SkRec: mlp_streamk.h)
    void* d_skp_tab; size_t skp_tab_cap; long long skp_key[8]; int skp_Jx[8], skp_Jmax, skp_L, skp_NSL; int persist_failed;   // mlp_persist.h: cached chunk-record table of the persistent stream-K rollout (key: the launch's shape) | a persistent launch timed out
    int res_failed;                                       // a resident launch gave up (its grid was not co-resident): this context stays on the step-wise path from then on
    int last_rollout_kernel;
    // metrpo_trpo_update_begin / _end: an update whose line search is still undecided on the host
    // metrpo_trpo_update_begin's outcome lands in pinned host memory straight from its last kernel (k_ls_publish: scal | lk | ls, then a
    // stamp); _end polls the stamp (no copy engine, no event, no blocking wait to wake up from).  Publishing from a side stream behind a device-scope
    // event was measured too: the second queue costs the update 35 us, more than the 15 us gap in front of the next rollout it removes.
    double* h_upd; unsigned long long upd_stamp;
    int upd_pending, upd_spec, upd_changed_in_end; metrpo_batch upd_batch; metrpo_trpo_params upd_params; metrpo_trpo_diag upd_diag;                              // which kernel family the last metrpo_rollout ran on: 0 generic, 1 head-per-wave MFMA, 2 cooperative MFMA, 3 step-wise GEMM, 4 resident
    hipStream_t side_stream[METRPO_MAX_PAR_ROUNDS - 1]; hipEvent_t ev_fork, ev_join[METRPO_MAX_PAR_ROUNDS - 1]; int side_ready;   // rollout_gemm.hip: independent rounds of a small-batch rollout run concurrently
    double* h_pinned;    // pinned host scratch for the per-trial read-back
    int n_sm;            // CU count (device property)
    int upd_tiles_per_wave;   // MFMA update kernels: at least this many 16-sample tiles per wave before another block is added (1; the option that set it was retired in round 6: experiments/upd_small.py)
    int n_cu_sched;      // CUs that actually ran this process's waves (probe.hip: census; 0 = not measured yet)
    int exclusive;       // the caller's metrpo_set_exclusive value (1 at metrpo_create); 0: the GPU is shared with other compute processes.  Read through ctx_exclusive(), which also honours option NO_RESIDENT
    std::string opt_val[OPT_COUNT]; bool opt_set[OPT_COUNT];   // METRPO_OPT_LIST: set by metrpo_create from the environment, then only by metrpo_set_option
    hipEvent_t fvp_ev[32]; int fvp_ev_n, fvp_ev_made;   // option TIME_FVP: events around the Fisher-vector-product kernel of launch_fvp_tail (metrpo_debug_fvp_us)
    std::string rollout_note;   // why the last metrpo_rollout left the fast dispatch table ("" when it did not): metrpo_rollout_note
    int fallback_logged;  // a rollout shape that fell off the fast dispatch table has been reported once (METRPO_VERBOSE)
    std::vector<void*> ws_retired; size_t ws_retired_bytes = 0;   // outgrown workspaces (ws_retire below): freed by metrpo_destroy, or by one sweep once they pass WS_RETIRED_MAX
    std::string err;
};

// Workspace growth inside a launch entry point (a larger B / N than any call before).  hipFree waits for the whole device -- every stream of the process -- while the
// ABI promises stream-ordered calls (include/metrpo.h, Threading): an outgrown buffer is RETIRED instead of freed.  Kernels already enqueued on any stream may still
// read it, which is exactly what retiring allows.  Retired buffers go at metrpo_destroy; a caller that sweeps ever larger shapes through ONE context pays one
// synchronising sweep whenever the retired bytes pass WS_RETIRED_MAX (or an allocation fails), a loop at fixed shapes never does.  hipMalloc does not wait for
// running work (tests/test_gpu_api.py::test_rollout_at_a_larger_batch_does_not_wait_for_other_streams).
constexpr size_t WS_RETIRED_MAX = (size_t)4 << 30;
static inline void ws_sweep(metrpo_ctx* c) {
    for (void* p : c->ws_retired) (void)hipFree(p);           // (the first hipFree waits for the device)
    c->ws_retired.clear(); c->ws_retired_bytes = 0;
}
static inline void ws_retire(metrpo_ctx* c, void* p) {
    if (!p) return;
    size_t sz = 0;
    if (hipMemPtrGetInfo(p, &sz) != hipSuccess) { (void)hipGetLastError(); sz = 0; }
    c->ws_retired.push_back(p); c->ws_retired_bytes += sz;
    if (c->ws_retired_bytes > WS_RETIRED_MAX) ws_sweep(c);
}
static inline hipError_t ws_alloc(metrpo_ctx* c, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && !c->ws_retired.empty()) { (void)hipGetLastError(); ws_sweep(c); e = hipMalloc(p, bytes); }
    return e;
}

// value of a switch, NULL when unset -- the same contract as the getenv() calls these replaced
static inline const char* ctx_opt(const metrpo_ctx* c, int id) { return c->opt_set[id] ? c->opt_val[id].c_str() : nullptr; }
// kernels that wait on other workgroups of their own launch may be selected: the caller said the device is its own (metrpo_set_exclusive) AND option NO_RESIDENT is unset
static inline bool ctx_exclusive(const metrpo_ctx* c) { return c->exclusive != 0 && ctx_opt(c, OPT_NO_RESIDENT) == nullptr; }
// the three-hidden-layer fused update kernels (policy_fused3.hip) serve this context's next update launch (not the VJP mode of the gradient kernels: GEMM path)
static inline bool f3_active(const metrpo_ctx* c) { return c->pol_f3 != 0 && c->pol_path == 1 && c->vjp_gm == nullptr && ctx_opt(c, OPT_NO_POL_FUSED3) == nullptr; }
const char* metrpo_opt_name(int id);
int metrpo_opt_id(const char* key);       // -1: unknown

struct RolloutK {          // device-side copy of metrpo_rollout_args (plain pointers)
    int B, T, H, sam_mode, determ, eval_all, n_pool;
    uint64_t seed, stream_offset;
    const float* pool;
    const float* eps;
    const int32_t* model_idx;
    const float* sel_noise;
    const int32_t* reset_idx;
    const int32_t* reset_model;
    float* obs; float* act; float* rew; float* mean; uint8_t* done; int32_t* tpath; float* last_obs;
    // continuation (metrpo_rollout_args ABI 2)
    int t0; const float* init_obs; const int32_t* init_ts; const int32_t* init_model;
    int32_t* last_ts; int32_t* last_model; const int32_t* stop;
    long long stop_batch; const double* stop_cum;     // in-launch stop rule (metrpo_rollout_args ABI 4): honoured by the persistent stream-K rollout only
    // tile migration of the cooperative kernel (rollout_coop.hip; ctx-owned hand-over slots, NULL elsewhere)
    int32_t* mig_flag; float* mig_obs; int32_t* mig_ts; int32_t* mig_model; int mig_epoch; double* mig_err;
    // merged rounds of the step-wise path (rollout_gemm.hip): vB > 0 -> the B rows of this launch are vR rounds of vB envs, row b = env b % vB of round
    // b / vB, whose steps are the rows t + (b / vB) * H of the trajectory tensors (vB envs per row) and of the draw counters
    int vB, vR;
};
// env index within its round (Philox stream, column of the trajectory row) | first step of the env's round | envs per trajectory row
#define RK_ENV(r, b) ((r).vB ? (b) % (r).vB : (b))
#define RK_TOFF(r, b) ((r).vB ? ((b) / (r).vB) * (r).H : 0)
#define RK_STRIDE(r) ((r).vB ? (r).vB : (r).B)
#define RK_LAST_ROUND(r, b) (!(r).vB || (b) / (r).vB == (r).vR - 1)

static inline RolloutK make_rollout_k(const metrpo_rollout_args* a) {
    RolloutK r;
    r.B = a->B; r.T = a->T; r.H = a->H; r.sam_mode = a->sam_mode; r.determ = a->determ; r.eval_all = a->eval_all_heads;
    r.n_pool = a->n_pool; r.seed = a->seed; r.stream_offset = a->stream_offset; r.pool = a->d_pool; r.eps = a->d_eps;
    r.model_idx = a->d_model_idx; r.sel_noise = a->d_sel_noise; r.reset_idx = a->d_reset_idx;
    r.reset_model = a->d_reset_model; r.obs = a->d_obs; r.act = a->d_act; r.rew = a->d_rew; r.mean = a->d_mean;
    r.done = a->d_done; r.tpath = a->d_tpath; r.last_obs = a->d_last_obs;
    r.t0 = a->t0; r.init_obs = a->d_init_obs; r.init_ts = a->d_init_ts; r.init_model = a->d_init_model;
    r.last_ts = a->d_last_ts; r.last_model = a->d_last_model; r.stop = a->d_stop;
    r.stop_batch = a->stop_batch; r.stop_cum = a->d_stop_cum;
    r.mig_flag = nullptr; r.mig_obs = nullptr; r.mig_ts = nullptr; r.mig_model = nullptr; r.mig_epoch = 0; r.mig_err = nullptr;
    r.vB = 0; r.vR = 0;
    return r;
}

struct PolK {
    const float* obs; const float* act; const float* adv; const float* old_mean; const float* old_ls;
    int ls_stride; const uint8_t* valid; long long N; float inv_n;
    const float* gm;         // non-NULL: VJP mode of the gradient kernels (bptt.hip): d objective / d mean [N][na] supplied, no loss terms
    const int* img_map;      // policy_mfma.hip: gather map of the LDS weight-fragment image (built once per ctx on the host)
    const double* skip;      // non-NULL: a line-search trial that leaves at once when skip[0] >= 0 (the search already stopped: CgTail::ls)
    float* imgval;           // policy_mfma.hip: non-NULL while metrpo_ctx::img_live -- gradient kernel: block 0 publishes its image here; MODE_FVPC: the image to copy
    float* hcache;           // policy_mfma.hip: hidden activations of (theta, batch): written by the gradient kernel, read by MODE_FVPC
};

int policy_mfma_select(const ProblemDesc& pd);
int policy_mfma_image_buffers(metrpo_ctx*);   // gather map, its inverse for the tangent entries and the image-value buffer of ctx->pol_mfma (idempotent)
struct CgTail;
int policy_mfma_launch(metrpo_ctx*, int idx, int mode, const metrpo_batch*, const float* theta, const float* v,
                       float* partials, int nblocks, hipStream_t);

int set_err(metrpo_ctx* c, int code, const std::string& msg);
#define HIP_TRY(c, expr)                                                                      \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return set_err((c), METRPO_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// launch helpers implemented in the individual .hip files
int launch_policy_actions(metrpo_ctx*, const float*, const float*, int, float*, float*, hipStream_t);
int launch_step(metrpo_ctx*, const float*, const float*, int, int, const int32_t*, const float*, float*, float*,
                uint8_t*, float*, hipStream_t);
int launch_rollout_generic(metrpo_ctx*, const metrpo_rollout_args*, hipStream_t);
bool gemm_path_applicable(const metrpo_ctx*);
int sched_cus(metrpo_ctx*, hipStream_t);
bool grid_is_coresident(metrpo_ctx*, const void* kernel, int threads, size_t lds, long long grid, hipStream_t);
int launch_rollout_gemm(metrpo_ctx*, const metrpo_rollout_args*, hipStream_t);
int launch_rollout_resident(metrpo_ctx*, const metrpo_rollout_args*, hipStream_t);   // METRPO_EUNSUPPORTED: not a shape / call of the resident kernel
int launch_bptt_grad(metrpo_ctx*, const float* init, int B, int T, double gamma, double* costs, double* grad, hipStream_t);
int launch_policy_adam(metrpo_ctx*, const double* grad, double lr, double b1, double b2, double eps, double clip_val, bool reset, hipStream_t);
int det_mfma_select(const metrpo_ctx*);
int launch_det_forward(metrpo_ctx*, int idx, const float* s0, int B, int T, double gamma, float* XS, float* WT, double* part, double* costs, hipStream_t);
int launch_det_backward(metrpo_ctx*, int idx, int B, int T, const float* XS, const float* WT, float* GM, hipStream_t);
int ensure_detpart(metrpo_ctx*, int B);
int ensure_detpart_n(metrpo_ctx*, size_t n_doubles);
int launch_det_cost_reduce(metrpo_ctx*, int n_part, const double* part, double* costs, hipStream_t, const double* err = nullptr);   // err: time-out cell of the launch that wrote the partials (NaN costs when set)
int launch_validation_resident(metrpo_ctx*, const float* s0, int Bv, int T, double gamma, double* costs, hipStream_t);   // rollout_resident.hip; METRPO_EUNSUPPORTED: not this shape
bool det_gemm_applicable(const metrpo_ctx*);
int launch_dg_forward(metrpo_ctx*, const float* s0, int B, int T, double gamma, float* XS, float* WT, double* costs, hipStream_t);
int launch_dg_backward(metrpo_ctx*, int B, int T, const float* XS, const float* WT, float* GM, hipStream_t);
int launch_policy_vjp(metrpo_ctx*, const float* obs, const float* gm, long long N, double* out, hipStream_t);
int launch_dyn_train_step(metrpo_ctx*, const float*, const float*, const metrpo_train_params*, double*, hipStream_t);
int launch_dyn_eval_losses(metrpo_ctx*, const float*, const float*, long long, double, double*, hipStream_t);
int launch_rms_accumulate(metrpo_ctx*, const float*, long long, int, double*, double*, hipStream_t);
int launch_rollout_mfma(metrpo_ctx*, const metrpo_rollout_args*, hipStream_t, int* coop = nullptr);   // returns METRPO_EUNSUPPORTED if no instantiation fits; *coop = 1: the cooperative kernel ran
int mfma_prepare_dynamics(metrpo_ctx*, hipStream_t);
int mfma_prepare_policy(metrpo_ctx*, hipStream_t);
int mfma_select_config(metrpo_ctx*);
int mfma_shape_config(const metrpo_ctx*);
int coop_select_config(metrpo_ctx*);
int launch_rollout_coop(metrpo_ctx*, int idx, const RolloutK&, hipStream_t, bool padded = false);
int launch_pad_dyn(metrpo_ctx*, hipStream_t);      // d_dyn -> d_dyn_pad (zero-padded 64 x 64 layout)
int launch_validation_cost(metrpo_ctx*, const float*, int, int, double, double*, hipStream_t);
int launch_gae(metrpo_ctx*, const float*, const float*, const uint8_t*, const int32_t*, int, int, const double*,
               double, double, float*, float*, uint8_t*, double*, hipStream_t);
int launch_center(metrpo_ctx*, float*, const uint8_t*, int64_t, const double*, hipStream_t);
int launch_process_begin(metrpo_ctx*, float*, double*, int64_t, hipStream_t);
int launch_sampler_progress(metrpo_ctx*, const uint8_t*, const int32_t*, int, int, int, long long, double*, double*, int32_t*, hipStream_t);
int launch_baseline_solve(metrpo_ctx*, const double* AtA, const double* Aty, double reg, double* coeffs, hipStream_t);
int launch_gram(metrpo_ctx*, const float*, const float*, const int32_t*, const uint8_t*, int64_t, double*, double*,
                hipStream_t);
int launch_loss_grad(metrpo_ctx*, const metrpo_batch*, double*, hipStream_t, const CgTail* tail = nullptr);
int comm_allreduce_f64(metrpo_ctx*, double* buf, long long count, hipStream_t);
// descriptor of the NEXT one-shot exchange (advances the sequence number); world = 0 when no peer-mapped transport is attached
XchgK xchg_next(metrpo_ctx*);
static inline XchgK xchg_none() { XchgK x = {}; return x; }
// scal[S_COMMERR] of the CG workspace (gout[1+P] | x r p z step [5P] | scal[8] | lk[2]): sticky error cell of the exchanges
static inline double* comm_err_cell(metrpo_ctx* c) { return c->d_cg + (size_t)(1 + c->pd.P) + 5 * (size_t)c->pd.P + 6; }
// time-out cell of the resident VALIDATION launches (behind scal | lk | ls): cleared in front of every such launch, so an earlier rollout's sticky S_ROLLERR
// cannot poison validation costs and a validation time-out cannot be mistaken for a rollout's
static inline double* val_err_cell(metrpo_ctx* c) { return comm_err_cell(c) + 8; }
// A rollout kernel reported a timed-out hand-over (scal[S_ROLLERR]): the trajectories of that launch are invalid.  The cell is cleared so the
// context can go on, and the resident kernel -- the one whose hand-overs need every workgroup of its grid on the chip at once -- is retired.
static inline int rollout_error_seen(metrpo_ctx* c, hipStream_t st) {
    (void)hipMemsetAsync(comm_err_cell(c) + 1, 0, sizeof(double), st);
    const bool was_resident = (c->last_rollout_kernel == 4);
    if (was_resident) c->res_failed = 1;
    if (c->last_rollout_kernel == 6) {
        c->persist_failed = 1;
        return set_err(c, METRPO_EHIP, "rollout: the persistent stream-K kernel's wait for a row block timed out (a workgroup of its grid never ran: is another process using "
                                       "this GPU?); the trajectories of that launch are invalid, later rollouts of this context use the launch-per-step path");
    }
    return set_err(c, METRPO_EHIP, was_resident ? "rollout: the resident kernel's hand-over timed out (a workgroup of its grid never ran: is another process using this GPU?); "
                                                  "the trajectories of that launch are invalid, later rollouts of this context use the step-wise path"
                                                : "rollout: a migrating tile's hand-over timed out (producer workgroup never ran); trajectories are invalid");
}
bool policy_gemm_applicable(const metrpo_ctx*, long long N);
int policy_f3_select(const ProblemDesc& pd);      // policy_fused3.hip: 1 when the three-hidden-layer kernels cover this policy shape
int policy_f3_launch(metrpo_ctx*, int mode, const metrpo_batch*, const float* theta, const float* vf, float* partials, int nblocks, hipStream_t);
int policy_gemm_run(metrpo_ctx*, int mode, const metrpo_batch*, const PolK&, const float* theta, const float* vf, const double* v64, double* out,
                    const CgTail* tail, hipStream_t);
int launch_fvp(metrpo_ctx*, const metrpo_batch*, const double*, double*, hipStream_t);
// vf = float copy of v already on the device (skips the conversion launch); v is still needed for the log_std rows
int launch_fvp_f32(metrpo_ctx*, const metrpo_batch*, const float* vf, const double* v, double* hv, hipStream_t);
// FVP + reduction + (in the reduction kernel's last block) the CG vector step described by `tail`
int launch_fvp_tail(metrpo_ctx*, const metrpo_batch*, const float* vf, const double* v, double* hv, const CgTail* tail, hipStream_t);
int launch_loss_kl(metrpo_ctx*, const metrpo_batch*, const float*, double*, hipStream_t, const CgTail* decide = nullptr);   // decide: op 4 tail (device-side accept test)
int run_trpo_update(metrpo_ctx*, const metrpo_batch*, const metrpo_trpo_params*, metrpo_trpo_diag*, double*,
                    double*, hipStream_t, int phase = 0, int spec = 0);
