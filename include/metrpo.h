/*
 * metrpo.h -- C ABI of libmetrpo.so: the MI355X (gfx950) implementation of the ME-TRPO
 * policy-optimisation inner loop (imagined rollout -> GAE -> TRPO CG/FVP update).
 *
 * The reference (thanard/me-trpo) is pure Python on TF1.4 + rllab and has NO FFI of its own
 * (SURVEY.md section 2); this header therefore declares the entry points a binding for that path
 * would need, one per reference operation, each citing the reference interface it replaces
 * (file:line under the reference tree).  `[rllab]` marks arithmetic that lives in the reference's
 * third-party dependency rllab (not vendored, no version pinned) and is reached at the cited
 * call site.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - every pointer named `d_*` is DEVICE memory owned by the caller (e.g. torch `data_ptr()`);
 *    the library never frees or retains it past the call, except the `set_*` calls which COPY.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, calls return
 *    without synchronising unless documented.  One ctx per GPU, one host thread per ctx, and the
 *    calls of a ctx on ONE stream at a time (they share ctx-owned workspaces: work of the same ctx
 *    enqueued on two streams is not ordered against itself).
 *  - no launch entry point frees device memory or synchronises the device: a ctx-owned workspace
 *    that has to grow (a B / N larger than any call before) is allocated with hipMalloc and the
 *    outgrown buffer is kept until metrpo_destroy (one synchronising sweep only once 4 GB of
 *    outgrown buffers have piled up in a context that keeps growing its shapes).
 *  - all functions return 0 (METRPO_OK) or a negative metrpo_status; no exception or abort
 *    crosses the ABI; metrpo_last_error() gives the message for the last failure on a ctx.
 *  - arithmetic is float32 on device (TF graph dtype of the reference); CG vectors, reductions
 *    and advantage statistics are float64, as the reference's host-side NumPy is.
 *  - trajectory tensors are TIME-MAJOR: element (t, b) of a [T][B][w] tensor is at
 *    ((t*B)+b)*w.  B = parallel imagined envs (the reference's n_envs), T = steps executed.
 */
#ifndef METRPO_H
#define METRPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define METRPO_ABI_VERSION 4
#define METRPO_MAX_LAYERS 6      /* hidden layers per MLP */

typedef struct metrpo_ctx metrpo_ctx;

typedef enum {
    METRPO_OK = 0,
    METRPO_EINVAL = -1,          /* bad dimension / argument */
    METRPO_ENULL = -2,           /* required pointer is NULL */
    METRPO_EHIP = -3,            /* HIP runtime error, see metrpo_last_error */
    METRPO_EUNSUPPORTED = -4,    /* configuration exceeds this build's limits (e.g. LDS) */
    METRPO_ESTATE = -5,          /* set_dynamics / set_policy not called yet */
    METRPO_UNSET = -100          /* metrpo_get_option: the key is known and has no value (not an error; distinct from METRPO_EINVAL = unknown key) */
} metrpo_status;

/* analytic reward / termination of the env families the reference ships (envs/com_*_env.py) */
typedef enum {
    METRPO_ENV_SWIMMER = 0,      /* envs/com_swimmer_env.py:112-114        */
    METRPO_ENV_HALF_CHEETAH = 1, /* envs/com_half_cheetah_env.py:72-75     */
    METRPO_ENV_ANT = 2,          /* envs/com_ant_env.py:77-83, is_done :88-101 */
    METRPO_ENV_HUMANOID = 3,     /* envs/com_simple_humanoid_env.py:105-109 */
    METRPO_ENV_HOPPER = 4,       /* envs/com_hopper_env.py:94-104          */
    METRPO_ENV_SNAKE = 5         /* envs/com_snake_env.py:81-84            */
} metrpo_env;

/* VecSimpleEnv.get_next_observation sampling modes, env_helpers.py:617-634 */
typedef enum {
    METRPO_SAM_STEP_RAND = 0, METRPO_SAM_EPS_RAND = 1, METRPO_SAM_MODEL_MEAN_STD = 2,
    METRPO_SAM_MODEL_MEAN = 3, METRPO_SAM_MODEL_MED = 4, METRPO_SAM_ONE_MODEL = 5
} metrpo_sam_mode;

typedef enum { METRPO_ACT_IDENTITY = 0, METRPO_ACT_RELU = 1, METRPO_ACT_TANH = 2 } metrpo_act;

/* Static shape of the problem (params/params-<env>.json keys in brackets).
 * Dynamics-model variant: prediction_type "state_change" only (training.py:257).  "second_derivative" (training.py:259-264), the "*_goal"
 * types (:265-268) and use_logit_weights (:234-242) have no field here and no kernel behind it: a host reading such a params file must refuse
 * it by name (me-trpo_amd/engine.py: Engine(prediction_type=..., use_logit_weights=...) raises). */
typedef struct {
    int32_t env;                              /* metrpo_env                                   */
    int32_t ns, na;                           /* state / action dims                          */
    int32_t n_models;                         /* K  [n_models]                                */
    int32_t dyn_n_hidden;                     /* [dynamics_model.hidden_layers]               */
    int32_t dyn_hidden[METRPO_MAX_LAYERS];
    int32_t dyn_act[METRPO_MAX_LAYERS];       /* metrpo_act per hidden layer [.nonlinearity]  */
    int32_t n_drop;                           /* 2: ignore_xy_input, 1: ignore_x_input, 0     */
    int32_t pol_n_hidden;                     /* [policy.hidden_layers], tanh, identity out   */
    int32_t pol_hidden[METRPO_MAX_LAYERS];
} metrpo_dims;

int32_t metrpo_abi_version(void);
const char* metrpo_status_string(int32_t status);

/* ---- context ------------------------------------------------------------------------------ */
int32_t metrpo_create(metrpo_ctx** out, int32_t device, const metrpo_dims* dims);
int32_t metrpo_destroy(metrpo_ctx* ctx);
/* exclusive = 0: other compute processes share this GPU (several ranks per device).  Kernels that wait on other workgroups of their own launch
 * (the resident rollout / validation kernels, the migrating-tile schedule) are then never selected; default 1 (0 when METRPO_NO_RESIDENT is set
 * in the environment).  With exclusive use they are still only launched when the runtime's occupancy answer times the CUs that actually
 * schedule this process's waves (a census: CU masks and partitions count) covers their grid.  No reference counterpart: the reference is
 * single-process and single-session (utils.py:229-232). */
int32_t metrpo_set_exclusive(metrpo_ctx* ctx, int32_t exclusive);
/* Variant / tuning switches of a context (ABI 4).  The reference steers its run with params-*.json and command-line flags (main.py:21-60); this library's
 * kernel-selection switches were process-global METRPO_<KEY> environment variables read inside the launch paths up to ABI 3.  They are now one table per
 * context: metrpo_create() fills the defaults ONCE from the environment (METRPO_<KEY>), afterwards only metrpo_set_option() changes them (value NULL
 * unsets a key; takes effect at the next launch) and metrpo_get_option() reports them (returns the value's length, METRPO_UNSET when unset, METRPO_EINVAL for an unknown
 * key).  Keys: the names metrpo_option_name(i), i = 0, 1, ... returns (NULL beyond the table), e.g. "STREAMK", "NO_STREAMK", "NO_RESIDENT", "SEQ_ROUNDS",
 * "PRE_GEMM"; lower case and a "METRPO_" prefix are accepted.  None of them changes results beyond the summation-order notes in DESIGN.md section 4. */
int32_t metrpo_set_option(metrpo_ctx* ctx, const char* key, const char* value);
int32_t metrpo_get_option(metrpo_ctx* ctx, const char* key, char* buf, int32_t cap);
const char* metrpo_option_name(int32_t index);
const char* metrpo_last_error(const metrpo_ctx* ctx);
/* floats per dynamics model: [W0 (n_in x h0, row-major), b0, W1, b1, ..., Wout, bout]          */
int32_t metrpo_dyn_param_count(const metrpo_ctx* ctx);
/* floats in the flat policy vector, [rllab] get_param_values order: W0,b0,...,Wout,bout,log_std */
int32_t metrpo_policy_param_count(const metrpo_ctx* ctx);

/* Replaces the K `dynamics_model` graph copies + RunningMeanStd read-outs the reference builds at
 * model_based_rl.py:91-97 from training.py:218-269 / running_mean_std.py:22-27.
 * d_params [K][dyn_param_count]; d_in_mean,d_in_std [ns+na]; d_diff_mean,d_diff_std [ns]. COPIES. */
int32_t metrpo_set_dynamics(metrpo_ctx* ctx, const float* d_params, const float* d_in_mean,
                            const float* d_in_std, const float* d_diff_mean, const float* d_diff_std,
                            void* stream);
/* [rllab] policy.set_param_values / get_param_values (used by ConjugateGradientOptimizer and by
 * the snapshot/restore at model_based_rl.py:1127-1129,1291-1293,1400).  d_theta [P]. COPIES. */
int32_t metrpo_set_policy(metrpo_ctx* ctx, const float* d_theta, void* stream);
int32_t metrpo_get_policy(metrpo_ctx* ctx, float* d_theta_out, void* stream);

/* ---- single step API (drop-in granularity of the reference's sampler loop) ---------------- */
/* [rllab] GaussianMLPPolicy.get_actions, called at samplers/vectorized_sampler.py:63.
 * d_obs [B][ns]; d_eps [B][na] N(0,1) draws or NULL (actions = mean, the determ=True path :64-65);
 * outputs d_actions [B][na] (unclipped), d_mean [B][na].  log_std is theta's tail. */
int32_t metrpo_policy_actions(metrpo_ctx* ctx, const float* d_obs, const float* d_eps, int32_t B,
                              float* d_actions, float* d_mean, void* stream);
/* VecSimpleEnv.step minus the horizon/reset bookkeeping: env_helpers.py:599-603 + 609-635.
 * d_s [B][ns]; d_a [B][na] UNCLIPPED (clipped inside, :599); d_model_idx [B] int32 (step_rand:
 * fresh draw, eps_rand: cur_model_idx; ignored otherwise, may be NULL); d_noise [B][ns] for
 * model_mean_std else NULL.  Outputs: d_s_next [B][ns], d_reward [B] (= -cost_np_vec, :601),
 * d_done [B] uint8 (= is_done(s',s'), :603), optional d_next_all [K][B][ns] (all heads, :612). */
int32_t metrpo_step(metrpo_ctx* ctx, const float* d_s, const float* d_a, int32_t B, int32_t sam_mode,
                    const int32_t* d_model_idx, const float* d_noise, float* d_s_next, float* d_reward,
                    uint8_t* d_done, float* d_next_all, void* stream);

/* ---- fused rollout: VectorizedSampler.obtain_samples (samplers/vectorized_sampler.py:45-116)
 *      driving VecSimpleEnv.reset/step (env_helpers.py:585-607) and policy.get_actions ---------- */
typedef struct {
    int32_t B, T, H;             /* envs, steps to execute, max_path_length                      */
    int32_t sam_mode;            /* metrpo_sam_mode                                              */
    int32_t determ;              /* 1: actions = mean (obtain_samples(determ=True), :64-65)      */
    int32_t eval_all_heads;      /* 1: evaluate all K heads every step as the reference does
                                    (env_helpers.py:612); 0: only the selected head where the
                                    mode allows (step_rand/eps_rand/one_model)                   */
    const float* d_pool;         /* [n_pool][ns] initial-state pool standing in for the real
                                    simulator's env.reset() (env_helpers.py:552-555)             */
    int32_t n_pool;
    uint64_t seed;               /* Philox seed for every draw not supplied below                */
    uint64_t stream_offset;      /* added to the Philox counter: rank*B for sharded runs         */
    /* parity mode: explicit draws (any may be NULL -> Philox)                                   */
    const float* d_eps;          /* [T][B][na]  policy noise (np.random.normal in get_actions)   */
    const int32_t* d_model_idx;  /* [T][B]      step_rand index (env_helpers.py:619)             */
    const float* d_sel_noise;    /* [T][B][ns]  model_mean_std noise (:626)                      */
    const int32_t* d_reset_idx;  /* [T+1][B]    pool row used when env b resets AFTER step t-1
                                                (row 0 = the initial reset(), :49/:585-595)      */
    const int32_t* d_reset_model;/* [T+1][B]    cur_model_idx drawn at that reset (:593)         */
    /* outputs, time-major                                                                      */
    float* d_obs;                /* [T][B][ns]  observation BEFORE the step (:91)                */
    float* d_act;                /* [T][B][na]  UNCLIPPED action (:92)                           */
    float* d_rew;                /* [T][B]                                                       */
    float* d_mean;               /* [T][B][na]  agent_infos['mean'] (log_std = theta tail)       */
    uint8_t* d_done;             /* [T][B]      done flag returned by step (incl. ts>=H, :604)   */
    int32_t* d_tpath;            /* [T][B]      0-based step index inside its path               */
    float* d_last_obs;           /* [B][ns]     state after the last step (optional, may be NULL) */
    /* ---- continuation (ABI 2; all zero / NULL = a fresh rollout starting with vec_env.reset()) ----
     * The reference's sampling loop `while n_samples < batch_size` (samplers/vectorized_sampler.py:60) runs an unknown number
     * of steps when the env terminates early (Ant).  A caller issues the steps in chunks: each further chunk resumes from the
     * d_last_* outputs of the previous one, and chunks enqueued after the stop condition was met do nothing.                   */
    int32_t t0;                  /* global index of this call's first step: Philox counters use t0 + t, so a chunked rollout
                                    draws exactly what one long call would; output tensors and the supplied-draw tensors are
                                    indexed by the LOCAL step t (row 0 of d_reset_* is unused when resuming)                */
    const float* d_init_obs;     /* [B][ns]     resume from these states instead of resetting (with d_init_ts/_model)       */
    const int32_t* d_init_ts;    /* [B]         steps already taken in each env's current path                              */
    const int32_t* d_init_model; /* [B]         cur_model_idx of each env (eps_rand, env_helpers.py:583,593)                */
    int32_t* d_last_ts;          /* [B]         outputs matching d_init_* (optional; may alias the d_init_* arrays)         */
    int32_t* d_last_model;
    const int32_t* d_stop;       /* optional device flag: if *d_stop != 0 when the call executes, nothing is computed or
                                    written (see metrpo_sampler_progress)                                                   */
    /* ---- in-launch stop rule (ABI 4; 0 / NULL = off) ----
     * stop_batch > 0: the loop condition of obtain_samples (`while n_samples < batch_size`, vectorized_sampler.py:60,104) may be
     * applied INSIDE the call: a kernel family that runs all T steps in one launch (the persistent stream-K rollout) counts the
     * samples of completed paths per step itself, starting from *d_stop_cum (the total of the steps in front of this call:
     * metrpo_sampler_progress's d_state[0]), and stops stepping behind the first step at which the total reaches stop_batch.
     * Rows beyond that step, d_last_* included, are then UNDEFINED; metrpo_sampler_progress over this call's d_done / d_tpath
     * still finds the stop step (it scans in step order and stops there).  Kernel families that launch per step ignore both fields
     * and run all T steps.                                                                                                       */
    int64_t stop_batch;
    const double* d_stop_cum;
} metrpo_rollout_args;
int32_t metrpo_rollout(metrpo_ctx* ctx, const metrpo_rollout_args* args, void* stream);
/* Which kernel family the last metrpo_rollout of this context ran on (-1 none yet; 0 thread-per-env, 1 head-per-wave fused, 2 cooperative fused, 3 step-wise
 * GEMM, 4 resident, 5 step-wise stream-K, 6 persistent stream-K), and -- when the shape fell off the fast dispatch table (2 x 64 nets with more heads than the
 * fused kernels hold: K > 10, Ant > 8, half-cheetah > 9, or K > 5 on a shared device; hidden widths that are neither 64 nor multiples of 256; INTEGRATION.md
 * section 9 has the table and the measured cost) -- why ("" otherwise; also printed once per context on stderr unless option QUIET is set).  The reference has one code path for every shape
 * (env_helpers.py:609-635); these two calls are how a caller learns which of this library's it got. */
int32_t metrpo_last_rollout_kernel(const metrpo_ctx* ctx);
const char* metrpo_rollout_note(const metrpo_ctx* ctx);

/* Loop condition of obtain_samples (samplers/vectorized_sampler.py:60,104): n_samples counts the samples of COMPLETED paths
 * only and is tested once per time step.  For the chunk [t0, t0+T) just rolled out (d_done, d_tpath [T][B]) this adds each
 * step's completed-path samples, sum_b done[t][b] * (tpath[t][b] + 1), to d_state[0] in step order; the first step at which
 * the total reaches batch_size is written to d_state[1] (global index, -1 until then) and *d_stop is set to 1.  A chunk
 * processed while *d_stop is already 1 changes nothing.  d_state [2] float64 (caller initialises {0, -1}), d_counts [T]
 * float64 scratch, d_stop int32 (caller zeroes).  The caller keeps steps 0..d_state[1]; paths still open there are dropped
 * by metrpo_gae's valid mask, as the reference drops them. */
int32_t metrpo_sampler_progress(metrpo_ctx* ctx, const uint8_t* d_done, const int32_t* d_tpath, int32_t T, int32_t B,
                                int32_t t0, int64_t batch_size, double* d_counts, double* d_state, int32_t* d_stop,
                                void* stream);

/* build_policy_graph forward (model_based_rl.py:106-151): per model i, deterministic clipped
 * policy, model i for the whole trajectory, cost_i = sum_t gamma^t mean_b cost (Ant: masked by the
 * running dones, :134-137).  d_s0 [Bv][ns] -> d_costs [K] (float64). */
int32_t metrpo_validation_cost(metrpo_ctx* ctx, const float* d_s0, int32_t Bv, int32_t T, double gamma,
                               double* d_costs, void* stream);

/* ---- BaseSampler.process_samples (samplers/base.py:48-104) -------------------------------- */
/* Per env column reverse scan: V = [rllab] LinearFeatureBaseline.predict (zeros if d_coeffs NULL),
 * delta = r + g*V' - V, adv = discount_cumsum(delta, g*lam), ret = discount_cumsum(r, g) (:57-64),
 * restarting at every done; samples of a trailing unfinished path get valid=0 (the reference
 * drops such paths, vectorized_sampler.py:60,104).  d_coeffs [2*ns+4] float64 or NULL.
 * d_stats [3] float64 is ACCUMULATED (caller zeroes): sum(adv), sum(adv^2), count over valid. */
int32_t metrpo_gae(metrpo_ctx* ctx, const float* d_obs, const float* d_rew, const uint8_t* d_done,
                   const int32_t* d_tpath, int32_t T, int32_t B, const double* d_coeffs, double gamma,
                   double lam, float* d_adv, float* d_ret, uint8_t* d_valid, double* d_stats, void* stream);
/* Opening launch of process_samples (samplers/base.py:48-104): d_old_log_std [na] <- max(log_std of the ctx policy, log 1e-6) -- the
 * agent_infos['log_std'] every sample of the batch carries ([rllab] GaussianMLPPolicy.dist_info, min_std = 1e-6; vectorized_sampler.py:72-77) -- and
 * d_acc [n_acc] float64 <- 0 (the accumulators metrpo_gae / metrpo_baseline_gram add into).  Either pointer may be NULL.  One launch, stream-ordered:
 * call it after the rollout was enqueued and before the policy is updated. */
int32_t metrpo_process_begin(metrpo_ctx* ctx, float* d_old_log_std, double* d_acc, int64_t n_acc, void* stream);
/* [rllab] util.center_advantages (base.py:82-83): adv <- (adv-mean)/(std+1e-8) over valid samples,
 * mean/std from d_stats (after the caller all-reduced it across ranks). */
int32_t metrpo_center_advantages(metrpo_ctx* ctx, float* d_adv, const uint8_t* d_valid, int64_t N,
                                 const double* d_stats, void* stream);
/* [rllab] LinearFeatureBaseline.fit normal equations (base.py:164-167): ACCUMULATES
 * d_AtA [F][F] += F^T F and d_Aty [F] += F^T ret over valid samples, F = 2*ns+4 features
 * [o, o^2, t/100, (t/100)^2, (t/100)^3, 1], o = clip(obs,-10,10).  float64. The F x F solve stays
 * with the caller (after its all-reduce). */
int32_t metrpo_baseline_gram(metrpo_ctx* ctx, const float* d_obs, const float* d_ret, const int32_t* d_tpath,
                             const uint8_t* d_valid, int64_t N, double* d_AtA, double* d_Aty, void* stream);
/* The solve of that fit, on the device ([rllab] linear_feature_baseline.py fit: lstsq(F^T F + reg I, F^T ret), reg x 10 while the
 * solution contains a NaN, at most 5 attempts): d_coeffs [F] float64 from d_AtA / d_Aty (summed over the ranks by the caller).
 * Stream-ordered, no host round trip: the coefficients go straight into the next metrpo_gae.  The system is square and nonsingular
 * (symmetric positive definite) for reg > 0, so elimination in the natural order returns lstsq's solution (to float64 conditioning). */
int32_t metrpo_baseline_solve(metrpo_ctx* ctx, const double* d_AtA, const double* d_Aty, double reg_coeff, double* d_coeffs, void* stream);

/* ---- NPO/TRPO update (algos/npo.py:68-111, algos/trpo.py:18-20; [rllab] ConjugateGradientOptimizer) */
typedef struct {
    const float* d_obs;          /* [N][ns]                                                      */
    const float* d_act;          /* [N][na] unclipped actions                                    */
    const float* d_adv;          /* [N]                                                          */
    const float* d_old_mean;     /* [N][na] agent_infos['mean']                                  */
    const float* d_old_log_std;  /* [N][na] (stride na) or [na] broadcast (stride 0)             */
    int32_t old_log_std_stride;
    const uint8_t* d_valid;      /* [N] or NULL (= all valid)                                    */
    int64_t N;                   /* local samples                                                */
    double inv_n_global;         /* 1 / (valid samples over ALL ranks): partial results are
                                    pre-scaled so that a sum all-reduce yields the global mean   */
} metrpo_batch;

/* f_loss + f_grad of the surrogate (npo.py:75): d_out[0] = loss partial, d_out[1..P] = gradient
 * partial (float64, this rank's share).  theta = the ctx policy. */
int32_t metrpo_loss_grad(metrpo_ctx* ctx, const metrpo_batch* batch, double* d_out, void* stream);
/* [rllab] PerlmutterHvp f_Hx_plain: Hessian(mean_kl) . v WITHOUT the reg_coeff*v term (added by the
 * caller after its all-reduce).  d_v [P] float64 in, d_hv [P] float64 out (this rank's share). */
int32_t metrpo_fvp(metrpo_ctx* ctx, const metrpo_batch* batch, const double* d_v, double* d_hv, void* stream);
/* f_loss_constraint at d_theta (float32 [P], or NULL = ctx policy): d_out[0]=loss, d_out[1]=mean_kl
 * partials (float64). */
int32_t metrpo_loss_kl(metrpo_ctx* ctx, const metrpo_batch* batch, const float* d_theta, double* d_out,
                       void* stream);

/* ---- multi-GPU: one process and one ctx per GPU, the env batch B sharded over the ranks.  The only exchanges on
 * the path are sum all-reduces of small float64 vectors.  Attach an RCCL communicator to the ctx and metrpo_trpo_update issues
 * its all-reduces itself (ncclAllReduce on the caller's stream, in-place, float64); the reference has no counterpart (it is a
 * single-process TF session, utils.py:229-232).  Bootstrap: rank 0 calls metrpo_comm_get_unique_id and ships the 128 bytes to
 * the other ranks by any side channel (bench.py: a torch.distributed broadcast); then EVERY rank calls metrpo_comm_init
 * (collective, blocks until all ranks arrive).  librccl is loaded on first use. */
#define METRPO_COMM_ID_BYTES 128
int32_t metrpo_comm_get_unique_id(void* id_out /* METRPO_COMM_ID_BYTES */);
int32_t metrpo_comm_init(metrpo_ctx* ctx, const void* id /* METRPO_COMM_ID_BYTES */, int32_t world, int32_t rank);
int32_t metrpo_comm_destroy(metrpo_ctx* ctx);
/* in-place sum over the ranks of the attached communicator, stream-ordered (advantage statistics, baseline normal equations) */
int32_t metrpo_allreduce_sum_f64(metrpo_ctx* ctx, double* d_buf, int64_t count, void* stream);

/* One-shot direct all-reduce (SURVEY.md 8e "collective choice on xGMI"): the exchanges above are latency-bound, and xGMI connects
 * every pair of GPUs directly, so each rank writes its vector straight into a slot of every peer's receive region (device memory
 * exported with hipIpcGetMemHandle and mapped by the peers) and adds the slots of its own region in rank order -- one fabric
 * traversal, deterministic, bit-identical on every rank, no host in the loop.  Inside metrpo_trpo_update the exchange rides in the
 * tail of the kernel that reduces the per-block partial rows, in front of the CG vector step.  Takes precedence over an RCCL
 * communicator when both are attached.  World size <= 8 (one node); the ranks may also share one GPU (tests).
 *   every rank: metrpo_comm_ipc_export(ctx, blob) -> all-gather the blobs by any side channel (Comm.attach_engine: torch.distributed)
 *   every rank: metrpo_comm_ipc_attach(ctx, blobs[world], world, rank)
 * A rank that never arrives makes the waiting kernels give up after METRPO_XCHG_TIMEOUT_MS (default 20 000) and the next
 * metrpo_trpo_update / metrpo_comm_check returns METRPO_EHIP instead of hanging the GPU. */
#define METRPO_COMM_IPC_BLOB_BYTES 128
int32_t metrpo_comm_ipc_export(metrpo_ctx* ctx, void* blob_out /* METRPO_COMM_IPC_BLOB_BYTES */);
int32_t metrpo_comm_ipc_attach(metrpo_ctx* ctx, const void* blobs /* world x METRPO_COMM_IPC_BLOB_BYTES */, int32_t world, int32_t rank);
int32_t metrpo_comm_ipc_detach(metrpo_ctx* ctx);
/* time limit of a waiting exchange kernel from now on (attach sets METRPO_XCHG_TIMEOUT_MS or 20 000) */
int32_t metrpo_comm_set_timeout_ms(metrpo_ctx* ctx, int64_t ms);
/* 0 = single rank, 1 = RCCL communicator, 2 = one-shot direct all-reduce */
int32_t metrpo_comm_transport(const metrpo_ctx* ctx);
/* synchronises `stream`; METRPO_EHIP if an exchange timed out since the transport was attached, or if a tile hand-over of the
 * cooperative rollout kernel gave up waiting (both are sticky device-side error cells, also checked by metrpo_trpo_update) */
int32_t metrpo_comm_check(metrpo_ctx* ctx, void* stream);

/* all-reduce(sum) hook for sharded runs WITHOUT an attached communicator (e.g. a gloo group in the CPU-side tests; takes
 * precedence over the communicator when both are given): called on the host while enqueuing, must reduce `count`
 * float64 values at device pointer d_buf in place across ranks, ordered after prior work on
 * `stream` and before later work on it.  NULL = single rank. */
typedef int32_t (*metrpo_allreduce_fn)(void* user, double* d_buf, int64_t count, void* stream);

typedef struct {
    double max_kl;               /* step_size (params trpo.step_size, npo.py:22,88)              */
    int32_t cg_iters;            /* 10                                                           */
    double reg_coeff;            /* 1e-5                                                         */
    double backtrack_ratio;      /* 0.8                                                          */
    int32_t max_backtracks;      /* 15                                                           */
    int32_t accept_violation;    /* 0                                                            */
    double residual_tol;         /* 1e-10 ([rllab] krylov.cg)                                    */
    metrpo_allreduce_fn allreduce;
    void* allreduce_user;
    int32_t explicit_final_hvp;  /* 0 (default): d.(H d) of the step scale is taken from the CG recurrence (H d = g - r, the
                                    identity krylov.cg maintains); 1: evaluate f_Hx(descent_direction) once more, as [rllab]
                                    ConjugateGradientOptimizer.optimize literally does.  Same value up to float32 rounding   */
} metrpo_trpo_params;

typedef struct {                 /* host-side diagnostics of one optimize() call                 */
    double loss_before, loss, kl, beta;
    int32_t n_backtrack, accepted, cg_iters_run;
} metrpo_trpo_diag;

/* One [rllab] ConjugateGradientOptimizer.optimize call (reached from algos/npo.py:111): loss_before,
 * flat gradient, cg_iters x FVP in krylov.cg, step scale, backtracking line search; on return the
 * ctx policy holds theta_new (or theta_prev if rejected).  SYNCHRONISES the stream once per
 * line-search trial (the accept test is a host decision in the reference too).
 * Optional parity outputs (may be NULL): d_g_out, d_dir_out [P] float64 (gradient, CG direction). */
int32_t metrpo_trpo_update(metrpo_ctx* ctx, const metrpo_batch* batch, const metrpo_trpo_params* params,
                           metrpo_trpo_diag* diag, double* d_g_out, double* d_dir_out, void* stream);

/* The same optimize() call (algos/npo.py:111) in two halves, for callers that want to keep enqueuing (the next iteration's
 * obtain_samples, model_based_rl.py:1171-1180) while the line search is being decided.
 *   _begin  gradient, CG solve and the first `spec_trials` (>= 1) trials of the backtracking loop; the loop's break test and the
 *           acceptance rule that follows it are evaluated ON THE DEVICE in the tail of each trial's reduction, an accepted trial's
 *           theta becomes the ctx policy at once, later speculative trials leave without work.  Does NOT synchronise; `batch`
 *           must stay valid until _end.  Launches that follow on `stream` see theta_new if one of these trials was accepted.
 *   _end    waits for _begin's work (not for what was enqueued after it), fills `diag`, and -- only if the search did not stop
 *           within the speculative trials -- runs trials spec_trials, spec_trials+1, ... exactly as metrpo_trpo_update does.
 *           *late_out (may be NULL) = 1 if the policy changed inside _end (accepted at one of those later trials): work enqueued
 *           between the two calls used theta_prev and has to be redone by the caller; 0 otherwise.
 * Results are those of metrpo_trpo_update bit for bit.  With a host all-reduce callback, an RCCL transport or the GEMM update path
 * the accept test cannot run on the device: _begin then performs the whole update synchronously and _end returns its diagnostics. */
int32_t metrpo_trpo_update_begin(metrpo_ctx* ctx, const metrpo_batch* batch, const metrpo_trpo_params* params, int32_t spec_trials,
                                 double* d_g_out, double* d_dir_out, void* stream);
int32_t metrpo_trpo_update_end(metrpo_ctx* ctx, metrpo_trpo_diag* diag, int32_t* late_out, void* stream);

/* ---- "next" rows of the scope table: ensemble dynamics training + normaliser statistics ---- */
typedef struct {
    double lr;                   /* dynamics_opt_params.learning_rate["scratch"|"refine"] (model_based_rl.py:905-918)      */
    double beta1, beta2, eps;    /* tf.train.AdamOptimizer defaults 0.9, 0.999, 1e-8 (:162)                                */
    double reg_constant;         /* dynamics_model.regularization.constant (training.py:271-282); SGD(lr) on it (:169-177) */
    int32_t batch_size;          /* dynamics_opt_params.batch_size: rows PER MODEL                                         */
} metrpo_train_params;

/* sess.run(dynamics_adam_init) (model_based_rl.py:912-918): zero the Adam moments and the step count. */
int32_t metrpo_dyn_train_reset(metrpo_ctx* ctx, void* stream);
/* One sess.run([dynamics_opt_op, dynamics_loss]) (model_based_rl.py:961-971; loss graph :39-71; optimizers :154-183).
 * d_x [batch_size*K][ns+na] = (state, action), d_y [batch_size*K][ns] = next state; the block is consumed as the
 * reference's np.reshape(x_batch, (batch_size, -1)) + utils.get_ith_tensor: model i trains on rows i, K+i, 2K+i, ...
 * The ctx dynamics weights are updated in place.  d_loss_out [K] float64 (optional): per-model loss BEFORE the update. */
int32_t metrpo_dyn_train_step(metrpo_ctx* ctx, const float* d_x, const float* d_y, const metrpo_train_params* params,
                              double* d_loss_out, void* stream);
/* dynamics_losses on np.tile(validation, n_models) (model_based_rl.py:933-945, 977-983): every model on all n rows.
 * d_losses [K] float64 = prediction loss + regulariser per model. */
int32_t metrpo_dyn_eval_losses(metrpo_ctx* ctx, const float* d_x, const float* d_y, int64_t n, double reg_constant,
                               double* d_losses, void* stream);
/* per-model savers (model_based_rl.py:499-509, 927-930, 1002-1004; recover_weights :871-878): read all / write one model. */
int32_t metrpo_get_dynamics(metrpo_ctx* ctx, float* d_params_out, void* stream);
int32_t metrpo_set_dynamics_model(metrpo_ctx* ctx, int32_t model, const float* d_params_model, void* stream);
/* RunningMeanStd read-outs changed (after input_rms.update / output_rms.update, model_based_rl.py:834-835). COPIES. */
int32_t metrpo_set_normalizers(metrpo_ctx* ctx, const float* d_in_mean, const float* d_in_std, const float* d_diff_mean,
                               const float* d_diff_std, void* stream);
/* RunningMeanStd.update (running_mean_std.py:35-42): d_sum[dim] += sum_rows x, d_sumsq[dim] += sum_rows x^2 (float64);
 * count += n, mean and the 0.1-floored std are the caller's (running_mean_std.py:22-27). */
int32_t metrpo_rms_accumulate(metrpo_ctx* ctx, const float* d_x, int64_t n, int32_t dim, double* d_sum, double* d_sumsq,
                              void* stream);

/* ---- "next" row of the scope table: BPTT policy update ('bptt' branch, model_based_rl.py:1181-1187) ---- */
/* Gradient of training_policy_cost = mean_i policy_costs[i] (model_based_rl.py:365) w.r.t. the policy parameters through the
 * unrolled graph of build_policy_graph (:106-151): x <- d_init [B][ns]; T times u = clip(policy_mean(x)), x' = model_i(x, u),
 * cost_i += gamma^t * cost_tf(x, u, x') (Ant: masked by the running `dones`).  `stochastic` is 0 (the 'bptt' branch).
 * d_costs [K] float64 (optional) = policy_costs[i] (same values as metrpo_validation_cost); d_grad [P] float64, log_std slots 0. */
int32_t metrpo_bptt_grad(metrpo_ctx* ctx, const float* d_init, int32_t B, int32_t T, double gamma, double* d_costs,
                         double* d_grad, void* stream);
/* sess.run(policy_adam_init) (model_based_rl.py:202-204): zero the policy optimizer's moments and step count. */
int32_t metrpo_policy_adam_reset(metrpo_ctx* ctx, void* stream);
/* policy_opt_op (get_policy_optimizer, model_based_rl.py:186-195): tf.clip_by_norm(grad, clip_val) per VARIABLE (W_l, b_l;
 * utils.py:262-276; clip_val <= 0: none) then tf.train.AdamOptimizer(lr).apply_gradients on the ctx policy parameters. */
int32_t metrpo_policy_adam_step(metrpo_ctx* ctx, const double* d_grad, double lr, double beta1, double beta2, double eps,
                                double clip_val, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* METRPO_H */
