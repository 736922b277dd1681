"""The fp32 tolerance the HIP path guarantees against the float64 oracle / the reference's float64 outputs -- ONE table, the one printed in
DESIGN.md section 5 ("Stated fp32 tolerances").  Every tests/test_gpu_*.py takes its bounds from here; a row may only be loosened here and in
DESIGN.md together, with the reason.  u = 2^-24 = 5.96e-8 is the fp32 unit round-off; "observed" is the worst case over the whole GPU suite
(a run of the suite with METRPO_TOL_REPORT=<file>, see conftest.py; round 4's is summarised in profiles/r04_tolerance_report.txt).

Inputs common to both sides are rounded to fp32 first (the device's storage type), so the bounds cover arithmetic only.  assert_allclose
semantics: |got - ref| <= atol + rtol * |ref| element by element; rel-L2 = ||got - ref||_2 / ||ref||_2.
"""

# --- one step of the model, inputs given (teacher-forced): row 1 -------------------------------------------------------------------------------------
# next state of every ensemble head, the selected next state, the policy mean, hidden widths <= 64.  Sums of <= 64 fp32 products by fmaf in a fixed
# order (error <= sqrt(n) u |w||h| ~ 5e-7 on O(1) pre-activations), tanh with <= 2 ulp, de-normalisation x std + mean.  Observed 6.6e-7 absolute.
# rtol 1e-5, atol 1e-6.
STEP = dict(rtol=1e-5, atol=1e-6)
# the analytic reward of the same step (env_helpers cost_np_vec): a handful of fp32 operations on the next state; same figure as the state
REWARD = dict(rtol=1e-5, atol=1e-6)
# the sampled action a = mean + exp(log_std) * eps (fp32 expf, one fma): row 1 with the exponential's 2 ulp on an O(1) product
ACTION = dict(rtol=1e-5, atol=2e-6)

# --- one step, wide nets (hidden 128 ... 1024; the MFMA paths): row 2 ------------------------------------------------------------------------------------
# sums 2x ... 16x longer, formed in the MFMA's own order (4 partial chains of k mod 16, chunks of 32): sqrt(n) growth -> 4x row 1's absolute bound,
# 2x its relative one.  Observed 8.6e-7 absolute.  SURVEY 8d has no separate figure (its row 1 figure is for the 2x64 nets).
WIDE = dict(rtol=2e-5, atol=5e-6)

# --- free-running rollouts: rows 3, 4 --------------------------------------------------------------------------------------------------------------
# device and oracle each feed their OWN states back.  t <= 10: SURVEY 8d keeps the single-step figure; the learned dynamics' Jacobian has norm ~1 on the
# fixtures, errors add up linearly: observed 7.5e-8 .. 4.8e-7.
FREE_RUN = dict(rtol=1e-5, atol=1e-6)
# beyond t = 10 (the reference's sampler runs: whole episodes of 100+ steps) SURVEY 8d asks for statistical agreement only; the suite still asserts
# element by element, at 10x the single-step figure (observed 3.2e-7 absolute)
LONG_RUN = dict(rtol=1e-4, atol=1e-5)
# two DEVICE kernel families free-running on the same draws (resident / step-wise / stream-K / generic): different summation orders of the
# same fp32 sums, a few steps: row 2's bound per step x steps <= 10
CROSS_KERNEL = dict(rtol=1e-4, atol=2e-5)

# --- per-model validation cost (discounted sum of T costs, mean over the batch): row 5 -----------------------------------------------------------------
# free-running T <= 15 steps, then a float64 batch mean: the per-step figure times the number of steps.  Observed 1.1e-6 on costs of O(10).
VALIDATION_COST = dict(rtol=2e-5, atol=2e-6)

# --- process_samples: rows 6, 7 -----------------------------------------------------------------------------------------------------------------
# returns: suffix sums of <= 1000 fp32 rewards by a wavefront scan (log-depth tree: error ~ log2(T) u |sum|); observed 1.3e-7
RETURNS = dict(rtol=1e-5, atol=1e-5)
# advantages before centring (GAE over deltas that hold the baseline prediction, an fp32 dot product of 2 ns + 4 features) and after centring
# ((a - mean) / (std + 1e-8), statistics summed in float64).  SURVEY 8d: atol 1e-4 after centring; observed 1.2e-6
ADVANTAGE = dict(rtol=1e-5, atol=5e-5)
ADVANTAGE_CENTRED = dict(rtol=1e-4, atol=1e-4)
# the refitted linear baseline: normal equations from fp32 products summed in float64, solved in float64; the fit's conditioning (kappa ~ 1e3 on the
# fixtures) multiplies the 2e-6 relative error of the moments.  Bound on predictions, relative to max |prediction|.  Observed 6.7e-5.
BASELINE_FIT = 1e-3
NORMAL_EQ = 2e-6            # entries of A^T A / A^T y relative to sqrt(G_ii G_jj): fp32 products, float64 sums

# --- the TRPO update: rows 8 .. 12 ----------------------------------------------------------------------------------------------------------------
LOSS_RTOL = 1e-5            # surrogate loss relative to max(1, |loss|): per-sample fp32, float64 block reduction (rtol 1e-5)
KL_ATOL = 1e-7              # mean KL at theta_old (atol 1e-7); at a trial theta also 1e-4 relative (KL is a difference of O(1) terms)
KL_RTOL = 1e-4
GRAD_REL_L2 = 1e-5          # gradient g (1e-5): per-sample back-propagation in fp32, sums over N in float64.  Observed 1.1e-6
FVP_REL_L2 = 1e-5           # Fisher-vector product (1e-4; held 10x tighter): tangent + back-propagation in fp32, float64 sums over N.  Observed 1.2e-7
CG_COS = 0.9999             # step direction d after 10 CG iterations (cosine >= 0.9999, rel-L2 <= 1e-3: ten FVPs amplify rounding by the
CG_REL_L2 = 3e-4            # Fisher matrix's condition number, ~1e3 on the fixtures).  Observed 7.4e-5; held at 3e-4
STEP_SCALE_RTOL = 1e-3      # beta = sqrt(2 delta / d.Hd): inherits d's figure
THETA_STEP_REL_L2 = 5e-4    # theta_new - theta_old = -ratio * beta * d: d's and beta's figures added.  Observed 8.3e-5
POST_UPDATE_RTOL = 5e-3     # loss / KL evaluated at the (slightly different) accepted theta

# --- several ranks against one (sharded sums exchanged in float64): row 13 ------------------------------------------------------------------------------
MULTI_RANK_GRAD = 2e-6      # x max |g|: the same float64 partial sums added in another order, over fp32 per-sample terms
MULTI_RANK_THETA = 2e-3     # x |step|: as THETA_STEP_REL_L2


def wide_or_step(hidden):
    """Row 1 for the 2x64-class nets, row 2 from 128 hidden units on."""
    return STEP if max(hidden) <= 64 else WIDE


# --- widened rows: BPTT policy update and ensemble training -----------------------------------------------------------------------------
BPTT_COST = VALIDATION_COST           # the same discounted cost sums, with the tape kept
BPTT_GRAD_REL_L2 = 3e-4               # gradient through T <= 30 chained fp32 Jacobians (dynamics o policy), per-variable: the one-step 1e-5 times T
DYN_LOSS = dict(rtol=3e-5, atol=1e-7)     # per-model training loss: mean over batch x ns squared errors in fp32 tiles, float64 across tiles
DYN_EVAL_LOSS = dict(rtol=5e-4, atol=1e-6)   # validation loss after training steps: weights already differ by Adam's sign-like early steps (below)
