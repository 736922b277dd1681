// placeholder until the MFMA fast path lands (replaced below in this round)
#include "metrpo_internal.h"
int mfma_select_config(metrpo_ctx*) { return -1; }
int mfma_prepare_dynamics(metrpo_ctx*, hipStream_t) { return METRPO_OK; }
int mfma_prepare_policy(metrpo_ctx*, hipStream_t) { return METRPO_OK; }
int launch_rollout_mfma(metrpo_ctx*, const metrpo_rollout_args*, hipStream_t) { return METRPO_EUNSUPPORTED; }
