// Cooperative rollout kernel (rollout_coop_kernel.h) for K = 8 heads at 2 x 64: one-workgroup-per-CU instantiations of the five envs (table row of rollout_coop.hip)
#include "rollout_coop_kernel.h"
COOP_WIDE_TABLE(kCoopK8, 8)
