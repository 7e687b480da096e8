// qmhip_lq.hip — the four instances of the K1b body (qm_lq_kernel, qm_lq_m18_kernel, qm_lq_ipm_kernel, qm_lq_dbg_kernel: k_lq.h) in a translation unit of their own.
// Why: LLVM's scheduling strategy is a per-module choice.  `max-ilp` takes 1.5 % off K1b (168 registers, 2.4 waves per SIMD: longer independent chains between its LDS
// hand-overs) and ADDS 5 % to the lone-wave kernels (WBC 388 -> 402 registers, K1a + 13 %), measured on one box (profiles/r06_ab_sched_strategy.log).  The kernels are launched
// from qmhip.hip through their host stubs (declared there by k_lq.h under QM_LQ_KERNELS_EXTERN); no relocatable device code is involved.
#include <hip/hip_runtime.h>
#define QM_LQ_ONLY_K1B 1
#include "../kernels/k_lq.h"
