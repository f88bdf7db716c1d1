// Software-pipelined fused posterior kernel, Matérn-5/2, Matérn-3/2 and RBF, 16 k-steps in the distance GEMM (d <= 62).
#include "bbh_fused.h"

#define BBH_LAUNCH(TBL, KND)                                                                      \
  do {                                                                                            \
    BBH_FUSED_ALLOW_LDS((bbh_fused_posterior_kernel<TBL, KND, 16>), lds);                          \
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<TBL, KND, 16>), grid, block, lds, s, a);        \
  } while (0)

void bbh_fused_launch_kd16(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a) {
  if (kind == BBH_KERNEL_RBF)  // RBF / Matérn-3/2 with a task / outputscale table take the plain form (caller)
    BBH_LAUNCH(false, BBH_KERNEL_RBF);
  else if (kind == BBH_KERNEL_MATERN32)
    BBH_LAUNCH(false, BBH_KERNEL_MATERN32);
  else if (has_tbl)
    BBH_LAUNCH(true, BBH_KERNEL_MATERN52);
  else
    BBH_LAUNCH(false, BBH_KERNEL_MATERN52);
}
