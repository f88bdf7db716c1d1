// Software-pipelined fused posterior kernel, Matérn-5/2, 12 k-steps in the distance GEMM (d <= 46).
#include "bbh_fused.h"

void bbh_fused_launch_kd12(bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a) {
  if (has_tbl) {
    BBH_FUSED_ALLOW_LDS((bbh_fused_posterior_kernel<true, BBH_KERNEL_MATERN52, 12>), lds);
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<true, BBH_KERNEL_MATERN52, 12>), grid, block, lds, s, a);
  } else {
    BBH_FUSED_ALLOW_LDS((bbh_fused_posterior_kernel<false, BBH_KERNEL_MATERN52, 12>), lds);
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<false, BBH_KERNEL_MATERN52, 12>), grid, block, lds, s, a);
  }
}
