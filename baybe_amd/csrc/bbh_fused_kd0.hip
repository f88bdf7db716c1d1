// Plain form of the fused posterior kernel: runtime k-steps, libm kernel function, every kernel kind.
#include "bbh_fused.h"

void bbh_fused_launch_kd0(bool has_tbl, bool m52, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a) {
  if (has_tbl && m52)
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<true, BBH_KERNEL_MATERN52, 0>), grid, block, lds, s, a);
  else if (has_tbl)
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<true, -1, 0>), grid, block, lds, s, a);
  else if (m52)
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<false, BBH_KERNEL_MATERN52, 0>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((bbh_fused_posterior_kernel<false, -1, 0>), grid, block, lds, s, a);
}
