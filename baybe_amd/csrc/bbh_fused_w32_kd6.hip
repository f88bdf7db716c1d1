// One-wave-per-SIMD form of the software-pipelined fused posterior kernel (WMAX = 32: windows of 32 column
// blocks, 256 accumulator registers), 6 k-steps in the distance GEMM (d <= 22), Matérn-5/2 without table.
#define BBH_CANDREG 1
#define BBH_DIST_ASM 1
#define BBH_MEAN_VALU_ONLY 1
#ifndef BBH_W32_REMAINDERS
#define BBH_W32_REMAINDERS 0
#endif
#include "bbh_fused.h"

bool bbh_fused_launch_w32(int kd, int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a) {
  if (kd != 6 || kind != BBH_KERNEL_MATERN52 || has_tbl) return false;
  if (grid.x == 0) return true;  // query: is there an instantiation for this model?
  BBH_FUSED_ALLOW_LDS((bbh_fused_posterior_kernel<false, BBH_KERNEL_MATERN52, 6, 32>), lds);
  hipLaunchKernelGGL((bbh_fused_posterior_kernel<false, BBH_KERNEL_MATERN52, 6, 32>), grid, block, lds, s, a);
  return true;
}
