// Cooperative form of the fused posterior kernel (bbh_coop.h), 6 k-steps in the distance GEMM (d <= 22), Matérn-5/2
// without task / outputscale table.
#include "bbh_coop.h"

bool bbh_coop_launch(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  if (kd != 6 || kind != BBH_KERNEL_MATERN52 || has_tbl) return false;
  if (grid.x == 0) return true;
  hipLaunchKernelGGL((bbh_coop_posterior_kernel<6, 0>), grid, dim3(256), lds, s, a);
  return true;
}
