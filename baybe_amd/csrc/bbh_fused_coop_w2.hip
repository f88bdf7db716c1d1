// Cooperative form of the fused posterior kernel with two candidate tiles per workgroup (bbh_coop.h, NT = 2).
#include "bbh_coop.h"

bool bbh_coop_launch_w2(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  if (kd != 6 || kind != BBH_KERNEL_MATERN52 || has_tbl) return false;
  if (grid.x == 0) return true;
  hipLaunchKernelGGL((bbh_coop_posterior_kernel<6, 0, 2>), grid, dim3(256), lds, s, a);
  return true;
}
