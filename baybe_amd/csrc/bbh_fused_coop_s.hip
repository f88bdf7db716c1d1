// Cooperative form of the fused posterior kernel (bbh_coop.h), instantiations for small models: n <= 256, i.e. only the last
// four rounds exist (GMIN = 4: half the accumulators, four workgroups per CU), and n <= 128 (GMIN = 6: five).  Matérn-5/2 with and without the task /
// outputscale table, 2 - 8 k-steps of the distance GEMM (d <= 30).
#include "bbh_coop.h"

#define BBH_COOP_SMALL_KD(KDV)                                                                              \
  if (kd == KDV) {                                                                                          \
    if (grid.x == 0) return true;                                                                           \
    if (a.g0 >= 6 && has_tbl) /* n <= 128 */                                                                \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 1, 1, 6>), grid, dim3(256), lds, s, a);             \
    else if (a.g0 >= 6)                                                                                     \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 0, 1, 6>), grid, dim3(256), lds, s, a);             \
    else if (has_tbl)                                                                                       \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 1, 1, 4>), grid, dim3(256), lds, s, a);             \
    else                                                                                                    \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 0, 1, 4>), grid, dim3(256), lds, s, a);             \
    return true;                                                                                            \
  }

bool bbh_coop_launch_small(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  if (kind != BBH_KERNEL_MATERN52) return false;
  BBH_COOP_SMALL_KD(2)
  BBH_COOP_SMALL_KD(4)
  BBH_COOP_SMALL_KD(6)
  BBH_COOP_SMALL_KD(8)
  return false;
}
