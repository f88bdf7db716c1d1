// Cooperative form with the generic kernel-value production (bbh_coopg.h): composite kernels and single kernels without a
// software-pipelined instantiation, n <= 512, d <= 30.
#include "bbh_coopg.h"

#define BBH_COOPG_CASE(KDV, FV)                                                                  \
  if (kd == KDV && F == FV) {                                                                    \
    if (grid.x == 0) return true;                                                                \
    hipLaunchKernelGGL((bbh_coopg_posterior_kernel<KDV, FV>), grid, dim3(256), lds, s, a);       \
    return true;                                                                                 \
  }

bool bbh_coopg_launch(int kd, int F, dim3 grid, size_t lds, hipStream_t s, const CoopGArgs& a) {
  BBH_COOPG_CASE(2, 1) BBH_COOPG_CASE(2, 2) BBH_COOPG_CASE(2, 3) BBH_COOPG_CASE(2, 4)
  BBH_COOPG_CASE(4, 1) BBH_COOPG_CASE(4, 2) BBH_COOPG_CASE(4, 3) BBH_COOPG_CASE(4, 4)
  BBH_COOPG_CASE(6, 1) BBH_COOPG_CASE(6, 2) BBH_COOPG_CASE(6, 3) BBH_COOPG_CASE(6, 4)
  BBH_COOPG_CASE(8, 1) BBH_COOPG_CASE(8, 2) BBH_COOPG_CASE(8, 3) BBH_COOPG_CASE(8, 4)
  return false;
}

#define BBH_COOPG_CROSS_CASE(KDV, FV)                                                         \
  if (kd == KDV && F == FV) {                                                                 \
    if (grid.x == 0) return true;                                                             \
    hipLaunchKernelGGL((bbh_coopg_cross_kernel<KDV, FV>), grid, dim3(256), 0, s, a);          \
    return true;                                                                              \
  }

// mean / cross-covariance pass (64 candidates per workgroup, one wave per 16)
bool bbh_coopg_cross_launch(int kd, int F, dim3 grid, hipStream_t s, const CoopGArgs& a) {
  BBH_COOPG_CROSS_CASE(2, 1) BBH_COOPG_CROSS_CASE(2, 2) BBH_COOPG_CROSS_CASE(2, 3) BBH_COOPG_CROSS_CASE(2, 4)
  BBH_COOPG_CROSS_CASE(4, 1) BBH_COOPG_CROSS_CASE(4, 2) BBH_COOPG_CROSS_CASE(4, 3) BBH_COOPG_CROSS_CASE(4, 4)
  BBH_COOPG_CROSS_CASE(6, 1) BBH_COOPG_CROSS_CASE(6, 2) BBH_COOPG_CROSS_CASE(6, 3) BBH_COOPG_CROSS_CASE(6, 4)
  BBH_COOPG_CROSS_CASE(8, 1) BBH_COOPG_CROSS_CASE(8, 2) BBH_COOPG_CROSS_CASE(8, 3) BBH_COOPG_CROSS_CASE(8, 4)
  return false;
}
