// Cooperative form of the fused posterior kernel (bbh_coop.h): instantiations with 8, 12 and 16 k-steps in the distance
// GEMM (d <= 62).
#include "bbh_coop.h"

bool bbh_coop_launch_b(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  BBH_COOP_DISPATCH_KD(8, 1)
  BBH_COOP_DISPATCH_KD(12, 1)
  BBH_COOP_DISPATCH_KD(16, 1)
  return false;
}
