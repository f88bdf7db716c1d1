// Two-sweep cooperative form of the fused posterior kernel (bbh_coop2.h, 512 < n <= 1024): instantiations with 8, 12 and
// 16 k-steps in the distance GEMM (d <= 62).
#include "bbh_coop2.h"

bool bbh_coop2_launch_b(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  BBH_COOP2_DISPATCH_KD(8)
  BBH_COOP2_DISPATCH_KD(12)
  BBH_COOP2_DISPATCH_KD(16)
  return false;
}
