// Two-sweep cooperative form of the fused posterior kernel (bbh_coop2.h, 512 < n <= 1024): instantiations with 2, 4 and 6
// k-steps in the distance GEMM (d <= 22), and the dispatcher over both translation units.
#include "bbh_coop2.h"

bool bbh_coop2_launch_a(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  BBH_COOP2_DISPATCH_KD(2)
  BBH_COOP2_DISPATCH_KD(4)
  BBH_COOP2_DISPATCH_KD(6)
  return false;
}

bool bbh_coop2_launch(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  return kd <= 6 ? bbh_coop2_launch_a(kd, kind, has_tbl, grid, lds, s, a) : bbh_coop2_launch_b(kd, kind, has_tbl, grid, lds, s, a);
}
