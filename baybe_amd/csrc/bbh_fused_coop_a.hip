// Cooperative form of the fused posterior kernel (bbh_coop.h): instantiations with 2, 4 and 6 k-steps in the distance
// GEMM (d <= 22), and the dispatcher over both translation units.
#include "bbh_coop.h"

bool bbh_coop_launch_a(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  BBH_COOP_DISPATCH_KD(2, 1)
  BBH_COOP_DISPATCH_KD(4, 1)
  BBH_COOP_DISPATCH_KD(6, 1)
  return false;
}

bool bbh_coop_launch(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a) {
  // BBH_COOP_SMALL=0 (read per launch: A/B runs) keeps the eight-round instantiation for small models too
  const char* env_small = getenv("BBH_COOP_SMALL");
  const bool small_ok = !(env_small && env_small[0] == '0');
  if (grid.x != 0 && a.g0 >= 4 && small_ok && bbh_coop_launch_small(kd, kind, has_tbl, grid, lds, s, a)) return true;
  return kd <= 6 ? bbh_coop_launch_a(kd, kind, has_tbl, grid, lds, s, a) : bbh_coop_launch_b(kd, kind, has_tbl, grid, lds, s, a);
}
