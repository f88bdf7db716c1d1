// Register-resident small-model form of the fused posterior kernel (bbh_small.h): instantiations for n <= 32 (NB = 1, 2).
#include "bbh_small.h"

#define BBH_SMALL_KVF(KDV, NBV)                                                              \
  {                                                                                          \
    const int kvf = (kind == BBH_KERNEL_RBF ? 2 : kind == BBH_KERNEL_MATERN32 ? 4 : 0) | (has_tbl ? 1 : 0); \
    if (kvf == 0) small_go<KDV, 0, NBV>(tiles, num_cu, s, a);                                \
    else if (kvf == 1) small_go<KDV, 1, NBV>(tiles, num_cu, s, a);                           \
    else if (kvf == 2) small_go<KDV, 2, NBV>(tiles, num_cu, s, a);                           \
    else if (kvf == 3) small_go<KDV, 3, NBV>(tiles, num_cu, s, a);                           \
    else if (kvf == 4) small_go<KDV, 4, NBV>(tiles, num_cu, s, a);                           \
    else return false;                                                                       \
    return true;                                                                             \
  }
#define BBH_SMALL_KD(NBV)            \
  if (kd == 2) BBH_SMALL_KVF(2, NBV) \
  if (kd == 4) BBH_SMALL_KVF(4, NBV) \
  if (kd == 6) BBH_SMALL_KVF(6, NBV) \
  if (kd == 8) BBH_SMALL_KVF(8, NBV)

bool bbh_small_launch_b(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a);
bool bbh_small_launch_c(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a);

bool bbh_small_launch(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a) {
  if (kind != BBH_KERNEL_MATERN52 && kind != BBH_KERNEL_MATERN32 && kind != BBH_KERNEL_RBF) return false;
  if (kind == BBH_KERNEL_MATERN32 && has_tbl) return false;  // (no such instantiation in any form)
  if (kd != 2 && kd != 4 && kd != 6 && kd != 8) return false;
  if (NB < 1 || NB > 8) return false;
  if (NB >= 5) return bbh_small_launch_c(kd, kind, has_tbl, NB, tiles, num_cu, s, a);
  if (tiles == 0) return true;
  if (NB >= 3) return bbh_small_launch_b(kd, kind, has_tbl, NB, tiles, num_cu, s, a);
  if (NB == 1) { BBH_SMALL_KD(1) }
  if (NB == 2) { BBH_SMALL_KD(2) }
  return false;
}
