// Register-resident small-model form of the fused posterior kernel (bbh_small.h): instantiations for 32 < n <= 64 (NB = 3, 4).
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

bool bbh_small_launch_b(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a) {
  if (NB == 3) { BBH_SMALL_KD(3) }
  if (NB == 4) { BBH_SMALL_KD(4) }
  return false;
}
