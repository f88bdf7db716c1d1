// Register-resident small-model form of the fused posterior kernel (bbh_small.h): instantiations for 64 < n <= 128 (NB = 5 ... 8,
// operand fragments in LDS, workgroups of eight waves), Matern-5/2 and RBF without / with the task / outputscale table.
#include "bbh_small.h"

#define BBH_SMALL_KVF(KDV, NBV)                                                              \
  {                                                                                          \
    const int kvf = (kind == BBH_KERNEL_RBF ? 2 : 0) | (has_tbl ? 1 : 0);                    \
    if (kvf == 0) small_go<KDV, 0, NBV>(tiles, num_cu, s, a);                                \
    else if (kvf == 1) small_go<KDV, 1, NBV>(tiles, num_cu, s, a);                           \
    else if (kvf == 2) small_go<KDV, 2, NBV>(tiles, num_cu, s, a);                           \
    else return false;                                                                       \
    return true;                                                                             \
  }
#define BBH_SMALL_KD(NBV)            \
  if (kd == 2) BBH_SMALL_KVF(2, NBV) \
  if (kd == 4) BBH_SMALL_KVF(4, NBV) \
  if (kd == 6) BBH_SMALL_KVF(6, NBV) \
  if (kd == 8) BBH_SMALL_KVF(8, NBV)

bool bbh_small_launch_c(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a) {
  if (kind == BBH_KERNEL_MATERN32 || (kind == BBH_KERNEL_RBF && has_tbl)) return false;
  if (tiles == 0) return true;
  if (NB == 5) { BBH_SMALL_KD(5) }
  if (NB == 6) { BBH_SMALL_KD(6) }
  if (NB == 7) { BBH_SMALL_KD(7) }
  if (NB == 8) { BBH_SMALL_KD(8) }
  return false;
}
