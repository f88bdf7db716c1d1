// Scrambled Sobol points exactly as torch.quasirandom.SobolEngine produces them (the engine botorch's SobolQMCNormalSampler
// draws its base samples from; sampler built at baybe/acquisition/_builder.py:195-334 through BoTorch's acquisition
// constructors), without the engine's per-dimension tensor loops: host code, integer arithmetic only, so "exactly" is bitwise.
//   * bbh_sobol_scramble: Owen-type linear scrambling of the direction numbers.  torch: a random lower-triangular 0/1 matrix
//     per dimension (unit diagonal), row p packed into the integer  dots[p] = sum_k ltm[p][k] 2^(MAXBIT - 1 - k)  ("cdot_pow2"),
//     and  v'_j = sum_p parity(popcount(dots[p] & v_j)) 2^(MAXBIT - 1 - p)  for each direction number v_j.  The engine spends 4 ms
//     per 768 dimensions in that loop nest (bit by bit); here it is one popcount per (j, p).
//   * bbh_sobol_draw: point 0 is the shift (rounded to single precision: the engine stores its first point in torch's default
//     dtype before converting to the requested one), point i = point i - 1 XOR v[rightmost zero bit of i - 1] (Gray-code order), scaled
//     by 2^-MAXBIT.
// The random bits (shift, ltm) still come from torch's generator, and the normal transform from torch.erfinv
// (baybe_amd/engine.py::sobol_normal_base_samples), which checks this path against the engine itself on first use.
#include <stdint.h>

#include <vector>

#include "../../include/baybe_hip.h"

#define BBH_SOBOL_MAXBIT 30

extern "C" int bbh_sobol_scramble(int64_t* state, const int64_t* ltm, int64_t dim) {
  if (!state || !ltm || dim < 1) return -1;
  for (int64_t d = 0; d < dim; d++) {
    const int64_t* m = ltm + d * BBH_SOBOL_MAXBIT * BBH_SOBOL_MAXBIT;
    uint64_t dots[BBH_SOBOL_MAXBIT];
    for (int p = 0; p < BBH_SOBOL_MAXBIT; p++) {
      uint64_t acc = 0;
      for (int k = 0; k < BBH_SOBOL_MAXBIT; k++) {
        const int64_t bit = (k == p) ? 1 : (k < p ? (m[p * BBH_SOBOL_MAXBIT + k] & 1) : 0);  // tril, unit diagonal
        acc |= (uint64_t)bit << (BBH_SOBOL_MAXBIT - 1 - k);
      }
      dots[p] = acc;
    }
    int64_t* v = state + d * BBH_SOBOL_MAXBIT;
    for (int j = 0; j < BBH_SOBOL_MAXBIT; j++) {
      const uint64_t vdj = (uint64_t)v[j];
      uint64_t t2 = 0;
      for (int p = 0; p < BBH_SOBOL_MAXBIT; p++)
        t2 |= (uint64_t)(__builtin_popcountll(dots[p] & vdj) & 1) << (BBH_SOBOL_MAXBIT - 1 - p);
      v[j] = (int64_t)t2;
    }
  }
  return 0;
}

extern "C" int bbh_sobol_draw(const int64_t* state, const int64_t* shift, int64_t n, int64_t dim, double* out) {
  if (!state || !shift || !out || n < 1 || dim < 1) return -1;
  const double scale = 1.0 / (double)(1ll << BBH_SOBOL_MAXBIT);
  // point-major walk over contiguous rows: vt[l][j] = v_l of dimension j, q[j] = the running integer point
  std::vector<int64_t> vt((size_t)BBH_SOBOL_MAXBIT * dim), q(shift, shift + dim);
  for (int64_t j = 0; j < dim; j++)
    for (int l = 0; l < BBH_SOBOL_MAXBIT; l++) vt[(size_t)l * dim + j] = state[j * BBH_SOBOL_MAXBIT + l];
  // the engine keeps its first point as `quasi / 2**MAXBIT` in torch's DEFAULT dtype (float32) before converting
  for (int64_t j = 0; j < dim; j++) out[j] = (double)(float)q[j] * scale;
  for (int64_t i = 1; i < n; i++) {
    const int l = __builtin_ctzll(~(uint64_t)(i - 1));  // rightmost zero bit of i - 1
    const int64_t* v = vt.data() + (size_t)l * dim;
    double* o = out + i * dim;
    for (int64_t j = 0; j < dim; j++) {
      q[j] ^= v[j];
      o[j] = (double)q[j] * scale;
    }
  }
  return 0;
}
