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

// ---- content key of host buffers (the resident candidate matrix is keyed on the CONTENT of the comp rep: every byte, every call) ----
// python-xxhash keeps the GIL, so "8 threads" hashed a 160 MB comp rep at one core's 32 GB/s: 5 ms of a 24 ms recommend().  Here:
// a multiply-fold hash (128-bit products folded to 64 bits, four independent lanes per 64-byte stripe - the construction of
// wyhash / rapidhash; change detection, not cryptography) over 4 MB pieces on std::threads, the pieces' digests folded in order.
#include <string.h>

#include <thread>
#include <vector>

namespace {
inline uint64_t bh_mum(uint64_t a, uint64_t b) {
  const __uint128_t r = (__uint128_t)a * b;
  return (uint64_t)r ^ (uint64_t)(r >> 64);
}
inline uint64_t bh_rd(const unsigned char* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
uint64_t bh_piece(const unsigned char* p, size_t n, uint64_t seed) {
  const uint64_t s0 = 0x2d358dccaa6c78a5ull, s1 = 0x8bb84b93962eacc9ull, s2 = 0x4b33a62ed433d4a3ull, s3 = 0x4d5a2da51de1aa47ull;
  uint64_t a0 = seed ^ s0, a1 = seed ^ s1, a2 = seed ^ s2, a3 = seed ^ s3;
  size_t i = 0;
  for (; i + 64 <= n; i += 64) {
    a0 = bh_mum(bh_rd(p + i) ^ s0, bh_rd(p + i + 8) ^ a0);
    a1 = bh_mum(bh_rd(p + i + 16) ^ s1, bh_rd(p + i + 24) ^ a1);
    a2 = bh_mum(bh_rd(p + i + 32) ^ s2, bh_rd(p + i + 40) ^ a2);
    a3 = bh_mum(bh_rd(p + i + 48) ^ s3, bh_rd(p + i + 56) ^ a3);
  }
  uint64_t tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  memcpy(tail, p + i, n - i);
  for (int k = 0; k < 8; k += 2) a0 = bh_mum(tail[k] ^ s1, tail[k + 1] ^ a0 ^ (uint64_t)(n - i));
  return bh_mum(a0 ^ a2 ^ (uint64_t)n, a1 ^ a3 ^ s2);
}
}  // namespace

extern "C" uint64_t bbh_content_key(const void* const* bufs, const int64_t* lens, int32_t nbuf, int32_t threads) {
  if (!bufs || !lens || nbuf < 1) return 0;
  constexpr size_t PIECE = 4u << 20;
  struct Task {
    const unsigned char* p;
    size_t n;
  };
  std::vector<Task> tasks;
  for (int b = 0; b < nbuf; b++) {
    const unsigned char* p = (const unsigned char*)bufs[b];
    size_t n = (size_t)(lens[b] > 0 ? lens[b] : 0);
    if (n == 0) tasks.push_back({p, 0});
    for (size_t o = 0; o < n; o += PIECE) tasks.push_back({p + o, n - o < PIECE ? n - o : PIECE});
  }
  std::vector<uint64_t> dig(tasks.size());
  int nt = threads < 1 ? 1 : threads;
  if ((size_t)nt > tasks.size()) nt = (int)tasks.size();
  auto work = [&](int t) {
    for (size_t k = (size_t)t; k < tasks.size(); k += (size_t)nt) dig[k] = bh_piece(tasks[k].p, tasks[k].n, 0x9e3779b97f4a7c15ull * (k + 1));
  };
  if (nt <= 1 || tasks.size() < 2) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  uint64_t acc = 0x243f6a8885a308d3ull ^ (uint64_t)tasks.size();
  for (uint64_t d : dig) acc = bh_mum(acc ^ d, 0x9e3779b97f4a7c15ull ^ d);
  return acc;
}
