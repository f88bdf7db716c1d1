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
#include <string.h>

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

// ---- the whole base-sample draw in native code (round 6) ---------------------------------------------------------------------------
// SobolQMCNormalSampler's base samples are  z = sqrt(2) erfinv(2 v - 1),  v = 0.5 + (1 - eps)(u - 0.5),  u = the engine's scrambled
// points.  Until round 5 the random bits came from torch's generator and the transform from torch.erfinv on the host: 19.9 ms of a
// qLogNEHVI pruning call (2048 x 768 values; profiles/r05_nehvi_setup_device.log), 0.3 ms per greedy step.  Here:
//   * the scrambling bits: torch's CPU generator is the standard MT19937 (at::mt19937, seeded with the low 32 bits of the seed), and
//     torch.randint(2, shape, generator=g) takes one 32-bit output per element, in element order, modulo 2 - so std::mt19937 yields the
//     engine's shift bits [dim, 30] and lower-triangular matrices [dim, 30, 30] bit for bit (checked against torch at first use,
//     baybe_amd/engine.py::_native_sobol_usable);
//   * the inverse error function: rational starting values + Newton steps on erf / erfc (bh_erfinv below) with the C library's
//     functions on the host and the device library's on the GPU - agreement with torch.erfinv (MKL) and between host and device is to
//     a few units in the last place, not bitwise;
//   * bbh_sobol_normal: everything on the host (small draws: S x q' of a greedy step);  bbh_sobol_normal_dev: generator and scrambling
//     on the host (integer work, 0.2 MB of direction numbers up), points and transform by one thread per value on the device, the
//     samples never exist on the host (pruning draw; consumed by bbh_nehvi_samples_dev).
#include <math.h>

#include <random>

#include "bbh_common.h"

namespace {

// the engine's scrambled direction numbers (transposed: vt[l][j]) and shift of every dimension, from the unscrambled state and the seed
void bh_sobol_prepare(const int64_t* state0, uint64_t seed, int64_t dim, std::vector<int64_t>& vt, std::vector<int64_t>& shift) {
  std::mt19937 gen((uint32_t)(seed & 0xffffffffull));
  shift.assign((size_t)dim, 0);
  for (int64_t j = 0; j < dim; j++) {
    int64_t acc = 0;
    for (int k = 0; k < BBH_SOBOL_MAXBIT; k++) acc |= (int64_t)(gen() & 1u) << k;  // shift_bits @ 2^k
    shift[(size_t)j] = acc;
  }
  vt.assign((size_t)BBH_SOBOL_MAXBIT * dim, 0);
  for (int64_t j = 0; j < dim; j++) {
    uint64_t dots[BBH_SOBOL_MAXBIT];
    for (int p = 0; p < BBH_SOBOL_MAXBIT; p++) {
      uint64_t acc = 0;
      for (int k = 0; k < BBH_SOBOL_MAXBIT; k++) {
        const uint64_t r = gen() & 1u;  // (every entry of the [30, 30] matrix is drawn; tril with unit diagonal keeps k < p)
        const uint64_t bit = (k == p) ? 1u : (k < p ? r : 0u);
        acc |= bit << (BBH_SOBOL_MAXBIT - 1 - k);
      }
      dots[p] = acc;
    }
    const int64_t* v = state0 + j * BBH_SOBOL_MAXBIT;
    for (int l = 0; l < BBH_SOBOL_MAXBIT; l++) {
      const uint64_t vdj = (uint64_t)v[l];
      uint64_t t2 = 0;
      for (int p = 0; p < BBH_SOBOL_MAXBIT; p++)
        t2 |= (uint64_t)(__builtin_popcountll(dots[p] & vdj) & 1) << (BBH_SOBOL_MAXBIT - 1 - p);
      vt[(size_t)l * dim + j] = (int64_t)t2;
    }
  }
}

// Inverse error function to the last bits.  torch.erfinv on the CPU is the vendor math library's (MKL vdErfInv in the builds BayBE
// runs on), i.e. correctly rounded up to an ulp or so; it cannot be reproduced bitwise, so what is asked here is the same accuracy:
// the rational starting values of G. Pavlis' erfinv.m (relative error < 1e-3), then Newton steps on erf in the centre and on
// erfc(|x|) = 1 - |y| in the tails - there `1 - |y|` is exact (Sterbenz) and erfc keeps its relative accuracy, where a step on
// erf(x) - y would lose log10(1 / erfc) digits (observed 2e-12 absolute at |x| ~ 4).  Three steps: 1e-3 -> 1e-6 x -> 1e-12 x^3 -> rounding.
template <bool DEV>
__host__ __device__ inline double bh_erfinv(double y) {
#pragma clang fp contract(off)
  const double a0 = 0.886226899, a1 = -1.645349621, a2 = 0.914624893, a3 = -0.140543331;
  const double b0 = -2.118377725, b1 = 1.442710462, b2 = -0.329097515, b3 = 0.012229801;
  const double c0 = -1.970840454, c1 = -1.624906493, c2 = 3.429567803, c3 = 1.641345311;
  const double d0 = 3.543889200, d1 = 1.637067800;
  const double ya = fabs(y);
  if (ya >= 1.0) return ya > 1.0 ? NAN : copysign(INFINITY, y);
  const double two_over_sqrt_pi = 1.1283791670955126;
  if (ya <= 0.7) {
    const double z = y * y;
    const double num = (((a3 * z + a2) * z + a1) * z + a0);
    const double dem = ((((b3 * z + b2) * z + b1) * z + b0) * z + 1.0);
    double x = y * num / dem;
    for (int it = 0; it < 3; it++) x = x - (erf(x) - y) / (two_over_sqrt_pi * exp(-x * x));
    return x;
  }
  const double c = 1.0 - ya;
  const double z = sqrt(-log(c / 2.0));
  const double num = ((c3 * z + c2) * z + c1) * z + c0;
  const double dem = (d1 * z + d0) * z + 1.0;
  double x = num / dem;
  for (int it = 0; it < 3; it++) x = x + (erfc(x) - c) / (two_over_sqrt_pi * exp(-x * x));
  return copysign(x, y);
}

template <bool DEV>
__host__ __device__ inline double bh_normal_of_uniform(double u) {
#pragma clang fp contract(off)
  const double one_minus_eps = 1.0 - 2.220446049250313e-16;
  const double v = 0.5 + one_minus_eps * (u - 0.5);
  return bh_erfinv<DEV>(2.0 * v - 1.0) * 1.4142135623730951;
}

// out[i][j], one thread per value: point i of dimension j is the shift XOR the direction numbers of the set bits of i's Gray code
__global__ __launch_bounds__(256) void bbh_sobol_normal_kernel(const int64_t* __restrict__ vt, const int64_t* __restrict__ shift, int64_t n,
                                                               int64_t dim, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * dim) return;
  const int64_t i = e / dim, j = e - i * dim;
  int64_t q = shift[j];
  uint64_t g = (uint64_t)i ^ ((uint64_t)i >> 1);
  while (g) {
    const int l = __builtin_ctzll(g);
    q ^= vt[(int64_t)l * dim + j];
    g &= g - 1;
  }
  const double scale = 1.0 / (double)(1ll << BBH_SOBOL_MAXBIT);
  const double u = (i == 0 ? (double)(float)q : (double)q) * scale;  // (the engine keeps its first point in single precision)
  out[e] = bh_normal_of_uniform<true>(u);
}

struct bh_sobol_state {
  int64_t* d_tab = nullptr;  // [31, dim]: vt then shift
  int64_t* h_tab = nullptr;  // pinned staging of the same
  size_t bytes = 0;
  hipEvent_t evt = nullptr;  // the staged copy has been consumed
};

}  // namespace

void bbh_sobol_destroy(bbh_handle* h) {
  bh_sobol_state* st = (bh_sobol_state*)h->sobol_state;
  if (!st) return;
  if (st->d_tab) hipFree(st->d_tab);
  if (st->h_tab) hipHostFree(st->h_tab);
  if (st->evt) hipEventDestroy(st->evt);
  delete st;
  h->sobol_state = nullptr;
}

extern "C" int bbh_sobol_normal(const int64_t* state0, uint64_t seed, int64_t n, int64_t dim, double* out) {
  if (!state0 || !out || n < 1 || dim < 1 || n > (1ll << BBH_SOBOL_MAXBIT)) return -1;
  std::vector<int64_t> vt, q;
  bh_sobol_prepare(state0, seed, dim, vt, q);
  const double scale = 1.0 / (double)(1ll << BBH_SOBOL_MAXBIT);
  for (int64_t j = 0; j < dim; j++) out[j] = bh_normal_of_uniform<false>((double)(float)q[(size_t)j] * scale);
  for (int64_t i = 1; i < n; i++) {
    const int l = __builtin_ctzll(~(uint64_t)(i - 1));
    const int64_t* v = vt.data() + (size_t)l * dim;
    double* o = out + i * dim;
    for (int64_t j = 0; j < dim; j++) {
      q[(size_t)j] ^= v[j];
      o[j] = bh_normal_of_uniform<false>((double)q[(size_t)j] * scale);
    }
  }
  return 0;
}

extern "C" int bbh_sobol_normal_dev(bbh_handle* h, const int64_t* state0, uint64_t seed, int64_t n, int64_t dim, double* out_dev) {
  if (!h) return -1;
  if (!state0 || !out_dev || n < 1 || dim < 1 || n > (1ll << BBH_SOBOL_MAXBIT)) {
    h->err = "bbh_sobol_normal_dev: bad arguments";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  bh_sobol_state* st = (bh_sobol_state*)h->sobol_state;
  if (!st) h->sobol_state = st = new bh_sobol_state();
  const size_t bytes = sizeof(int64_t) * (size_t)(BBH_SOBOL_MAXBIT + 1) * (size_t)dim;
  if (!st->evt) BBH_HIP_TRY(h, hipEventCreateWithFlags(&st->evt, hipEventDisableTiming));
  else BBH_HIP_TRY(h, hipEventSynchronize(st->evt));
  if (bytes > st->bytes) {
    if (st->d_tab) hipFree(st->d_tab);
    if (st->h_tab) hipHostFree(st->h_tab);
    st->d_tab = st->h_tab = nullptr;
    st->bytes = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&st->d_tab, bytes));
    BBH_HIP_TRY(h, hipHostMalloc((void**)&st->h_tab, bytes, hipHostMallocDefault));
    st->bytes = bytes;
  }
  std::vector<int64_t> vt, shift;
  bh_sobol_prepare(state0, seed, dim, vt, shift);
  memcpy(st->h_tab, vt.data(), sizeof(int64_t) * vt.size());
  memcpy(st->h_tab + vt.size(), shift.data(), sizeof(int64_t) * shift.size());
  BBH_HIP_TRY(h, hipMemcpyAsync(st->d_tab, st->h_tab, bytes, hipMemcpyHostToDevice, h->stream));
  BBH_HIP_TRY(h, hipEventRecord(st->evt, h->stream));
  const int64_t total = n * dim;
  hipLaunchKernelGGL(bbh_sobol_normal_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, st->d_tab,
                     st->d_tab + (size_t)BBH_SOBOL_MAXBIT * dim, n, dim, out_dev);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- content key of host buffers (the resident candidate matrix is keyed on the CONTENT of the comp rep: every byte, every call) ----
// python-xxhash keeps the GIL, so "8 threads" hashed a 160 MB comp rep at one core's 32 GB/s: 5 ms of a 24 ms recommend().  Here:
// a multiply-fold hash (128-bit products folded to 64 bits, four independent lanes per 64-byte stripe - the construction of
// wyhash / rapidhash in its protected form; change detection, not cryptography: equal keys mean equal content up to a 2^-64 chance) over 4 MB pieces on std::threads, the pieces' digests folded in order.
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
  // "protected" folds: a bare multiply-fold collapses when one factor is zero - a word equal to the lane secret would
  // erase the lane's history and hide the next word.  Every step therefore also carries the previous state and both words forward
  // outside the product (rotate + add), so no input value makes the new state independent of the other word or of the history.
  auto fold = [](uint64_t x, uint64_t y, uint64_t a, uint64_t sec) {
    return bh_mum(x ^ sec, y ^ a) ^ (((a << 29) | (a >> 35)) + y + (x << 1 | x >> 63));
  };
  for (; i + 64 <= n; i += 64) {
    a0 = fold(bh_rd(p + i), bh_rd(p + i + 8), a0, s0);
    a1 = fold(bh_rd(p + i + 16), bh_rd(p + i + 24), a1, s1);
    a2 = fold(bh_rd(p + i + 32), bh_rd(p + i + 40), a2, s2);
    a3 = fold(bh_rd(p + i + 48), bh_rd(p + i + 56), a3, s3);
  }
  uint64_t tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  memcpy(tail, p + i, n - i);
  for (int k = 0; k < 8; k += 2) a0 = fold(tail[k], tail[k + 1] ^ (uint64_t)(n - i), a0, s1);
  const uint64_t lo = a0 ^ a2 ^ (uint64_t)n, hi = a1 ^ a3 ^ s2;
  return bh_mum(lo | 1ull, hi | 2ull) ^ (lo + ((hi << 31) | (hi >> 33)));
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
  for (uint64_t d : dig) acc = bh_mum((acc ^ d) | 1ull, (0x9e3779b97f4a7c15ull + d) | 2ull) ^ (((acc << 27) | (acc >> 37)) + d);
  return acc;
}
