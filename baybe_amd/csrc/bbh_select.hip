// Selection kernels: q' = 1 qLogEI with sample slices, chunk keys and the one-pass top-k.
//
// What the reference does here (baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126 -> botorch's
// optimize_acqf_discrete): evaluate the acquisition function on every candidate and take the first-index argmax
// (restated in oracle/gp_oracle.py: qlogei_q1, topk_first_index).  On the device that is, per selection step, a fixed cost on
// top of the posterior pass; at 1e5 rows it used to be 70 us (one thread per candidate over all S samples: 1.5 waves per SIMD) +
// 45 us (k sequential LDS argmax rounds, twice) + two device-to-host copies.  This file replaces it with
//
//   bbh_qlogei_q1s_kernel   one workgroup per CHUNK of 64 m candidates; the S base samples are split over the four waves (lane =
//                           candidate), so 1e5 rows put 6 waves on every SIMD.  Per sample only the "fat" term 0.1 / (1 + t^2) is
//                           evaluated - four samples share one v_rcp_f64 - because the softplus part of fatplus() is a sum over a
//                           contiguous run of the SORTED base samples: its count and z-sum come from two binary searches in
//                           suffix-sum tables (t = a + b z is monotone in z), and the few samples with -750 <= t <= 20 are
//                           evaluated exactly.  The wave that finalises the scores also reduces them to the chunk's key
//                           (max score, first index achieving it).
//   bbh_chunk_keys_kernel   the same keys for a score vector that came from another kernel.
//   bbh_select_kernel       ONE workgroup, no k-round loops: (1) every thread takes the best key of its strided chunks, (2) the
//                           k-th best key of small thread GROUPS is a lower bound T2 of the k-th best chunk key, so (3) only the
//                           winning groups rescan their chunks for keys >= T2 into an LDS list, (4) ranking by counting gives the k best chunks -
//                           every top-k element lies in one of them, because any other chunk's elements are below k chunk maxima -
//                           (5) their elements >= the k-th chunk's key go to an LDS list and (6) are ranked by counting.  Keys are
//                           (score descending, index ascending): a strict order, so duplicates cannot inflate any list and ties
//                           resolve to the lowest index exactly as torch.argmax / the oracle do.  NaN never wins.  The k results go
//                           to device memory or straight to a host-mapped buffer (no copy engine in the step).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "bbh_common.h"

#define SEL_TILE 64
#define SEL_MAX_CHUNKS 4096
#define SEL_CAP 3072        // entries of the LDS candidate list of bbh_select_kernel (16 bytes each)
#define SEL_KEYS_PER_THREAD (SEL_MAX_CHUNKS / 256)
#define SEL_NANFLAG 0x40000000
#define SEL_BATCH 8         // global loads in flight per thread in the scans of bbh_select_kernel
#define Q1S_MAX_S 1024      // sample-sliced form: four tables of S doubles in LDS
#define Q1_LOG_TAU_RELU -13.815510557964274  // log(1e-6)
#define Q1_INV_TAU 1e6

// (score descending, index ascending): does a beat b?
__device__ __forceinline__ bool sel_beats(double av, int64_t ai, double bv, int64_t bi) { return av > bv || (av == bv && ai < bi); }

__device__ __forceinline__ bool sel_beats32(double av, int ai, double bv, int bi) { return av > bv || (av == bv && ai < bi); }

// 1x1 psd_safe_cholesky: v <= 0 (or NaN) -> add jitter 1e-8, 1e-7, 1e-6 (oracle/gp_oracle.py:_safe_sqrt_var)
__device__ __forceinline__ double sel_safe_sd(double v) {
  if (!(v > 0.0)) {
    v += 1e-8;
    if (!(v > 0.0)) {
      v += 1e-7;
      if (!(v > 0.0)) v += 1e-6;
    }
  }
  return sqrt(fmax(v, 0.0));
}

// wave-wide (max score, first index) of one value per lane; result in every lane
__device__ __forceinline__ void sel_wave_best(double& v, int64_t& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const int64_t oi = __shfl_xor(i, o, 64);
    if (oi >= 0 && (i < 0 || sel_beats(ov, oi, v, i))) {
      v = ov;
      i = oi;
    }
  }
}

// sum over four samples of 1 / (1 + t^2) with ONE reciprocal: 1/d0 + 1/d1 + 1/d2 + 1/d3 = ((d0 + d1) d2 d3 + (d2 + d3) d0 d1) / (d0 d1 d2 d3).
// d in [1, 1e77) for |t| < 1e38 (t = (objective - best_f) / 1e-6): no overflow in the product of four.  v_rcp_f64 seeds 2^-26, one
// Newton step: <= 2^-46 relative; all terms positive.
__device__ __forceinline__ double sel_fat4(double a, double b, double z0, double z1, double z2, double z3) {
  const double t0 = fma(b, z0, a), t1 = fma(b, z1, a), t2 = fma(b, z2, a), t3 = fma(b, z3, a);
  const double d0 = fma(t0, t0, 1.0), d1 = fma(t1, t1, 1.0), d2 = fma(t2, t2, 1.0), d3 = fma(t3, t3, 1.0);
  const double p01 = d0 * d1, p23 = d2 * d3;
  const double num = fma(d2 + d3, p01, (d0 + d1) * p23);
  const double den = p01 * p23;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  return num * r;
}
__device__ __forceinline__ double sel_fat1(double a, double b, double z) {
  const double t = fma(b, z, a);
  const double d = fma(t, t, 1.0);
  double r = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r, 1.0), r, r);
}

// tab: device tables [2][S + 1]: Z = the base samples oriented by the objective's sign (sign * z) and sorted ascending, SF[j] = sum of
// Z[j ..] (summed from the top), SF[S] = 0.  key_v / key_i [chunks].
// LDS: tables [2][S + 1] | partial fat sums [2][3][64]
struct Q1SArgs {
  int64_t N;
  int S, tiles_per_chunk;
  double best_f, sign;
};
__global__ __launch_bounds__(256) void bbh_qlogei_q1s_kernel(const double* __restrict__ mean, const double* __restrict__ var,
                                                             const double* __restrict__ tab, const uint8_t* __restrict__ alive,
                                                             double* __restrict__ scores, double* __restrict__ key_v,
                                                             int64_t* __restrict__ key_i, const Q1SArgs A) {
  extern __shared__ double lds[];
  const int S = A.S, S1 = S + 1;
  double* s_part = lds + 2 * S1;
  for (int e = threadIdx.x; e < 2 * S1; e += 256) lds[e] = tab[e];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // sample slices in quads of four; the finalising wave (0) takes a smaller one: it also runs the searches, the logarithm and the
  // key reduction (~250 instructions against ~5.5 per sample)
  const int nq = S >> 2;
  const int q_w0 = (nq >= 64) ? (nq >> 2) - 3 * (nq >> 5) : nq >> 2;  // S = 512: 128 quads -> 20 | 36 36 36
  const int rest = nq - q_w0;
  const int q_lo = (wave == 0) ? 0 : q_w0 + (rest * (wave - 1)) / 3;
  const int q_hi = (wave == 0) ? q_w0 : q_w0 + (rest * wave) / 3;
  const double* __restrict__ zs_g = tab;  // global copy: uniform addresses -> scalar loads
  __syncthreads();
  double best_v = -INFINITY;
  int64_t best_i = -1;
  const int64_t chunk0 = (int64_t)blockIdx.x * A.tiles_per_chunk * SEL_TILE;
  for (int t = 0; t < A.tiles_per_chunk; t++) {
    const int64_t i = chunk0 + (int64_t)t * SEL_TILE + lane;
    if (chunk0 + (int64_t)t * SEL_TILE >= A.N) break;  // uniform
    const bool in = i < A.N;
    const double mu = in ? mean[i] : 0.0;
    const double vr = in ? var[i] : 1.0;
    // t = (sign (mu + sd z) - best_f) / tau = a + b Z with Z = sign z (|sign| = 1) and b = sd / tau >= 0: monotone in sorted Z
    const double a = (A.sign * mu - A.best_f) * Q1_INV_TAU;
    const double b = sel_safe_sd(vr) * Q1_INV_TAU;
    double f0 = 0.0, f1 = 0.0;
    int q = q_lo;
    for (; q + 1 < q_hi; q += 2) {
      const double* z = zs_g + 4 * q;
      f0 += sel_fat4(a, b, z[0], z[1], z[2], z[3]);
      f1 += sel_fat4(a, b, z[4], z[5], z[6], z[7]);
    }
    if (q < q_hi) {
      const double* z = zs_g + 4 * q;
      f0 += sel_fat4(a, b, z[0], z[1], z[2], z[3]);
    }
    double fat = f0 + f1;
    // softplus part: t_j = a + b Z[j] is non-decreasing in j, so {t_j > 20} is a suffix and {-750 <= t_j <= 20} the run before it,
    // whose terms are evaluated one by one (torch's softplus below its threshold).  On a well-converged model (sd ~ 1e-3: b = sd /
    // tau ~ 1e3) that run is a third of the samples for most lanes: every wave finds the run (two binary searches, ~20 LDS reads)
    // and sums its own quarter of it - wave 0 alone walked up to S log1p(exp()) per candidate while three waves sat at the barrier
    // (ADVICE r4).  NaN inputs have no run (the score is NaN anyway).
    const double babs = b;
    const double* Z = lds;
    const double* SF = lds + S1;
    const bool num = (a == a) && (b == b);
    int lo = 0, hi = num ? S : 0;  // first j with t_j > 20
    while (__builtin_amdgcn_ballot_w64(lo < hi) != 0) {
      if (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (fma(babs, Z[mid], a) > 20.0) hi = mid; else lo = mid + 1;
      }
    }
    const int j_hi = num ? lo : 0;
    lo = 0, hi = j_hi;   // first j with t_j >= -750
    while (__builtin_amdgcn_ballot_w64(lo < hi) != 0) {
      if (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (!(fma(babs, Z[mid], a) < -750.0)) hi = mid; else lo = mid + 1;
      }
    }
    const int run = j_hi - lo;
    const int r0 = lo + (run * wave) / 4, r1 = lo + (run * (wave + 1)) / 4;
    double spw = 0.0;
    for (int j = r0; __builtin_amdgcn_ballot_w64(j < r1) != 0; j++)
      if (j < r1) spw += log1p(exp(fma(babs, Z[j], a)));  // torch softplus below its threshold of 20
    double* part = s_part + (t & 1) * 384;
    if (wave != 0) {
      part[(wave - 1) * 64 + lane] = fat;
      part[192 + (wave - 1) * 64 + lane] = spw;
    }
    __syncthreads();
    if (wave == 0) {
      for (int s = nq * 4; s < S; s++) fat += sel_fat1(a, b, zs_g[s]);  // S not a multiple of four
      fat = ((fat + part[lane]) + part[64 + lane]) + part[128 + lane];
      double sp = (j_hi < S && num) ? fma(babs, SF[j_hi], a * (double)(S - j_hi)) : 0.0;
      sp += ((spw + part[192 + lane]) + part[256 + lane]) + part[320 + lane];
      double score = Q1_LOG_TAU_RELU + log(fma(0.1, fat, sp)) - log((double)S);
      if (!(a == a) || !(b == b)) score = NAN;
      if (in && alive && !alive[i]) score = -INFINITY;
      if (in) scores[i] = score;
      double cv = score;
      int64_t ci = (in && score == score) ? i : -1;
      sel_wave_best(cv, ci);
      if (ci >= 0 && (best_i < 0 || sel_beats(cv, ci, best_v, best_i))) {
        best_v = cv;
        best_i = ci;
      }
    }
  }
  if (threadIdx.x == 0) {
    key_v[blockIdx.x] = (best_i >= 0) ? best_v : -INFINITY;
    key_i[blockIdx.x] = best_i;
  }
}

// chunk keys of an arbitrary score vector: one wave per chunk
__global__ __launch_bounds__(256) void bbh_chunk_keys_kernel(const double* __restrict__ scores, int64_t N, int chunk, int chunks,
                                                             double* __restrict__ key_v, int64_t* __restrict__ key_i) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= chunks) return;
  const int lane = threadIdx.x & 63;
  const int64_t base = (int64_t)c * chunk;
  const int64_t end = (base + chunk < N) ? base + chunk : N;
  double v = -INFINITY;
  int64_t idx = -1;
  for (int64_t i = base + lane; i < end; i += 64) {
    const double x = scores[i];
    if (x == x && (idx < 0 || x > v)) {  // ascending i per lane: equal scores keep the earlier index
      v = x;
      idx = i;
    }
  }
  sel_wave_best(v, idx);
  if (lane == 0) {
    key_v[c] = (idx >= 0) ? v : -INFINITY;
    key_i[c] = idx;
  }
}

// workgroup-wide best key (every thread passes its own; all get the winner)
__device__ __forceinline__ void sel_block_best(double& v, int64_t& i, double* sv, int64_t* si) {
  sel_wave_best(v, i);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  v = sv[0], i = si[0];
  for (int w = 1; w < 4; w++)
    if (si[w] >= 0 && (i < 0 || sel_beats(sv[w], si[w], v, i))) v = sv[w], i = si[w];
}

__global__ __launch_bounds__(256) void bbh_select_kernel(const double* __restrict__ scores, int64_t N, int chunk, int chunks,
                                                         const double* __restrict__ key_v, const int64_t* __restrict__ key_i, int k,
                                                         double* __restrict__ outv, int64_t* __restrict__ outi) {
  // Latency budget: the kernel is one workgroup, so every DEPENDENT trip to memory (~1.5 us: the keys were written by other XCDs)
  // shows in full.  There are two: the chunk keys (all SEL_KEYS_PER_THREAD of a thread in flight at once, kept in registers for
  // step 3) and the elements of the k selected chunks (one flat index space, all loads in flight at once).
  __shared__ double s_tv[256];
  __shared__ int s_tc[256];
  __shared__ int s_rank[256];
  __shared__ double s_lv[SEL_CAP];
  __shared__ int64_t s_li[SEL_CAP];  // chunk entries: (id << 32) | offset of the chunk's first maximum; element entries: global index
  __shared__ double s_selv[64];
  __shared__ int64_t s_selc[64];
  __shared__ int s_cnt;
  __shared__ double s_thr_v;
  __shared__ int s_thr_c;
  const int tid = threadIdx.x;
  if (tid < k) {
    outv[tid] = -INFINITY;
    outi[tid] = -1;
  }
  if (tid == 0) {
    s_cnt = 0;
    s_thr_v = -INFINITY;
    s_thr_c = 0x7fffffff;
  }
  const int kk = (k < chunks) ? k : chunks;
  // (1) best chunk of every thread; chunk ids carry a flag when the chunk holds no number at all (they rank after every real chunk)
  double rv[SEL_KEYS_PER_THREAD];
  int64_t ri[SEL_KEYS_PER_THREAD];
#pragma unroll
  for (int u = 0; u < SEL_KEYS_PER_THREAD; u++) {
    const int c = tid + u * 256;
    rv[u] = (c < chunks) ? key_v[c] : -INFINITY;
    ri[u] = (c < chunks) ? key_i[c] : -1;
  }
  double mv = -INFINITY;
  int mc = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < SEL_KEYS_PER_THREAD; u++) {
    const int c = tid + u * 256;
    const int id = (ri[u] < 0) ? (c | SEL_NANFLAG) : c;
    if (c < chunks && sel_beats32(rv[u], id, mv, mc)) {
      mv = rv[u];
      mc = id;
    }
  }
  // (2) the k-th best of G GROUP keys (groups of gs neighbouring threads, G = 256 / gs >= k) is a lower bound T2 of the k-th best
  // chunk key: ranking G keys by counting costs G compares per key - 64 x 64 for k <= 16 instead of 256 x 256
  // (only groups whose threads all own a chunk count: with few chunks the groups shrink until kk of them hold real keys)
  const int nthr = (chunks < 256) ? chunks : 256;
  const int gs = (kk <= 16 && nthr / 4 >= kk) ? 4 : ((kk <= 32 && nthr / 2 >= kk) ? 2 : 1);
  const int G = 256 / gs;
  double gv = mv;
  int gc = mc;
  for (int o = 1; o < gs; o <<= 1) {
    const double ov = __shfl_xor(gv, o, 64);
    const int oc = __shfl_xor(gc, o, 64);
    if (sel_beats32(ov, oc, gv, gc)) gv = ov, gc = oc;
  }
  if ((tid & (gs - 1)) == 0) {
    s_tv[tid / gs] = gv;
    s_tc[tid / gs] = gc;
  }
  __syncthreads();
  if (tid < G) {
    const double kv = s_tv[tid];
    const int kc = s_tc[tid];
    int r = 0;
#pragma unroll 8
    for (int j = 0; j < G; j++) r += sel_beats32(s_tv[j], s_tc[j], kv, kc) ? 1 : 0;
    s_rank[tid] = r;
    if (r == kk - 1) {
      s_thr_v = kv;
      s_thr_c = kc;
    }
  }
  __syncthreads();
  // (3) chunks >= T2, from the threads of the kk winning groups only (every other thread's chunks are below T2)
  const double t2v = s_thr_v;
  const int t2c = s_thr_c;
  if (s_rank[tid / gs] < kk) {
#pragma unroll
    for (int u = 0; u < SEL_KEYS_PER_THREAD; u++) {
      const int c = tid + u * 256;
      const int id = (ri[u] < 0) ? (c | SEL_NANFLAG) : c;
      if (c < chunks && !sel_beats32(t2v, t2c, rv[u], id)) {
        const int pos = atomicAdd(&s_cnt, 1);  // <= kk * gs * SEL_KEYS_PER_THREAD <= 1024 entries
        s_lv[pos] = rv[u];
        s_li[pos] = ((int64_t)id << 32) | (uint32_t)((ri[u] < 0) ? 0 : (ri[u] - (int64_t)c * chunk));
      }
    }
  }
  __syncthreads();
  const int c1 = s_cnt;
  // (4) the kk best chunks, in order
  for (int e = tid; e < c1; e += 256) {
    const double v = s_lv[e];
    const int64_t id = s_li[e];
    int r = 0;
#pragma unroll 8
    for (int j = 0; j < c1; j++) r += sel_beats(s_lv[j], s_li[j], v, id) ? 1 : 0;  // (unrolled: the LDS reads of a batch are in flight together)
    if (r < kk) {
      s_selv[r] = v;
      s_selc[r] = id;
    }
  }
  __syncthreads();
  if (tid == 0) s_cnt = 0;
  // threshold element: the max element of the k-th chunk (everything when there are fewer than k chunks)
  double ev = -INFINITY;
  int64_t ei = INT64_MAX;
  {
    const int64_t last = s_selc[kk - 1];
    if (kk == k && !((last >> 32) & SEL_NANFLAG)) {
      ev = s_selv[kk - 1];
      ei = (last >> 32) * (int64_t)chunk + (int64_t)(last & 0xffffffff);
    }
  }
  __syncthreads();
  // (5) elements >= the threshold element, over the flat index space (selected chunk, offset); 32-bit index arithmetic (a 64-bit
  // division by the run-time chunk size is ~100 instructions, sixteen of them per thread were a quarter of the kernel)
  const uint32_t total = (uint32_t)kk * (uint32_t)chunk;  // <= 64 * 64 * ceil(N / 262144): fits for N < 2^43
  for (uint32_t e0 = tid; e0 < total; e0 += 256 * SEL_BATCH) {
    double bx[SEL_BATCH];
    int64_t bg[SEL_BATCH];
#pragma unroll
    for (int u = 0; u < SEL_BATCH; u++) {
      const uint32_t e = e0 + u * 256;
      bx[u] = NAN;
      bg[u] = -1;
      if (e < total) {
        const uint32_t sl = e / (uint32_t)chunk;
        const int64_t cid = s_selc[sl] >> 32;
        const int64_t g = cid * chunk + (e - sl * (uint32_t)chunk);
        if (!(cid & SEL_NANFLAG) && g < N) {
          bx[u] = scores[g];
          bg[u] = g;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SEL_BATCH; u++) {
      const double x = bx[u];
      if (x == x && !sel_beats(ev, ei, x, bg[u])) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < SEL_CAP) {
          s_lv[pos] = x;
          s_li[pos] = bg[u];
        }
      }
    }
  }
  __syncthreads();
  const int c2 = s_cnt;
  if (c2 > SEL_CAP) {
    // more elements above the k-th chunk's key than the list holds (k close to 64 on flat scores): k rounds over the selected
    // chunks, each taking the best element strictly below the previous winner - the order is strict, nothing has to be marked
    __shared__ double s_wv[4];
    __shared__ int64_t s_wi[4];
    double pv = 0.0;
    int64_t pi = -1;  // no previous winner
    for (int r = 0; r < k; r++) {
      double bv = -INFINITY;
      int64_t bi = -1;
      for (uint32_t e = tid; e < total; e += 256) {
        const uint32_t sl = e / (uint32_t)chunk;
        const int64_t cid = s_selc[sl] >> 32;
        const int64_t g = cid * chunk + (e - sl * (uint32_t)chunk);
        if ((cid & SEL_NANFLAG) || g >= N) continue;
        const double x = scores[g];
        if (x == x && (pi < 0 || sel_beats(pv, pi, x, g)) && (bi < 0 || sel_beats(x, g, bv, bi))) {
          bv = x;
          bi = g;
        }
      }
      sel_block_best(bv, bi, s_wv, s_wi);
      if (bi < 0) break;
      if (tid == 0) {
        outv[r] = bv;
        outi[r] = bi;
      }
      pv = bv, pi = bi;
    }
    return;
  }
  // (6) rank by counting
  for (int e = tid; e < c2; e += 256) {
    const double v = s_lv[e];
    const int64_t id = s_li[e];
    int r = 0;
#pragma unroll 8
    for (int j = 0; j < c2; j++) r += sel_beats(s_lv[j], s_li[j], v, id) ? 1 : 0;
    if (r < k) {
      outv[r] = v;
      outi[r] = id;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct bbh_select_state {
  double* d_key_v = nullptr;   // [SEL_MAX_CHUNKS]
  int64_t* d_key_i = nullptr;  // [SEL_MAX_CHUNKS]
  double* d_out_v = nullptr;   // [64] device-side results (bbh_topk_device)
  int64_t* d_out_i = nullptr;  // [64]
  // host-mapped result block: 16 bytes unused | values [64] | indices [64]
  void* h_res = nullptr;
  void* h_res_dev = nullptr;   // the device's address of h_res
  // sorted base-sample tables of the sample-sliced qLogEI kernel
  double* d_tab = nullptr;
  double* h_tab = nullptr;     // pinned staging copy
  hipEvent_t tab_evt = nullptr;
  size_t tab_elems = 0;
  std::vector<double> z_last;  // the z the tables were built from
  double z_sign = 0.0;
};

static bbh_select_state* sel_state(bbh_handle* h) {
  if (!h->select_state) h->select_state = new bbh_select_state();
  return (bbh_select_state*)h->select_state;
}

void bbh_select_destroy(bbh_handle* h) {
  if (!h->select_state) return;
  bbh_select_state* st = (bbh_select_state*)h->select_state;
  if (st->d_key_v) hipFree(st->d_key_v);
  if (st->d_key_i) hipFree(st->d_key_i);
  if (st->d_out_v) hipFree(st->d_out_v);
  if (st->d_out_i) hipFree(st->d_out_i);
  if (st->h_res) hipHostFree(st->h_res);
  if (st->d_tab) hipFree(st->d_tab);
  if (st->h_tab) hipHostFree(st->h_tab);
  if (st->tab_evt) hipEventDestroy(st->tab_evt);
  delete st;
  h->select_state = nullptr;
}

static int sel_ensure(bbh_handle* h, bbh_select_state* st) {
  if (st->d_key_v) return 0;
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_key_v, sizeof(double) * SEL_MAX_CHUNKS));
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_key_i, sizeof(int64_t) * SEL_MAX_CHUNKS));
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_out_v, sizeof(double) * 64));
  BBH_HIP_TRY(h, hipMalloc((void**)&st->d_out_i, sizeof(int64_t) * 64));
  // results straight to the host: the select kernel's stores travel over the link, one stream synchronisation ends the step
  // (BBH_SELECT_MAPPED=0: device buffer + copy, for A/B)
  const char* mp = getenv("BBH_SELECT_MAPPED");
  if (mp && mp[0] == '0') {
    st->h_res = nullptr;
  } else if (hipHostMalloc(&st->h_res, 16 + 64 * 16, hipHostMallocMapped) == hipSuccess) {
    if (hipHostGetDevicePointer(&st->h_res_dev, st->h_res, 0) != hipSuccess) {
      hipHostFree(st->h_res);
      st->h_res = st->h_res_dev = nullptr;
    }
  } else {
    st->h_res = nullptr;
  }
  return 0;
}

static inline int sel_tiles_per_chunk(int64_t N) {
  const int64_t tiles = (N + SEL_TILE - 1) / SEL_TILE;
  return (int)((tiles + SEL_MAX_CHUNKS - 1) / SEL_MAX_CHUNKS);
}

// Keys must be current in st->d_key_* for (scores_dev, N).  With vals_host the results go to the host (one synchronisation);
// otherwise they stay in st->d_out_* (*vals_dev / *idx_dev), nothing is waited for.
static int sel_finish(bbh_handle* h, bbh_select_state* st, const double* scores_dev, int64_t N, int64_t k, double* vals_host,
                      int64_t* idx_host, double** vals_dev, int64_t** idx_dev) {
  const int tpc = sel_tiles_per_chunk(N);
  const int chunk = tpc * SEL_TILE;
  const int chunks = (int)((N + chunk - 1) / chunk);
  const bool to_host = vals_host != nullptr;
  const bool mapped = to_host && st->h_res_dev;
  double* ov = mapped ? (double*)((char*)st->h_res_dev + 16) : st->d_out_v;
  int64_t* oi = mapped ? (int64_t*)((char*)st->h_res_dev + 16 + 64 * 8) : st->d_out_i;
  {
    bbh_timed_scope timed(h, BBH_TIMED_SELECT);
    hipLaunchKernelGGL(bbh_select_kernel, dim3(1), dim3(256), 0, h->stream, scores_dev, N, chunk, chunks, st->d_key_v, st->d_key_i, (int)k,
                       ov, oi);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  if (mapped) {
    BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(vals_host, (char*)st->h_res + 16, sizeof(double) * k);
    memcpy(idx_host, (char*)st->h_res + 16 + 64 * 8, sizeof(int64_t) * k);
  } else if (to_host) {
    BBH_HIP_TRY(h, hipMemcpyAsync(vals_host, st->d_out_v, sizeof(double) * k, hipMemcpyDeviceToHost, h->stream));
    BBH_HIP_TRY(h, hipMemcpyAsync(idx_host, st->d_out_i, sizeof(int64_t) * k, hipMemcpyDeviceToHost, h->stream));
    BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  } else {
    *vals_dev = st->d_out_v;
    *idx_dev = st->d_out_i;
  }
  return 0;
}

static int sel_check(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k) {
  if (!scores_dev || N < 1 || k < 1 || k > N || k > 64) {
    h->err = "bbh_topk: bad arguments (1 <= k <= min(N, 64))";
    return -1;
  }
  return 0;
}

static int sel_keys(bbh_handle* h, bbh_select_state* st, const double* scores_dev, int64_t N) {
  const int tpc = sel_tiles_per_chunk(N);
  const int chunk = tpc * SEL_TILE;
  const int chunks = (int)((N + chunk - 1) / chunk);
  bbh_timed_scope timed(h, BBH_TIMED_SELECT);
  hipLaunchKernelGGL(bbh_chunk_keys_kernel, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, h->stream, scores_dev, N, chunk, chunks,
                     st->d_key_v, st->d_key_i);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

int bbh_topk_rounds_device(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double** vals_dev, int64_t** idx_dev);
int bbh_topk_rounds(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double* vals_host, int64_t* idx_host);
int bbh_argmax_rounds(bbh_handle* h, const double* scores_dev, int64_t N, double* best_val_host, int64_t* best_idx_host);

// k best scores on the device: *vals_dev / *idx_dev point into the handle's selection state (valid until its next use)
int bbh_topk_device(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double** vals_dev, int64_t** idx_dev) {
  if (!h->select_on) return bbh_topk_rounds_device(h, scores_dev, N, k, vals_dev, idx_dev);
  int rc = sel_check(h, scores_dev, N, k);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  bbh_select_state* st = sel_state(h);
  if ((rc = sel_ensure(h, st)) || (rc = sel_keys(h, st, scores_dev, N))) return rc;
  return sel_finish(h, st, scores_dev, N, k, nullptr, nullptr, vals_dev, idx_dev);
}

extern "C" int bbh_topk(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double* vals_host, int64_t* idx_host) {
  if (!h) return -1;
  if (!vals_host || !idx_host) {
    h->err = "bbh_topk: bad arguments (1 <= k <= min(N, 64))";
    return -1;
  }
  if (!h->select_on) return bbh_topk_rounds(h, scores_dev, N, k, vals_host, idx_host);
  int rc = sel_check(h, scores_dev, N, k);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  bbh_select_state* st = sel_state(h);
  if ((rc = sel_ensure(h, st)) || (rc = sel_keys(h, st, scores_dev, N))) return rc;
  return sel_finish(h, st, scores_dev, N, k, vals_host, idx_host, nullptr, nullptr);
}

extern "C" int bbh_argmax(bbh_handle* h, const double* scores_dev, int64_t N, double* best_val_host, int64_t* best_idx_host) {
  if (!h) return -1;
  if (!scores_dev || N < 1 || !best_val_host || !best_idx_host) {
    h->err = "bbh_argmax: bad arguments";
    return -1;
  }
  if (!h->select_on) return bbh_argmax_rounds(h, scores_dev, N, best_val_host, best_idx_host);
  return bbh_topk(h, scores_dev, N, 1, best_val_host, best_idx_host);
}

// base samples oriented by the objective's sign, sorted, with suffix sums; rebuilt only when (z, sign) changes (the q' = 1 passes of a
// campaign's selection steps mostly reuse one draw)
static int sel_tables(bbh_handle* h, bbh_select_state* st, const double* z_host, int S, double sign) {
  const size_t S1 = (size_t)S + 1;
  const double sg = (sign < 0.0) ? -1.0 : 1.0;
  if (st->z_last.size() == (size_t)S && st->z_sign == sg && memcmp(st->z_last.data(), z_host, sizeof(double) * S) == 0) return 0;
  if (!st->h_tab || 2 * S1 > st->tab_elems) {
    if (st->d_tab) hipFree(st->d_tab);
    if (st->h_tab) hipHostFree(st->h_tab);
    st->d_tab = st->h_tab = nullptr;
    st->tab_elems = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&st->d_tab, sizeof(double) * 2 * S1));
    BBH_HIP_TRY(h, hipHostMalloc((void**)&st->h_tab, sizeof(double) * 2 * S1, hipHostMallocDefault));
    st->tab_elems = 2 * S1;
  } else if (st->tab_evt) {
    BBH_HIP_TRY(h, hipEventSynchronize(st->tab_evt));  // the previous staged copy has been consumed
  }
  double* Z = st->h_tab;
  double* SF = Z + S1;
  for (int j = 0; j < S; j++) Z[j] = sg * z_host[j];
  std::sort(Z, Z + S);
  Z[S] = 0.0;
  SF[S] = 0.0;
  for (int j = S - 1; j >= 0; j--) SF[j] = SF[j + 1] + Z[j];
  if (!st->tab_evt) BBH_HIP_TRY(h, hipEventCreateWithFlags(&st->tab_evt, hipEventDisableTiming));
  BBH_HIP_TRY(h, hipMemcpyAsync(st->d_tab, st->h_tab, sizeof(double) * 2 * S1, hipMemcpyHostToDevice, h->stream));
  BBH_HIP_TRY(h, hipEventRecord(st->tab_evt, h->stream));
  st->z_last.assign(z_host, z_host + S);
  st->z_sign = sg;
  return 0;
}

// q' = 1 qLogEI through the sample-sliced kernel; leaves the chunk keys of scores_dev in the selection state.  Returns 1 when the
// form does not apply (S too large for the LDS tables), negative on errors.
int bbh_qlogei_q1_sliced(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host, int64_t S,
                         double best_f, double sign, const uint8_t* alive_dev, double* scores_dev) {
  if (S > Q1S_MAX_S || S < 4) return 1;
  bbh_select_state* st = sel_state(h);
  int rc = sel_ensure(h, st);
  if (rc || (rc = sel_tables(h, st, z_host, (int)S, sign))) return rc;
  const size_t lds = sizeof(double) * (2 * ((size_t)S + 1) + 2 * 6 * 64);  // tables + two generations of (fat, softplus-run) partials of waves 1-3
  Q1SArgs a;
  a.N = N, a.S = (int)S, a.best_f = best_f, a.sign = sign, a.tiles_per_chunk = sel_tiles_per_chunk(N);
  const int chunk = a.tiles_per_chunk * SEL_TILE;
  const int chunks = (int)((N + chunk - 1) / chunk);
  bbh_timed_scope timed(h, BBH_TIMED_Q1);
  hipLaunchKernelGGL(bbh_qlogei_q1s_kernel, dim3((unsigned)chunks), dim3(256), lds, h->stream, mean_dev, var_dev, st->d_tab, alive_dev,
                     scores_dev, st->d_key_v, st->d_key_i, a);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

int bbh_qlogei_q1_rounds(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host, int64_t S,
                         double best_f, double sign, const uint8_t* alive_dev, double* scores_dev);

extern "C" int bbh_qlogei_q1_topk(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host,
                                  int64_t S, double best_f, double sign, const uint8_t* alive_dev, double* scores_dev, int64_t k,
                                  double* vals_host, int64_t* idx_host) {
  if (!h) return -1;
  if (!mean_dev || !var_dev || !z_host || !scores_dev || N < 1 || S < 1 || S > 8192 || !vals_host || !idx_host) {
    h->err = "bbh_qlogei_q1_topk: bad arguments (1 <= S <= 8192, N >= 1)";
    return -1;
  }
  int rc = sel_check(h, scores_dev, N, k);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  bbh_select_state* st = sel_state(h);
  rc = (h->q1_sliced && h->select_on) ? bbh_qlogei_q1_sliced(h, mean_dev, var_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev) : 1;
  if (rc < 0) return rc;
  if (rc == 1) {  // one thread per candidate, then the keys in their own pass
    if ((rc = bbh_qlogei_q1_rounds(h, mean_dev, var_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev))) return rc;
    if (!h->select_on) return bbh_topk_rounds(h, scores_dev, N, k, vals_host, idx_host);
    if ((rc = sel_ensure(h, st)) || (rc = sel_keys(h, st, scores_dev, N))) return rc;
  }
  return sel_finish(h, st, scores_dev, N, k, vals_host, idx_host, nullptr, nullptr);
}
