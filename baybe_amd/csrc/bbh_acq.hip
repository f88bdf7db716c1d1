// Monte-Carlo acquisition (qLogEI) and selection kernels.
//
// qLogExpectedImprovement as BayBE builds it (baybe/acquisition/acqfs.py:219-223,
// baybe/acquisition/_builder.py:195-265): fat=True, tau_relu=1e-6, tau_max=1e-2, Sobol-normal
// base samples shared by all candidates; objective = sign * sample (minimisation = -1,
// baybe/objectives/base.py:99-105).  Maths restated in oracle/gp_oracle.py (log_fatplus, fatmax,
// logmeanexp, psd_safe_cholesky jitter).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "bbh_common.h"

#define TAU_RELU 1e-6
#define TAU_MAX 1e-2
#define LOG_TAU_RELU -13.815510557964274  // log(1e-6)
#ifndef BBH_PENDING_LSE
#define BBH_PENDING_LSE 0
#endif
typedef float bbh_f2 __attribute__((ext_vector_type(2)));
#ifndef BBH_PENDING_PK
#define BBH_PENDING_PK 1  // (with BBH_PENDING_FAST) the u_r^2 terms of the fat maximum in packed single precision, two points at a time
#endif
#ifndef BBH_PENDING_FAST
#define BBH_PENDING_FAST 1  // joint q'-batch qLogEI kernels: reduced-precision logarithms where the power tau_max = 0.01 absorbs them
#endif

// fatplus(x; tau) / tau = softplus(t) + 0.1 / (1 + t^2),  t = x / tau; torch softplus threshold 20
template <int NEWTON = 2>
__device__ __forceinline__ double bbh_fatplus_core(double t) {
  // softplus: t / tau_relu is huge in magnitude for almost every sample, so the log1p(exp) branch is rare.
  // It is entered through a wave-uniform test (ballot): a per-lane branch in unrolled callers is
  // if-converted by the compiler into "always evaluate both sides", i.e. ~50 extra VALU per call.
  double sp = (t > 20.0) ? t : 0.0;
  const bool mid = !(t > 20.0) && !(t < -750.0);
  if (__builtin_amdgcn_ballot_w64(mid) != 0) {
    if (mid) sp = log1p(exp(t));
  }
  // 0.1 / (1 + t^2) without the IEEE division sequence (div_scale / div_fmas / div_fixup, ~15 VALU
  // of the ~22 per sample): v_rcp_f64 seed (2^-26) and two Newton steps; 1 + t^2 is in [1, 1e40) for
  // every reachable t, so no scaling is needed.  Relative error <= 2 ulp.
  const double d = fma(t, t, 1.0);
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  if (NEWTON > 1) y = fma(fma(-d, y, 1.0), y, y);  // (one step: <= 2^-46 relative - enough where the caller's own terms are single precision)
  y = (d < INFINITY) ? y : 0.0;  // |t| = inf (unbounded cell): the Newton step would produce inf * 0
  return fma(0.1, y, sp);
}

// 1x1 psd_safe_cholesky: v <= 0 (or NaN) -> add jitter 1e-8, 1e-7, 1e-6
__device__ __forceinline__ double bbh_safe_sd(double v) {
  if (!(v > 0.0)) {
    v += 1e-8;
    if (!(v > 0.0)) {
      v += 1e-7;
      if (!(v > 0.0)) v += 1e-6;
    }
  }
  return sqrt(fmax(v, 0.0));
}

// q' = 1:  score = log( mean_s fatplus(sign (mu + sd z_s) - best_f) )
//               = logmeanexp_s log_fatplus(...)   (all terms positive, no cancellation)
__global__ __launch_bounds__(256) void bbh_qlogei_q1_kernel(const double* __restrict__ mean,
                                                            const double* __restrict__ var, int64_t N,
                                                            const double* __restrict__ z, int S, double best_f,
                                                            double sign, const uint8_t* __restrict__ alive,
                                                            double* __restrict__ scores) {
  extern __shared__ double s_z[];
  for (int s = threadIdx.x; s < S; s += blockDim.x) s_z[s] = z[s];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  const double inv_tau = 1.0 / TAU_RELU;
  const double a = (sign * mean[i] - best_f) * inv_tau;
  const double b = sign * bbh_safe_sd(var[i]) * inv_tau;
  // four independent partial sums: the per-sample chain (fma, compare, reciprocal seed, two Newton steps) is ~10
  // dependent instructions; one accumulator left the kernel latency-bound at half the VALU rate (0.34 ms per 1e6 x 512)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int s = 0;
  for (; s + 3 < S; s += 4) {
    s0 += bbh_fatplus_core<1>(fma(b, s_z[s], a));
    s1 += bbh_fatplus_core<1>(fma(b, s_z[s + 1], a));
    s2 += bbh_fatplus_core<1>(fma(b, s_z[s + 2], a));
    s3 += bbh_fatplus_core<1>(fma(b, s_z[s + 3], a));
  }
  for (; s < S; s++) s0 += bbh_fatplus_core<1>(fma(b, s_z[s], a));
  const double sum = (s0 + s1) + (s2 + s3);
  scores[i] = log(TAU_RELU) + log(sum) - log((double)S);
}

// q' = 1 + p with pending points.  Thread-private packed lower-triangular Cholesky in LDS
// (element e of thread t at s_L[e * 64 + t]); exact psd_safe_cholesky semantics (jitter retries).
#define QMAX 16
#define QTRI (QMAX * (QMAX + 1) / 2)
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }

__global__ __launch_bounds__(64) void bbh_qlogei_pending_kernel(
    const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ cross, int64_t N, int p,
    const double* __restrict__ mean_p, const double* __restrict__ cov_pp, const double* __restrict__ z, int S,
    double best_f, double sign, const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  __shared__ double s_L[QTRI * 64];
  __shared__ double s_mp[QMAX];
  __shared__ double s_cpp[QMAX * QMAX];
  const int t = threadIdx.x;
  const int q = p + 1;
  for (int e = t; e < p; e += 64) s_mp[e] = mean_p[e];
  for (int e = t; e < p * p; e += 64) s_cpp[e] = cov_pp[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 64 + t;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  double* L = s_L + t;  // L[tri(i,j) * 64]
  const double v0 = var[i];
  // Cholesky of Sigma = [[v0, c^T], [c, cov_pp]] with diagonal jitter retries
  double jitter = 0.0;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; attempt++) {
    ok = true;
    for (int r = 0; r < q && ok; r++) {
      for (int c = 0; c <= r; c++) {
        double s;
        if (r == 0)
          s = v0;
        else if (c == 0)
          s = cross[i * p + (r - 1)];
        else
          s = s_cpp[(r - 1) * p + (c - 1)];
        if (r == c) s += jitter;
        for (int k = 0; k < c; k++) s -= L[tri(r, k) * 64] * L[tri(c, k) * 64];
        if (r == c) {
          if (!(s > 0.0)) {
            ok = false;
            break;
          }
          L[tri(r, r) * 64] = sqrt(s);
        } else {
          L[tri(r, c) * 64] = s / L[tri(c, c) * 64];
        }
      }
    }
    if (!ok) jitter = 1e-8 * pow(10.0, (double)attempt);
  }
  if (!ok) {
    scores[i] = NAN;  // not PSD even with jitter 1e-6 (gpytorch raises NotPSDError)
    return;
  }
  const double m0 = mean[i];
  const double inv_tau = 1.0 / TAU_RELU;
  double sum = 0.0;  // sum_s exp(fatmax_s - ref), streaming log-sum-exp
  double ref = -INFINITY;
  for (int s = 0; s < S; s++) {
    const double* zs = z + (int64_t)s * q;
    // li_j = log_fatplus(sign * y_j - best_f), y = m + L z
    double li[QMAX];
    double mx = -INFINITY;
#pragma unroll 1
    for (int r = 0; r < q; r++) {
      double y = (r == 0) ? m0 : s_mp[r - 1];
      for (int c = 0; c <= r; c++) y = fma(L[tri(r, c) * 64], zs[c], y);
      const double tt = (sign * y - best_f) * inv_tau;
      const double v = log(TAU_RELU) + log(bbh_fatplus_core(tt));
      li[r] = v;
      mx = fmax(mx, v);
    }
    // fatmax over the q' points: mx + tau log sum_j (2 / (2 + (mx - li_j)/tau))^2
    double acc = 0.0;
#pragma unroll 1
    for (int r = 0; r < q; r++) {
      const double u = 2.0 / (2.0 + (mx - li[r]) / TAU_MAX);
      acc = fma(u, u, acc);
    }
    const double fm = mx + TAU_MAX * log(acc);
    if (fm > ref) {
      sum = sum * exp(ref - fm) + 1.0;
      ref = fm;
    } else {
      sum += exp(fm - ref);
    }
  }
  scores[i] = ref + log(sum) - log((double)S);
}

// log(x) for positive, finite, normal x (the fat-softplus values and sums of squares below): exponent and
// mantissa by v_frexp, m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m - 1)/(m + 1) through a
// v_rcp_f64 seed with two Newton steps, odd series to s^19 (|s| <= 0.1716: truncation 2e-17), two-word ln 2.
// ~26 VALU instead of ~40 for the library call; relative error <= 3e-16 (also next to x = 1, where the
// result is 2 s (1 + R) with no cancellation).
__device__ __forceinline__ double bbh_fast_recip(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = fma(fma(-d, y, 1.0), y, y);
  return fma(fma(-d, y, 1.0), y, y);
}
__device__ __forceinline__ double bbh_fast_log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lowm = m < 0.70710678118654752440;
  m = lowm ? 2.0 * m : m;
  e = lowm ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * bbh_fast_recip(2.0 + f);
  const double z = s * s;
  double r = fma(z, 1.0 / 19.0, 1.0 / 17.0);
  r = fma(r, z, 1.0 / 15.0);
  r = fma(r, z, 1.0 / 13.0);
  r = fma(r, z, 1.0 / 11.0);
  r = fma(r, z, 1.0 / 9.0);
  r = fma(r, z, 1.0 / 7.0);
  r = fma(r, z, 1.0 / 5.0);
  r = fma(r, z, 1.0 / 3.0);
  const double lm = fma(2.0 * s * z, r, 2.0 * s);
  const double ed = (double)e;
  const double r2 = fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
  return (x < INFINITY) ? r2 : x;  // log(inf) = inf (NaN propagates)
}

// Register-resident form of the kernel above for q' = Q <= 14 (208 VGPRs at Q = 8, 2 waves per SIMD from Q = 9,
// 1 from Q = 10, no spills up to Q = 14; Q = 15, 16 spill and run no faster than the LDS form): the packed Cholesky
// factor (Q (Q + 1) / 2 doubles) and the per-sample values live in registers (every index is a
// compile-time constant), the base samples z [S, Q] in LDS (broadcast reads).  The LDS form needs 69 KB
// per 64 threads - one wave per two SIMDs - and took 30 ms per greedy step on 1e6 candidates, six times the
// fused posterior; this one runs 256-thread workgroups at full occupancy.  The arithmetic (operation order
// included) is that of the LDS form, so both give bit-identical scores.
// SAMPLE SLICES (gridDim.y > 1, linear-domain form only): blockIdx.y takes the samples [y per, (y + 1) per) and writes the partial sum
// of its terms to partial[y][i]; bbh_pending_finish_kernel adds the slices in a fixed order and takes the logarithm.  With one
// thread per candidate a 1e5-row candidate set is 1563 wavefronts, 1.5 per SIMD, each one long dependent chain: the slices bring
// the launch to the occupancy a 1e6-row set has (the joint Cholesky factor is recomputed per slice: Q^3 / 3 flops against
// per x Q x 40).
template <int Q>
__global__ __launch_bounds__(256) void bbh_qlogei_pending_q_kernel(
    const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ cross, int64_t N,
    const double* __restrict__ mean_p, const double* __restrict__ cov_pp, const double* __restrict__ z, int S_total,
    double best_f, double sign, const uint8_t* __restrict__ alive, double* __restrict__ scores, double* __restrict__ partial) {
  extern __shared__ double s_zq[];  // [S * Q] base samples of this slice, then mean_p [Q - 1], cov_pp [(Q - 1)^2]
  constexpr int P = Q - 1;
  const int per = (S_total + (int)gridDim.y - 1) / (int)gridDim.y;
  const int s_begin = (int)blockIdx.y * per;
  const int S = (s_begin + per <= S_total) ? per : (S_total > s_begin ? S_total - s_begin : 0);  // samples of this slice
  z += (int64_t)s_begin * Q;
  if (partial) scores = partial + (int64_t)blockIdx.y * N;
  double* s_mp = s_zq + (int64_t)per * Q;
  double* s_cpp = s_mp + P;
  for (int e = threadIdx.x; e < S * Q; e += 256) s_zq[e] = z[e];
  for (int e = threadIdx.x; e < P; e += 256) s_mp[e] = mean_p[e];
  for (int e = threadIdx.x; e < P * P; e += 256) s_cpp[e] = cov_pp[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = partial ? 0.0 : -INFINITY;
    return;
  }
  double L[Q * (Q + 1) / 2];
  double A[Q * (Q + 1) / 2];  // lower triangle of Sigma = [[v0, c^T], [c, cov_pp]]
  A[0] = var[i];
#pragma unroll
  for (int r = 1; r < Q; r++) {
    A[r * (r + 1) / 2] = cross[i * P + (r - 1)];
#pragma unroll
    for (int c = 1; c <= r; c++) A[r * (r + 1) / 2 + c] = s_cpp[(r - 1) * P + (c - 1)];
  }
  double jitter = 0.0;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; attempt++) {
    ok = true;
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
      for (int c = 0; c <= r; c++) {
        double sacc = A[r * (r + 1) / 2 + c];
        if (r == c) sacc += jitter;
#pragma unroll
        for (int k = 0; k < c; k++) sacc -= L[r * (r + 1) / 2 + k] * L[c * (c + 1) / 2 + k];
        if (r == c) {
          if (!(sacc > 0.0)) ok = false;  // the rest of this attempt is discarded (values may be NaN)
          L[r * (r + 1) / 2 + r] = sqrt(sacc);
        } else {
          L[r * (r + 1) / 2 + c] = sacc / L[c * (c + 1) / 2 + c];
        }
      }
    }
    if (!ok) jitter = 1e-8 * pow(10.0, (double)attempt);
  }
  if (!ok) {
    scores[i] = NAN;  // not PSD even with jitter 1e-6 (gpytorch raises NotPSDError)
    return;
  }
  double m[Q];
  m[0] = mean[i];
#pragma unroll
  for (int r = 1; r < Q; r++) m[r] = s_mp[r - 1];
  const double inv_tau = 1.0 / TAU_RELU;
  // exp(fatmax_s) = exp(mx_s) acc_s^tau_max with exp(mx_s) = tau_relu max_r fatplus_core(t_sr): the per-sample
  // exponential of the streaming log-sum-exp and the logarithm of the largest term are not needed.  The terms
  // tau_relu^-1 exp(fatmax_s) lie in [1e-41, 1e7) for every reachable t, so the plain sum neither overflows nor
  // loses its smallest terms to underflow (BBH_PENDING_LSE=1 at compile time restores the streaming form).
  double sum = 0.0;
#if BBH_PENDING_LSE
  double ref = -INFINITY;
#endif
#if BBH_PENDING_FAST && !BBH_PENDING_LSE
  // exp(fatmax_s) / tau_relu = fmx acc^tau_max with fmx = max_r fatplus(t_sr) and acc = sum_r u_r^2, u_r = 2 tau / (2 tau + D_r),
  // D_r = log(fmx / fatplus(t_sr)) >= 0.  fmx enters linearly and is kept to full precision, but acc enters through the power
  // tau_max = 0.01: an absolute error of 1e-7 in u_r^2 changes the sample's term by 1e-9 relative.  So D_r never needs a
  // double-precision logarithm:  rho = fatplus_r / fmx, x = 1 - rho;  x < 0.15: D = -log1p(-x) = x + x^2/2 + ... + x^7/7
  // (truncation 3e-8 at the edge, where u = 0.11 and du^2/dD = 100 u^3 = 0.13);  otherwise D = -ln 2 log2f(rho) in single
  // precision (error ~1e-7; rho below the float range gives D = inf, u = 0: u^2 < 6e-8 there anyway).  The reciprocal is the
  // bare v_rcp_f64 seed (2^-26), log(acc) for acc in [1, Q] single precision as well.  Per sample and point ~33 instead of
  // ~55 VALU instructions, per sample 13 instead of 34 (BBH_PENDING_FAST=0 at compile time restores the former sequence).
  for (int s = 0; s < S; s++) {
    const double* zs = s_zq + (int64_t)s * Q;
    double zr[Q], fp[Q];
#pragma unroll
    for (int c = 0; c < Q; c++) zr[c] = zs[c];
    double fmx = 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      double y = m[r];
#pragma unroll
      for (int c = 0; c <= r; c++) y = fma(L[r * (r + 1) / 2 + c], zr[c], y);
      const double tt = (sign * y - best_f) * inv_tau;
      fp[r] = bbh_fatplus_core<1>(tt);
      fmx = fmax(fmx, fp[r]);
    }
#if BBH_PENDING_PK
    // The same in packed single precision, two points per instruction: rho_r = fp_r / fmx and x_r = (fmx - fp_r) / fmx as two
    // single-precision quotients (the difference in double precision: exact next to rho = 1, where u^2 is sensitive) sharing one
    // v_rcp_f32; series, log2, reciprocal and square on float2 values.  u^2 <= 1 to ~1e-7 relative: 1e-9 in the sample's term.
    const float rcf = __builtin_amdgcn_rcpf((float)fmx);
    bbh_f2 accp = {0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < Q; r += 2) {
      const int r1 = (r + 1 < Q) ? r + 1 : r;
      bbh_f2 x, rho;
      x.x = (float)(fmx - fp[r]);
      x.y = (float)(fmx - fp[r1]);
      rho.x = (float)fp[r];
      rho.y = (float)fp[r1];
      x = x * rcf;
      rho = rho * rcf;
      bbh_f2 ds = x * (1.0f / 7.0f) + (1.0f / 6.0f);
      ds = ds * x + 0.2f;
      ds = ds * x + 0.25f;
      ds = ds * x + (1.0f / 3.0f);
      ds = ds * x + 0.5f;
      ds = ds * x + 1.0f;
      ds = ds * x;
      bbh_f2 den;
      den.x = ((x.x < 0.15f) ? ds.x : -0.6931471805599453f * __log2f(rho.x)) + (float)(2.0 * TAU_MAX);
      den.y = ((x.y < 0.15f) ? ds.y : -0.6931471805599453f * __log2f(rho.y)) + (float)(2.0 * TAU_MAX);
      bbh_f2 u;
      u.x = __builtin_amdgcn_rcpf(den.x);  // den = inf (rho below the float range) -> 0
      u.y = (r + 1 < Q) ? __builtin_amdgcn_rcpf(den.y) : 0.0f;  // odd Q: the last point's twin does not count
      u = u * (float)(2.0 * TAU_MAX);
      accp = u * u + accp;
    }
    const float accf = accp.x + accp.y;
    const double w = (TAU_MAX * 0.6931471805599453) * (double)__log2f(accf);
    double e = fma(w, 1.0 / 120.0, 1.0 / 24.0);
    e = fma(e, w, 1.0 / 6.0);
    e = fma(e, w, 0.5);
    e = fma(e, w, 1.0);
    e = fma(e, w, 1.0);
    sum = fma(fmx, e, sum);
#else
    double inv = __builtin_amdgcn_rcp(fmx);
    inv = fma(fma(-fmx, inv, 1.0), inv, inv);
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      const double rho = fp[r] * inv;
      const double x = 1.0 - rho;
      double ds = fma(x, 1.0 / 7.0, 1.0 / 6.0);
      ds = fma(ds, x, 0.2);
      ds = fma(ds, x, 0.25);
      ds = fma(ds, x, 1.0 / 3.0);
      ds = fma(ds, x, 0.5);
      ds = fma(ds, x, 1.0);
      ds *= x;
      const double dl = -0.6931471805599453 * (double)__log2f((float)rho);
      const double dd = (x < 0.15) ? ds : dl;
      const double u = (2.0 * TAU_MAX) * __builtin_amdgcn_rcp(2.0 * TAU_MAX + dd);  // dd = inf -> 0
      acc = fma(u, u, acc);
    }
    // acc in [1, Q]: acc^0.01 = exp(w), w = 0.01 ln acc in [0, 0.028): Taylor to the 5th power (7e-13 at w = 0.028)
    const double w = (TAU_MAX * 0.6931471805599453) * (double)__log2f((float)acc);
    double e = fma(w, 1.0 / 120.0, 1.0 / 24.0);
    e = fma(e, w, 1.0 / 6.0);
    e = fma(e, w, 0.5);
    e = fma(e, w, 1.0);
    e = fma(e, w, 1.0);
    sum = fma(fmx, e, sum);
#endif
  }
#else
  for (int s = 0; s < S; s++) {
    const double* zs = s_zq + (int64_t)s * Q;
    double zr[Q], li[Q], fp[Q];
#pragma unroll
    for (int c = 0; c < Q; c++) zr[c] = zs[c];
    double fmx = 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      double y = m[r];
#pragma unroll
      for (int c = 0; c <= r; c++) y = fma(L[r * (r + 1) / 2 + c], zr[c], y);
      const double tt = (sign * y - best_f) * inv_tau;
      fp[r] = bbh_fatplus_core(tt);
      fmx = fmax(fmx, fp[r]);
    }
    double mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      li[r] = bbh_fast_log_pos(fp[r]);
      mx = fmax(mx, li[r]);
    }
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      const double u = (2.0 * TAU_MAX) * bbh_fast_recip(2.0 * TAU_MAX + (mx - li[r]));
      acc = fma(u, u, acc);
    }
#if BBH_PENDING_LSE
    const double fm = fma(TAU_MAX, bbh_fast_log_pos(acc), mx + LOG_TAU_RELU);
    if (fm > ref) {
      sum = sum * exp(ref - fm) + 1.0;
      ref = fm;
    } else {
      sum += exp(fm - ref);
    }
#else
    // acc in [1, Q]: acc^0.01 = exp(0.01 log acc), argument in [0, 0.021) - Taylor to the 6th power (1e-17)
    const double w = TAU_MAX * bbh_fast_log_pos(acc);
    double e = fma(w, 1.0 / 720.0, 1.0 / 120.0);
    e = fma(e, w, 1.0 / 24.0);
    e = fma(e, w, 1.0 / 6.0);
    e = fma(e, w, 0.5);
    e = fma(e, w, 1.0);
    e = fma(e, w, 1.0);
    sum = fma(fmx, e, sum);
#endif
  }
#endif
#if BBH_PENDING_LSE
  scores[i] = ref + log(sum) - log((double)S);
#else
  scores[i] = partial ? sum : LOG_TAU_RELU + log(sum) - log((double)S_total);
#endif
}

__global__ __launch_bounds__(256) void bbh_pending_finish_kernel(const double* __restrict__ partial, int slices, int64_t N, int S,
                                                                 const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double total = 0.0;
  for (int k = 0; k < slices; k++) total += partial[(int64_t)k * N + i];  // NaN (no positive definite factor) propagates
  scores[i] = (alive && !alive[i]) ? -INFINITY : LOG_TAU_RELU + log(total) - log((double)S);
}

template <int Q>
static void bbh_launch_pending_q(hipStream_t st, const double* mean, const double* var, const double* cross, int64_t N,
                                 const double* mp, const double* cpp, const double* z, int S, double best_f, double sign,
                                 const uint8_t* alive, double* scores, int slices, double* partial) {
  const int per = (S + slices - 1) / slices;
  const size_t lds = sizeof(double) * ((size_t)per * Q + (Q - 1) + (size_t)(Q - 1) * (Q - 1));
  hipLaunchKernelGGL((bbh_qlogei_pending_q_kernel<Q>), dim3((unsigned)((N + 255) / 256), (unsigned)slices), dim3(256), lds, st, mean, var,
                     cross, N, mp, cpp, z, S, best_f, sign, alive, scores, slices > 1 ? partial : nullptr);
  if (slices > 1)
    hipLaunchKernelGGL(bbh_pending_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, partial, slices, N, S, alive, scores);
}

// ---- first-index argmax -----------------------------------------------------------------------
__device__ __forceinline__ void amax_combine(double& v, int64_t& i, double ov, int64_t oi) {
  // NaN never wins; ties -> lower index
  const bool better = (ov > v) || (ov == v && oi < i) || (i < 0 && oi >= 0);
  if (oi >= 0 && better) {
    v = ov;
    i = oi;
  }
}

__global__ __launch_bounds__(256) void bbh_argmax_stage1(const double* __restrict__ scores, int64_t N,
                                                         double* __restrict__ pv, int64_t* __restrict__ pi) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  double v = -INFINITY;
  int64_t idx = -1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const double x = scores[i];
    if (x == x) amax_combine(v, idx, x, i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_down(v, o, 64);
    const int64_t oi = __shfl_down(idx, o, 64);
    amax_combine(v, idx, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) amax_combine(v, idx, sv[w], si[w]);
    pv[blockIdx.x] = v;
    pi[blockIdx.x] = idx;
  }
}

__global__ __launch_bounds__(256) void bbh_argmax_stage2(const double* __restrict__ pv, const int64_t* __restrict__ pi,
                                                         int nblocks, double* __restrict__ outv,
                                                         int64_t* __restrict__ outi) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  double v = -INFINITY;
  int64_t idx = -1;
  for (int b = threadIdx.x; b < nblocks; b += 256) amax_combine(v, idx, pv[b], pi[b]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_down(v, o, 64);
    const int64_t oi = __shfl_down(idx, o, 64);
    amax_combine(v, idx, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) amax_combine(v, idx, sv[w], si[w]);
    *outv = v;
    *outi = idx;
  }
}

__global__ void bbh_mask_one_kernel(double* scores, const int64_t* idx) {
  if (*idx >= 0) scores[*idx] = -INFINITY;
}

#define ARGMAX_BLOCKS 1024

static int bbh_ensure_red(bbh_handle* h) {
  if (!h->d_red) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_red, sizeof(double) * (ARGMAX_BLOCKS + 64)));
  if (!h->d_redi) BBH_HIP_TRY(h, hipMalloc((void**)&h->d_redi, sizeof(int64_t) * (ARGMAX_BLOCKS + 64)));
  return 0;
}

// The handle's pinned staging buffer: at least `bytes` large and no longer read by an earlier staged copy.  The caller fills it,
// enqueues its hipMemcpyAsync calls from it on h->stream and then calls bbh_stage_done.  (A synchronous hipMemcpy from pageable
// memory is not an alternative: four 4 KB copies of bbh_set_model_ex took 13 - 26 ms in a process holding a 1e5-row search space -
// profiles/r06_set_model_probe.log.)
void* bbh_stage_pinned(bbh_handle* h, size_t bytes) {
  if (!h->z_evt) {
    if (hipEventCreateWithFlags(&h->z_evt, hipEventDisableTiming) != hipSuccess) return nullptr;
  } else if (hipEventSynchronize(h->z_evt) != hipSuccess) {  // previous staged copy has been consumed
    return nullptr;
  }
  if (bytes > h->zstage_bytes) {
    if (h->h_zstage) hipHostFree(h->h_zstage);
    h->h_zstage = nullptr;
    h->zstage_bytes = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes;
    if (hipHostMalloc((void**)&h->h_zstage, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    h->zstage_bytes = want;
  }
  return h->h_zstage;
}

int bbh_stage_done(bbh_handle* h) {
  BBH_HIP_TRY(h, hipEventRecord(h->z_evt, h->stream));
  return 0;
}

int bbh_upload_z(bbh_handle* h, const double* z_host, size_t count) {
  const size_t bytes = sizeof(double) * count;
  if (bytes > h->z_bytes) {
    if (h->d_z) hipFree(h->d_z);
    h->d_z = nullptr;
    h->z_bytes = 0;
    BBH_HIP_TRY(h, hipMalloc((void**)&h->d_z, bytes));
    h->z_bytes = bytes;
  }
  // Staged through a pinned buffer of the handle, so that the call neither waits for the kernels already in
  // the stream (a synchronous copy would: it is ordered behind them) nor depends on the caller's buffer.
  void* stage = bbh_stage_pinned(h, bytes);
  if (!stage) {
    h->err = "bbh_upload_z: no pinned staging buffer";
    (void)hipGetLastError();
    return -2;
  }
  memcpy(stage, z_host, bytes);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_z, stage, bytes, hipMemcpyHostToDevice, h->stream));
  return bbh_stage_done(h);
}

int bbh_qlogei_q1_sliced(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host, int64_t S,
                         double best_f, double sign, const uint8_t* alive_dev, double* scores_dev);  // bbh_select.hip
int bbh_qlogei_q1_rounds(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host, int64_t S,
                         double best_f, double sign, const uint8_t* alive_dev, double* scores_dev);

extern "C" int bbh_qlogei_q1(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N,
                             const double* z_host, int64_t S, double best_f, double sign, const uint8_t* alive_dev,
                             double* scores_dev) {
  if (!h) return -1;
  if (!mean_dev || !var_dev || !z_host || !scores_dev || N < 0 || S < 1 || S > 8192) {
    h->err = "bbh_qlogei_q1: bad arguments (1 <= S <= 8192)";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (h->q1_sliced) {  // sample-sliced form (bbh_select.hip); 1 = does not apply (S beyond its LDS tables)
    const int rc = bbh_qlogei_q1_sliced(h, mean_dev, var_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev);
    if (rc <= 0) return rc;
  }
  return bbh_qlogei_q1_rounds(h, mean_dev, var_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev);
}

// one thread per candidate over all S samples (any S; A/B partner of the sliced form: BBH_Q1_SLICED=0)
int bbh_qlogei_q1_rounds(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N, const double* z_host, int64_t S,
                         double best_f, double sign, const uint8_t* alive_dev, double* scores_dev) {
  int rc = bbh_upload_z(h, z_host, (size_t)S);
  if (rc) return rc;
  bbh_timed_scope timed(h, BBH_TIMED_Q1);
  hipLaunchKernelGGL(bbh_qlogei_q1_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), sizeof(double) * S, h->stream,
                     mean_dev, var_dev, N, h->d_z, (int)S, best_f, sign, alive_dev, scores_dev);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// p <= 15 pending points with the pending statistics given explicitly (the handle's own, or a caller's - see
// bbh_qlogei_pending_big)
static int bbh_qlogei_pending_impl(bbh_handle* h, const double* mean_dev, const double* var_dev, const double* cross_dev, int64_t N,
                                   int p, const double* mp_host, const double* cpp_host, const double* z_host, int64_t S,
                                   double best_f, double sign, const uint8_t* alive_dev, double* scores_dev);

extern "C" int bbh_qlogei_pending(bbh_handle* h, const double* mean_dev, const double* var_dev,
                                  const double* cross_dev, int64_t N, const double* z_host, int64_t S, double best_f,
                                  double sign, const uint8_t* alive_dev, double* scores_dev) {
  if (!h) return -1;
  if (h->p < 1) {
    h->err = "bbh_qlogei_pending: bad arguments / no pending points set";
    return -1;
  }
  return bbh_qlogei_pending_impl(h, mean_dev, var_dev, cross_dev, N, h->p, h->pend_mean.data(), h->pend_cov.data(), z_host, S, best_f,
                                 sign, alive_dev, scores_dev);
}

static int bbh_qlogei_pending_impl(bbh_handle* h, const double* mean_dev, const double* var_dev, const double* cross_dev, int64_t N,
                                   int p, const double* mp_host, const double* cpp_host, const double* z_host, int64_t S,
                                   double best_f, double sign, const uint8_t* alive_dev, double* scores_dev) {
  if (!mean_dev || !var_dev || !cross_dev || !z_host || !scores_dev || !mp_host || !cpp_host || N < 0 || S < 1 || p < 1 ||
      p > BBH_MAX_PENDING) {
    h->err = "bbh_qlogei_pending: bad arguments / no pending points set";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  // device buffer: z [S, 1+p] followed by the pending posterior mean_p [p], cov_pp [p,p]
  int rc;
  std::vector<double> buf((size_t)S * (p + 1) + p + (size_t)p * p);
  memcpy(buf.data(), z_host, sizeof(double) * S * (p + 1));
  memcpy(buf.data() + S * (p + 1), mp_host, sizeof(double) * p);
  memcpy(buf.data() + S * (p + 1) + p, cpp_host, sizeof(double) * p * p);
  rc = bbh_upload_z(h, buf.data(), buf.size());
  if (rc) return rc;
  const double* dz = h->d_z;
  const double* dmp = dz + S * (p + 1);
  const double* dcpp = dmp + p;
  const bool fits = sizeof(double) * ((size_t)S * (p + 1) + p + (size_t)p * p) <= 60 * 1024;  // z in LDS
  // sample slices for small candidate sets (~16 waves per SIMD; linear-domain form; each slice at least 32 samples)
  int slices = 1;
#if BBH_PENDING_FAST && !BBH_PENDING_LSE
  if (fits && !h->pending_lds_form && p + 1 <= 14) {
    int64_t want = ((int64_t)16 * 4 * h->num_cu * 64) / (h->slice_rows > 0 ? h->slice_rows : N);
    if (const char* e = getenv("BBH_PENDING_SLICES")) want = atoi(e);
    if (want > S / 32) want = S / 32;
    slices = (int)(want < 1 ? 1 : (want > 32 ? 32 : want));
    if (slices > 1) {
      rc = bbh_ensure_ws(h, sizeof(double) * (size_t)slices * (size_t)N);
      if (rc) return rc;
    }
  }
#endif
  bbh_timed_scope timed(h, BBH_TIMED_PENDING);
#define BBH_PENDING_Q(QV)                                                                                        \
  case QV:                                                                                                       \
    bbh_launch_pending_q<QV>(h->stream, mean_dev, var_dev, cross_dev, N, dmp, dcpp, dz, (int)S, best_f, sign,    \
                             alive_dev, scores_dev, slices, h->d_ws);                                            \
    break;
  switch (fits && !h->pending_lds_form ? p + 1 : 0) {
    BBH_PENDING_Q(2)
    BBH_PENDING_Q(3)
    BBH_PENDING_Q(4)
    BBH_PENDING_Q(5)
    BBH_PENDING_Q(6)
    BBH_PENDING_Q(7)
    BBH_PENDING_Q(8)
    BBH_PENDING_Q(9)
    BBH_PENDING_Q(10)
    BBH_PENDING_Q(11)
    BBH_PENDING_Q(12)
    BBH_PENDING_Q(13)
    BBH_PENDING_Q(14)
    default:
      hipLaunchKernelGGL(bbh_qlogei_pending_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, h->stream, mean_dev,
                         var_dev, cross_dev, N, p, dmp, dcpp, dz, (int)S, best_f, sign, alive_dev, scores_dev);
  }
#undef BBH_PENDING_Q
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

// ---- joint q'-batches beyond 16 points ------------------------------------------------------------------------------------
// The reference has no cap on batch_size + pending experiments (botorch/discrete.py:120-126); the register form above holds
// q' <= 14, the LDS form q' <= 16.  For larger q' the thread-private Cholesky factor (q' (q' + 1) / 2 doubles: 4.2 KB at q' = 32)
// lives in a global workspace, element e of candidate i at Lws[e N + i] (coalesced across the wave), the per-sample
// log-fat-softplus values in LDS ([q'][64]); mean_p / cov_pp / z are read from global memory at wave-uniform addresses.  Same
// arithmetic and operation order as bbh_qlogei_pending_kernel (log-domain streaming form, psd_safe_cholesky's jitter retries).
// Memory-bound on the factor (q'^2 / 2 loads per sample): ~0.1 s per 1e5 candidates at q' = 32 - a correctness path for the
// rare large batch, not a tuned one.
#define QBIG_MAX 64
#define QBIG_WS_BYTES ((size_t)512 << 20)  // workspace bound of bbh_qlogei_pending_big (env BBH_QBIG_WS_MB for tests)
__global__ __launch_bounds__(64) void bbh_qlogei_pending_big_kernel(
    const double* __restrict__ mean, const double* __restrict__ var, const double* __restrict__ cross, int64_t N, int p,
    const double* __restrict__ mean_p, const double* __restrict__ cov_pp, const double* __restrict__ z, int S,
    double best_f, double sign, const uint8_t* __restrict__ alive, double* __restrict__ scores, double* __restrict__ Lws) {
  extern __shared__ double s_li[];  // [q][64]
  const int t = threadIdx.x;
  const int q = p + 1;
  const int64_t i = (int64_t)blockIdx.x * 64 + t;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  double* L = Lws + i;  // L[tri(r, c) * N]
  const double v0 = var[i];
  double jitter = 0.0;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; attempt++) {
    ok = true;
    for (int r = 0; r < q && ok; r++) {
      for (int c = 0; c <= r; c++) {
        double sacc;
        if (r == 0)
          sacc = v0;
        else if (c == 0)
          sacc = cross[i * p + (r - 1)];
        else
          sacc = cov_pp[(r - 1) * p + (c - 1)];
        if (r == c) sacc += jitter;
        for (int k = 0; k < c; k++) sacc -= L[(int64_t)tri(r, k) * N] * L[(int64_t)tri(c, k) * N];
        if (r == c) {
          if (!(sacc > 0.0)) {
            ok = false;
            break;
          }
          L[(int64_t)tri(r, r) * N] = sqrt(sacc);
        } else {
          L[(int64_t)tri(r, c) * N] = sacc / L[(int64_t)tri(c, c) * N];
        }
      }
    }
    if (!ok) jitter = 1e-8 * pow(10.0, (double)attempt);
  }
  if (!ok) {
    scores[i] = NAN;
    return;
  }
  const double m0 = mean[i];
  const double inv_tau = 1.0 / TAU_RELU;
  double sum = 0.0, ref = -INFINITY;
  for (int s = 0; s < S; s++) {
    const double* zs = z + (int64_t)s * q;
    double mx = -INFINITY;
    for (int r = 0; r < q; r++) {
      double y = (r == 0) ? m0 : mean_p[r - 1];
      for (int c = 0; c <= r; c++) y = fma(L[(int64_t)tri(r, c) * N], zs[c], y);
      const double tt = (sign * y - best_f) * inv_tau;
      const double v = log(TAU_RELU) + log(bbh_fatplus_core(tt));
      s_li[r * 64 + t] = v;
      mx = fmax(mx, v);
    }
    double acc = 0.0;
    for (int r = 0; r < q; r++) {
      const double u = 2.0 / (2.0 + (mx - s_li[r * 64 + t]) / TAU_MAX);
      acc = fma(u, u, acc);
    }
    const double fm = mx + TAU_MAX * log(acc);
    if (fm > ref) {
      sum = sum * exp(ref - fm) + 1.0;
      ref = fm;
    } else {
      sum += exp(fm - ref);
    }
  }
  scores[i] = ref + log(sum) - log((double)S);
}

extern "C" int bbh_qlogei_pending_big(bbh_handle* h, const double* mean_dev, const double* var_dev, const double* cross_dev,
                                      int64_t N, int64_t p, const double* mean_p_host, const double* cov_pp_host,
                                      const double* z_host, int64_t S, double best_f, double sign, const uint8_t* alive_dev,
                                      double* scores_dev) {
  if (!h) return -1;
  if (!mean_dev || !var_dev || !cross_dev || !mean_p_host || !cov_pp_host || !z_host || !scores_dev || N < 0 || S < 1 || p < 1 ||
      p + 1 > QBIG_MAX) {
    h->err = "bbh_qlogei_pending_big: bad arguments (1 <= p <= 63 pending points)";
    return -1;
  }
  if (N == 0) return 0;
  if (p <= BBH_MAX_PENDING)  // the register / LDS kernels, with the caller's statistics instead of the handle's
    return bbh_qlogei_pending_impl(h, mean_dev, var_dev, cross_dev, N, (int)p, mean_p_host, cov_pp_host, z_host, S, best_f, sign,
                                   alive_dev, scores_dev);
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const int64_t q = p + 1;
  std::vector<double> buf((size_t)S * q + p + (size_t)p * p);
  memcpy(buf.data(), z_host, sizeof(double) * S * q);
  memcpy(buf.data() + S * q, mean_p_host, sizeof(double) * p);
  memcpy(buf.data() + S * q + p, cov_pp_host, sizeof(double) * p * p);
  int rc = bbh_upload_z(h, buf.data(), buf.size());
  if (rc) return rc;
  // The per-candidate factor workspace is q'(q' + 1) / 2 doubles: 16 GB for 1e6 candidates at q' = 64.  The candidates are walked in
  // chunks whose workspace stays below QBIG_WS_BYTES (the handle's grow-only workspace outlives the call, also in the handle pool);
  // chunks are multiples of 64 rows, every chunk addresses its own rows from 0.
  const int64_t tri_q = q * (q + 1) / 2;
  size_t ws_bound = QBIG_WS_BYTES;
  if (const char* e = getenv("BBH_QBIG_WS_MB")) ws_bound = (size_t)std::max(1LL, atoll(e)) << 20;
  int64_t chunk = (int64_t)(ws_bound / (sizeof(double) * (size_t)tri_q)) / 64 * 64;
  if (chunk < 64) chunk = 64;
  if (chunk > N) chunk = N;
  rc = bbh_ensure_ws(h, sizeof(double) * (size_t)tri_q * (size_t)chunk);
  if (rc) return rc;
  const double* dz = h->d_z;
  bbh_timed_scope timed(h, BBH_TIMED_PENDING);
  for (int64_t c0 = 0; c0 < N; c0 += chunk) {
    const int64_t nc = std::min(chunk, N - c0);
    hipLaunchKernelGGL(bbh_qlogei_pending_big_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(64), sizeof(double) * q * 64, h->stream,
                       mean_dev + c0, var_dev + c0, cross_dev + c0 * p, nc, (int)p, dz + S * q, dz + S * q + p, dz, (int)S, best_f, sign,
                       alive_dev ? alive_dev + c0 : nullptr, scores_dev + c0, h->d_ws);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

int bbh_argmax_rounds(bbh_handle* h, const double* scores_dev, int64_t N, double* best_val_host,
                      int64_t* best_idx_host) {
  if (!h) return -1;
  if (!scores_dev || N < 1 || !best_val_host || !best_idx_host) {
    h->err = "bbh_argmax: bad arguments";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  int rc = bbh_ensure_red(h);
  if (rc) return rc;
  int nblocks = (int)((N + 255) / 256);
  if (nblocks > ARGMAX_BLOCKS) nblocks = ARGMAX_BLOCKS;
  hipLaunchKernelGGL(bbh_argmax_stage1, dim3(nblocks), dim3(256), 0, h->stream, scores_dev, N, h->d_red, h->d_redi);
  hipLaunchKernelGGL(bbh_argmax_stage2, dim3(1), dim3(256), 0, h->stream, h->d_red, h->d_redi, nblocks,
                     h->d_red + ARGMAX_BLOCKS, h->d_redi + ARGMAX_BLOCKS);
  BBH_HIP_TRY(h, hipMemcpyAsync(best_val_host, h->d_red + ARGMAX_BLOCKS, sizeof(double), hipMemcpyDeviceToHost,
                                h->stream));
  BBH_HIP_TRY(h, hipMemcpyAsync(best_idx_host, h->d_redi + ARGMAX_BLOCKS, sizeof(int64_t), hipMemcpyDeviceToHost,
                                h->stream));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

// ---- single-pass top-k -----------------------------------------------------------------------------
// stage 1: each workgroup owns a chunk of <= TOPK_CHUNK scores in LDS and extracts its k best by k
// rounds of workgroup-argmax + mask; stage 2: one workgroup merges the (blocks x k) survivors.
#define TOPK_CHUNK 4096
#define TOPK_MAXK 64

__device__ __forceinline__ void bbh_wg_argmax(double& v, int64_t& idx, double* sv, int64_t* si) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_down(v, o, 64);
    const int64_t oi = __shfl_down(idx, o, 64);
    amax_combine(v, idx, ov, oi);
  }
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = v;
    si[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  v = sv[0];
  idx = si[0];
  for (int w = 1; w < 4; w++) amax_combine(v, idx, sv[w], si[w]);
  __syncthreads();
}

__global__ __launch_bounds__(256) void bbh_topk_stage1(const double* __restrict__ scores, int64_t N, int k,
                                                       double* __restrict__ pv, int64_t* __restrict__ pi) {
  __shared__ double s_val[TOPK_CHUNK];
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
  const int cnt = (int)((N - base < TOPK_CHUNK) ? N - base : TOPK_CHUNK);
  for (int e = threadIdx.x; e < TOPK_CHUNK; e += 256) s_val[e] = (e < cnt) ? scores[base + e] : NAN;
  __syncthreads();
  for (int j = 0; j < k; j++) {
    double v = -INFINITY;
    int64_t idx = -1;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      const double x = s_val[e];
      if (x == x) amax_combine(v, idx, x, base + e);
    }
    bbh_wg_argmax(v, idx, sv, si);
    if (threadIdx.x == 0) {
      pv[(int64_t)blockIdx.x * k + j] = v;
      pi[(int64_t)blockIdx.x * k + j] = idx;
      if (idx >= 0) s_val[idx - base] = NAN;  // taken
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bbh_topk_stage2(double* __restrict__ pv, const int64_t* __restrict__ pi, int64_t cnt,
                                                       int k, double* __restrict__ outv, int64_t* __restrict__ outi) {
  __shared__ double sv[4];
  __shared__ int64_t si[4];
  __shared__ int64_t s_pos;
  for (int j = 0; j < k; j++) {
    double v = -INFINITY;
    int64_t idx = -1, pos = -1;
    for (int64_t e = threadIdx.x; e < cnt; e += 256) {
      const double x = pv[e];
      const int64_t gi = pi[e];
      if (gi >= 0 && x == x) {
        const int64_t before = idx;
        amax_combine(v, idx, x, gi);
        if (idx != before) pos = e;
      }
    }
    double bv = v;
    int64_t bi = idx;
    bbh_wg_argmax(bv, bi, sv, si);
    if (idx == bi && bi >= 0 && pos >= 0) s_pos = pos;  // unique owner: global indices are distinct
    __syncthreads();
    if (threadIdx.x == 0) {
      outv[j] = bv;
      outi[j] = bi;
      if (bi >= 0) pv[s_pos] = NAN;
    }
    __syncthreads();
  }
}

// k best scores on the device by k rounds of workgroup argmax (the form before bbh_select.hip; BBH_SELECT=0 routes bbh_topk /
// bbh_argmax here for A/B): *vals_dev / *idx_dev point into the handle's workspace (valid until its next use)
int bbh_topk_rounds_device(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double** vals_dev, int64_t** idx_dev) {
  if (!scores_dev || N < 1 || k < 1 || k > N || k > TOPK_MAXK) {
    h->err = "bbh_topk: bad arguments (1 <= k <= min(N, 64))";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const int64_t blocks = (N + TOPK_CHUNK - 1) / TOPK_CHUNK;
  const size_t need = (sizeof(double) + sizeof(int64_t)) * (size_t)(blocks * k + k);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* pv = h->d_ws;
  double* outv = pv + blocks * k;
  int64_t* pi = (int64_t*)(outv + k);
  int64_t* outi = pi + blocks * k;
  hipLaunchKernelGGL(bbh_topk_stage1, dim3((unsigned)blocks), dim3(256), 0, h->stream, scores_dev, N, (int)k, pv, pi);
  hipLaunchKernelGGL(bbh_topk_stage2, dim3(1), dim3(256), 0, h->stream, pv, pi, blocks * k, (int)k, outv, outi);
  BBH_HIP_TRY(h, hipGetLastError());
  *vals_dev = outv;
  *idx_dev = outi;
  return 0;
}

int bbh_topk_rounds(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double* vals_host,
                    int64_t* idx_host) {
  if (!h) return -1;
  if (!vals_host || !idx_host) {
    h->err = "bbh_topk: bad arguments (1 <= k <= min(N, 64))";
    return -1;
  }
  double* outv = nullptr;
  int64_t* outi = nullptr;
  int rc = bbh_topk_rounds_device(h, scores_dev, N, k, &outv, &outi);
  if (rc) return rc;
  BBH_HIP_TRY(h, hipMemcpyAsync(vals_host, outv, sizeof(double) * k, hipMemcpyDeviceToHost, h->stream));
  BBH_HIP_TRY(h, hipMemcpyAsync(idx_host, outi, sizeof(int64_t) * k, hipMemcpyDeviceToHost, h->stream));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int bbh_score_qlogei(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, const double* z_host,
                                int64_t S, double best_f, double sign, const uint8_t* alive_dev, double* mean_dev,
                                double* var_dev, double* scores_dev) {
  if (!h) return -1;
  if (!h->factorized || N < 0 || (N > 0 && !X_dev) || ldx < h->desc.d || !z_host || S < 1 || S > 4096 || !scores_dev) {
    h->err = "bbh_score_qlogei: model not factorised / bad arguments (1 <= S <= 4096)";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  if (bbh_materialised_only(h)) {  // no fused form: materialised-K* posterior, then the stand-alone scoring kernel
    double *tm = mean_dev, *tv = var_dev, *scratch = nullptr;
    if (!tm || !tv) {
      BBH_HIP_TRY(h, hipMalloc((void**)&scratch, sizeof(double) * 2 * (size_t)N));
      tm = tm ? tm : scratch;
      tv = tv ? tv : scratch + N;
    }
    int rc2 = bbh_launch_fused(h, X_dev, N, ldx, tm, tv, nullptr, true);
    if (!rc2) rc2 = bbh_qlogei_q1(h, tm, tv, N, z_host, S, best_f, sign, alive_dev, scores_dev);
    if (scratch) {
      hipStreamSynchronize(h->stream);
      hipFree(scratch);
    }
    return rc2;
  }
  int rc = bbh_upload_z(h, z_host, (size_t)S);
  if (rc) return rc;
  h->fuse_qz = h->d_z;
  h->fuse_S = (int)S;
  h->fuse_best_f = best_f;
  h->fuse_sign = sign;
  h->fuse_alive = alive_dev;
  h->fuse_scores = scores_dev;
  rc = bbh_launch_fused(h, X_dev, N, ldx, mean_dev, var_dev, nullptr, true);
  h->fuse_qz = nullptr;
  h->fuse_scores = nullptr;
  return rc;
}

// =================================================================================================
// qLogNEHVI (q' = 1): per MC sample, log-sum over the cells of that sample's box decomposition of
//   sum_o fatmin( log_fatplus(f_o - lo_o; tau_relu), loglen_o; tau_max ),  then logmeanexp over samples.
// One thread per candidate; the cell data is uniform across the wave (scalar / broadcast loads).
// =================================================================================================
struct NehviArgs {
  const double* tmat[BBH_MAX_OBJECTIVES];
  const double* var[BBH_MAX_OBJECTIVES];
  double sign[BBH_MAX_OBJECTIVES];
  int m;
  int64_t N;
  int S;
  const double* zx;        // [S, m]
  const int64_t* cell_off; // [S + 1]
  const double* cell_lo;   // [ncells, m]
  const double* cell_ll;   // [ncells, m]
  int sample_major;        // tmat[o] is [S, N] (bbh_posterior_columns_sm) instead of [N, S]
  const uint8_t* alive;
  double* scores;
};

__device__ __forceinline__ double bbh_fatmin2(double a, double b) {
  // min(a,b) - tau log(1 + (alpha / (alpha + |a-b|/tau))^alpha), alpha = 2, tau = 1e-2
  const double mn = fmin(a, b);
  double diff = fabs(a - b);
  if (!(diff == diff)) diff = INFINITY;  // (-inf) - (-inf)
  // u = 2 / (2 + diff / tau) and log1p(u^2) through the short reciprocal / logarithm sequences (u^2 in (0, 1]:
  // forming 1 + u^2 costs at most 1e-16 absolute, times tau = 1e-2); diff = inf gives u = 0 exactly
  const double u = (diff < INFINITY) ? (2.0 * TAU_MAX) * bbh_fast_recip(fma(2.0, TAU_MAX, diff)) : 0.0;
  return fma(-TAU_MAX, bbh_fast_log_pos(fma(u, u, 1.0)), mn);
}

template <int M>
__global__ __launch_bounds__(256) void bbh_qlognehvi_kernel(const NehviArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  if (a.alive && !a.alive[i]) {
    a.scores[i] = -INFINITY;
    return;
  }
  double sd[M], sg[M];
  const double* trow[M];
#pragma unroll
  for (int o = 0; o < M; o++) {
    sd[o] = bbh_safe_sd(a.var[o][i]);
    sg[o] = a.sign[o];
    trow[o] = a.sample_major ? a.tmat[o] + i : a.tmat[o] + i * (int64_t)a.S;
  }
  const int64_t tstride = a.sample_major ? a.N : 1;
  const double inv_tau = 1.0 / TAU_RELU;
  const double log_tau = log(TAU_RELU);
  double sref = -INFINITY, ssum = 0.0;  // streaming log-sum-exp over MC samples
  for (int s = 0; s < a.S; s++) {
    double f[M];
#pragma unroll
    for (int o = 0; o < M; o++) f[o] = sg[o] * fma(sd[o], a.zx[(int64_t)s * M + o], trow[o][s * tstride]);
    double cref = -INFINITY, csum = 0.0;  // streaming log-sum-exp over the cells of this sample
    const int64_t c0 = a.cell_off[s], c1 = a.cell_off[s + 1];
    for (int64_t c = c0; c < c1; c++) {
      double la = 0.0;
#pragma unroll
      for (int o = 0; o < M; o++) {
        const double t = (f[o] - a.cell_lo[c * M + o]) * inv_tau;
        const double li = log_tau + bbh_fast_log_pos(bbh_fatplus_core(t));
        la += bbh_fatmin2(li, a.cell_ll[c * M + o]);
      }
      if (la > cref) {
        csum = csum * exp(cref - la) + 1.0;
        cref = la;
      } else if (la > -INFINITY) {
        csum += exp(la - cref);
      }
    }
    const double v = (cref > -INFINITY) ? cref + log(csum) : -INFINITY;
    if (v > sref) {
      ssum = ssum * exp(sref - v) + 1.0;
      sref = v;
    } else if (v > -INFINITY) {
      ssum += exp(v - sref);
    }
  }
  a.scores[i] = (sref > -INFINITY) ? sref + log(ssum) - log((double)a.S) : -INFINITY;
}

// The same value with the sums over cells and samples taken in the LINEAR domain:
//   score = log( (1/S) sum_s sum_c prod_o exp(fatmin(li_o, ll_o)) ),
//   exp(fatmin(a, b)) = min(e^a, e^b) (1 + u^2)^(-tau_max),  u = 2 tau_max / (2 tau_max + |a - b|),
// where e^a = tau_relu fatplus(t) is at hand BEFORE any logarithm is taken and e^b is the cell's side length.  Every term is a
// product of m side lengths >= ~1e-25 (the fat tail 0.1 tau_relu / (1 + t^2) at |t| <= 1e10), far inside the fp64 range, so
// nothing needs the log domain; what the log-domain form pays per (cell, target) - two double-precision logarithms, and per
// cell an exponential for the streaming log-sum-exp - shrinks to series and single-precision logarithms (see the loop body: the
// power tau_max = 0.01 absorbs their error).  BBH_NEHVI_LOG=1 selects the log-domain kernel (A/B, tests).
// One thread per candidate and SAMPLE SLICE (blockIdx.y of gridDim.y slices): with one thread per candidate a 1e5-row
// candidate set is 1563 wavefronts - 1.5 per SIMD, each a long dependent chain - and the kernel ran at a fifth of the VALU
// issue rate; the slices bring the launch to >= 8 waves per SIMD.  All lanes of a wave work on the same samples, so the
// cell data stays wave-uniform (scalar loads).  The slices' partial sums are combined in a fixed order by
// bbh_qlognehvi_finish_kernel (sums of positive terms in the linear domain: no atomics, reproducible).
// delta = 1 - (1 + u^2)^(-tau_max) <= 0.007 of TWO (cell, target) terms at once, in packed single precision (v_pk_fma_f32 /
// v_pk_mul_f32): the term is min(A, B) (1 - delta), so a relative error of ~1e-6 in delta is 7e-9 in the term - the same budget
// the double-precision form below spends on its single-precision logarithms.  rho = min / max, x1 = 1 - rho (taken in double
// precision, where it is exact next to rho = 1).  rho = 0 (cell unbounded above, or a ratio below the single-precision range):
// log2 -> -inf, |li - ll| = inf, u = 0, delta = 0.
__device__ __forceinline__ bbh_f2 bbh_fatmin_delta2(bbh_f2 rho, bbh_f2 x1) {
  bbh_f2 ds = x1 * (1.0f / 7.0f) + (1.0f / 6.0f);
  ds = ds * x1 + 0.2f;
  ds = ds * x1 + 0.25f;
  ds = ds * x1 + (1.0f / 3.0f);
  ds = ds * x1 + 0.5f;
  ds = ds * x1 + 1.0f;
  ds = ds * x1;  // -log1p(-x1), x1 < 0.15
  bbh_f2 dl;
  dl.x = -0.6931471805599453f * __log2f(rho.x);
  dl.y = -0.6931471805599453f * __log2f(rho.y);
  bbh_f2 den;
  den.x = (x1.x < 0.15f ? ds.x : dl.x) + (float)(2.0 * TAU_MAX);
  den.y = (x1.y < 0.15f ? ds.y : dl.y) + (float)(2.0 * TAU_MAX);
  bbh_f2 u;
  u.x = __builtin_amdgcn_rcpf(den.x);
  u.y = __builtin_amdgcn_rcpf(den.y);
  u = u * (float)(2.0 * TAU_MAX);
  const bbh_f2 w = u * u;
  bbh_f2 lps = w * -0.25f + (1.0f / 3.0f);
  lps = lps * w - 0.5f;
  lps = lps * w + 1.0f;
  lps = lps * w;  // log1p(w), w < 0.03
  bbh_f2 lp;
  lp.x = (w.x < 0.03f) ? lps.x : 0.6931471805599453f * __log2f(1.0f + w.x);
  lp.y = (w.y < 0.03f) ? lps.y : 0.6931471805599453f * __log2f(1.0f + w.y);
  const bbh_f2 x = lp * (float)TAU_MAX;  // in [0, 0.00694]
  bbh_f2 d = x * (-1.0f / 24.0f) + (1.0f / 6.0f);
  d = d * x - 0.5f;
  d = d * x + 1.0f;
  return d * x;  // 1 - exp(-x)
}

// PK: the (1 + u^2)^(-tau_max) factors in packed single precision, two cells at a time (default); PK = false is the double-precision
// sequence (BBH_NEHVI_PK=0: A/B, tests)
template <int M, bool PK>
__global__ __launch_bounds__(256) void bbh_qlognehvi_lin_kernel(const NehviArgs a, const double* __restrict__ cell_len,
                                                                double* __restrict__ partial) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  const int per = (a.S + (int)gridDim.y - 1) / (int)gridDim.y;
  const int s_begin = (int)blockIdx.y * per, s_end = (s_begin + per < a.S) ? s_begin + per : a.S;
  if (a.alive && !a.alive[i]) {
    partial[(int64_t)blockIdx.y * a.N + i] = 0.0;
    return;
  }
  double sd[M], sg[M];
  const double* trow[M];
  const int64_t tstride = a.sample_major ? a.N : 1;  // between consecutive samples of this candidate
#pragma unroll
  for (int o = 0; o < M; o++) {
    sd[o] = bbh_safe_sd(a.var[o][i]);
    sg[o] = a.sign[o];
    trow[o] = a.sample_major ? a.tmat[o] + i : a.tmat[o] + i * (int64_t)a.S;
  }
  const double inv_tau = 1.0 / TAU_RELU;
  double total = 0.0;
  for (int s = s_begin; s < s_end; s++) {
    double f[M];
#pragma unroll
    for (int o = 0; o < M; o++) f[o] = sg[o] * fma(sd[o], a.zx[(int64_t)s * M + o], trow[o][s * tstride]);
    const int64_t c0 = a.cell_off[s], c1 = a.cell_off[s + 1];
    double ssum = 0.0;
    int64_t c = c0;
    if (PK) {
      for (; c < c1; c += 2) {
        const int64_t cb = (c + 1 < c1) ? c + 1 : c;  // odd cell count: the last cell is paired with itself, its twin is not added
        double pa = 1.0, pb = 1.0;
#pragma unroll
        for (int o = 0; o < M; o++) {
          double mn2[2];
          bbh_f2 rho, x1, rc;
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int64_t cc = h ? cb : c;
            const double B = cell_len[cc * M + o];
            const double A = TAU_RELU * bbh_fatplus_core<1>((f[o] - a.cell_lo[cc * M + o]) * inv_tau);
            const double mn = fmin(A, B), mxv = fmax(A, B);
            // rho = min / max and x1 = 1 - rho = (max - min) / max, each with its own relative accuracy (x1 next to rho = 1, where the
            // term is sensitive; rho next to 0, where delta ~ 1e-8 would vanish in 1 - x1): the difference in double precision,
            // the quotients in single precision - no double-precision reciprocal
            const float gap = (float)(mxv - mn), mnf = (float)mn, mxf = (float)mxv;
            mn2[h] = mn;
            if (h) {
              x1.y = gap;
              rho.y = mnf;
              rc.y = __builtin_amdgcn_rcpf(mxf);
            } else {
              x1.x = gap;
              rho.x = mnf;
              rc.x = __builtin_amdgcn_rcpf(mxf);
            }
          }
          x1 = x1 * rc;  // max = inf (cell unbounded above): inf * 0 = NaN -> 1;  rho = min * 0 = 0
          rho = rho * rc;
          x1.x = (x1.x == x1.x) ? x1.x : 1.0f;
          x1.y = (x1.y == x1.y) ? x1.y : 1.0f;
          const bbh_f2 dl = bbh_fatmin_delta2(rho, x1);
          pa *= fma(-mn2[0], (double)dl.x, mn2[0]);
          pb *= fma(-mn2[1], (double)dl.y, mn2[1]);
        }
        ssum += pa;
        if (c + 1 < c1) ssum += pb;
      }
    }
    for (; !PK && c < c1; c++) {
      double prod = 1.0;
#pragma unroll
      for (int o = 0; o < M; o++) {
        const double B = cell_len[c * M + o];
        const double fp = bbh_fatplus_core((f[o] - a.cell_lo[c * M + o]) * inv_tau);
        const double A = TAU_RELU * fp;
        // |li - ll| = -log(rho), rho = min(A, B) / max(A, B): a 7-term series of -log1p(-x) next to rho = 1 (x = 1 - rho <
        // 0.15, where the term is sensitive: d log(term) = u^3 d|li - ll|), single precision beyond (u < 0.11 there: an error of
        // 1e-7 moves the term by 1e-10 relative).  B = inf (cell unbounded above): rho = 0, u = 0.  No double-precision
        // logarithm is left: inside a wave some lane is nearly always close to rho = 1, so a wave-uniform skip never skipped.
        const double mn = fmin(A, B), mxv = fmax(A, B);
        double inv = __builtin_amdgcn_rcp(mxv);
        inv = fma(fma(-mxv, inv, 1.0), inv, inv);
        const double rho = (mxv < INFINITY) ? mn * inv : 0.0;
        const double x1 = 1.0 - rho;
        double ds = fma(x1, 1.0 / 7.0, 1.0 / 6.0);
        ds = fma(ds, x1, 0.2);
        ds = fma(ds, x1, 0.25);
        ds = fma(ds, x1, 1.0 / 3.0);
        ds = fma(ds, x1, 0.5);
        ds = fma(ds, x1, 1.0);
        ds *= x1;
        const double dl = -0.6931471805599453 * (double)__log2f((float)rho);  // rho = 0 -> inf
        const double diff = (x1 < 0.15) ? ds : dl;
        const double u = (2.0 * TAU_MAX) * __builtin_amdgcn_rcp(fma(2.0, TAU_MAX, diff));  // bare seed (2^-26); diff = inf -> 0
        const double w = u * u;
        // log1p(w), w in [0, 1]: series below 0.03 (error w^5 / 5 < 5e-9), single precision above (1.2e-7) - times tau_max = 0.01
        const double lps = w * fma(w, fma(w, fma(w, -0.25, 1.0 / 3.0), -0.5), 1.0);
        const double lpf = 0.6931471805599453 * (double)__log2f((float)(1.0 + w));
        const double lp = (w < 0.03) ? lps : lpf;
        const double x = TAU_MAX * lp;  // in [0, 0.00694]
        const double e = fma(x, fma(x, fma(x, fma(x, fma(x, -1.0 / 120.0, 1.0 / 24.0), -1.0 / 6.0), 0.5), -1.0), 1.0);  // exp(-x)
        prod *= mn * e;
      }
      ssum += prod;
    }
    total += ssum;
  }
  partial[(int64_t)blockIdx.y * a.N + i] = total;
}

__global__ __launch_bounds__(256) void bbh_qlognehvi_finish_kernel(const double* __restrict__ partial, int slices, int64_t N, int S,
                                                                   const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double total = 0.0;
  for (int k = 0; k < slices; k++) total += partial[(int64_t)k * N + i];
  scores[i] = (total > 0.0 && !(alive && !alive[i])) ? log(total) - log((double)S) : -INFINITY;
}

static int bbh_qlognehvi_impl(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                             const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                             const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                             const uint8_t* alive_dev, double* scores_dev, bool sample_major);

extern "C" int bbh_qlognehvi(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                             const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                             const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                             const uint8_t* alive_dev, double* scores_dev) {
  return bbh_qlognehvi_impl(h, m, N, tmat_dev, var_dev, sign_host, zx_host, S, cell_off_host, cell_lo_host, cell_loglen_host, alive_dev,
                            scores_dev, false);
}

extern "C" int bbh_qlognehvi_sm(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                                const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                                const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                                const uint8_t* alive_dev, double* scores_dev) {
  return bbh_qlognehvi_impl(h, m, N, tmat_dev, var_dev, sign_host, zx_host, S, cell_off_host, cell_lo_host, cell_loglen_host, alive_dev,
                            scores_dev, true);
}

// launches with every operand on the device: a.zx, a.cell_off, a.cell_lo, a.cell_ll set by the caller, len_dev = side lengths
static int bbh_qlognehvi_run(bbh_handle* h, NehviArgs& a, const double* len_dev) {
  const int m = a.m;
  const int64_t N = a.N, S = a.S;
  dim3 grid((unsigned)((N + 255) / 256)), block(256);
  bbh_timed_scope timed(h, BBH_TIMED_NEHVI);
  const char* env_log = getenv("BBH_NEHVI_LOG");
  if (!(env_log && env_log[0] == '1')) {  // linear-domain sums (default)
    const double* len = len_dev;
    // sample slices: ~16 waves per SIMD (4 SIMDs per CU; 11.3 / 10.4 / 10.0 / 9.8 ms for 6 / 12 / 24 / 48 slices at 1e5 candidates), each slice at least 8 samples
    const int64_t srows = h->slice_rows > 0 ? h->slice_rows : N;
    int64_t slices = ((int64_t)16 * 4 * h->num_cu * 64 + srows - 1) / srows;
    if (const char* e = getenv("BBH_NEHVI_SLICES")) slices = atoi(e);
    if (slices > S / 8) slices = S / 8;
    if (slices > 64) slices = 64;
    if (slices < 1) slices = 1;
    int rc = bbh_ensure_ws(h, sizeof(double) * (size_t)slices * (size_t)N);
    if (rc) return rc;
    dim3 sgrid(grid.x, (unsigned)slices);
    const char* env_pk = getenv("BBH_NEHVI_PK");
    if (env_pk && env_pk[0] == '0') {
      switch (m) {
        case 1: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<1, false>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        case 2: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<2, false>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        case 3: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<3, false>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        default: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<4, false>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
      }
    } else {
      switch (m) {
        case 1: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<1, true>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        case 2: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<2, true>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        case 3: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<3, true>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
        default: hipLaunchKernelGGL((bbh_qlognehvi_lin_kernel<4, true>), sgrid, block, 0, h->stream, a, len, h->d_ws); break;
      }
    }
    hipLaunchKernelGGL(bbh_qlognehvi_finish_kernel, grid, block, 0, h->stream, h->d_ws, (int)slices, N, (int)S, a.alive, a.scores);
    BBH_HIP_TRY(h, hipGetLastError());
    return 0;
  }
  switch (m) {
    case 1: hipLaunchKernelGGL(bbh_qlognehvi_kernel<1>, grid, block, 0, h->stream, a); break;
    case 2: hipLaunchKernelGGL(bbh_qlognehvi_kernel<2>, grid, block, 0, h->stream, a); break;
    case 3: hipLaunchKernelGGL(bbh_qlognehvi_kernel<3>, grid, block, 0, h->stream, a); break;
    default: hipLaunchKernelGGL(bbh_qlognehvi_kernel<4>, grid, block, 0, h->stream, a); break;
  }
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

static void bbh_nehvi_fill_args(NehviArgs& a, int32_t m, int64_t N, const double* const* tmat_dev, const double* const* var_dev,
                                const double* sign_host, int64_t S, const uint8_t* alive_dev, double* scores_dev, bool sample_major) {
  for (int o = 0; o < BBH_MAX_OBJECTIVES; o++) {
    a.tmat[o] = o < m ? tmat_dev[o] : nullptr;
    a.var[o] = o < m ? var_dev[o] : nullptr;
    a.sign[o] = o < m ? sign_host[o] : 1.0;
  }
  a.m = m;
  a.N = N;
  a.S = (int)S;
  a.alive = alive_dev;
  a.scores = scores_dev;
  a.sample_major = sample_major ? 1 : 0;
}

static int bbh_qlognehvi_impl(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                             const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                             const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                             const uint8_t* alive_dev, double* scores_dev, bool sample_major) {
  if (!h) return -1;
  if (m < 1 || m > BBH_MAX_OBJECTIVES || N < 0 || S < 1 || !tmat_dev || !var_dev || !sign_host || !zx_host ||
      !cell_off_host || !scores_dev) {
    h->err = "bbh_qlognehvi: bad arguments (1 <= m <= 4)";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const int64_t ncells = cell_off_host[S];
  if (ncells < 0 || (ncells > 0 && (!cell_lo_host || !cell_loglen_host))) {
    h->err = "bbh_qlognehvi: inconsistent cell arrays";
    return -1;
  }
  // one upload: zx [S*m] | cell_lo [ncells*m] | cell_ll [ncells*m] | cell_len [ncells*m] | cell_off [S+1] (int64, 8-byte slots)
  const size_t nd = (size_t)S * m + 3 * (size_t)ncells * m;
  std::vector<double> buf(nd + (size_t)S + 1);
  memcpy(buf.data(), zx_host, sizeof(double) * S * m);
  if (ncells > 0) {
    memcpy(buf.data() + S * m, cell_lo_host, sizeof(double) * ncells * m);
    memcpy(buf.data() + S * m + ncells * m, cell_loglen_host, sizeof(double) * ncells * m);
    double* len = buf.data() + S * m + 2 * ncells * m;  // side lengths e^ll (inf for cells unbounded above)
    for (int64_t e = 0; e < ncells * m; e++) len[e] = exp(cell_loglen_host[e]);
  }
  memcpy(buf.data() + nd, cell_off_host, sizeof(int64_t) * (S + 1));
  int rc = bbh_upload_z(h, buf.data(), buf.size());
  if (rc) return rc;
  NehviArgs a;
  bbh_nehvi_fill_args(a, m, N, tmat_dev, var_dev, sign_host, S, alive_dev, scores_dev, sample_major);
  a.zx = h->d_z;
  a.cell_lo = h->d_z + S * m;
  a.cell_ll = h->d_z + S * m + ncells * m;
  a.cell_off = (const int64_t*)(h->d_z + nd);
  return bbh_qlognehvi_run(h, a, h->d_z + S * m + 2 * ncells * m);
}

// bbh_qlognehvi_sm against the device-resident cell lists of bbh_cells_build_dev on this handle: only the candidate's base
// samples zx_host [S, m] travel.
extern "C" int bbh_qlognehvi_cells(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                                   const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                                   const uint8_t* alive_dev, double* scores_dev) {
  if (!h) return -1;
  const bbh_nehvi_state* st = (const bbh_nehvi_state*)h->nehvi_state;
  if (m < 1 || m > BBH_MAX_OBJECTIVES || N < 0 || S < 1 || !tmat_dev || !var_dev || !sign_host || !zx_host || !scores_dev || !st ||
      st->S != S || st->m != m) {
    h->err = "bbh_qlognehvi_cells: bad arguments, or no cell lists for this S and m on the handle (bbh_cells_build_dev)";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  int rc = bbh_upload_z(h, zx_host, (size_t)S * m);
  if (rc) return rc;
  NehviArgs a;
  bbh_nehvi_fill_args(a, m, N, tmat_dev, var_dev, sign_host, S, alive_dev, scores_dev, true);
  a.zx = h->d_z;
  a.cell_off = st->off();
  a.cell_lo = st->lo();
  a.cell_ll = st->ll();
  return bbh_qlognehvi_run(h, a, st->len());
}

// ---- Pareto frequency of the baseline points over MC samples (pruning) ---------------------------
template <int M>
__global__ __launch_bounds__(256) void bbh_pareto_freq_kernel(const double* __restrict__ obj, int n, const double* __restrict__ ref,
                                                              unsigned long long* __restrict__ counts) {
  extern __shared__ double s_obj[];  // [n, M] of this sample
  const double* src = obj + (int64_t)blockIdx.x * n * M;
  for (int e = threadIdx.x; e < n * M; e += blockDim.x) s_obj[e] = src[e];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double yi[M];
    bool above = true;
#pragma unroll
    for (int o = 0; o < M; o++) {
      yi[o] = s_obj[i * M + o];
      above = above && (yi[o] > ref[o]);
    }
    bool dominated = false;
    for (int j = 0; j < n && !dominated; j++) {
      bool ge = true, gt = false;
#pragma unroll
      for (int o = 0; o < M; o++) {
        const double yj = s_obj[j * M + o];
        ge = ge && (yj >= yi[o]);
        gt = gt || (yj > yi[o]);
      }
      dominated = ge && gt;
    }
    if (above && !dominated) atomicAdd(&counts[i], 1ULL);
  }
}

static int bbh_pareto_frequency_impl(bbh_handle* h, const double* obj, bool on_device, int64_t S, int64_t n, int32_t m,
                                     const double* ref_host, int64_t* counts_host) {
  if (!h) return -1;
  if (!obj || !ref_host || !counts_host || S < 1 || n < 1 || n > 8192 / (m > 0 ? m : 1) || m < 1 ||
      m > BBH_MAX_OBJECTIVES) {
    h->err = "bbh_pareto_frequency: bad arguments (n * m <= 8192, 1 <= m <= 4)";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const size_t nobj = on_device ? 0 : (size_t)S * n * m;
  int rc = bbh_ensure_ws(h, sizeof(double) * (nobj + m + n));
  if (rc) return rc;
  double* d_obj = h->d_ws;
  double* d_ref = d_obj + nobj;
  unsigned long long* d_cnt = (unsigned long long*)(d_ref + m);
  if (!on_device) BBH_HIP_TRY(h, hipMemcpyAsync(d_obj, obj, sizeof(double) * nobj, hipMemcpyHostToDevice, h->stream));
  const double* src = on_device ? obj : d_obj;
  BBH_HIP_TRY(h, hipMemcpyAsync(d_ref, ref_host, sizeof(double) * m, hipMemcpyHostToDevice, h->stream));
  BBH_HIP_TRY(h, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * n, h->stream));
  const size_t lds = sizeof(double) * n * m;
  dim3 grid((unsigned)S), block(256);
  switch (m) {
    case 1: hipLaunchKernelGGL(bbh_pareto_freq_kernel<1>, grid, block, lds, h->stream, src, (int)n, d_ref, d_cnt); break;
    case 2: hipLaunchKernelGGL(bbh_pareto_freq_kernel<2>, grid, block, lds, h->stream, src, (int)n, d_ref, d_cnt); break;
    case 3: hipLaunchKernelGGL(bbh_pareto_freq_kernel<3>, grid, block, lds, h->stream, src, (int)n, d_ref, d_cnt); break;
    default: hipLaunchKernelGGL(bbh_pareto_freq_kernel<4>, grid, block, lds, h->stream, src, (int)n, d_ref, d_cnt); break;
  }
  BBH_HIP_TRY(h, hipGetLastError());
  BBH_HIP_TRY(h, hipMemcpyAsync(counts_host, d_cnt, sizeof(int64_t) * n, hipMemcpyDeviceToHost, h->stream));
  BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int bbh_pareto_frequency(bbh_handle* h, const double* obj_host, int64_t S, int64_t n, int32_t m,
                                    const double* ref_host, int64_t* counts_host) {
  return bbh_pareto_frequency_impl(h, obj_host, false, S, n, m, ref_host, counts_host);
}

// ... with the objective samples already on the device (bbh_nehvi_samples wrote them): nothing but the counts travels
extern "C" int bbh_pareto_frequency_dev(bbh_handle* h, const double* obj_dev, int64_t S, int64_t n, int32_t m,
                                        const double* ref_host, int64_t* counts_host) {
  return bbh_pareto_frequency_impl(h, obj_dev, true, S, n, m, ref_host, counts_host);
}

// =================================================================================================
// Other acquisition functions of baybe/acquisition/acqfs.py:161-290 on the same (mean, var, cross)
// inputs.  MC family (BoTorch SampleReducingMCAcquisitionFunction): mean_s max_j u(obj_sj);
// analytic family: closed forms in (mu~, sigma), sigma^2 clamped at 1e-12 as BoTorch does.
// =================================================================================================
__device__ __forceinline__ double bbh_mc_utility(int kind, double obj, double m, double best_f, double cu) {
  switch (kind) {
    case BBH_ACQ_QEI: return fmax(obj - best_f, 0.0);
    case BBH_ACQ_QPI: return 1.0 / (1.0 + exp(-(obj - best_f) * 1e3));
    case BBH_ACQ_QSR: return obj;
    case BBH_ACQ_QUCB: return m + cu * fabs(obj - m);
    default: return cu * fabs(obj - m);  // QPSTD
  }
}

__global__ __launch_bounds__(256) void bbh_mc_q1_kernel(int kind, const double* __restrict__ mean, const double* __restrict__ var,
                                                        int64_t N, const double* __restrict__ z, int S, double zbar,
                                                        double best_f, double sign, double cu,
                                                        const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  extern __shared__ double s_z[];
  for (int s = threadIdx.x; s < S; s += blockDim.x) s_z[s] = z[s];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  const double a = sign * mean[i], b = sign * bbh_safe_sd(var[i]);
  const double m = fma(b, zbar, a);
  double sum = 0.0;
  for (int s = 0; s < S; s++) sum += bbh_mc_utility(kind, fma(b, s_z[s], a), m, best_f, cu);
  scores[i] = sum / (double)S;
}

__global__ __launch_bounds__(64) void bbh_mc_pending_kernel(int kind, const double* __restrict__ mean, const double* __restrict__ var,
                                                            const double* __restrict__ cross, int64_t N, int p,
                                                            const double* __restrict__ mean_p, const double* __restrict__ cov_pp,
                                                            const double* __restrict__ z, const double* __restrict__ zbar, int S,
                                                            double best_f, double sign, double cu,
                                                            const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  __shared__ double s_L[QTRI * 64];
  __shared__ double s_mp[QMAX];
  __shared__ double s_cpp[QMAX * QMAX];
  const int t = threadIdx.x;
  const int q = p + 1;
  for (int e = t; e < p; e += 64) s_mp[e] = mean_p[e];
  for (int e = t; e < p * p; e += 64) s_cpp[e] = cov_pp[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 64 + t;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  double* L = s_L + t;
  const double v0 = var[i];
  double jitter = 0.0;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; attempt++) {
    ok = true;
    for (int r = 0; r < q && ok; r++) {
      for (int c = 0; c <= r; c++) {
        double s = (r == 0) ? v0 : (c == 0 ? cross[i * p + (r - 1)] : s_cpp[(r - 1) * p + (c - 1)]);
        if (r == c) s += jitter;
        for (int k = 0; k < c; k++) s -= L[tri(r, k) * 64] * L[tri(c, k) * 64];
        if (r == c) {
          if (!(s > 0.0)) {
            ok = false;
            break;
          }
          L[tri(r, r) * 64] = sqrt(s);
        } else {
          L[tri(r, c) * 64] = s / L[tri(c, c) * 64];
        }
      }
    }
    if (!ok) jitter = 1e-8 * pow(10.0, (double)attempt);
  }
  if (!ok) {
    scores[i] = NAN;
    return;
  }
  const double m0 = mean[i];
  double mbar[QMAX];  // per-point sample means of the objective
#pragma unroll 1
  for (int r = 0; r < q; r++) {
    double y = (r == 0) ? m0 : s_mp[r - 1];
    for (int c = 0; c <= r; c++) y = fma(L[tri(r, c) * 64], zbar[c], y);
    mbar[r] = sign * y;
  }
  double sum = 0.0;
  for (int s = 0; s < S; s++) {
    const double* zs = z + (int64_t)s * q;
    double mx = -INFINITY;
#pragma unroll 1
    for (int r = 0; r < q; r++) {
      double y = (r == 0) ? m0 : s_mp[r - 1];
      for (int c = 0; c <= r; c++) y = fma(L[tri(r, c) * 64], zs[c], y);
      mx = fmax(mx, bbh_mc_utility(kind, sign * y, mbar[r], best_f, cu));
    }
    sum += mx;
  }
  scores[i] = sum / (double)S;
}

// Register-resident form of bbh_mc_pending_kernel for q' = Q <= 14 (see bbh_qlogei_pending_q_kernel): same
// arithmetic in the same order, factor and per-point values in registers, base samples in LDS.
template <int Q>
__global__ __launch_bounds__(256) void bbh_mc_pending_q_kernel(int kind, const double* __restrict__ mean,
                                                              const double* __restrict__ var, const double* __restrict__ cross,
                                                              int64_t N, const double* __restrict__ mean_p,
                                                              const double* __restrict__ cov_pp, const double* __restrict__ z,
                                                              const double* __restrict__ zbar, int S, double best_f, double sign,
                                                              double cu, const uint8_t* __restrict__ alive,
                                                              double* __restrict__ scores) {
  extern __shared__ double s_zq[];  // [S * Q] base samples, then zbar [Q], mean_p [Q - 1], cov_pp [(Q - 1)^2]
  constexpr int P = Q - 1;
  double* s_zb = s_zq + (int64_t)S * Q;
  double* s_mp = s_zb + Q;
  double* s_cpp = s_mp + P;
  for (int e = threadIdx.x; e < S * Q; e += 256) s_zq[e] = z[e];
  for (int e = threadIdx.x; e < Q; e += 256) s_zb[e] = zbar[e];
  for (int e = threadIdx.x; e < P; e += 256) s_mp[e] = mean_p[e];
  for (int e = threadIdx.x; e < P * P; e += 256) s_cpp[e] = cov_pp[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  double L[Q * (Q + 1) / 2], A[Q * (Q + 1) / 2];
  A[0] = var[i];
#pragma unroll
  for (int r = 1; r < Q; r++) {
    A[r * (r + 1) / 2] = cross[i * P + (r - 1)];
#pragma unroll
    for (int c = 1; c <= r; c++) A[r * (r + 1) / 2 + c] = s_cpp[(r - 1) * P + (c - 1)];
  }
  double jitter = 0.0;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; attempt++) {
    ok = true;
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
      for (int c = 0; c <= r; c++) {
        double sacc = A[r * (r + 1) / 2 + c];
        if (r == c) sacc += jitter;
#pragma unroll
        for (int k = 0; k < c; k++) sacc -= L[r * (r + 1) / 2 + k] * L[c * (c + 1) / 2 + k];
        if (r == c) {
          if (!(sacc > 0.0)) ok = false;
          L[r * (r + 1) / 2 + r] = sqrt(sacc);
        } else {
          L[r * (r + 1) / 2 + c] = sacc / L[c * (c + 1) / 2 + c];
        }
      }
    }
    if (!ok) jitter = 1e-8 * pow(10.0, (double)attempt);
  }
  if (!ok) {
    scores[i] = NAN;
    return;
  }
  double m[Q], mbar[Q];
  m[0] = mean[i];
#pragma unroll
  for (int r = 1; r < Q; r++) m[r] = s_mp[r - 1];
#pragma unroll
  for (int r = 0; r < Q; r++) {
    double y = m[r];
#pragma unroll
    for (int c = 0; c <= r; c++) y = fma(L[r * (r + 1) / 2 + c], s_zb[c], y);
    mbar[r] = sign * y;
  }
  double sum = 0.0;
  for (int s = 0; s < S; s++) {
    const double* zs = s_zq + (int64_t)s * Q;
    double zr[Q];
#pragma unroll
    for (int c = 0; c < Q; c++) zr[c] = zs[c];
    double mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < Q; r++) {
      double y = m[r];
#pragma unroll
      for (int c = 0; c <= r; c++) y = fma(L[r * (r + 1) / 2 + c], zr[c], y);
      mx = fmax(mx, bbh_mc_utility(kind, sign * y, mbar[r], best_f, cu));
    }
    sum += mx;
  }
  scores[i] = sum / (double)S;
}

template <int Q>
static void bbh_launch_mc_pending_q(hipStream_t st, int kind, const double* mean, const double* var, const double* cross,
                                    int64_t N, const double* mp, const double* cpp, const double* z, const double* zbar, int S,
                                    double best_f, double sign, double cu, const uint8_t* alive, double* scores) {
  const size_t lds = sizeof(double) * ((size_t)S * Q + Q + (Q - 1) + (size_t)(Q - 1) * (Q - 1));
  hipLaunchKernelGGL((bbh_mc_pending_q_kernel<Q>), dim3((unsigned)((N + 255) / 256)), dim3(256), lds, st, kind, mean, var,
                     cross, N, mp, cpp, z, zbar, S, best_f, sign, cu, alive, scores);
}

__device__ __forceinline__ double bbh_log_h(double u) {
  // log(phi(u) + u Phi(u)), asymptotic branch for u < -1 (BoTorch _log_ei_helper)
  const double inv_sqrt2 = 0.7071067811865476, half_log_2pi = 0.9189385332046727;
  if (u > -1.0) {
    const double phi = exp(-0.5 * u * u - half_log_2pi), Phi = 0.5 * erfc(-u * inv_sqrt2);
    return log(phi + u * Phi);
  }
  const double au = fabs(u);
  const double logphi = -0.5 * u * u - half_log_2pi;
  if (u > -1e6) return logphi + log1p(-au * erfcx(au * inv_sqrt2) * 1.2533141373155003);
  return logphi - 2.0 * log(au);
}

__global__ void bbh_analytic_kernel(int kind, const double* __restrict__ mean, const double* __restrict__ var, int64_t N,
                                    double best_f, double sign, double beta, int maximize,
                                    const uint8_t* __restrict__ alive, double* __restrict__ scores) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (alive && !alive[i]) {
    scores[i] = -INFINITY;
    return;
  }
  const double mt = sign * mean[i];
  const double v = var[i];
  double out;
  if (kind == BBH_ACQ_PM) {
    out = mt;
  } else if (kind == BBH_ACQ_PSTD) {
    const double sd = sqrt(fmax(v, 0.0));
    out = maximize ? sd : -sd;
  } else {
    const double sd = sqrt(fmax(v, 1e-12));
    const double u = (mt - best_f) / sd;
    if (kind == BBH_ACQ_UCB)
      out = mt + sqrt(beta) * sd;
    else if (kind == BBH_ACQ_EI)
      out = sd * (exp(-0.5 * u * u - 0.9189385332046727) + u * 0.5 * erfc(-u * 0.7071067811865476));
    else if (kind == BBH_ACQ_PI)
      out = 0.5 * erfc(-u * 0.7071067811865476);
    else
      out = log(sd) + bbh_log_h(u);  // LogEI
  }
  scores[i] = out;
}

static double bbh_mc_cu(int kind, double beta) {
  if (kind == BBH_ACQ_QUCB) return sqrt(beta * 3.141592653589793 / 2.0);
  if (kind == BBH_ACQ_QPSTD) return sqrt(3.141592653589793 / 2.0);
  return 0.0;
}

extern "C" int bbh_mc_acq_q1(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev, int64_t N,
                             const double* z_host, int64_t S, double best_f, double sign, double beta,
                             const uint8_t* alive_dev, double* scores_dev) {
  if (!h) return -1;
  if (kind == BBH_ACQ_QLOGEI) return bbh_qlogei_q1(h, mean_dev, var_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev);
  if (kind < BBH_ACQ_QEI || kind > BBH_ACQ_QPSTD || !mean_dev || !var_dev || !z_host || !scores_dev || N < 0 || S < 1 ||
      S > 8192) {
    h->err = "bbh_mc_acq_q1: bad arguments";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  double zbar = 0.0;
  for (int64_t s = 0; s < S; s++) zbar += z_host[s];
  zbar /= (double)S;
  int rc = bbh_upload_z(h, z_host, (size_t)S);
  if (rc) return rc;
  bbh_timed_scope timed(h, BBH_TIMED_Q1);
  hipLaunchKernelGGL(bbh_mc_q1_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), sizeof(double) * S, h->stream, kind,
                     mean_dev, var_dev, N, h->d_z, (int)S, zbar, best_f, sign, bbh_mc_cu(kind, beta), alive_dev, scores_dev);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

extern "C" int bbh_mc_acq_pending(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev,
                                  const double* cross_dev, int64_t N, const double* z_host, int64_t S, double best_f,
                                  double sign, double beta, const uint8_t* alive_dev, double* scores_dev) {
  if (!h) return -1;
  if (kind == BBH_ACQ_QLOGEI)
    return bbh_qlogei_pending(h, mean_dev, var_dev, cross_dev, N, z_host, S, best_f, sign, alive_dev, scores_dev);
  const int p = h->p;
  if (kind < BBH_ACQ_QEI || kind > BBH_ACQ_QPSTD || !mean_dev || !var_dev || !cross_dev || !z_host || !scores_dev ||
      N < 0 || S < 1 || p < 1) {
    h->err = "bbh_mc_acq_pending: bad arguments / no pending points set";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const int q = p + 1;
  std::vector<double> buf((size_t)S * q + q + p + (size_t)p * p, 0.0);
  memcpy(buf.data(), z_host, sizeof(double) * S * q);
  double* zb = buf.data() + S * q;
  for (int64_t s = 0; s < S; s++)
    for (int c = 0; c < q; c++) zb[c] += z_host[s * q + c];
  for (int c = 0; c < q; c++) zb[c] /= (double)S;
  memcpy(zb + q, h->pend_mean.data(), sizeof(double) * p);
  memcpy(zb + q + p, h->pend_cov.data(), sizeof(double) * p * p);
  int rc = bbh_upload_z(h, buf.data(), buf.size());
  if (rc) return rc;
  const double* dz = h->d_z;
  const double* dzb = dz + S * q;
  const double* dmp = dzb + q;
  const double* dcpp = dmp + p;
  const double cu = bbh_mc_cu(kind, beta);
  const bool fits = sizeof(double) * buf.size() <= 60 * 1024;  // base samples in LDS
  bbh_timed_scope timed(h, BBH_TIMED_PENDING);
#define BBH_MC_PENDING_Q(QV)                                                                                     \
  case QV:                                                                                                       \
    bbh_launch_mc_pending_q<QV>(h->stream, kind, mean_dev, var_dev, cross_dev, N, dmp, dcpp, dz, dzb, (int)S,    \
                                best_f, sign, cu, alive_dev, scores_dev);                                        \
    break;
  switch (fits && !h->pending_lds_form ? q : 0) {
    BBH_MC_PENDING_Q(2)
    BBH_MC_PENDING_Q(3)
    BBH_MC_PENDING_Q(4)
    BBH_MC_PENDING_Q(5)
    BBH_MC_PENDING_Q(6)
    BBH_MC_PENDING_Q(7)
    BBH_MC_PENDING_Q(8)
    BBH_MC_PENDING_Q(9)
    BBH_MC_PENDING_Q(10)
    BBH_MC_PENDING_Q(11)
    BBH_MC_PENDING_Q(12)
    BBH_MC_PENDING_Q(13)
    BBH_MC_PENDING_Q(14)
    default:
      hipLaunchKernelGGL(bbh_mc_pending_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, h->stream, kind, mean_dev,
                         var_dev, cross_dev, N, p, dmp, dcpp, dz, dzb, (int)S, best_f, sign, cu, alive_dev, scores_dev);
  }
#undef BBH_MC_PENDING_Q
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}

extern "C" int bbh_analytic_acq(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev, int64_t N,
                                double best_f, double sign, double beta, int32_t maximize, const uint8_t* alive_dev,
                                double* scores_dev) {
  if (!h) return -1;
  if (kind < BBH_ACQ_PM || kind > BBH_ACQ_PI || !mean_dev || !var_dev || !scores_dev || N < 0) {
    h->err = "bbh_analytic_acq: bad arguments";
    return -1;
  }
  if (N == 0) return 0;
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(bbh_analytic_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, kind, mean_dev, var_dev,
                     N, best_f, sign, beta, (int)maximize, alive_dev, scores_dev);
  BBH_HIP_TRY(h, hipGetLastError());
  return 0;
}
