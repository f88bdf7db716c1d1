// Cooperative form of the fused posterior kernel with a GENERIC kernel-value production: composite kernels (ProductKernel /
// AdditiveKernel of up to four stationary ARD factors, baybe/kernels/composite.py:60-91) and the single kernels that have no
// software-pipelined instantiation (rational quadratic, piecewise polynomial, Linear, Polynomial, Periodic; baybe/kernels/basic.py:
// 20-46, 73-163, 203-216), n <= 512.
//
// These models used to take the materialised-K* path for every posterior-shaped call: [K(X*, X)] written to and re-read
// from HBM per 16384-candidate chunk (8 n bytes per candidate and pass, ~47 x the algorithmic traffic), an elementwise
// kernel on libm sqrt / exp, a plain (not triangular) GEMM and an epilogue - 23.5 ms per 1e6 candidates at n = 512 against
// 4.7 ms for the fused single-kernel form.  Here the structure of bbh_coop.h is kept - one workgroup per 16 candidates, the
// column blocks of L^-T dealt to the four waves, every kernel value computed once per tile by the wave whose turn it is and
// handed to the others through LDS - and only the production differs: per factor one distance GEMM against that factor's own
// training fragments (its lengthscales) and candidate fragments, one evaluation of that factor's kernel function (the staged
// rsq / minimax sequences of bbh_fused.h for Matérn-5/2, -3/2 and RBF; libm for the others), then the product / sum, the
// per-factor scales and the task / outputscale table.  The production is NOT interleaved with the variance MFMAs (it runs as a
// block at the start of a group, before the wave touches its operand ring), so a wave's fp64 pipe idles through the
// dependency chains of one production per group; the other waves of the SIMD fill them.
#pragma once
#include "bbh_coop.h"

// One entry of a factor's candidate-side feature map: feature e = 4 k + q of the factor's GEMM operand is
//   op 0  x[src] * scl + ofs           (scaled coordinate; its square enters the factor's |b|^2)
//   op 1  cos(x[src] * scl + ofs)      op 2  sin(...)      (periodic factors: angle 2 pi xn / p)
//   op 3  the constant ofs             op 4  |b|^2 of the factor            op 5  0
// The metric of a factor is ONE dot product of such features with the factor's training fragments:
//   distance kinds   |a|^2 + |b|^2 - 2 a.b   train [-2 a, |a|^2, 1]            candidate [b, 1, |b|^2]
//   dot kinds        a.b (Linear, Polynomial; a = xn / w)   train [a, 0, 0]     candidate [b, 1, |b|^2]  (|b|^2 is k(x, x)'s argument)
//   periodic         sum_j sin^2(pi (xn_j - xn'_j) / p_j) / l_j = sum_j (1 - cos al_j cos be_j - sin al_j sin be_j) / (2 l_j)
//                    train [-cos al / 2l, -sin al / 2l, sum_j 1 / 2l_j]         candidate [cos be, sin be, 1]
// (padding rows of the training fragments carry 1e8 in the slot that meets the candidate's constant 1: metric 1e8 -> value 0)
struct CoopGFeat {
  double scl, ofs;
  int src, op;
};

struct CoopGArgs {
  CoopArgs c;
  int F, combine, has_tbl, jb;
  int kind[BBH_MAX_FACTORS];
  int grp[BBH_MAX_FACTORS];       // term of the sum each factor multiplies into (bbh_combine)
  double fos[BBH_MAX_FACTORS];    // per-factor scales (1 for a single kernel)
  double alpha[BBH_MAX_FACTORS];  // RQ alpha / polynomial offset per factor
  const double* trainfrag_f;      // [F][nb + 1][KD][64]
  int64_t tf_stride;              // doubles per factor
  const CoopGFeat* feat;          // [F][4 KD]
  double prior_k0;                // k(x, x) without the table / outputscale when no factor is a dot kind: prod_f fos_f or sum_f fos_f
  int has_dot;                    // some factor is Linear / Polynomial: k(x, x) is per candidate
};

// one k-block's kernel values (4 per lane) for all factors, combined
template <int KD, int F>
__device__ __forceinline__ void coopg_produce(const CoopGArgs& g, const WaveCtx& c, const double (&cf)[F][KD], int tb,
                                              double (&out)[4]) {
  double uv[4][BBH_MAX_FACTORS];  // [value][factor] os_f k_f
  bool pad[4] = {false, false, false, false};
#pragma unroll
  for (int f = 0; f < F; f++) {
    double tfv[KD];
    kvp_load<KD>(g.trainfrag_f + (int64_t)f * g.tf_stride + c.l, tb, tfv);
    d4 da = {0.0, 0.0, 0.0, 0.0}, db = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < KD; k += 2) {
      da = mfma_f64(tfv[k], cf[f][k], da);
      if (k + 1 < KD) db = mfma_f64(tfv[k + 1], cf[f][k + 1], db);
    }
    double kv[4];
    const int kind = g.kind[f];  // wave-uniform
    if (kind == BBH_KERNEL_MATERN52) {
      kv_all<0>(c, tb, da, db, kv);
    } else if (kind == BBH_KERNEL_RBF) {
      kv_all<2>(c, tb, da, db, kv);
    } else if (kind == BBH_KERNEL_MATERN32) {
      kv_all<4>(c, tb, da, db, kv);
    } else if (BBH_KIND_IS_DOT(kind)) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double sdot = da[r] + db[r];
        kv[r] = (kind == BBH_KERNEL_LINEAR) ? sdot : bbh_powi(sdot + g.alpha[f], kind - BBH_KERNEL_POLY1 + 1);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double r2 = fmax(da[r] + db[r], 0.0);
        if (kind == BBH_KERNEL_RQ)
          kv[r] = exp(-g.alpha[f] * log1p(r2 / (2.0 * g.alpha[f])));
        else if (kind == BBH_KERNEL_PERIODIC)
          kv[r] = exp(-2.0 * r2);
        else
          kv[r] = bbh_piecewise(kind - BBH_KERNEL_PIECEWISE0, g.jb, r2, false);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (f == 0) pad[r] = (da[r] + db[r]) > 1e7;  // padding rows carry |a|^2 = 1e8: exactly 0 for every kernel kind
      uv[r][f] = g.fos[f] * kv[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double v = pad[r] ? 0.0 : (F > 1 ? bbh_combine(F, g.combine, g.grp, uv[r]) : uv[r][0]);
    if (g.has_tbl) v *= c.tbl[c.tc * c.T + c.taskext[16 * tb + 4 * r + c.q]];
    out[r] = v;
  }
}

template <int KD, int F>
__global__ __launch_bounds__(256, 2) void bbh_coopg_posterior_kernel(const CoopGArgs g) {
  const CoopArgs& ca = g.c;
  const FusedArgs& a = ca.f;
  extern __shared__ __attribute__((aligned(16))) double s_mem[];  // alpha [16 nb] | kv [2][4][256] | red [2][4][16]
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnd = l & 15, q = l >> 4;
  double* s_alpha = s_mem;
  double* s_kv = s_alpha + 16 * a.nb;
  double* s_red = s_kv + BBH_COOP_KV_TILE;
  const int64_t tile0 = (int64_t)blockIdx.x * 16;

  // candidate fragments per factor through the factor's feature map (see CoopGFeat)
  const int64_t row = (tile0 + cnd < a.N) ? tile0 + cnd : a.N - 1;
  const double* xr = a.X + row * a.ldx;
  for (int s = threadIdx.x; s < 16 * a.nb; s += 256) s_alpha[s] = a.meanB[(int64_t)s * 16];
  double cf[F][KD];
  double kself_u[BBH_MAX_FACTORS] = {1.0, 1.0, 1.0, 1.0};  // os_f k_f(x, x) of candidate cnd (dot kinds: not a constant)
#pragma unroll
  for (int f = 0; f < F; f++) {
    double nbsum = 0.0;
    int ops[KD];
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const CoopGFeat ft = g.feat[(f * KD + k) * 4 + q];
      const double x = (ft.src >= 0) ? xr[a.numcol_identity ? ft.src : a.numcol[ft.src]] : 0.0;
      double v = fma(x, ft.scl, ft.ofs);
      if (ft.op == 0) nbsum = fma(v, v, nbsum);
      if (ft.op == 1) v = cos(v);
      if (ft.op == 2) v = sin(v);
      if (ft.op >= 4) v = 0.0;
      cf[f][k] = v;
      ops[k] = ft.op;
    }
    nbsum += __shfl_xor(nbsum, 16, 64);
    nbsum += __shfl_xor(nbsum, 32, 64);
#pragma unroll
    for (int k = 0; k < KD; k++)
      if (ops[k] == 4) cf[f][k] = nbsum;
    const int kindf = g.kind[f];
    const double ks = BBH_KIND_IS_DOT(kindf) ? ((kindf == BBH_KERNEL_LINEAR) ? nbsum : bbh_powi(nbsum + g.alpha[f], kindf - BBH_KERNEL_POLY1 + 1)) : 1.0;
    kself_u[f] = g.fos[f] * ks;
  }
  const double kself = F > 1 ? bbh_combine(F, g.combine, g.grp, kself_u) : kself_u[0];  // without table / outer outputscale
  int tc = 0;
  if (g.has_tbl && a.task_col >= 0) {
    tc = (int)xr[a.task_col];
    tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
  }
  WaveCtx c[1];
  c[0].tf = a.trainfrag + l;
  c[0].candl = nullptr;
  c[0].mb = nullptr;
  c[0].tbl = a.tasktbl;
  c[0].taskext = a.taskext;
  c[0].kvc = nullptr;
  c[0].kvl = (bbh_lds_double*)nullptr;
  c[0].nl = 0;
  c[0].ncache = 0;
  c[0].al = (const bbh_lds_double*)nullptr;
  c[0].kd = KD;
  c[0].kind = a.kind;
  c[0].T = a.T;
  c[0].tc = tc;
  c[0].q = q;
  c[0].l = l;
  c[0].dn = a.dn;

  bbh_lds_double* kvb = (bbh_lds_double*)(s_kv + l);
  const bbh_lds_double* alq = (const bbh_lds_double*)(s_alpha + q);
  const int g0 = ca.g0;
  double accm[1] = {0.0};
  d4 acc[1][BBH_COOP_ROUNDS];
#pragma unroll
  for (int s = 0; s < BBH_COOP_ROUNDS; s++) acc[0][s] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* rs = ca.rstream + (int64_t)w * ca.frags * 64;
  d2 ring[BBH_COOP_PAIRS];
  static_for<0, BBH_COOP_PAIRS>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    coop_gload2<(i % 4) * 1024>(ring[i], rs + (i / 4) * 512, (unsigned)l * 16u);
  });
  {  // the first group's kernel values: wave w produces k-block w
    double kv0[4];
    coopg_produce<KD, F>(g, c[0], cf, w, kv0);
    __syncthreads();  // alpha is in LDS
#pragma unroll
    for (int r = 0; r < 4; r++) {
      kvb[((g0 & 1) * 4 + w) * 256 + r * 64] = kv0[r];
      accm[0] = fma(kv0[r], alq[16 * w + 4 * r], accm[0]);
    }
  }
  __syncthreads();

  static_for<0, BBH_COOP_ROUNDS>([&](auto gc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value;
    if (G >= g0) {  // wave-uniform
      const int cw = (G & 1) ? 3 - w : w;
      if constexpr (G + 1 < BBH_COOP_ROUNDS) {  // produce k-block tbn for the next group first: its loads and its dependency
        const int tbn = 4 * (G + 1 - g0) + w;   // chains are out of the way before the counted waits of the operand ring begin
        double kvn[4];
        coopg_produce<KD, F>(g, c[0], cf, tbn, kvn);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          kvb[(((G + 1) & 1) * 4 + w) * 256 + r * 64] = kvn[r];
          accm[0] = fma(kvn[r], alq[16 * tbn + 4 * r], accm[0]);
        }
      }
      coop_group<G, KD, 0, false, 1>(c, rs, nullptr, kvb + (G & 1) * 4 * 256, nullptr, nullptr, 0, cw, acc, ring, accm);
      rs += (int64_t)16 * (BBH_COOP_ROUNDS - G) * 64;
      if constexpr (G + 1 < BBH_COOP_ROUNDS) __syncthreads();
    }
  });

  double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < BBH_COOP_ROUNDS; s++)
#pragma unroll
    for (int r = 0; r < 4; r++) ss[r] = fma(acc[0][s][r], acc[0][s][r], ss[r]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double v = ss[r];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    ss[r] = v;
  }
  double mp = accm[0];
  mp += __shfl_xor(mp, 16, 64);
  mp += __shfl_xor(mp, 32, 64);
  double* red_v = s_red;
  double* red_m = red_v + 64;
  if (cnd == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) red_v[w * 16 + q + 4 * r] = ss[r];
  }
  if (q == 0) red_m[w * 16 + cnd] = mp;
  __syncthreads();
  if (threadIdx.x < 16) {
    const int m = threadIdx.x;
    const int64_t gidx = tile0 + m;
    const double sv = (red_v[m] + red_v[16 + m]) + (red_v[32 + m] + red_v[48 + m]);
    const double sm = (red_m[m] + red_m[16 + m]) + (red_m[32 + m] + red_m[48 + m]);
    const double k0 = g.has_dot ? kself : g.prior_k0;  // (lane m of wave 0: candidate m's own k(x, x))
    double pv = a.prior_scale * k0, mc = a.mean_const;
    if (g.has_tbl) {
      pv = a.tasktbl[tc * a.T + tc] * k0;
      if (a.taskmean) mc = a.taskmean[tc];
    }
    if (gidx < a.N) {
      if (a.mean) a.mean[gidx] = a.ybar + a.ysd * (mc + sm);
      if (a.var) a.var[gidx] = a.ysd * a.ysd * (pv - sv);
    }
  }
}

// ---- mean / cross-covariance pass with the generic production (round 4) ---------------------------------------------------------
// Steps >= 2 of a greedy batch need, per candidate, the posterior mean and the cross-covariances with the p <= 15 pending points:
//   [mean | cross_1 .. cross_p] = K*(x, [X ; P]) . [alpha | -beta_1 .. -beta_p ; 0 | e_1 .. e_p]      (d_meanB, rows np.. = unit vectors)
// i.e. one 16-column contraction of the kernel values - no variance GEMM.  Composite / non-stationary models used to materialise
// K* for it (23 ms per 1e6 candidates and step at n = 512); here one wave takes 16 candidates, walks the nb (+1 pending) k-blocks,
// produces each block's kernel values with coopg_produce (the pending points sit in block nb of every factor's training fragments,
// written by bbh_pending_set) and feeds four MFMAs per block.  Same epilogue as the windowed form's mean-only pass (bbh_fused.h).
template <int KD, int F>
__global__ __launch_bounds__(256, 2) void bbh_coopg_cross_kernel(const CoopGArgs g) {
  const FusedArgs& a = g.c.f;
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnd = l & 15, q = l >> 4;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + w) * 16;
  if (tile0 >= a.N) return;
  const int64_t row = (tile0 + cnd < a.N) ? tile0 + cnd : a.N - 1;
  const double* xr = a.X + row * a.ldx;
  double cf[F][KD];
#pragma unroll
  for (int f = 0; f < F; f++) {
    double nbsum = 0.0;
    int ops[KD];
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const CoopGFeat ft = g.feat[(f * KD + k) * 4 + q];
      const double x = (ft.src >= 0) ? xr[a.numcol_identity ? ft.src : a.numcol[ft.src]] : 0.0;
      double v = fma(x, ft.scl, ft.ofs);
      if (ft.op == 0) nbsum = fma(v, v, nbsum);
      if (ft.op == 1) v = cos(v);
      if (ft.op == 2) v = sin(v);
      if (ft.op >= 4) v = 0.0;
      cf[f][k] = v;
      ops[k] = ft.op;
    }
    nbsum += __shfl_xor(nbsum, 16, 64);
    nbsum += __shfl_xor(nbsum, 32, 64);
#pragma unroll
    for (int k = 0; k < KD; k++)
      if (ops[k] == 4) cf[f][k] = nbsum;
  }
  int tc = 0;
  if (g.has_tbl && a.task_col >= 0) {
    tc = (int)xr[a.task_col];
    tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
  }
  WaveCtx c;
  c.tf = a.trainfrag + l;
  c.candl = nullptr;
  c.mb = a.meanB + l;
  c.tbl = a.tasktbl;
  c.taskext = a.taskext;
  c.kvc = nullptr;
  c.kvl = (bbh_lds_double*)nullptr;
  c.nl = 0;
  c.ncache = 0;
  c.al = (const bbh_lds_double*)nullptr;
  c.kd = KD;
  c.kind = a.kind;
  c.T = a.T;
  c.tc = tc;
  c.q = q;
  c.l = l;
  c.dn = a.dn;
  d4 accm = {0.0, 0.0, 0.0, 0.0};
  for (int tb = 0; tb < a.nb_ext; tb++) {
    double mbv[4], kv[4];
#pragma unroll
    for (int r = 0; r < 4; r++) mbv[r] = c.mb[(int64_t)(4 * tb + r) * 64];
    coopg_produce<KD, F>(g, c, cf, tb, kv);
#pragma unroll
    for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], mbv[r], accm);
  }
  // epilogue: lane (q, cnd), reg r  <->  candidate q + 4 r, column cnd
  const double s2 = a.ysd * a.ysd;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = q + 4 * r;
    const int tcm = __shfl(tc, m, 64);
    const int64_t gi = tile0 + m;
    const double mc = (g.has_tbl && a.taskmean) ? a.taskmean[tcm] : a.mean_const;
    if (gi < a.N) {
      if (cnd == 0) {
        if (a.mean) a.mean[gi] = a.ybar + a.ysd * (mc + accm[r]);
      } else if (cnd <= a.p && a.cross) {
        a.cross[gi * a.p + (cnd - 1)] = s2 * accm[r];
      }
    }
  }
}

// Instantiations (bbh_fused_coopg.hip): 2, 4, 6, 8 k-steps of the distance GEMM (d <= 30), 1 - 4 factors.
// false: no instantiation for this model; grid.x == 0 only asks.
bool bbh_coopg_launch(int kd, int F, dim3 grid, size_t lds, hipStream_t s, const CoopGArgs& a);
bool bbh_coopg_cross_launch(int kd, int F, dim3 grid, hipStream_t s, const CoopGArgs& a);
