// Device code of the fused posterior kernel (see bbh_panel.hip for the design notes).  It lives in a
// header so that the instantiations can be compiled as separate translation units in parallel
// (bbh_fused_kd{0,2,4,6,8,12,16}.hip): the unrolled triangular region makes each one minutes of compile time.
#pragma once
#include <math.h>
#include <string.h>

#include <type_traits>

#include "bbh_common.h"

#define BBH_MAX_PASS 64
#ifndef BBH_W32_REMAINDERS
#define BBH_W32_REMAINDERS 0
#endif

__device__ __forceinline__ double bbh_kfun_p(int kind, double r2) {
  if (kind == BBH_KERNEL_RBF) return exp(-0.5 * r2);
  const double r = sqrt(r2);
  if (kind == BBH_KERNEL_MATERN52) return (1.0 + BBH_SQRT5 * r + (5.0 / 3.0) * r2) * exp(-BBH_SQRT5 * r);
  if (kind == BBH_KERNEL_MATERN32) return (1.0 + BBH_SQRT3 * r) * exp(-BBH_SQRT3 * r);
  return exp(-r);
}

struct FusedArgs {
  const double* X;
  int64_t N, ldx;
  const double* trainfrag;
  const double* rfrag;
  const double* meanB;
  const double* scl;
  const double* ofs;
  const int* numcol;
  const double* tasktbl;
  const double* taskmean;  // [T] per-task constant means (hadamard models), nullptr = mean_const for every task
  const int* taskext;
  double* mean;
  double* var;
  double* cross;
  const int64_t* pass_off;  // [npass] element offset of each pass in rfrag
  const int* pass_w;        // [npass] window width in 16-column blocks
  int npass;
  int kind, dn, kd, nb, nb_ext, task_col, T, p, with_var;
  double ybar, ysd, mean_const, prior_scale;
  // fused qLogEI epilogue (q' = 1): disabled when qz == nullptr
  const double* qz;
  int qS;
  double q_best_f, q_sign;
  const uint8_t* q_alive;
  double* q_scores;
  // kernel-value cache of the multi-pass form (one slab per resident wave), see pass_body_p
  double* kvcache;
  int* slab_flags;  // [nslab] 0 = free, 1 = owned by a resident wave (kvcache slabs are claimed per wave)
  int nslab, nxcc;  // slabs in total, XCD partitions
  int ncache;    // k-blocks cached per wave = first column block of the last pass
  int64_t nblk;  // 64-candidate blocks = workgroups
  int nl;         // kernel-value cache: k-blocks [0, nl) live in wave-private LDS, [nl, ncache) in the global slab
  int has_tbl;    // task / outputscale table in use
  int mean_valu;  // no pending columns: the mean contraction runs on the VALU against alpha in LDS
  int numcol_identity;  // numcol[j] == j for every numerical column (cooperative forms: skips the index load)
};

typedef __attribute__((address_space(3))) double bbh_lds_double;

struct WaveCtx {
  const double* tf;     // trainfrag + lane
  const double* candl;  // this wave's candidate fragments in LDS, + lane
  const double* mb;     // meanB + lane
  const double* tbl;
  const int* taskext;
  double* kvc;  // this wave's kernel-value cache slab, + lane, shifted so that it is indexed by k-block like kvl
  bbh_lds_double* kvl;  // LDS part of the cache (k-blocks < nl), + lane (explicit address space: a select
                        // between this and the global slab pointer would otherwise become flat accesses,
                        // which count in vmcnt and lgkmcnt and wreck the counted waits of the operand ring)
  int nl, ncache;  // k-blocks [0, ncache) are cached at all
  const bbh_lds_double* al;  // alpha in LDS + (lane >> 4), or null: mean through the MFMA form (pending columns)
  int kd, kind, T, tc, q, l, dn;
  double cf[16];  // BBH_CANDREG: the wave's candidate fragments in registers (pipelined forms, KD <= 16)
};

// Per translation unit (defined before this header is included): BBH_CANDREG = 1 keeps the candidate fragments of
// the distance GEMM in registers instead of re-reading them from wave-private LDS for every k-block (the LDS
// latency is exposed when a SIMD holds a single wave).
#ifndef BBH_CANDREG
#define BBH_CANDREG 0
#endif
// BBH_DIST_ASM = 1 (with BBH_CANDREG): the distance MFMAs are VGPR-form inline assembly, see mfma_f64_v below
#ifndef BBH_DIST_ASM
#define BBH_DIST_ASM 0
#endif
// BBH_MEAN_VALU_ONLY = 1: the translation unit's kernels are only launched with the mean contraction on the VALU
// (FusedArgs::mean_valu), so the MFMA form of the mean - and its accumulator - is not compiled in.
#ifndef BBH_MEAN_VALU_ONLY
#define BBH_MEAN_VALU_ONLY 0
#endif

// KIND >= 0: compile-time kernel kind (branch-free fast path); KIND < 0: runtime c.kind.
template <bool HAS_TBL, int KIND>
__device__ __forceinline__ void compute_kv(const WaveCtx& c, int tb, double (&kv)[4]) {
  d4 da = {0.0, 0.0, 0.0, 0.0}, db = {0.0, 0.0, 0.0, 0.0};
  const double* tf = c.tf + (int64_t)tb * c.kd * 64;
  int k = 0;
  for (; k + 3 < c.kd; k += 4) {  // loads first, then the MFMA chain (two independent accumulators)
    const double t0 = tf[k * 64], t1 = tf[(k + 1) * 64], t2 = tf[(k + 2) * 64], t3 = tf[(k + 3) * 64];
    const double c0 = c.candl[k * 64], c1 = c.candl[(k + 1) * 64], c2 = c.candl[(k + 2) * 64],
                 c3 = c.candl[(k + 3) * 64];
    da = mfma_f64(t0, c0, da);
    db = mfma_f64(t1, c1, db);
    da = mfma_f64(t2, c2, da);
    db = mfma_f64(t3, c3, db);
  }
  for (; k < c.kd; k++) da = mfma_f64(tf[k * 64], c.candl[k * 64], da);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const double r2 = fmax(da[r] + db[r], 0.0);
    double v;
    if (KIND == BBH_KERNEL_MATERN52) {
      const double rr = sqrt(r2);
      v = (1.0 + BBH_SQRT5 * rr + (5.0 / 3.0) * r2) * exp(-BBH_SQRT5 * rr);
    } else if (c.kind == BBH_KERNEL_MATERN12) {
      // exp(-r) is not smooth at r = 0: the |a|^2 + |b|^2 - 2ab form has an absolute error of ~1e-16 (|a|^2 +
      // |b|^2) in r2, i.e. ~1e-11 / r in r - harmless except for (nearly) coinciding points.  Lanes with
      // r2 < 1e-2 recompute the distance from direct differences of the same operands (a_i = -0.5 * A_aug,
      // b_c from LDS; exact zeros for coinciding points); the test is wave-uniform so that the loop is only
      // entered where some lane needs it (it used to run for every pair: 15 ms instead of 6 for this kernel).
      double d2 = r2;
      const bool nearp = r2 < 1e-2;
      if (__builtin_amdgcn_ballot_w64(nearp) != 0) {
        if (nearp) {
          const double* tfb = tf - c.l;
          const double* cb = c.candl - c.l;
          const int il = c.q + 4 * r, cl = c.l & 15;
          d2 = 0.0;
          for (int dim = 0; dim < c.dn; dim++) {
            const int o = (dim >> 2) * 64 + (dim & 3) * 16;
            const double df = cb[o + cl] + 0.5 * tfb[o + il];
            d2 = fma(df, df, d2);
          }
        }
      }
      v = (r2 > 1e7) ? 0.0 : exp(-sqrt(d2));  // r2 > 1e7 marks padding (A_aug norm slot = 1e8)
    } else {
      v = bbh_kfun_p(c.kind, r2);
    }
    if (HAS_TBL) v *= c.tbl[c.tc * c.T + c.taskext[16 * tb + 4 * r + c.q]];
    kv[r] = v;
  }
}

// Triangular region of a pass, k-block j0 + TT, as a compile-time recursion over TT (a plain
// `#pragma unroll` over tt exceeds the unroll threshold and would demote acc[] to scratch).
template <int W, int D, int TT, bool HAS_TBL, bool DO_MEAN, int KIND>
__device__ __forceinline__ void diag_steps(const WaveCtx& c, const double* rf, int j0, d4 (&acc)[W], double (&ring)[D],
                                           d4& accm) {
  if constexpr (TT < W) {
    constexpr int TOTAL = 2 * W * (W + 1);
    constexpr int BASE = 4 * (TT * W - (TT * (TT - 1)) / 2);  // fragments consumed before this k-block
    constexpr int CNT = W - TT;
    double kv[4];
    double mbv[4];
    if (DO_MEAN) {
#pragma unroll
      for (int r = 0; r < 4; r++) mbv[r] = c.mb[(int64_t)(4 * (j0 + TT) + r) * 64];
    }
    compute_kv<HAS_TBL, KIND>(c, j0 + TT, kv);
    if (DO_MEAN) {
#pragma unroll
      for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], mbv[r], accm);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int jj = 0; jj < CNT; jj++) {
        const int i = BASE + r * CNT + jj;
        acc[TT + jj] = mfma_f64(kv[r], ring[i % D], acc[TT + jj]);
        if (i + D < TOTAL) ring[i % D] = rf[(i + D) * 64];
        if (i % D == D - 1) __builtin_amdgcn_sched_barrier(0);  // keep the prefetch distance at D
      }
    diag_steps<W, D, TT + 1, HAS_TBL, DO_MEAN, KIND>(c, rf, j0, acc, ring, accm);
  }
}

// One pass over the column-block window [j0, j0 + W): accumulates ||v||^2 contributions into ss
// and (DO_MEAN) the mean/cross columns into accm.
//
// The R fragments of a pass are stored in exactly the order the MFMAs consume them (k-block tb,
// k-step r, column block jj), so the operand stream is one linear walk.  It is software-pipelined
// through a register ring of D fragments: fragment i + D is requested right after MFMA i has
// consumed ring slot i % D (all indices are compile-time constants; 4 W and the triangular total
// 2 W (W + 1) are multiples of D).  The first D fragments of a k-block are thus already in flight
// while its kernel values are computed.
template <int W, int D, bool HAS_TBL, bool DO_MEAN, int KIND>
__device__ __forceinline__ void pass_body(const WaveCtx& c, const double* rf, int j0, double (&ss)[4], d4& accm) {
  static_assert((4 * W) % D == 0 && (2 * W * (W + 1)) % D == 0, "ring depth must divide the stream");
  d4 acc[W];
#pragma unroll
  for (int jj = 0; jj < W; jj++) acc[jj] = (d4){0.0, 0.0, 0.0, 0.0};
  double ring[D];
#pragma unroll
  for (int i = 0; i < D; i++) ring[i] = rf[i * 64];
  // rectangular region: every column block of the window is active
  for (int tb = 0; tb < j0; tb++) {
    double kv[4];
    double mbv[4];
    if (DO_MEAN) {  // requested first (oldest in the in-order vmcnt queue): its wait never drains the ring
#pragma unroll
      for (int r = 0; r < 4; r++) mbv[r] = c.mb[(int64_t)(4 * tb + r) * 64];
    }
    compute_kv<HAS_TBL, KIND>(c, tb, kv);
    if (DO_MEAN) {  // before the ring loop, so that mbv is dead while the ring is live
#pragma unroll
      for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], mbv[r], accm);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4 * W; i++) {
      acc[i % W] = mfma_f64(kv[i / W], ring[i % D], acc[i % W]);
      ring[i % D] = rf[(i + D) * 64];
      if (i % D == D - 1) __builtin_amdgcn_sched_barrier(0);  // keep the prefetch distance at D
    }
    rf += 4 * W * 64;
  }
  // triangular region: k-block j0 + tt only reaches column blocks jj >= tt
  diag_steps<W, D, 0, HAS_TBL, DO_MEAN, KIND>(c, rf, j0, acc, ring, accm);
#pragma unroll
  for (int jj = 0; jj < W; jj++)
#pragma unroll
    for (int r = 0; r < 4; r++) ss[r] = fma(acc[jj][r], acc[jj][r], ss[r]);
}

// =====================================================================================================
// Software-pipelined pass (KD = k-steps of the distance GEMM known at compile time, Matérn-5/2 /
// generic kind alike).  Ablation on the 1e6 x 20 x 512 workload: variance-GEMM stream alone 4.27 ms,
// kernel-value stage alone 2.25 ms, un-pipelined kernel 5.96 ms — i.e. only a quarter of the VALU work
// was hidden by the second wave of the SIMD.  Here every wave computes the kernel values of k-block
// tb+1 in slices placed between the four k-steps of k-block tb's MFMAs (same basic block, fenced with
// sched_barrier so the slices stay where they are): slice 0 issues the training-fragment loads and the
// distance MFMAs, slices 1-2 evaluate two kernel values each, slice 3 carries the mean MFMAs.
// =====================================================================================================
template <int KD>
__device__ __forceinline__ void kvp_load(const double* tf_lane, int tb, double (&tfv)[KD]) {  // tf_lane = trainfrag + lane
  const double* tf = tf_lane + (int64_t)tb * KD * 64;
#pragma unroll
  for (int k = 0; k < KD; k++) tfv[k] = tf[k * 64];
}
template <int KD>
__device__ __forceinline__ void kvp_load(const WaveCtx& c, int tb, double (&tfv)[KD]) {
  kvp_load<KD>(c.tf, tb, tfv);
}

#if BBH_DIST_ASM
// VGPR-form MFMA through inline assembly.  With 32 column blocks the variance accumulators fill the whole
// AGPR half of the register file (256); the compiler gives every MFMA *intrinsic* of such a kernel an AGPR
// destination, so the two distance accumulators would push accumulators out to VGPRs and back around every
// use.  The "v" constraints keep these in arch VGPRs.  Hazards: the compiler does not see an MFMA here, so the
// consumers are placed by hand - the two accumulator chains alternate (a dependent MFMA follows a full 16-pass
// MFMA), and the VALU reads of the results happen after the next variance MFMA in program order (kblock_p).
__device__ __forceinline__ void mfma_f64_v0(d4& acc, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_f64_v(d4& acc, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
#endif

// r2 = da + db: the sum is left to the caller (first kernel-value micro-step), see above
template <int KD>
__device__ __forceinline__ void kvp_dist(const WaveCtx& c, const double (&tfv)[KD], d4& da, d4& db) {
#if BBH_CANDREG && !BBH_DIST_ASM
  da = (d4){0.0, 0.0, 0.0, 0.0};
  db = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < KD; k += 2) {
    da = mfma_f64(tfv[k], c.cf[k], da);
    if (k + 1 < KD) db = mfma_f64(tfv[k + 1], c.cf[k + 1], db);
  }
#elif BBH_CANDREG
  static_assert(KD >= 2, "two accumulator chains");
  mfma_f64_v0(da, tfv[0], c.cf[0]);
  mfma_f64_v0(db, tfv[1], c.cf[1]);
#pragma unroll
  for (int k = 2; k < KD; k += 2) {
    mfma_f64_v(da, tfv[k], c.cf[k]);
    if (k + 1 < KD) mfma_f64_v(db, tfv[k + 1], c.cf[k + 1]);
  }
#else
  da = (d4){0.0, 0.0, 0.0, 0.0};
  db = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < KD; k += 2) {
    da = mfma_f64(tfv[k], c.candl[k * 64], da);
    if (k + 1 < KD) db = mfma_f64(tfv[k + 1], c.candl[(k + 1) * 64], db);
  }
#endif
}

// Kernel values of NU (= 4) scaled squared distances in lockstep, cut into BBH_KV_STEPS micro-steps of NU to
// 2 NU VALU instructions so that the caller can place them between MFMAs in program order (the compiler's
// scheduler clumps library sqrt()/exp() calls behind the MFMAs even when asked to interleave them
// with sched_group_barrier).  Matérn-5/2: k(r2) = (1 + s + s^2/3) exp(-s), s = sqrt(5 r2); Matérn-3/2:
// (1 + s) exp(-s), s = sqrt(3 r2); RBF: exp(-r2 / 2) (the sqrt steps are skipped):
//   sqrt: v_rsq_f64 seed and one Newton step (error 1.5 eps^2 = 3e-16; the Goldschmidt + Newton form
//         with error O(eps^4) is kept behind BBH_KV_SQRT_NR=0 and measured 1 % slower);
//   exp:  -s = k ln2 + r, |r| <= ln2/2, degree-11 minimax polynomial (3e-18), scaled with v_ldexp_f64 (huge s, e.g. the
//         padding marker r2 = 1e8, underflows to 0 there); r2 = 0 is handled by a 1e-300 floor.
// Measured against the libm form on 1e6 x 512 values: tests/test_gpu_parity.py::test_pipelined_kernel_matches_plain_form, scripts/gpu_kv_accuracy.py.
#define BBH_KV_STEPS 18
#ifndef BBH_KV_SQRT_NR
#define BBH_KV_SQRT_NR 1  // 1: one Newton step on the v_rsq_f64 seed (4 VALU); 0: Goldschmidt + Newton (7 VALU)
#endif
#ifndef BBH_KV_NU
#define BBH_KV_NU 4  // values evaluated in lockstep (independent dependency chains per micro-step)
#endif
template <int NU>
struct KvState {
  double t[NU], y[NU], g[NU], h[NU], kf[NU], q[NU], tv[NU];
  int ki[NU], te[NU];
};

#define BBH_KV_EACH for (int u = 0; u < NU; u++)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int KVF, int NU, int step>
__device__ __forceinline__ void kv_micro(KvState<NU>& P, const WaveCtx& c, int tb, int r0, const d4& da, const d4& db,
                                         double (&out)[4]) {
  constexpr bool HAS_TBL = (KVF & 1) != 0, RBFK = (KVF & 2) != 0;  // table multiply / RBF instead of Matern
  constexpr bool M32K = (KVF & 4) != 0;  // Matern-3/2: k = (1 + s) exp(-s), s = sqrt(3 r2)  (5/2: 1 + s + s^2/3, s = sqrt(5 r2))
  constexpr double KC1 = M32K ? 3.0 : 5.0, KC2 = M32K ? 0.0 : 1.0 / 3.0;
  constexpr double LOG2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01,
                   LN2_LO = 1.90821492927058770002e-10;
  // register roles: t = 5 r2, later the reduced argument r; g = sqrt estimate, later s; h = half
  // reciprocal sqrt, later the exp polynomial; y = rsq seed / residuals
  switch (step) {
    case 0:
#if BBH_DIST_ASM
      // The distance MFMAs are inline assembly (kvp_dist): the compiler's hazard recogniser does not know that da / db
      // come from a 16-pass MFMA, whose VALU consumers must be >= 19 wait states behind it.  These issue
      // slots overlap with the variance MFMA that precedes this micro-step in program order.
      asm volatile("s_nop 15");
      asm volatile("s_nop 3");
#endif
#pragma unroll
      BBH_KV_EACH {
        const double r2 = da[r0 + u] + db[r0 + u];
        if (RBFK)
          P.g[u] = __builtin_fmin(__builtin_fmax(0.5 * r2, 0.0), 800.0);  // RBF: s = r2 / 2, no sqrt
        else
          P.t[u] = __builtin_fmax(KC1 * r2, 1e-300);
        if (HAS_TBL) P.te[u] = c.taskext[16 * tb + 4 * (r0 + u) + c.q];
      }
      break;
    case 1:
      if (RBFK) break;
#pragma unroll
      BBH_KV_EACH P.y[u] = __builtin_amdgcn_rsq(P.t[u]);
      break;
#if BBH_KV_SQRT_NR  // s = u + (u/2)(1 - u y), u = t y: error 1.5 eps^2 with eps = 2^-26 of v_rsq_f64
    case 2:
      if (RBFK) break;
#pragma unroll
      BBH_KV_EACH P.g[u] = P.t[u] * P.y[u];
      break;
    case 3:
      if (RBFK) break;
#pragma unroll
      BBH_KV_EACH P.y[u] = fma(-P.g[u], P.y[u], 1.0);
      break;
    case 4:
      if (RBFK) break;
#pragma unroll
      BBH_KV_EACH P.h[u] = 0.5 * P.g[u];
      break;
    case 5: break;
    case 6:
      if (RBFK) break;
#pragma unroll
      BBH_KV_EACH P.g[u] = fma(P.h[u], P.y[u], P.g[u]);  // g = s from here on (no clamp: v_ldexp_f64 takes any exponent to 0)
      break;
#else
    case 2:
#pragma unroll
      BBH_KV_EACH P.g[u] = P.t[u] * P.y[u];
#pragma unroll
      BBH_KV_EACH P.h[u] = 0.5 * P.y[u];
      break;
    case 3:
#pragma unroll
      BBH_KV_EACH P.y[u] = fma(-P.h[u], P.g[u], 0.5);
      break;
    case 4:
#pragma unroll
      BBH_KV_EACH P.g[u] = fma(P.g[u], P.y[u], P.g[u]);
#pragma unroll
      BBH_KV_EACH P.h[u] = fma(P.h[u], P.y[u], P.h[u]);
      break;
    case 5:
#pragma unroll
      BBH_KV_EACH P.y[u] = fma(-P.g[u], P.g[u], P.t[u]);
      break;
    case 6:
#pragma unroll
      BBH_KV_EACH P.g[u] = __builtin_fmin(fma(P.y[u], P.h[u], P.g[u]), 800.0);  // g = s from here on
      break;
#endif
    case 7:
#pragma unroll
      BBH_KV_EACH P.kf[u] = __builtin_rint(P.g[u] * -LOG2E);
      if (!RBFK) {
#pragma unroll
        BBH_KV_EACH P.q[u] = M32K ? 1.0 : fma(P.g[u], KC2, 1.0);
      }
      break;
    case 8:
#pragma unroll
      BBH_KV_EACH P.t[u] = fma(P.kf[u], -LN2_HI, -P.g[u]);  // t = reduced argument r from here on
      if (!RBFK) {
#pragma unroll
        BBH_KV_EACH P.q[u] = fma(P.q[u], P.g[u], 1.0);
      }
      break;
    case 9:
#pragma unroll
      BBH_KV_EACH P.t[u] = fma(P.kf[u], -LN2_LO, P.t[u]);
#pragma unroll
      BBH_KV_EACH {
        P.ki[u] = (int)P.kf[u];
        if (HAS_TBL) P.tv[u] = c.tbl[c.tc * c.T + P.te[u]];
      }
      break;
    // exp(r) on |r| <= ln2 / 2: degree-11 minimax polynomial of the relative error (Remez in 60-digit arithmetic:
    // 3.1e-18; 2.3e-17 with the coefficients rounded to fp64) - two FMAs fewer per value than the degree-13 Taylor form
    case 10:
#pragma unroll
      BBH_KV_EACH P.h[u] = fma(2.4994304884701822e-08, P.t[u], 2.763229327960932e-07);
      break;
#define BBH_KV_HORNER2(CA, CB)                      \
  _Pragma("unroll") BBH_KV_EACH P.h[u] = fma(P.h[u], P.t[u], CA); \
  _Pragma("unroll") BBH_KV_EACH P.h[u] = fma(P.h[u], P.t[u], CB);
    case 11: BBH_KV_HORNER2(2.7557622530873466e-06, 2.4801486521427063e-05) break;
    case 12: BBH_KV_HORNER2(0.00019841269432679235, 0.001388888895122399) break;
    case 13: BBH_KV_HORNER2(0.00833333333355927, 0.04166666666649277) break;
    case 14: BBH_KV_HORNER2(0.1666666666666617, 0.5000000000000018) break;
    case 15: BBH_KV_HORNER2(1.0, 1.0) break;
#undef BBH_KV_HORNER2
    case 16:
      if (!RBFK) {
#pragma unroll
        BBH_KV_EACH P.h[u] *= P.q[u];
      }
      break;
    default:
#pragma unroll
      BBH_KV_EACH {
        double v = __builtin_ldexp(P.h[u], P.ki[u]);
        if (HAS_TBL) v *= P.tv[u];
        out[r0 + u] = v;
      }
      break;
  }
}
#undef BBH_KV_EACH

// all micro-steps of the four values back to back (first k-block of a pass)
template <int KVF>
__device__ __forceinline__ void kv_all(const WaveCtx& c, int tb, const d4& da, const d4& db, double (&out)[4]) {
  KvState<4> P;
  static_for<0, BBH_KV_STEPS>([&](auto st) __attribute__((always_inline)) {
    kv_micro<KVF, 4, decltype(st)::value>(P, c, tb, 0, da, db, out);
  });
}

// one k-block: CNT column blocks starting at accumulator TT; kv = values of this block, kvn = values of
// the next block (computed here when NEXT); fragments are read from rf (+ lane) in consumption order
// NEXT: how the kernel values of k-block tb + 1 are obtained while this block's MFMAs run:
//   BBH_NEXT_NONE (last block of a pass), BBH_NEXT_COMPUTE (micro-steps between the MFMAs),
//   BBH_NEXT_LOAD (from the wave's kernel-value cache: an earlier pass computed and stored them).
// store: write this block's values to the cache (diagonal blocks of every pass but the last).
#ifndef BBH_RING
#define BBH_RING 8
#endif
#define BBH_NEXT_NONE 0
#define BBH_NEXT_COMPUTE 1
#define BBH_NEXT_LOAD 2
template <int W, int CNT, int TT, int KD, int KVF, bool DO_MEAN, int NEXT, int BASE, int REM>
__device__ __forceinline__ void kblock_p(const WaveCtx& c, const double* rf, int tb, const double (&kv)[4],
                                         double (&kvn)[4], d4 (&acc)[W], d4& accm, double (&ring)[BBH_RING], bool store) {
  // The R fragments of a pass are one linear stream (k-block, k-step, column block); they flow through
  // a BBH_RING-deep register ring that is never drained inside a pass: the fragment BBH_RING ahead is
  // requested right after a slot is consumed, across k-block boundaries too.  A block holds 4 CNT
  // fragments, so the slot of its first fragment is BASE in {0, 4} - a compile-time constant.
  // (2 waves x 8 loads in flight per SIMD cover the L2 latency; depth 4 measurably does not.)
  constexpr int D = BBH_RING;
  constexpr int TOT = 4 * CNT;
  double tfv[KD];
  d4 dsa, dsb;  // the two accumulator chains of the next block's distance GEMM
  double mbv[4];
  KvState<BBH_KV_NU> P;
  if (NEXT == BBH_NEXT_COMPUTE) kvp_load<KD>(c, tb + 1, tfv);
  if (NEXT == BBH_NEXT_LOAD) {
    if (tb + 1 < c.nl) {  // wave-uniform
#pragma unroll
      for (int rr = 0; rr < 4; rr++) kvn[rr] = c.kvl[(tb + 1) * 256 + rr * 64];
    } else {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) kvn[rr] = c.kvc[(int64_t)(tb + 1) * 256 + rr * 64];
    }
  }
  if (store && tb < c.ncache) {  // wave-uniform
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      if (tb < c.nl)
        c.kvl[tb * 256 + rr * 64] = kv[rr];
      else
        c.kvc[(int64_t)tb * 256 + rr * 64] = kv[rr];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if (DO_MEAN && r == 3) {
      if (BBH_MEAN_VALU_ONLY || c.al) {  // wave-uniform: alpha[16 tb + 4 rr + (lane >> 4)] from LDS (lgkmcnt, not the vmcnt ring)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) mbv[rr] = c.al[16 * tb + 4 * rr];
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) mbv[rr] = c.mb[(int64_t)(4 * tb + rr) * 64];
      }
    }
    static_for<0, CNT>([&](auto jc) __attribute__((always_inline)) {
      constexpr int jj = decltype(jc)::value;
      constexpr int i = r * CNT + jj;
      acc[TT + jj] = mfma_f64(kv[r], ring[(BASE + i) % D], acc[TT + jj]);
      if (i + D < TOT + REM) ring[(BASE + i) % D] = rf[(i + D) * 64];
      if constexpr (NEXT == BBH_NEXT_COMPUTE && (r == 1 || r == 2)) {
        if constexpr (BBH_KV_NU == 4) {  // four values in lockstep, micro-steps spread over both slices
          constexpr int m = (r - 1) * CNT + jj;
          static_for<(m * BBH_KV_STEPS) / (2 * CNT), ((m + 1) * BBH_KV_STEPS) / (2 * CNT)>(
              [&](auto st) __attribute__((always_inline)) {
                kv_micro<KVF, BBH_KV_NU, decltype(st)::value>(P, c, tb + 1, 0, dsa, dsb, kvn);
              });
        } else {  // two values per slice
          static_for<(jj * BBH_KV_STEPS) / CNT, ((jj + 1) * BBH_KV_STEPS) / CNT>(
              [&](auto st) __attribute__((always_inline)) {
                kv_micro<KVF, BBH_KV_NU, decltype(st)::value>(P, c, tb + 1, 2 * (r - 1), dsa, dsb, kvn);
              });
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (NEXT == BBH_NEXT_COMPUTE && r == 0) kvp_dist<KD>(c, tfv, dsa, dsb);
    if (DO_MEAN && r == 3) {
      if (BBH_MEAN_VALU_ONLY || c.al) {
        // Without pending points only column 0 of [alpha | -beta] is wanted: 4 FMAs on this lane's slice of
        // the k range instead of 4 MFMAs whose other 15 columns are zeros (the two cost 20 vs 256 cycles
        // of the shared DP pipe); accm[0] carries the partial sum, reduced over the four lanes of a
        // candidate after the last pass.
#pragma unroll
        for (int rr = 0; rr < 4; rr++) accm[0] = fma(kv[rr], mbv[rr], accm[0]);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) accm = mfma_f64(kv[rr], mbv[rr], accm);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

template <int W, int TT, int KD, int KVF, bool DO_MEAN>
__device__ __forceinline__ void diag_steps_p(const WaveCtx& c, const double* rf, int j0, double (&kv)[4], d4 (&acc)[W],
                                             d4& accm, double (&ring)[BBH_RING], bool store) {
  if constexpr (TT < W) {
    double kvn[4];
    constexpr int NEXT = (TT + 1 < W) ? BBH_NEXT_COMPUTE : BBH_NEXT_NONE;
    constexpr int BASE = (4 * (TT * W - (TT * (TT - 1)) / 2)) % BBH_RING;  // slot of the first fragment
    constexpr int REM = 2 * (W - TT) * (W - TT - 1);  // fragments of this pass after this block
    kblock_p<W, W - TT, TT, KD, KVF, DO_MEAN, NEXT, BASE, REM>(c, rf, j0 + TT, kv, kvn, acc, accm, ring, store);
    if (NEXT) {
#pragma unroll
      for (int r = 0; r < 4; r++) kv[r] = kvn[r];
    }
    diag_steps_p<W, TT + 1, KD, KVF, DO_MEAN>(c, rf + 4 * (W - TT) * 64, j0, kv, acc, accm, ring, store);
  }
}

// One pass over the column-block window [j0, j0 + W).  Kernel values are computed once per k-block
// and wave: the diagonal blocks [j0, j0 + W) of every pass but the last store theirs to the wave's
// cache slab (CACHE), and the rectangular region of later passes (k-blocks < j0) reads them back one
// block ahead instead of redoing the distance GEMM and the Matérn evaluation (fp64 VALU work is not
// hidden behind fp64 MFMAs on gfx950 - both issue to the same DP pipe, scripts/mfma_valu_overlap_probe.hip).
template <int W, int KD, int KVF, bool DO_MEAN>
__device__ __forceinline__ void pass_body_p(const WaveCtx& c, const double* rf, int j0, double (&ss)[4], d4& accm,
                                            bool cache) {
  d4 acc[W];
#pragma unroll
  for (int jj = 0; jj < W; jj++) acc[jj] = (d4){0.0, 0.0, 0.0, 0.0};
  double kv[4], kvn[4], ring[BBH_RING];
#pragma unroll
  for (int i = 0; i < BBH_RING; i++) ring[i] = rf[i * 64];
  if (cache && j0 > 0) {  // cache: the launch has more than one pass and slabs are in use (wave-uniform)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (c.nl > 0)
        kv[r] = c.kvl[r * 64];
      else
        kv[r] = c.kvc[r * 64];
    }
  } else {  // first k-block of the pass: not overlapped
    double tfv[KD];
    d4 dsa, dsb;
    kvp_load<KD>(c, 0, tfv);
    kvp_dist<KD>(c, tfv, dsa, dsb);
    kv_all<KVF>(c, 0, dsa, dsb, kv);
  }
  // rectangular region (k-blocks left of the window); the block after it is this pass's first diagonal
  // block, whose values nobody has computed yet
  const int nload = cache ? (j0 < c.ncache ? j0 : c.ncache) - 1 : 0;  // blocks whose successor is cached
  int tb = 0;
  for (; tb < nload; tb++) {  // next block's values come from the cache
    kblock_p<W, W, 0, KD, KVF, DO_MEAN, BBH_NEXT_LOAD, 0, 1 << 20>(c, rf, tb, kv, kvn, acc, accm, ring, false);
#pragma unroll
    for (int r = 0; r < 4; r++) kv[r] = kvn[r];
    rf += 4 * W * 64;
  }
  for (; tb < j0; tb++) {  // next block's values are computed between this block's MFMAs
    kblock_p<W, W, 0, KD, KVF, DO_MEAN, BBH_NEXT_COMPUTE, 0, 1 << 20>(c, rf, tb, kv, kvn, acc, accm, ring, false);
#pragma unroll
    for (int r = 0; r < 4; r++) kv[r] = kvn[r];
    rf += 4 * W * 64;
  }
  diag_steps_p<W, 0, KD, KVF, DO_MEAN>(c, rf, j0, kv, acc, accm, ring, cache && !DO_MEAN);
#pragma unroll
  for (int jj = 0; jj < W; jj++)
#pragma unroll
    for (int r = 0; r < 4; r++) ss[r] = fma(acc[jj][r], acc[jj][r], ss[r]);
}

// WMAX: column blocks per pass.  16: 128 accumulator registers, two waves per SIMD (256 registers each).
// 32: 256 accumulator registers (the AGPR half of the 512-register budget of a lone wave), one wave per SIMD:
// n <= 512 is a single pass - no kernel-value cache, no recomputation, no spills.
template <bool HAS_TBL, int KIND, int KD, int WMAX = 16>
__global__ __launch_bounds__(256, (WMAX > 16 ? 1 : 2)) void bbh_fused_posterior_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) double s_cand[];  // [4 waves][kd][64] (+ z[qS])
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cnd = l & 15, q = l >> 4;
  double* s_z = s_cand + 4 * (int64_t)a.kd * 64;
  double* s_alpha = s_z + (a.qz ? a.qS : 0);
  if (a.qz || a.mean_valu) {  // the only workgroup barrier, before any wave may leave
    if (a.qz)
      for (int s = threadIdx.x; s < a.qS; s += 256) s_z[s] = a.qz[s];
    if (a.mean_valu)
      for (int s = threadIdx.x; s < 16 * a.nb; s += 256) s_alpha[s] = a.meanB[(int64_t)s * 16];
    __syncthreads();
  }
  // per-wave state only from here on: no workgroup barrier is used below.  (A persistent variant - two
  // workgroups per CU walking the candidate blocks - measured 5 % slower than one workgroup per block:
  // the hardware dispatcher balances the tail better.)
  const int64_t blk = blockIdx.x;
  const int64_t tile0 = (blk * 4 + w) * 16;
  if (tile0 >= a.N) return;  // whole wave out of range
  const int64_t row = (tile0 + cnd < a.N) ? tile0 + cnd : a.N - 1;
  const double* xr = a.X + row * a.ldx;
  double* candw = s_cand + (int64_t)w * a.kd * 64;

  // ---- candidate fragments: b = x * scl + ofs, augmented with [1, |b|^2] -------------------
  // (loads in independent groups of four k-steps with clamped indices: as `if (dim < dn) v = fma(xr[numcol[dim]], ...)`
  // every k-step was its own divergent block with two dependent memory round trips)
  double nbsum = 0.0;
  for (int k0 = 0; k0 < a.kd; k0 += 4) {
    int xcol[4];
    double xval[4], xscl[4], xofs[4];
#pragma unroll
    for (int u = 0; u < 4; u++) xcol[u] = a.numcol[(4 * (k0 + u) + q < a.dn) ? 4 * (k0 + u) + q : a.dn - 1];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int dimc = (4 * (k0 + u) + q < a.dn) ? 4 * (k0 + u) + q : a.dn - 1;
      xval[u] = xr[xcol[u]];
      xscl[u] = a.scl[dimc];
      xofs[u] = a.ofs[dimc];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (k0 + u < a.kd) {
        double v = 0.0;
        if (4 * (k0 + u) + q < a.dn) {
          v = fma(xval[u], xscl[u], xofs[u]);
          nbsum = fma(v, v, nbsum);
        }
        candw[(k0 + u) * 64 + l] = v;
      }
    }
  }
  nbsum += __shfl_xor(nbsum, 16, 64);
  nbsum += __shfl_xor(nbsum, 32, 64);
  {
    const int k1 = a.dn >> 2, q1 = a.dn & 3;  // slot dn: 1.0
    if (q == q1) candw[k1 * 64 + l] = 1.0;
    const int k2 = (a.dn + 1) >> 2, q2 = (a.dn + 1) & 3;  // slot dn + 1: |b|^2
    if (q == q2) candw[k2 * 64 + l] = nbsum;
  }
  int tc = 0;
  constexpr bool has_tbl = HAS_TBL;
  if (has_tbl && a.task_col >= 0) {
    tc = (int)xr[a.task_col];
    tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
  }

  WaveCtx c;
  c.tf = a.trainfrag + l;
  c.candl = candw + l;
  c.mb = a.meanB + l;
  c.tbl = a.tasktbl;
  c.taskext = a.taskext;
  // Claim a kernel-value cache slab for this wave.  The pool is partitioned by XCD (HW_REG_XCC_ID): a
  // slab is only ever touched through one XCD's L2, so re-use by a later wave needs no L2 write-back
  // (an agent-scope release fence per wave costs 30 % of the kernel).  Each partition has twice as
  // many slabs as the XCD can hold resident waves (8 per CU), so linear probing from a hashed start
  // ends after a few attempts.
  int slab = 0;
  if (a.kvcache) {
    if (l == 0) {
      const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) % (unsigned)a.nxcc;  // XCC_ID[3:0]
      const unsigned per = (unsigned)(a.nslab / a.nxcc);
      unsigned sidx = (unsigned)(((uint64_t)(blk * 4 + w) * 2654435761ull) % per);
      int* flags = a.slab_flags + xcc * per;
      while (atomicCAS(&flags[sidx], 0, 1) != 0) sidx = (sidx + 1 == per) ? 0u : sidx + 1;
      slab = (int)(xcc * per + sidx);
    }
    slab = __builtin_amdgcn_readfirstlane(slab);
  }
  c.kvc = a.kvcache ? a.kvcache + ((int64_t)slab * (a.ncache - a.nl) - a.nl) * 256 + l : nullptr;
  c.kvl = (bbh_lds_double*)(s_alpha + (a.mean_valu ? 16 * a.nb : 0) + (int64_t)w * a.nl * 256 + l);
  c.nl = a.nl;
  c.ncache = a.ncache;
  c.al = a.mean_valu ? (const bbh_lds_double*)(s_alpha + q) : (const bbh_lds_double*)nullptr;
  c.kd = a.kd;
  c.kind = a.kind;
  c.T = a.T;
  c.tc = tc;
  c.q = q;
  c.l = l;
  c.dn = a.dn;
#if BBH_CANDREG
  if constexpr (KD > 0) {  // lane l wrote exactly the LDS words it reads back: wave-private, no barrier needed
#pragma unroll
    for (int k = 0; k < KD; k++) c.cf[k] = candw[k * 64 + l];
  }
#endif

  double ss[4] = {0.0, 0.0, 0.0, 0.0};
  d4 accm = {0.0, 0.0, 0.0, 0.0};
  double kv[4];

  if (a.with_var) {
    int j0 = 0;
    for (int ps = 0; ps < a.npass; ps++) {
      const int W = a.pass_w[ps];
      const double* rf = a.rfrag + a.pass_off[ps] + l;
      const bool last = (ps == a.npass - 1);
      if constexpr (KD > 0 && WMAX == 32) {  // one wave per SIMD: windows of 32 column blocks (remainders: 8, 16, 24)
        constexpr int KVF = (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0);
        const bool use_cache = (a.ncache > 0);
        if (!last) {
          pass_body_p<32, KD, KVF, false>(c, rf, j0, ss, accm, use_cache);
        } else {
          switch (W) {
#if BBH_W32_REMAINDERS
            case 8: pass_body_p<8, KD, KVF, true>(c, rf, j0, ss, accm, use_cache); break;
            case 16: pass_body_p<16, KD, KVF, true>(c, rf, j0, ss, accm, use_cache); break;
            case 24: pass_body_p<24, KD, KVF, true>(c, rf, j0, ss, accm, use_cache); break;
#endif
            default: pass_body_p<32, KD, KVF, true>(c, rf, j0, ss, accm, use_cache); break;
          }
        }
      } else if constexpr (KD > 0) {  // software-pipelined passes
        const bool use_cache = (a.ncache > 0);
        if (!last) {
          pass_body_p<16, KD, (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0), false>(c, rf, j0, ss, accm, use_cache);
        } else {
          switch (W) {
            case 4: pass_body_p<4, KD, (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0), true>(c, rf, j0, ss, accm, use_cache); break;
            case 8: pass_body_p<8, KD, (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0), true>(c, rf, j0, ss, accm, use_cache); break;
            case 12: pass_body_p<12, KD, (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0), true>(c, rf, j0, ss, accm, use_cache); break;
            default: pass_body_p<16, KD, (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0), true>(c, rf, j0, ss, accm, use_cache); break;
          }
        }
      } else if (!last) {
        pass_body<16, 8, HAS_TBL, false, KIND>(c, rf, j0, ss, accm);
      } else {
        switch (W) {
          case 4: pass_body<4, 8, HAS_TBL, true, KIND>(c, rf, j0, ss, accm); break;
          case 8: pass_body<8, 8, HAS_TBL, true, KIND>(c, rf, j0, ss, accm); break;
          case 12: pass_body<12, 8, HAS_TBL, true, KIND>(c, rf, j0, ss, accm); break;
          default: pass_body<16, 8, HAS_TBL, true, KIND>(c, rf, j0, ss, accm); break;
        }
      }
      j0 += W;
    }
    // pending block(s): mean/cross columns only
    if constexpr (WMAX <= 16) {
      for (int tb = a.nb; tb < a.nb_ext; tb++) {
        compute_kv<HAS_TBL, KIND>(c, tb, kv);
#pragma unroll
        for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], c.mb[(int64_t)(4 * tb + r) * 64], accm);
      }
    }
  } else if constexpr (KD > 0 && WMAX <= 16) {
    // mean-only pass (cross-covariances with the pending points of a greedy step, conditional means of qLogNEHVI):
    // the staged kernel-value evaluation of the pipelined form instead of the libm one (1.65 -> ~1.1 ms per 1e6
    // candidates on the bench shape); nothing to overlap it with but the other wave of the SIMD
    constexpr int KVF = (HAS_TBL ? 1 : 0) | (KIND == BBH_KERNEL_RBF ? 2 : 0) | (KIND == BBH_KERNEL_MATERN32 ? 4 : 0);
    for (int tb = 0; tb < a.nb_ext; tb++) {
      double tfv[KD], mbv[4];
      d4 dsa, dsb;
      kvp_load<KD>(c, tb, tfv);
#pragma unroll
      for (int r = 0; r < 4; r++) mbv[r] = c.mb[(int64_t)(4 * tb + r) * 64];
      kvp_dist<KD>(c, tfv, dsa, dsb);
      kv_all<KVF>(c, tb, dsa, dsb, kv);
#pragma unroll
      for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], mbv[r], accm);
    }
  } else if constexpr (WMAX <= 16) {  // (the one-wave form is launched for variance passes without pending columns only)
    for (int tb = 0; tb < a.nb_ext; tb++) {
      compute_kv<HAS_TBL, KIND>(c, tb, kv);
#pragma unroll
      for (int r = 0; r < 4; r++) accm = mfma_f64(kv[r], c.mb[(int64_t)(4 * tb + r) * 64], accm);
    }
  }

  if (a.kvcache && l == 0) atomicExch(&a.slab_flags[slab], 0);  // every cached value has been read back
  if (c.al && a.with_var) {  // VALU mean: lane (q, cnd) holds a quarter of candidate cnd's sum in accm[0]
    double mp = accm[0];
    mp += __shfl_xor(mp, 16, 64);
    mp += __shfl_xor(mp, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; r++) accm[r] = __shfl(mp, q + 4 * r, 64);  // epilogue layout: reg r <-> candidate q + 4 r
  }
  // ---- epilogue: lane (q, cnd), reg r  <->  candidate q + 4 r, column cnd -------------------
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double s = ss[r];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    ss[r] = s;
  }
  const double s2 = a.ysd * a.ysd;
  double mval[4], vval[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int m = q + 4 * r;  // candidate within the tile
    const int tcm = __shfl(tc, m, 64);
    const int64_t gi = tile0 + m;
    double pv = a.prior_scale;
    if (has_tbl) pv = a.tasktbl[tcm * a.T + tcm];
    const double mc = (has_tbl && a.taskmean) ? a.taskmean[tcm] : a.mean_const;
    mval[r] = a.ybar + a.ysd * (mc + accm[r]);  // meaningful in the cnd == 0 lanes
    vval[r] = s2 * (pv - ss[r]);
    if (gi < a.N) {
      if (cnd == 0) {
        if (a.mean) a.mean[gi] = mval[r];
        if (a.with_var && a.var) a.var[gi] = vval[r];
      } else if (cnd <= a.p && a.cross) {
        a.cross[gi * a.p + (cnd - 1)] = s2 * accm[r];
      }
    }
  }
  // ---- fused qLogEI (q' = 1): lane (q, cnd) evaluates samples s = q, q+4, ... of candidate cnd ---
  if (a.qz && a.with_var) {
    double mu = 0.0, vr = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {  // candidate cnd = q' + 4 r' lives in lane 16 q', register r'
      const double tm = __shfl(mval[r], (cnd & 3) * 16, 64);
      const double tv = __shfl(vval[r], (cnd & 3) * 16, 64);
      if ((cnd >> 2) == r) {
        mu = tm;
        vr = tv;
      }
    }
    if (!(vr > 0.0)) {  // 1x1 psd_safe_cholesky jitter rule
      vr += 1e-8;
      if (!(vr > 0.0)) {
        vr += 1e-7;
        if (!(vr > 0.0)) vr += 1e-6;
      }
    }
    const double inv_tau = 1e6;  // 1 / tau_relu
    const double ca = (a.q_sign * mu - a.q_best_f) * inv_tau;
    const double cb = a.q_sign * sqrt(fmax(vr, 0.0)) * inv_tau;
    double sum = 0.0;
    for (int s = q; s < a.qS; s += 4) {
      const double t = fma(cb, s_z[s], ca);
      double sp;
      if (t > 20.0)
        sp = t;
      else if (t < -750.0)
        sp = 0.0;
      else
        sp = log1p(exp(t));
      const double dd = fma(t, t, 1.0);  // as bbh_fatplus_core (bbh_acq.hip): rcp seed + two Newton steps
      double yy = __builtin_amdgcn_rcp(dd);
      yy = fma(fma(-dd, yy, 1.0), yy, yy);
      yy = fma(fma(-dd, yy, 1.0), yy, yy);
      sum += fma(0.1, yy, sp);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const int64_t gi = tile0 + cnd;
    if (q == 0 && gi < a.N) {
      double sc = log(1e-6) + log(sum) - log((double)a.qS);
      if (a.q_alive && !a.q_alive[gi]) sc = -INFINITY;
      a.q_scores[gi] = sc;
    }
  }
}


// dynamic LDS beyond 64 KB has to be requested per kernel (the kernel-value cache may use up to half a CU's LDS)
#define BBH_FUSED_ALLOW_LDS(KERNEL, BYTES)                                                              \
  do {                                                                                                 \
    static size_t allowed[64]; /* per kernel instantiation (one expansion each) and device */           \
    int dev_ = 0;                                                                                      \
    (void)hipGetDevice(&dev_);                                                                         \
    dev_ &= 63;                                                                                        \
    if ((size_t)(BYTES) > 48 * 1024 && (size_t)(BYTES) > allowed[dev_]) {                              \
      (void)hipFuncSetAttribute((const void*)(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)); \
      allowed[dev_] = (size_t)(BYTES);                                                                 \
    }                                                                                                  \
  } while (0)

// one launcher per translation unit (KD = compile-time k-steps of the distance GEMM; 0 = runtime k-steps,
// every kernel kind).  m52: Matérn-5/2 instantiation, otherwise the runtime-kind one (KD = 0 only).
// KD > 0: kind selects the RBF / Matérn-3/2 instantiation (both without table only), otherwise the Matérn-5/2 one
// with or without the task / outputscale table.  (A run-time table flag inside the micro-steps cost the default kernel 7 %, so
// it stays a template parameter; Matérn-1/2 and RBF / Matérn-3/2 with a table take the plain form.)
void bbh_fused_launch_kd0(bool has_tbl, bool m52, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd2(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd4(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd6(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd8(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd12(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
void bbh_fused_launch_kd16(int kind, bool has_tbl, dim3 grid, dim3 block, size_t lds, hipStream_t s, const FusedArgs& a);
