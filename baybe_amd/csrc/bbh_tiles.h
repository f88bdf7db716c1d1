// 64 x 64 tile toolkit of the fit's dataflow kernels (bbh_linalg.hip: bbh_potrf_tiles_kernel, bbh_fitflow.hip: bbh_fit_flow_kernel):
// 16 x 16-blocked MFMA products on LDS tiles, the factor-and-invert of a diagonal tile, tile loads / stores, and the
// publish / wait protocol on epoch-stamped flags.  256 threads per workgroup, LDS tiles [64][PD_LD].
#pragma once
#include <type_traits>

#include "bbh_common.h"

__device__ __forceinline__ void pd_publish(int* flag, int epoch);

// ---- the same 64x64 diagonal block, blocked 16 x 16 on the fp64 MFMA ------------------------------------
// The register form above is one wave walking 64 dependent pivots and then 64 substitution steps: 46 us per block, 8
// blocks in sequence at n = 512 = 47 % of a fit evaluation (profiles/r02_fit_kernel_stats.csv).  Here the block is a
// 4 x 4 grid of 16 x 16 sub-blocks in LDS: per block column J the diagonal sub-block is factorised and inverted in the
// registers of 16 lanes (16 pivots), the sub-diagonal panel is one MFMA chain per sub-block against that inverse, the
// trailing sub-blocks take rank-16 updates on the MFMA, and L^-1 is assembled block column by block column from the
// diagonal inverses - 4 short dependent stages instead of 128 long ones.
//   fragment layouts (bbh_common.h): A (16x4) lane l <- A[l & 15][4 ks + (l >> 4)],  B (4x16) lane l <- B[4 ks + (l >> 4)][l & 15],
//   C lane l, reg r <-> C[(l >> 4) + 4 r][l & 15]
#define PD_LD 66  // LDS row pitch (doubles)
__device__ __forceinline__ d4 pd_mul_nt(const double (*a)[PD_LD], int ar, int ac, const double (*b)[PD_LD], int br, int bc, int l) {
  // C = A[ar.., ac..] (16x16) * B[br.., bc..]^T (16x16):  C[m][n] = sum_k A[m][k] B[n][k]
  d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 4; ks++)
    c = mfma_f64(a[ar + (l & 15)][ac + 4 * ks + (l >> 4)], b[br + (l & 15)][bc + 4 * ks + (l >> 4)], c);
  return c;
}
__device__ __forceinline__ d4 pd_mul_nn(const double (*a)[PD_LD], int ar, int ac, const double (*b)[PD_LD], int br, int bc, int l, d4 c) {
  // C += A[ar.., ac..] (16x16) * B[br.., bc..] (16x16):  C[m][n] += sum_k A[m][k] B[k][n]
#pragma unroll
  for (int ks = 0; ks < 4; ks++)
    c = mfma_f64(a[ar + (l & 15)][ac + 4 * ks + (l >> 4)], b[br + 4 * ks + (l >> 4)][bc + (l & 15)], c);
  return c;
}

// Row broadcasts for the 16 x 16 diagonal sub-blocks: the 16 lanes of a DPP row hold the 16 rows of the sub-block, and
// gfx90a+ has `row_newbcast:n` (lane n of every row to all its lanes) on the 64-bit v_mov and v_fmac.  One instruction
// instead of two v_readlane_b32 + an SGPR operand: the readlane form of this code was 796 readlanes and 272 hazard nops in
// 2 500 instructions and spilled SGPRs into VGPR lanes.  The `s_nop 1` in front of every DPP instruction covers the "VALU
// write -> DPP read" hazard (2 wait states) whatever the compiler places before the statement.
#ifndef BBH_DIAG_DPP
#define BBH_DIAG_DPP 1
#endif
template <int LANE>
__device__ __forceinline__ double pd_bcast(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(LANE));
  return r;
}
// acc += bcast_LANE(src) * (-mul).  GUARD: src may have been written by the instruction just before (first member of an
// update series: the scaled pivot column); the others read a register that has been at rest for many instructions.
template <int LANE, bool GUARD>
__device__ __forceinline__ void pd_fmac_bcast_neg(double& acc, double src, double mul) {
  if constexpr (GUARD)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(LANE));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc)
                 : "v"(src), "v"(mul), "n"(LANE));
}
template <int I, int N, class F>
__device__ __forceinline__ void pd_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pd_static_for<I + 1, N>(f);
  }
}

// Factor and invert the 64 x 64 SPD block held in LDS array a (in place: lower factor L, strict upper zeroed) into x =
// L^-1 (lower); s is scratch.  256 threads, all LDS arrays [64][PD_LD]; x must be zero on entry.  row0: global index of
// the block's first row (failure reports row0 + pivot + 1 through *info).
// nbk < 4 (one-workgroup fit evaluation of a model with n <= 16 nbk rows): the sub-blocks from nbk on are identity in a AND in x on
// entry and are left alone.
__device__ __forceinline__ void pd_factor_block(double (*a)[PD_LD], double (*x)[PD_LD], double (*s)[PD_LD], int64_t row0, int* info,
                                                int* early_flag = nullptr, int epoch = 0, int nbk = 4) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  for (int jb = 0; jb < nbk; jb++) {
    // (tile-dataflow caller: stores issued before this call have landed by now - publish them without a stall)
    if (jb == 1 && early_flag) pd_publish(early_flag, epoch);
    const int o = 16 * jb;
    if (w == 0) {  // diagonal sub-block: lane i < 16 holds row i; pivots travel by v_readlane
      double row[16];
      const int i = l & 15;
#pragma unroll
      for (int k = 0; k < 16; k++) row[k] = a[o + i][o + k];
      int bad = 0;
      double rd[16];
      double xc[16];
#if BBH_DIAG_DPP
      pd_static_for<0, 16>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const double djj = pd_bcast<j>(row[j]);
        bad = (!(djj > 0.0) && bad == 0) ? j + 1 : bad;
        double rs = __builtin_amdgcn_rsq(djj);
        rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
        rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
        rd[j] = rs;  // the same in every lane: 1 / l_jj
        row[j] = (i == j) ? djj * rs : row[j] * rs;
        pd_static_for<j + 1, 16>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          pd_fmac_bcast_neg<k, k == j + 1>(row[k], row[j], row[j]);  // row[k] -= l_kj * row[j]   (meaningful for i >= k)
        });
      });
      if (l == 0 && bad) atomicCAS(info, 0, (int)(row0 + o + bad));
      if (l < 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) a[o + i][o + k] = (k <= i) ? row[k] : 0.0;
      }
      // inverse of the 16 x 16 factor: lane c < 16 owns column c (forward substitution, l_rk = lane r of row[k])
      pd_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        double acc = (r == i) ? 1.0 : 0.0;
        pd_static_for<0, r>([&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          pd_fmac_bcast_neg<r, false>(acc, row[k], xc[k]);
        });
        xc[r] = (r >= i) ? acc * rd[r] : 0.0;
      });
#else
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const double djj = bbh_readlane_f64(row[j], j);
        bad = (!(djj > 0.0) && bad == 0) ? j + 1 : bad;
        double rs = __builtin_amdgcn_rsq(djj);
        rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
        rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
        rd[j] = rs;  // wave-uniform 1 / l_jj
        row[j] = (i == j) ? djj * rs : row[j] * rs;
#pragma unroll
        for (int k = j + 1; k < 16; k++) {
          const double lkj = bbh_readlane_f64(row[j], k);
          row[k] = fma(-row[j], lkj, row[k]);  // meaningful for i >= k
        }
      }
      if (l == 0 && bad) atomicCAS(info, 0, (int)(row0 + o + bad));
      if (l < 16) {
#pragma unroll
        for (int k = 0; k < 16; k++) a[o + i][o + k] = (k <= i) ? row[k] : 0.0;
      }
      // inverse of the 16 x 16 factor: lane c < 16 owns column c (forward substitution, L rows by readlane)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; k++) acc = fma(-bbh_readlane_f64(row[k], r), xc[k], acc);  // l_rk = row r, entry k
        xc[r] = (r >= i) ? acc * rd[r] : 0.0;
      }
#endif
      if (l < 16) {
#pragma unroll
        for (int r = 0; r < 16; r++) x[o + r][o + i] = xc[r];
      }
    }
    __syncthreads();
    // panel: L_IJ = A_IJ T_J^T for the sub-blocks below the diagonal one, one wave each, in place (a wave's LDS
    // operations complete in order: its operand reads precede its writes)
    for (int ib = jb + 1 + w; ib < nbk; ib += 4) {
      const d4 c = pd_mul_nt(a, 16 * ib, o, x, o, o, l);
#pragma unroll
      for (int r = 0; r < 4; r++) a[16 * ib + (l >> 4) + 4 * r][o + (l & 15)] = c[r];
    }
    __syncthreads();
    // trailing update: A_IK -= L_IJ L_KJ^T for jb < K <= I (at most 6 sub-blocks, dealt to the waves)
    int cnt = 0;
    for (int ib = jb + 1; ib < nbk; ib++)
      for (int kb = jb + 1; kb <= ib; kb++, cnt++) {
        if ((cnt & 3) != w) continue;
        const d4 c = pd_mul_nt(a, 16 * ib, o, a, 16 * kb, o, l);
#pragma unroll
        for (int r = 0; r < 4; r++) a[16 * ib + (l >> 4) + 4 * r][16 * kb + (l & 15)] -= c[r];
      }
    __syncthreads();
  }
  // zero the strict upper triangle of L
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    if (j > i) a[i][j] = 0.0;
  }
  __syncthreads();
  // L^-1 by block columns: X_IJ = -T_I (sum_{K=J}^{I-1} L_IK X_KJ), wave J owns block column J (rows I = J+1..3 in turn)
  {
    const int jb = w, oj = 16 * jb;
    for (int ib = jb + 1; ib < nbk; ib++) {
      d4 c = {0.0, 0.0, 0.0, 0.0};
      for (int kb = jb; kb < ib; kb++) c = pd_mul_nn(a, 16 * ib, 16 * kb, x, 16 * kb, oj, l, c);
#pragma unroll
      for (int r = 0; r < 4; r++) s[16 * ib + (l >> 4) + 4 * r][oj + (l & 15)] = c[r];
      // (wave-private region of s and x: block column jb; LDS operations of one wave complete in order)
      d4 e = {0.0, 0.0, 0.0, 0.0};
      e = pd_mul_nn(x, 16 * ib, 16 * ib, s, 16 * ib, oj, l, e);  // T_I is the diagonal sub-block of x
#pragma unroll
      for (int r = 0; r < 4; r++) x[16 * ib + (l >> 4) + 4 * r][oj + (l & 15)] = -e[r];
    }
  }
  __syncthreads();
}

// ---- the same factor-and-invert with look-ahead (the tile-dataflow kernels' form; four sub-block rows) -----------------------
// pd_factor_block runs its stages behind one another: diagonal sub-block (wave 0 alone, three waves at the barrier) -> panel ->
// trailing update (two rounds over four waves) -> next stage, then zeroes the upper triangle and assembles L^-1 block column by
// block column: 15 barriers and 14.7-16 us per tile, 60 % of a step of the Cholesky chain (profiles/r05_flow_trace_512.log).  Here
// wave 0 owns the critical chain - after the panel of stage j it updates only the NEXT diagonal sub-block and goes straight on to
// factor it - while waves 1-3 take the rest of the trailing update and every product of the inverse whose operands are final:
//   after panel(j):    w0: A_{j+1,j+1} -= L_{j+1,j} L_{j+1,j}^T, then diag(j+1)        w1-3: the other trailing sub-blocks, P_ij products
//   after diag(j+1):   w1-3: panel(j+1) = A_{i,j+1} T_{j+1}^T, and X_{j+1,k} = -T_{j+1} P_{j+1,k}
// with P_ij = sum_{k=j}^{i-1} L_ik X_kj (X_jj = T_j).  Eight barriers; what follows the last diagonal sub-block is one product.
// x needs no initialisation (the upper sub-blocks are zeroed here); a's upper off-diagonal sub-blocks are zeroed at the end.
__device__ __forceinline__ void pd_diag16(double (*a)[PD_LD], double (*x)[PD_LD], int o, int64_t row0, int* info) {
  const int l = threadIdx.x & 63, i = l & 15;
  double row[16], rd[16], xc[16];
#pragma unroll
  for (int k = 0; k < 16; k++) row[k] = a[o + i][o + k];
  int bad = 0;
  pd_static_for<0, 16>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const double djj = pd_bcast<j>(row[j]);
    bad = (!(djj > 0.0) && bad == 0) ? j + 1 : bad;
    double rs = __builtin_amdgcn_rsq(djj);
    rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
    rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
    rd[j] = rs;  // the same in every lane: 1 / l_jj
    row[j] = (i == j) ? djj * rs : row[j] * rs;
    pd_static_for<j + 1, 16>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      pd_fmac_bcast_neg<k, k == j + 1>(row[k], row[j], row[j]);  // row[k] -= l_kj * row[j]   (meaningful for i >= k)
    });
  });
  if (l == 0 && bad) atomicCAS(info, 0, (int)(row0 + o + bad));
  if (l < 16) {
#pragma unroll
    for (int k = 0; k < 16; k++) a[o + i][o + k] = (k <= i) ? row[k] : 0.0;
  }
  // inverse of the 16 x 16 factor: lane c < 16 owns column c (forward substitution, l_rk = lane r of row[k]); two chains per row
  pd_static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    double acc0 = (r == i) ? 1.0 : 0.0, acc1 = 0.0;
    pd_static_for<0, r>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr ((k & 1) == 0)
        pd_fmac_bcast_neg<r, false>(acc0, row[k], xc[k]);
      else
        pd_fmac_bcast_neg<r, false>(acc1, row[k], xc[k]);
    });
    xc[r] = (r >= i) ? (acc0 + acc1) * rd[r] : 0.0;
  });
  if (l < 16) {
#pragma unroll
    for (int r = 0; r < 16; r++) x[o + r][o + i] = xc[r];
  }
}

// early_flag: a tile stored before the call is published from inside (its stores have landed by then); early_wt: it was stored
// write-through (pd_store_tile_wt) - drain + flag, no release fence
__device__ __forceinline__ void pd_factor_block4(double (*a)[PD_LD], double (*x)[PD_LD], double (*s)[PD_LD], int64_t row0, int* info,
                                                 int* early_flag = nullptr, int epoch = 0, bool early_wt = false) {
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  auto panel = [&](int ib, int jb) {  // L_ib,jb = A_ib,jb T_jb^T (in place)
    const d4 c = pd_mul_nt(a, 16 * ib, 16 * jb, x, 16 * jb, 16 * jb, l);
#pragma unroll
    for (int r = 0; r < 4; r++) a[16 * ib + (l >> 4) + 4 * r][16 * jb + (l & 15)] = c[r];
  };
  auto trail = [&](int ib, int kb, int jb) {  // A_ib,kb -= L_ib,jb L_kb,jb^T
    const d4 c = pd_mul_nt(a, 16 * ib, 16 * jb, a, 16 * kb, 16 * jb, l);
#pragma unroll
    for (int r = 0; r < 4; r++) a[16 * ib + (l >> 4) + 4 * r][16 * kb + (l & 15)] -= c[r];
  };
  auto pprod = [&](int ib, int jb) {  // P_ib,jb = sum_{kb = jb}^{ib - 1} L_ib,kb X_kb,jb  -> s
    d4 c = {0.0, 0.0, 0.0, 0.0};
    for (int kb = jb; kb < ib; kb++) c = pd_mul_nn(a, 16 * ib, 16 * kb, x, 16 * kb, 16 * jb, l, c);
#pragma unroll
    for (int r = 0; r < 4; r++) s[16 * ib + (l >> 4) + 4 * r][16 * jb + (l & 15)] = c[r];
  };
  auto xblock = [&](int ib, int jb) {  // X_ib,jb = -T_ib P_ib,jb
    d4 e = {0.0, 0.0, 0.0, 0.0};
    e = pd_mul_nn(x, 16 * ib, 16 * ib, s, 16 * ib, 16 * jb, l, e);
#pragma unroll
    for (int r = 0; r < 4; r++) x[16 * ib + (l >> 4) + 4 * r][16 * jb + (l & 15)] = -e[r];
  };
  auto zero_upper = [&](double (*m)[PD_LD]) {  // the six upper off-diagonal sub-blocks, by the 192 threads of waves 1-3
    for (int e = t - 64; e < 6 * 256; e += 192) {
      const int blk = e >> 8, r = (e >> 4) & 15, c = e & 15;
      const int ib = blk < 3 ? 0 : blk < 5 ? 1 : 2, jb = blk < 3 ? 1 + blk : blk < 5 ? blk - 1 : 3;
      m[16 * ib + r][16 * jb + c] = 0.0;
    }
  };
  // ---- stage 0 ----
  if (w == 0)
    pd_diag16(a, x, 0, row0, info);
  else
    zero_upper(x);
  __syncthreads();  // B0
  if (w > 0) panel(w, 0);
  __syncthreads();  // B1
  if (w == 0) {
    trail(1, 1, 0);
    pd_diag16(a, x, 16, row0, info);
  } else if (w == 1) {
    trail(2, 1, 0);
    trail(2, 2, 0);
  } else if (w == 2) {
    trail(3, 1, 0);
    trail(3, 2, 0);
  } else {
    trail(3, 3, 0);
    pprod(1, 0);
  }
  // (tile-dataflow caller: stores issued before this call have landed by now - publish them without a stall)
  if (early_flag) {
    if (early_wt)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  }
  __syncthreads();  // B2
  if (early_flag && t == 0) {
    if (!early_wt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2: cdna guide, Guideline 16 pitfall 12)
    }
    __hip_atomic_store(early_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- stage 1 ----
  if (w == 1) panel(2, 1);
  if (w == 2) panel(3, 1);
  if (w == 3) xblock(1, 0);
  __syncthreads();  // B3
  if (w == 0) {
    trail(2, 2, 1);
    pd_diag16(a, x, 32, row0, info);
  } else if (w == 1) {
    trail(3, 2, 1);
    pprod(2, 1);
  } else if (w == 2) {
    trail(3, 3, 1);
  } else {
    pprod(2, 0);
  }
  __syncthreads();  // B4
  // ---- stage 2 ----
  if (w == 1) panel(3, 2);
  if (w == 2) xblock(2, 0);
  if (w == 3) xblock(2, 1);
  __syncthreads();  // B5
  if (w == 0) {
    trail(3, 3, 2);
    pd_diag16(a, x, 48, row0, info);
  } else if (w == 1) {
    pprod(3, 0);
  } else if (w == 2) {
    pprod(3, 1);
  } else {
    pprod(3, 2);
  }
  if (w > 0) zero_upper(a);
  __syncthreads();  // B6
  // ---- stage 3: the last row of L^-1 ----
  if (w > 0) xblock(3, w - 1);
  __syncthreads();  // B7
}

// ---- Gram tiles produced inside the factorisation's workgroups (fit evaluations: no Gram launch, no HBM round trip) --------------
// bbh_kbase, inlined (a call from a loop with many live registers saves them to scratch around it)
__device__ __forceinline__ double pd_kbase(int kind, double r2, int jb, double alpha) {
  if (kind == BBH_KERNEL_LINEAR) return r2;
  if (BBH_KIND_IS_POLY(kind)) return bbh_powi(r2 + alpha, kind - BBH_KERNEL_POLY1 + 1);
  if (kind == BBH_KERNEL_RBF) return exp(-0.5 * r2);
  if (kind == BBH_KERNEL_RQ) return exp(-alpha * log1p(r2 / (2.0 * alpha)));
  if (kind >= BBH_KERNEL_PIECEWISE0 && kind <= BBH_KERNEL_PIECEWISE3) return bbh_piecewise(kind - BBH_KERNEL_PIECEWISE0, jb, r2, false);
  const double r = sqrt(r2);
  if (kind == BBH_KERNEL_MATERN52) return (1.0 + BBH_SQRT5 * r + (5.0 / 3.0) * r2) * exp(-BBH_SQRT5 * r);
  if (kind == BBH_KERNEL_MATERN32) return (1.0 + BBH_SQRT3 * r) * exp(-BBH_SQRT3 * r);
  return exp(-r);
}
#define PD_GRAM_MAXD 32
#define PD_GRAM_MAXTH 64
#define PD_GRAM_MAXTHV 52  // theta entries that travel as kernel arguments (bbh_fit_flow_eligible: <= 49)
struct pd_gram_src {     // by value; the single-kernel models bbh_fit_flow_eligible admits (one factor, shared noise / mean, <= 4 tasks)
  const double* xnT;     // [dn][np] normalised training inputs, transposed
  const int* task;       // [np]
  const double* nmask;   // [np]
  const double* theta;   // [tl] (device or host-mapped); null: thv below
  double thv[PD_GRAM_MAXTHV];  // theta by value (kernel argument: no copy, no load from host memory)
  int n, np, dn, T, tl, kind, use_os, jb, alpha_off;
  int d_sc1;             // row heads read D_{I-1} with sc1 loads and skip the acquire fence (write-through launches; set outside the Gram form too)
  long long* dbg;        // BBH_TILE_STAMPS=1: [tiles][8] wall_clock64 stamps (bbh_tiles_trace_read); null otherwise
};
struct pd_gram_lds {
  double th[PD_GRAM_MAXTH];
  double inv[PD_GRAM_MAXD];
  double nm[64];
  int tr[64], tc[64];
};
// theta -> LDS, once per workgroup
__device__ __forceinline__ void pd_gram_init(pd_gram_lds& g, const pd_gram_src& gs) {
  const int t = threadIdx.x;
  if (t < gs.tl) g.th[t] = gs.theta ? gs.theta[t] : gs.thv[t < PD_GRAM_MAXTHV ? t : 0];
  __syncthreads();
  if (t < gs.dn) g.inv[t] = 1.0 / g.th[3 + t];
  __syncthreads();
}
// rows of block I of the transposed inputs -> flat [dn][64]
__device__ __forceinline__ void pd_gram_stage(double* dst, const pd_gram_src& gs, int I) {
  for (int e = threadIdx.x; e < gs.dn * 64; e += 256) dst[e] = gs.xnT[(int64_t)(e >> 6) * gs.np + I * 64 + (e & 63)];
}
__device__ __forceinline__ void pd_gram_meta(pd_gram_lds& g, const pd_gram_src& gs, int I, int K) {
  const int t = threadIdx.x;
  if (t < 64) {
    g.tr[t] = gs.task[I * 64 + t];
    g.nm[t] = gs.nmask[I * 64 + t];
  } else if (t < 128) {
    g.tc[t - 64] = gs.task[K * 64 + (t - 64)];
  }
}
// Tile (I, K) of K + s2 M: out[a][b] = os B[ta][tb] k(r_ab) + noise nmask_a [a == b]; identity on the padding (the arithmetic of
// bbh_gram_kernel).  xr / xc: the staged inputs of block I / K.  Thread t owns row t >> 2 and the columns (t & 3) + 4 m; four columns
// are in flight together (four independent distance sums and kernel-function chains: the loop is latency-bound otherwise).
// KIND: the kernel kind as a compile-time constant for the common ones (-1: any kind, through pd_kbase).  With the kind a run-time
// value inside the entry loop the compiler evaluates pd_kbase's whole chain of alternatives - exp, log1p, the piecewise polynomials'
// power loops - around every entry: 22 us per tile (profiles/r05_tile_gram.log), 1.4 us per entry and wave.
template <int KIND>
__device__ __forceinline__ void pd_gram_tile_k(double (*out)[PD_LD], const double* xr, const double* xc, int I, int K, const pd_gram_src& gs,
                                               const pd_gram_lds& g) {
  const int kind = KIND >= 0 ? KIND : gs.kind, dn = gs.dn;
  const double os = gs.use_os ? g.th[2] : 1.0;
  const double kalpha = gs.alpha_off >= 0 ? g.th[gs.alpha_off] : 1.0;
  const int i = threadIdx.x >> 2, part = threadIdx.x & 3, ga = I * 64 + i;
  const bool dot = KIND >= 0 ? false : BBH_KIND_IS_DOT(kind);
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    double r2[4] = {0.0, 0.0, 0.0, 0.0};
    const double* xcq = xc + part + 16 * q;
    for (int c = 0; c < dn; c++) {
      const double xa = xr[c * 64 + i], iv = g.inv[c];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const double xb = xcq[c * 64 + 4 * u];
        if (dot) {
          r2[u] += (xa * iv) * (xb * iv);
        } else {
          const double df = (xa - xb) * iv;
          r2[u] += df * df;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int j = part + 16 * q + 4 * u, gb = K * 64 + j;
      double v;
      if (ga >= gs.n || gb >= gs.n) {
        v = (ga == gb) ? 1.0 : 0.0;
      } else {
        if (KIND == BBH_KERNEL_MATERN52) {
          const double r = sqrt(r2[u]);
          v = (1.0 + BBH_SQRT5 * r + (5.0 / 3.0) * r2[u]) * exp(-BBH_SQRT5 * r);
        } else if (KIND == BBH_KERNEL_MATERN32) {
          const double r = sqrt(r2[u]);
          v = (1.0 + BBH_SQRT3 * r) * exp(-BBH_SQRT3 * r);
        } else if (KIND == BBH_KERNEL_MATERN12) {
          v = exp(-sqrt(r2[u]));
        } else if (KIND == BBH_KERNEL_RBF) {
          v = exp(-0.5 * r2[u]);
        } else {
          v = pd_kbase(kind, r2[u], gs.jb, kalpha);
        }
        if (gs.use_os) v *= os;
        if (gs.T > 1) v *= g.th[3 + dn + g.tr[i] * gs.T + g.tc[j]];
        if (ga == gb) v += g.th[0] * g.nm[i];
      }
      out[i][j] = v;
    }
  }
}
__device__ __forceinline__ void pd_gram_tile(double (*out)[PD_LD], const double* xr, const double* xc, int I, int K, const pd_gram_src& gs,
                                             const pd_gram_lds& g) {
  switch (gs.kind) {
    case BBH_KERNEL_MATERN52: pd_gram_tile_k<BBH_KERNEL_MATERN52>(out, xr, xc, I, K, gs, g); break;
    case BBH_KERNEL_MATERN32: pd_gram_tile_k<BBH_KERNEL_MATERN32>(out, xr, xc, I, K, gs, g); break;
    case BBH_KERNEL_MATERN12: pd_gram_tile_k<BBH_KERNEL_MATERN12>(out, xr, xc, I, K, gs, g); break;
    case BBH_KERNEL_RBF: pd_gram_tile_k<BBH_KERNEL_RBF>(out, xr, xc, I, K, gs, g); break;
    default: pd_gram_tile_k<-1>(out, xr, xc, I, K, gs, g); break;
  }
}

// =====================================================================================================================
// The whole factorisation L = chol(A), X = L^-1 of an np x np matrix (np <= 1024) in ONE launch: tile dataflow.
// One workgroup per 64 x 64 tile, all resident at once (<= 256 workgroups, one per CU: 135 KB of LDS each):
//   L-tile (I, K), K <= I-2: a = A_IK;  for J < K: a -= L_IJ L_KJ^T as soon as both are published;  then wait for D_K,
//                           L_IK = a D_K^T, publish.
//   row head I:             the tiles (I, I-1) and (I, I) together: the same updates for both, then L_{I,I-1} = a_left
//                           D_{I-1}^T, a_diag -= L_{I,I-1} L_{I,I-1}^T, factor + invert (pd_factor_block), publish D_I.
//   X-tile (I, J), I > J:   acc = sum_{K = J}^{I-1} L_IK X_KJ as the operands appear (X_JJ = D_J), then X_IJ = -D_I acc.
// "Published" = tile written, __threadfence(), flag[tile] = epoch (release); consumers poll the flag (acquire) with a
// bounded number of polls - if the workgroups are ever not co-resident (another kernel holding CUs) the launch gives up
// (*info = -7) instead of hanging and the caller falls back to the launch-per-step path.  The dependency chain is
// nbk x (factor + one 64^3 panel product + one 64^3 update) inside one kernel instead of 8 x 3 dependent launches.
// =====================================================================================================================
// (urgent: unused - backing off the pollers whose flag is not about to flip (0.4 us between polls) was measured: the delays add up along
// the off-critical chains until they are critical, 484 -> 712 us at n = 1024; profiles/r05_tile_gram.log)
// lazy: a waiter that starts polling long before its flag can flip (the roles behind the factorisation in the single-launch fit
// evaluation: ~190 workgroups from the first microsecond on) sleeps ~0.85 us between polls - at full rate they take memory bandwidth
// from the critical chain; the price is up to one sleep at the moment the flag flips.
__device__ __forceinline__ bool pd_wait_n(const int* flag, int epoch, int* info, int spin_limit, bool urgent = false, bool acquire = true,
                                          long long* polls_out = nullptr, bool lazy = false) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1, it = 0;
    // Polls are RELAXED loads at device scope: an acquire load per poll invalidates the XCD's L2 every time, and with every
    // waiting workgroup of a launch polling (130 of them in the one-launch fit evaluation) those invalidations slowed the
    // critical path's own tile loads and write-backs - hand-offs took up to 19 us instead of 2 (profiles/r05_flow_trace_*).
    // The acquire side is the fence every thread executes after the flag has flipped.
    // (polls as device-scope atomic read-modify-writes instead of loads were measured: no difference)
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      if (++it > spin_limit || ((it & 15) == 0 && __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == -7)) {
        __hip_atomic_store(info, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      if (urgent) {
        // (the one waiter on the critical path: no sleep between polls)
      } else if (lazy) {
        __builtin_amdgcn_s_sleep(32);
      } else if (it < 64) {
        __builtin_amdgcn_s_sleep(1);
      } else {
        __builtin_amdgcn_s_sleep(4);
      }
    }
    // acquire: ONE agent-scope fence per workgroup (buffer_inv of the CU's L1 and the XCD's L2 - caches the four waves share); every
    // thread executing it meant four invalidations per workgroup and wait, and with 144 workgroups released by one flag the loads
    // behind them took 13 us (profiles/r05_flow_tail_trace.log)
    // (acquire = false: the caller reads the payload with sc1 loads, which do not look at this CU's L1 - valid when the producer stored it
    // write-through, cdna guide Guideline 16; saves the 1.7 us of buffer_inv on a critical hand-off)
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (polls_out) *polls_out = it;
    s_ok = ok;
  }
  __syncthreads();  // (workgroup-scope ordering: the other waves' loads follow the fence)
  const bool ok = s_ok != 0;
  __syncthreads();  // (s_ok is reused by the next wait)
  return ok;
}
// release: every wave waits for its own stores (workgroup scope: they are in L2 then), ONE agent-scope fence writes the L2 back
__device__ __forceinline__ void pd_publish(int* flag, int epoch) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (restated where the compiler cannot drop it: cdna guide, Guideline 16 pitfall 12)
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// The write-through form (cdna guide, Guideline 16 R1): a payload stored with sc1 stores (pd_store_tile_wt) leaves the XCD's L2 as it is
// written, so publishing it needs no release fence (buffer_wbl2 writes back every dirty line of the L2: 1.7 us clean, 6.5 us with a
// fresh 32 KB tile) - every storing wave drains its stores, one lane stores the flag.  Consumers are unchanged (pd_wait_n: relaxed
// poll, one agent acquire, plain loads).
__device__ __forceinline__ void pd_publish_wt(int* flag, int epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (16-byte accesses: tile rows start 16-byte aligned in global memory - ld is a multiple of 64 - and in LDS, pitch 528 B)
typedef double pd_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pd_load_tile(double (*dst)[PD_LD], const double* src, int64_t ld) {
#pragma unroll
  for (int e = threadIdx.x; e < 2048; e += 256)
    *(pd_d2*)&dst[e >> 5][2 * (e & 31)] = *(const pd_d2*)(src + (int64_t)(e >> 5) * ld + 2 * (e & 31));
}
// a contiguous 64 x 64 tile (ld = 64) through sc1 loads: past this CU's L1, so no acquire fence is needed for a tile its producer stored
// write-through.  src must be wave-uniform (buffer descriptor).
__device__ __forceinline__ void pd_load_tile_sc1(double (*dst)[PD_LD], const double* src) {
  typedef unsigned int pd_u4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, 64 * 64 * 8, 0x00020000);
#pragma unroll
  for (int e = threadIdx.x; e < 2048; e += 256) {
    const pd_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, e * 16, 0, 16);
    *(pd_d2*)&dst[e >> 5][2 * (e & 31)] = __builtin_bit_cast(pd_d2, w);
  }
}
__device__ __forceinline__ void pd_store_tile(double* dst, int64_t ld, const double (*src)[PD_LD], double scale) {
#pragma unroll
  for (int e = threadIdx.x; e < 2048; e += 256) {
    pd_d2 v = *(const pd_d2*)&src[e >> 5][2 * (e & 31)];
    v *= scale;
    *(pd_d2*)(dst + (int64_t)(e >> 5) * ld + 2 * (e & 31)) = v;
  }
}
__device__ __forceinline__ void pd_store_tile_wt(double* dst, int64_t ld, const double (*src)[PD_LD], double scale) {
#pragma unroll
  for (int e = threadIdx.x; e < 2048; e += 256) {
    pd_d2 v = *(const pd_d2*)&src[e >> 5][2 * (e & 31)];
    v *= scale;
    double* p = dst + (int64_t)(e >> 5) * ld + 2 * (e & 31);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  }
}
// c (+)= sign * a b^T (NT) or a b (NN), 64 x 64 x 64, the 16 output sub-blocks dealt to the four waves.  SKIP names
// structural zeros / don't-cares at 16 x 16 sub-block granularity:
//   PD_B_LOWER (NT): b is lower triangular, b[n][k] = 0 for k > n      -> k-blocks kb <= nb only
//   PD_A_LOWER (NN): a is lower triangular, a[m][k] = 0 for k > m      -> k-blocks kb <= mb only
//   PD_OUT_LOWER:    only the lower sub-blocks of c (mb >= nb) are read afterwards (symmetric update of a diagonal tile)
enum { PD_FULL = 0, PD_B_LOWER = 1, PD_A_LOWER = 2, PD_OUT_LOWER = 3 };
template <bool NT, bool ACCUM, int SKIP>
__device__ __forceinline__ void pd_gemm64(double (*c)[PD_LD], const double (*a)[PD_LD], const double (*b)[PD_LD], double sign) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  // sub-blocks are dealt so that every wave gets the same number of k-blocks: lower-only output - the ten lower sub-blocks
  // in turn; triangular b - one sub-block of every block column; triangular a - one of every block row
  constexpr int NSB = SKIP == PD_OUT_LOWER ? 10 : 16;
  for (int sb = w; sb < NSB; sb += 4) {
    int mb, nb;
    if (SKIP == PD_OUT_LOWER) {
      nb = sb < 4 ? 0 : sb < 7 ? 1 : sb < 9 ? 2 : 3;
      mb = sb - (nb == 0 ? 0 : nb == 1 ? 3 : nb == 2 ? 5 : 6);
    } else if (SKIP == PD_A_LOWER) {
      mb = sb >> 2, nb = sb & 3;
    } else {
      mb = sb & 3, nb = sb >> 2;
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
      if (SKIP == PD_B_LOWER && kb > nb) continue;
      if (SKIP == PD_A_LOWER && kb > mb) continue;
      if (NT) {
        const d4 p = pd_mul_nt(a, 16 * mb, 16 * kb, b, 16 * nb, 16 * kb, l);
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] += p[r];
      } else {
        acc = pd_mul_nn(a, 16 * mb, 16 * kb, b, 16 * kb, 16 * nb, l, acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double* out = &c[16 * mb + (l >> 4) + 4 * r][16 * nb + (l & 15)];
      *out = ACCUM ? *out + sign * acc[r] : sign * acc[r];
    }
  }
}


// ---- tiles of M = K^-1 = X^T X (X = L^-1, lower block triangular): M_IJ = sum_{K >= I} X_KI^T X_KJ, X_KK = D_K, I >= J ---------------
// The role of the fit's dataflow tail (bbh_fitflow.hip) and - for matrices small enough that these workgroups are co-resident with the
// factorisation's - extra workgroups of bbh_potrf_tiles_kernel: K^-1's tiles are then built while the factorisation runs, as the rows
// of X appear.  Two LDS tiles for the operands, the sum in MFMA accumulators, the next k-step's operands on their way into registers
// while this one's MFMAs run.
#define PD_MT_STRIDE 32  // row pitch of the per-tile tables shared with the dataflow fit evaluation (flagsM, apart): block rows I < 32
struct pd_mt_args {
  int nM;                // M-tile workgroups appended to the factorisation's grid (0: none)
  double* M;             // [np][ld] lower tiles (LOO: both triangles)
  int64_t ld;
  double* apart;         // [(I * PD_MT_STRIDE + J) * 128] alpha partials: row part M_IJ r_J, column part M_IJ^T r_I
  const double* ystd;    // [np]
  const double* theta;   // null: cmean
  double cmean;          // constant mean (theta[1])
  int n, loo;
  int* flagsM;           // [I * PD_MT_STRIDE + J], stamped with flow_epoch
  int* doneM;            // cumulative counter of finished M-tiles
  int flow_epoch;
};
__device__ __forceinline__ void pd_mma_tn(d4 (&acc)[4], const double (*a)[PD_LD], const double (*b)[PD_LD]) {  // acc += a^T b
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int sb = w + 4 * q, mb = sb & 3, nb = sb >> 2;
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
        acc[q] = mfma_f64(a[16 * kb + 4 * ks + (l >> 4)][16 * mb + (l & 15)], b[16 * kb + 4 * ks + (l >> 4)][16 * nb + (l & 15)], acc[q]);
  }
}
__device__ __forceinline__ void pd_ld_regs(pd_d2 (&r)[8], const double* src, int64_t ld) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int e = threadIdx.x + 256 * k;
    r[k] = *(const pd_d2*)(src + (int64_t)(e >> 5) * ld + 2 * (e & 31));
  }
}
__device__ __forceinline__ void pd_st_regs(double (*dst)[PD_LD], const pd_d2 (&r)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int e = threadIdx.x + 256 * k;
    *(pd_d2*)&dst[e >> 5][2 * (e & 31)] = r[k];
  }
}
// WAIT(K) -> bool: the operands of k-step K (D_I for K == I, X_KI and - off the diagonal - X_KJ) are visible.  vr / vc: 64 doubles of LDS
// each (residuals of block I / J).  On return the tile and its alpha partials are stored (plain stores); the caller publishes.
template <class WAIT>
__device__ __forceinline__ bool pd_mtile_core(double (*t0)[PD_LD], double (*t1)[PD_LD], double* vr, double* vc, int I, int J, int nbk, const double* D,
                                              const double* X, int64_t ldx, const pd_mt_args& ma, double cmean, WAIT&& wait) {
  const int t = threadIdx.x;
  if (t < 64) {
    const int g = I * 64 + t;
    vr[t] = g < ma.n ? ma.ystd[g] - cmean : 0.0;
  } else if (t < 128) {
    const int g = J * 64 + (t - 64);
    vc[t - 64] = g < ma.n ? ma.ystd[g] - cmean : 0.0;
  }
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  pd_d2 rb[8], rc[8];
  auto fetch = [&](int K) -> bool {
    if (!wait(K)) return false;
    if (K == I)
      pd_ld_regs(rb, D + (int64_t)I * 4096, 64);
    else
      pd_ld_regs(rb, X + (int64_t)(K * 64) * ldx + I * 64, ldx);
    if (I != J) pd_ld_regs(rc, X + (int64_t)(K * 64) * ldx + J * 64, ldx);
    return true;
  };
  if (!fetch(I)) return false;
  for (int K = I; K < nbk; K++) {
    pd_st_regs(t0, rb);
    if (I != J) pd_st_regs(t1, rc);
    __syncthreads();
    if (K + 1 < nbk && !fetch(K + 1)) return false;
    pd_mma_tn(acc, t0, I != J ? t1 : t0);
    __syncthreads();
  }
  double(*a)[PD_LD] = t0;
  {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int sb = w + 4 * q, mb = sb & 3, nb = sb >> 2;
#pragma unroll
      for (int r = 0; r < 4; r++) a[16 * mb + (l >> 4) + 4 * r][16 * nb + (l & 15)] = acc[q][r];
    }
  }
  __syncthreads();
  pd_store_tile(ma.M + (int64_t)(I * 64) * ma.ld + J * 64, ma.ld, a, 1.0);
  if (ma.loo && I != J) {  // the leave-one-out terms contract whole rows of M: keep both triangles
    for (int e = t; e < 4096; e += 256) {
      const int i = e >> 6, j = e & 63;
      ma.M[(int64_t)(J * 64 + i) * ma.ld + I * 64 + j] = a[j][i];
    }
  }
  // alpha partials: row part (M_IJ r_J) and, off the diagonal, column part (M_IJ^T r_I); four threads per entry
  {
    const int row = t >> 2, part = t & 3;
    double s1 = 0.0, s2 = 0.0;
    for (int j = part; j < 64; j += 4) {
      s1 = fma(a[row][j], vc[j], s1);
      s2 = fma(a[j][row], vr[j], s2);
    }
    s1 += __shfl_xor(s1, 1, 64);
    s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64);
    s2 += __shfl_xor(s2, 2, 64);
    if (part == 0) {
      double* ap = ma.apart + (int64_t)(I * PD_MT_STRIDE + J) * 128;
      ap[row] = s1;
      ap[64 + row] = (I != J) ? s2 : 0.0;
    }
  }
  return true;
}
