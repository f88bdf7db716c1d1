// Register-resident form of the fused posterior kernel for small models: n <= 128 training points (operands in LDS beyond 32) (where BayBE campaigns start and
// where its backtesting loops live: simulation/core.py:144-201 calls recommend() thousands of times at n = 10 ... 100).
//
// Why another form.  The cooperative form (bbh_coop.h) deals 16-candidate tiles to workgroups of four waves and streams L^-T from
// memory; at n <= 64 a tile is ~40 variance MFMAs against a per-tile set-up of loads, two barriers and an LDS exchange - it measured
// 0.28 (n = 64) and 0.095 (n = 32) of the fp64 peak on its own flops (profiles/r03_ab_small_n.log).  Here the whole model lives in
// the wave:
//   L^-T        its (k-block, k-step, column-block) fragments of the lower triangle - NB (NB + 1) / 2 x 4 of them, 40 for n = 64 -
//               are loaded ONCE per wave into registers (80 VGPRs) and stay there;
//   X_train, alpha  the augmented training fragments of the distance GEMM and the mean weights sit in LDS (read-only, shared
//               by the workgroup's waves: 6.5 KB for n = 64, d <= 14);
//   candidates  every wave walks over 16-candidate tiles on its own (persistent launch, no barrier after the set-up): the next
//               tile's rows are requested before the current tile is computed;
//   per tile    for each of the NB training blocks: KD distance MFMAs (train x candidates, whose accumulator layout IS the
//               A-operand layout of the next GEMM), four kernel values per lane on the VALU (the micro-step chain of bbh_fused.h),
//               the mean as a 4-term dot product, 4 (NB - tb) variance MFMAs against the register-resident fragments.
// Work per 16 candidates at n = 64, d = 10: 12 + 40 MFMAs (3 328 cycles of the fp64 pipe) + 16 kernel values per lane (~2 200
// cycles on the same pipe): the kernel values are as much of the pipe as the matrix products - the bound of this shape.
// Variance passes without pending columns only (the first pass of every selection step; mean-only / cross-covariance passes keep the
// cooperative form).
#pragma once
#define BBH_CANDREG 1
#include "bbh_fused.h"

struct SmallArgs {
  FusedArgs f;
  const double* rsmall;  // [pairs (tb, j >= tb) in tb-major order][4 k-steps][64 lanes]
};

__host__ __device__ constexpr int small_pair_index(int NB, int tb, int j) { return tb * NB - (tb * (tb - 1)) / 2 + (j - tb); }

// lane l <- R[k = 16 tb + 4 r + (l >> 4)][col = 16 j + (l & 15)] = X[col][k]  (X = L^-1, lower triangular)
static __global__ void bbh_pack_small_kernel(const double* __restrict__ X, int64_t np, int NB, double* __restrict__ out) {
  const int tb = blockIdx.x, j = blockIdx.y, r = blockIdx.z, l = threadIdx.x;
  if (j < tb) return;
  const int64_t k = 16 * (int64_t)tb + 4 * r + (l >> 4), col = 16 * (int64_t)j + (l & 15);
  out[((int64_t)small_pair_index(NB, tb, j) * 4 + r) * 64 + l] = (k <= col) ? X[col * np + k] : 0.0;
}

// RLDS: the operand fragments live in LDS instead of registers (NB >= 3: 80 VGPRs less, three waves per SIMD instead of two - the
// kernel-value chains and the MFMA -> VALU hand-overs of one wave are bubbles that only other waves fill)
#ifndef BBH_SMALL_RLDS
#define BBH_SMALL_RLDS 1
#endif
// register budget as waves per SIMD: the table variants carry the per-lane task lookups of four kernel values on top.
// 64 < n <= 128 (NB = 5 ... 8): the operand fragments are 45 - 74 KB of LDS, one workgroup per CU - of eight waves, so that every
// SIMD still holds two.
__host__ __device__ constexpr int small_waves(int KD, int KVF, int NB) {
  return NB >= 5 ? 2 : (KVF & 1) ? (KD >= 6 ? 2 : 3) : (NB <= 2 ? (KD >= 6 ? 3 : 4) : (BBH_SMALL_RLDS ? 3 : 2));
}
__host__ __device__ constexpr int small_threads(int NB) { return NB >= 5 ? 512 : 256; }
template <int KD, int KVF, int NB>
__global__ __launch_bounds__(small_threads(NB), small_waves(KD, KVF, NB)) void bbh_small_posterior_kernel(const SmallArgs sa) {
  const FusedArgs& a = sa.f;
  constexpr int NP = NB * (NB + 1) / 2;
  constexpr bool RLDS = (BBH_SMALL_RLDS && NB >= 3) || NB >= 5;
  constexpr int NT = small_threads(NB), NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) double s_mem[];  // training fragments [NB][KD][64] | alpha [16 NB] | (RLDS) operand fragments [NP][4][64]
  double* s_tf = s_mem;
  double* s_alpha = s_mem + NB * KD * 64;
  double* s_r = s_alpha + 16 * NB;
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnd = l & 15, q = l >> 4;
  for (int e = threadIdx.x; e < NB * KD * 64; e += NT) s_tf[e] = a.trainfrag[e];
  for (int e = threadIdx.x; e < 16 * NB; e += NT) s_alpha[e] = a.meanB[(int64_t)e * 16];
  double rfr[RLDS ? 1 : NP * 4];
  if constexpr (RLDS) {
    for (int e = threadIdx.x; e < NP * 4 * 64; e += NT) s_r[e] = sa.rsmall[e];
  } else {
#pragma unroll
    for (int i = 0; i < NP * 4; i++) rfr[i] = sa.rsmall[(int64_t)i * 64 + l];
  }
  // per-lane column map and scaling of the candidate fragments: feature 4 k + q of candidate cnd
  int xcol[KD];
  double xscl[KD], xofs[KD];
#pragma unroll
  for (int k = 0; k < KD; k++) {
    const int dimc = (4 * k + q < a.dn) ? 4 * k + q : a.dn - 1;
    xcol[k] = a.numcol_identity ? dimc : a.numcol[dimc];
    xscl[k] = a.scl[dimc];
    xofs[k] = a.ofs[dimc];
  }
  const int64_t ntiles = (a.N + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * NW;
  int64_t tile = (int64_t)blockIdx.x * NW + w;
  double xv[KD], xtask = 0.0;
  {
    const int64_t row = (tile * 16 + cnd < a.N) ? tile * 16 + cnd : a.N - 1;
    const double* xr = a.X + row * a.ldx;
#pragma unroll
    for (int k = 0; k < KD; k++) xv[k] = xr[xcol[k]];
    if constexpr ((KVF & 1) != 0)
      if (a.task_col >= 0) xtask = xr[a.task_col];
  }
  __syncthreads();  // LDS tables are complete; no barrier from here on
  WaveCtx c;
  c.tf = nullptr, c.candl = nullptr, c.mb = nullptr, c.tbl = a.tasktbl, c.taskext = a.taskext, c.kvc = nullptr;
  c.kvl = (bbh_lds_double*)nullptr, c.nl = 0, c.ncache = 0, c.al = (const bbh_lds_double*)nullptr;
  c.kd = KD, c.kind = a.kind, c.T = a.T, c.q = q, c.l = l, c.dn = a.dn, c.tc = 0;
  for (; tile < ntiles; tile += stride) {
    // candidate fragments of this tile: b = x * scl + ofs, augmented with [1, |b|^2]
    double nbsum = 0.0;
#pragma unroll
    for (int k = 0; k < KD; k++) {
      double v = 0.0;
      if (4 * k + q < a.dn) {
        v = fma(xv[k], xscl[k], xofs[k]);
        nbsum = fma(v, v, nbsum);
      }
      c.cf[k] = v;
    }
    nbsum += __shfl_xor(nbsum, 16, 64);
    nbsum += __shfl_xor(nbsum, 32, 64);
#pragma unroll
    for (int k = 0; k < KD; k++) {
      if (4 * k + q == a.dn) c.cf[k] = 1.0;
      if (4 * k + q == a.dn + 1) c.cf[k] = nbsum;
    }
    if constexpr ((KVF & 1) != 0) {
      int tc = 0;
      if (a.task_col >= 0) {
        tc = (int)xtask;
        tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
      }
      c.tc = tc;
    }
    // the next tile's rows (clamped: the loads are unconditional)
    {
      const int64_t nt = tile + stride;
      const int64_t row = (nt * 16 + cnd < a.N) ? nt * 16 + cnd : a.N - 1;
      const double* xr = a.X + row * a.ldx;
#pragma unroll
      for (int k = 0; k < KD; k++) xv[k] = xr[xcol[k]];
      if constexpr ((KVF & 1) != 0)
        if (a.task_col >= 0) xtask = xr[a.task_col];
    }
    d4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
    double accm = 0.0;
    // The training fragments and alpha are re-read from LDS for every tile.  They are loop-invariant, so the compiler would hoist
    // all NB (KD + 4) of them into registers - up to 96 VGPRs next to the 80 of the operand fragments: spills.  The LDS addresses
    // pass through an empty asm per tile, which hides the invariance and costs nothing.
    const bbh_lds_double* tfp = (const bbh_lds_double*)(s_tf + l);
    const bbh_lds_double* alp = (const bbh_lds_double*)(s_alpha + q);
    const bbh_lds_double* rlp = (const bbh_lds_double*)(s_r + l);
    asm volatile("" : "+v"(tfp), "+v"(alp), "+v"(rlp));
    static_for<0, NB>([&](auto tbc) __attribute__((always_inline)) {
      constexpr int tb = decltype(tbc)::value;
      double tfv[KD], kv[4];
#pragma unroll
      for (int k = 0; k < KD; k++) tfv[k] = tfp[(tb * KD + k) * 64];
      d4 da, db;
      kvp_dist<KD>(c, tfv, da, db);
      kv_all<KVF>(c, tb, da, db, kv);
#pragma unroll
      for (int r = 0; r < 4; r++) accm = fma(kv[r], alp[16 * tb + 4 * r], accm);
      static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        static_for<tb, NB>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          if constexpr (RLDS)
            acc[j] = mfma_f64(kv[r], rlp[(small_pair_index(NB, tb, j) * 4 + r) * 64], acc[j]);
          else
            acc[j] = mfma_f64(kv[r], rfr[small_pair_index(NB, tb, j) * 4 + r], acc[j]);
        });
      });
    });
    // ||v||^2 over the column blocks (registers) and the 16 columns of a block (lanes): lanes with cnd == 0 hold candidate q + 4 r
    double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < NB; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) ss[r] = fma(acc[j][r], acc[j][r], ss[r]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double v = ss[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      ss[r] = v;
    }
    accm += __shfl_xor(accm, 16, 64);
    accm += __shfl_xor(accm, 32, 64);  // every lane: the mean sum of candidate cnd
    // lane (q = 0, cnd): candidate cnd's variance sum sits in lane 16 (cnd & 3), register cnd >> 2
    double sv = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const double g = __shfl(ss[r], 16 * (cnd & 3), 64);
      if ((cnd >> 2) == r) sv = g;
    }
    const int64_t gi = tile * 16 + cnd;
    if (q == 0 && gi < a.N) {
      double pv = a.prior_scale, mc = a.mean_const;
      if constexpr ((KVF & 1) != 0) {
        pv = a.tasktbl[c.tc * a.T + c.tc];
        if (a.taskmean) mc = a.taskmean[c.tc];
      }
      if (a.mean) a.mean[gi] = a.ybar + a.ysd * (mc + accm);
      if (a.var) a.var[gi] = a.ysd * a.ysd * (pv - sv);
    }
  }
}

__host__ inline size_t small_lds_bytes(int kd, int NB) {
  return sizeof(double) * ((size_t)NB * kd * 64 + 16 * (size_t)NB + (((BBH_SMALL_RLDS && NB >= 3) || NB >= 5) ? (size_t)NB * (NB + 1) / 2 * 4 * 64 : 0));
}

// Persistent launch: exactly as many workgroups as the device holds at once (a queued workgroup would start when the others are
// done and double the launch time).  The occupancy query's answer can be one workgroup per CU too high when a kernel uses more
// than 80 SGPRs: 256-thread workgroups are admitted up to min(API, 8, floor(800 / (ceil(sgpr / 16) 16 + 16))) per CU
// (MI355X_MICROARCH.md, "Residency").  Every instantiation here reports 104 - 106 SGPRs (hipcc -S, .sgpr_count), i.e. an SGPR bound
// of 6 workgroups per CU, and the register budgets (small_waves) ask for at most 4: the API's answer is the residency.  Round 4
// first took one off above 2 per CU "to be safe" - which ran the n <= 64 variants at two waves per SIMD instead of three.  (Dynamic
// tile tickets from a device-wide atomic counter, which would make the question moot, cost 13 ns per draw - device-scope atomics are
// executed at the memory side across the 8 XCDs: 0.8 ms per 1e6 candidates whatever n is; dropped.)
template <int KD, int KVF, int NB>
static void small_go(int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a) {
  static int per_cu = 0;
  const size_t lds = small_lds_bytes(KD, NB);
  if (!per_cu) {
    int n = 0;
    constexpr int NT = small_threads(NB);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)bbh_small_posterior_kernel<KD, KVF, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bbh_small_posterior_kernel<KD, KVF, NB>, NT, lds) != hipSuccess || n < 1) n = 1;
    per_cu = n > 5 ? n - 1 : n;  // (beyond the SGPR bound's reach the API may over-report by one)
  }
  constexpr int NW = small_threads(NB) / 64;
  int64_t blocks = (tiles + NW - 1) / NW;
  if (blocks > (int64_t)per_cu * num_cu) blocks = (int64_t)per_cu * num_cu;
  hipLaunchKernelGGL((bbh_small_posterior_kernel<KD, KVF, NB>), dim3((unsigned)blocks), dim3(small_threads(NB)), lds, s, a);
}

// false: no instantiation for this model; tiles == 0 only asks
bool bbh_small_launch(int kd, int kind, bool has_tbl, int NB, int64_t tiles, int num_cu, hipStream_t s, const SmallArgs& a);
