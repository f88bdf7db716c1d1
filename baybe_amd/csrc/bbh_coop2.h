// Two-sweep cooperative form of the fused posterior kernel: one workgroup (4 waves) per tile of 16 candidates,
// 512 < n <= 1024 (the transfer-learning configuration of BASELINE.json: ICM over 4 tasks, n = 1024).
//
// The one-sweep form (bbh_coop.h) deals the <= 32 column blocks of L^-T to the four waves, 8 accumulator blocks each.
// Here the <= 64 column blocks are covered in two sweeps over the same 8 accumulator blocks per wave:
//
//   sweep A   column blocks [0, 4 RA), RA = nb / 4 - 8 rounds, against the k-blocks [0, 4 RA): the triangular schedule of
//             the one-sweep form (rounds g0 .. 7 of its numbering, g0 = 16 - nb / 4).  Every kernel value is computed once per
//             tile, as there - the waves take turns, one k-block in four, in micro-steps between their MFMAs - but is
//             stored into an ARCHIVE in LDS (2 KB per k-block, <= 64 KB) instead of a double-buffered slot.
//   rect      column blocks [4 RA, nb) (all 8 slots) against the k-blocks [0, 4 RA) again: their kernel values are read
//             back from the archive - no distance GEMM, no kernel function, a pure MFMA + operand stream.
//   sweep B   column blocks [4 RA, nb) against the k-blocks [4 RA, nb): the one-sweep form once more (8 rounds).
//
// The windowed form (bbh_fused.h) needs nb / 16 passes per wave with a per-wave kernel-value cache (LDS + global slabs)
// and has 64-candidate workgroups: at N = 1e5 that is 1563 workgroups for 512 resident slots - a quarter of the launch is
// tail.  Here a workgroup is 16 candidates (6250 of them), every kernel value of a tile is computed once, nothing is
// cached outside LDS, and the operand slices of the three phases form ONE stream per wave in consumption order, so the
// register ring of the one-sweep form flows through the phase boundaries without draining.
#pragma once
#include "bbh_coop.h"

// fragments of one wave's slice: sweep A + rect + sweep B + the tail the open-ended ring requests but never uses
__host__ __device__ constexpr int64_t coop2_frags(int g0) {
  return (int64_t)(coop_frags_before(BBH_COOP_ROUNDS) - coop_frags_before(g0)) + (int64_t)(BBH_COOP_ROUNDS - g0) * 128 +
         coop_frags_before(BBH_COOP_ROUNDS) + 2 * BBH_COOP_PAIRS;
}

// 4 archived k-blocks against the 8 column-block slots of the second sweep
__device__ __forceinline__ void coop2_rect_group(const double* rs, const bbh_lds_double* kva, unsigned lane16,
                                                 d4 (&acc)[BBH_COOP_ROUNDS], d2 (&ring)[BBH_COOP_PAIRS]) {
  constexpr int NP = BBH_COOP_PAIRS;
  double kv[4], kvx[4];
#pragma unroll
  for (int r = 0; r < 4; r++) kv[r] = kva[r * 64];
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < 3) {
#pragma unroll
      for (int r = 0; r < 4; r++) kvx[r] = kva[(i + 1) * 256 + r * 64];
    }
    static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      static_for<0, BBH_COOP_ROUNDS>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr int f = (i * 4 + r) * BBH_COOP_ROUNDS + s;
        constexpr int pr = f / 2;
        if constexpr (f % 2 == 0) coop_vmwait2<NP - 1>(ring[pr % NP]);
        acc[s] = mfma_f64(kv[r], ring[pr % NP][f % 2], acc[s]);
        if constexpr (f % 2 == 1) {
          constexpr int np = pr + NP;
          coop_gload2<(np % 4) * 1024>(ring[pr % NP], rs + (np / 4) * 512, lane16);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if constexpr (i < 3) {
#pragma unroll
      for (int r = 0; r < 4; r++) kv[r] = kvx[r];
    }
  });
}

template <int KD, int KVF>
__global__ __launch_bounds__(256, 2) void bbh_coop2_posterior_kernel(const CoopArgs ca) {
  const FusedArgs& a = ca.f;
  // alpha [16 nb] | archive [max(4 (8 - g0), 8) k-blocks][4 x 64] | red [2][4][16]
  // (sweep B's double-buffered exchange slots [2][4 k-blocks][4 x 64] re-use the first 16 KB of the archive, which is dead by
  // then: with 64 KB of archive at n = 1024 a separate 16 KB would push the workgroup past half a CU's LDS)
  extern __shared__ __attribute__((aligned(16))) double s_mem[];
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnd = l & 15, q = l >> 4;
  const int g0 = ca.g0;  // 16 - nb / 4 in [0, 7]
  const int RA = BBH_COOP_ROUNDS - g0;  // rounds (= k-block groups) of sweep A
  double* s_alpha = s_mem;
  double* s_arch = s_alpha + 16 * a.nb;
  double* s_kv = s_arch;
  double* s_red = s_arch + (RA > 2 ? RA : 2) * 4 * 256;
  const int64_t tile0 = (int64_t)blockIdx.x * 16;

  WaveCtx c[1];
  int xcol[KD];
  double xval[KD], xscl[KD], xofs[KD];
  const int64_t row = (tile0 + cnd < a.N) ? tile0 + cnd : a.N - 1;
  const double* xr = a.X + row * a.ldx;
  if (a.numcol_identity) {
#pragma unroll
    for (int k = 0; k < KD; k++) xcol[k] = (4 * k + q < a.dn) ? 4 * k + q : a.dn - 1;
  } else {
#pragma unroll
    for (int k = 0; k < KD; k++) xcol[k] = a.numcol[(4 * k + q < a.dn) ? 4 * k + q : a.dn - 1];
  }
#pragma unroll
  for (int k = 0; k < KD; k++) {
    const int dimc = (4 * k + q < a.dn) ? 4 * k + q : a.dn - 1;
    xval[k] = xr[xcol[k]];
    xscl[k] = a.scl[dimc];
    xofs[k] = a.ofs[dimc];
  }
  for (int s = threadIdx.x; s < 16 * a.nb; s += 256) s_alpha[s] = a.meanB[(int64_t)s * 16];
  {
    double nbsum = 0.0;
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const int dim = 4 * k + q;
      double v = 0.0;
      if (dim < a.dn) {
        v = fma(xval[k], xscl[k], xofs[k]);
        nbsum = fma(v, v, nbsum);
      }
      c[0].cf[k] = v;
    }
    nbsum += __shfl_xor(nbsum, 16, 64);
    nbsum += __shfl_xor(nbsum, 32, 64);
#pragma unroll
    for (int k = 0; k < KD; k++) {
      if (4 * k + q == a.dn) c[0].cf[k] = 1.0;
      if (4 * k + q == a.dn + 1) c[0].cf[k] = nbsum;
    }
  }
  int tc = 0;
  if constexpr ((KVF & 1) != 0) {
    if (a.task_col >= 0) {
      tc = (int)xr[a.task_col];
      tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
    }
  }
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

  bbh_lds_double* arch = (bbh_lds_double*)(s_arch + l);  // [k-block][4 values x 64 lanes]
  bbh_lds_double* kvb = (bbh_lds_double*)(s_kv + l);     // [buffer][k-block of the group][4 values x 64 lanes]
  const bbh_lds_double* alq = (const bbh_lds_double*)(s_alpha + q);  // alpha[16 tb + 4 r + q]
  double accm[1] = {0.0};
  d4 acc[1][BBH_COOP_ROUNDS];
#pragma unroll
  for (int s = 0; s < BBH_COOP_ROUNDS; s++) acc[0][s] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* rs = ca.rstream + (int64_t)w * ca.frags * 64;
  const unsigned lane16 = (unsigned)l * 16u;
  d2 ring[BBH_COOP_PAIRS];
  static_for<0, BBH_COOP_PAIRS>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    coop_gload2<(i % 4) * 1024>(ring[i], rs + (i / 4) * 512, lane16);
  });
  {  // the first group's kernel values: wave w produces k-block w, not overlapped with anything
    double kv0[4], tfv0[KD];
    d4 dsa, dsb;
    kvp_load<KD>(c[0], w, tfv0);
    kvp_dist<KD>(c[0], tfv0, dsa, dsb);
    kv_all<KVF>(c[0], w, dsa, dsb, kv0);
    __syncthreads();  // alpha is in LDS
#pragma unroll
    for (int r = 0; r < 4; r++) {
      arch[w * 256 + r * 64] = kv0[r];
      accm[0] = fma(kv0[r], alq[16 * w + 4 * r], accm[0]);
    }
  }
  __syncthreads();  // k-blocks 0 .. 3 are complete in the archive

  int gi = 0;  // k-block group being consumed (k-blocks 4 gi .. 4 gi + 3)
  double ss[4] = {0.0, 0.0, 0.0, 0.0};
  // The two sweeps run through ONE copy of the group code (a two-trip loop that is not unrolled): instantiated twice, the
  // register allocator carried the second copy's constants and addresses across the first and spilled 50 - 67 VGPRs
  // (0.42 GB of scratch stores per 1e5-candidate launch); as one copy the kernel needs no scratch at all.
#pragma nounroll
  for (int sw = 0; sw < 2; sw++) {
    const int gs = sw ? 0 : g0;  // sweep A: rounds g0 .. 7 of the one-sweep numbering; sweep B: all eight
    static_for<0, BBH_COOP_ROUNDS>([&](auto gc) __attribute__((always_inline)) {
      constexpr int G = decltype(gc)::value;
      if (G >= gs) {  // wave-uniform
        const int cw = (G & 1) ? 3 - w : w;
        const int tbn = 4 * (gi + 1) + w;  // k-block this wave produces for the next group
        // (the last group has a single column-block slot - no MFMA slots to host a production; sweep B's first group is
        // produced after the rectangular part)
        constexpr bool PRODUCE = G + 1 < BBH_COOP_ROUNDS;
        // sweep A: kernel values are archived, k-block tb in slot tb; sweep B: double-buffered exchange slots (the archive's first 16 KB)
        const int cur = sw ? (G & 1) * 4 : 4 * gi;
        const int nxt = sw ? ((G + 1) & 1) * 4 + w : tbn;
        coop_group<G, KD, KVF, PRODUCE, 1, true>(c, rs, a.trainfrag + (int64_t)tbn * KD * 64, arch + cur * 256, arch + nxt * 256,
                                                 alq + 16 * tbn, tbn, cw, acc, ring, accm);
        rs += (int64_t)16 * (BBH_COOP_ROUNDS - G) * 64;
        gi++;
        if (PRODUCE || sw == 0) __syncthreads();
      }
    });
    if (sw == 0) {
#pragma unroll
      for (int s = 0; s < BBH_COOP_ROUNDS; s++) {
#pragma unroll
        for (int r = 0; r < 4; r++) ss[r] = fma(acc[0][s][r], acc[0][s][r], ss[r]);
        acc[0][s] = (d4){0.0, 0.0, 0.0, 0.0};
      }
      // ---- rectangular part: the archived k-blocks against the column blocks of sweep B ----
      for (int kg = 0; kg < RA; kg++) {
        coop2_rect_group(rs, arch + 4 * kg * 256, lane16, acc[0], ring);
        rs += (int64_t)16 * BBH_COOP_ROUNDS * 64;
      }
      __syncthreads();  // every wave is done with the archive: its first 16 KB become sweep B's exchange slots
      {  // sweep B's first group: wave w produces k-block 4 RA + w, not overlapped (as the very first group)
        const int tb0 = 4 * gi + w;
        double tfv[KD], kv0[4];
        d4 dsa, dsb;
        kvp_load<KD>(c[0], tb0, tfv);
        kvp_dist<KD>(c[0], tfv, dsa, dsb);
        kv_all<KVF>(c[0], tb0, dsa, dsb, kv0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          kvb[w * 256 + r * 64] = kv0[r];
          accm[0] = fma(kv0[r], alq[16 * tb0 + 4 * r], accm[0]);
        }
      }
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last requests (the unused tail of the slice)

  // ---- ||v||^2 over this wave's column blocks, then over the 16 columns of a block (lanes), then over the waves ----
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
    ss[r] = v;  // candidate q + 4 r
  }
  double mp = accm[0];
  mp += __shfl_xor(mp, 16, 64);
  mp += __shfl_xor(mp, 32, 64);
  double* red_v = s_red;  // [4 waves][16 candidates]
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
    double pv = a.prior_scale, mc = a.mean_const;
    if constexpr ((KVF & 1) != 0) {
      pv = a.tasktbl[tc * a.T + tc];  // lane m of wave 0 holds candidate m's task (cnd = m, q = 0)
      if (a.taskmean) mc = a.taskmean[tc];
    }
    if (gidx < a.N) {
      if (a.mean) a.mean[gidx] = a.ybar + a.ysd * (mc + sm);
      if (a.var) a.var[gidx] = a.ysd * a.ysd * (pv - sv);
    }
  }
}

// Instantiations (bbh_fused_coop2_a.hip: 2, 4, 6 k-steps of the distance GEMM; _b: 8, 12, 16): Matérn-5/2 with and without
// the task / outputscale table.  false: no instantiation for this model; grid.x == 0 only asks.
bool bbh_coop2_launch(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);
bool bbh_coop2_launch_a(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);
bool bbh_coop2_launch_b(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);

#define BBH_COOP2_DISPATCH_KD(KDV)                                                                             \
  if (kd == KDV) {                                                                                             \
    if (kind != BBH_KERNEL_MATERN52) return false;                                                             \
    if (grid.x == 0) return true;                                                                              \
    if (has_tbl) {                                                                                             \
      BBH_FUSED_ALLOW_LDS((bbh_coop2_posterior_kernel<KDV, 1>), lds);                                          \
      hipLaunchKernelGGL((bbh_coop2_posterior_kernel<KDV, 1>), grid, dim3(256), lds, s, a);                    \
    } else {                                                                                                   \
      BBH_FUSED_ALLOW_LDS((bbh_coop2_posterior_kernel<KDV, 0>), lds);                                          \
      hipLaunchKernelGGL((bbh_coop2_posterior_kernel<KDV, 0>), grid, dim3(256), lds, s, a);                    \
    }                                                                                                          \
    return true;                                                                                               \
  }
