// Cooperative form of the fused posterior kernel: one workgroup (4 waves) per tile of 16 candidates, n <= 512.
//
// The two-pass form (bbh_fused.h, 16-block windows, one wave per tile) needs every kernel value of the first 16
// k-blocks twice at n = 512; its wave-private LDS cache holds 8 of them, the other 8 are recomputed (48 MFMAs + 1088
// VALU instructions per tile, ~4 % of the fp64 pipe), and its 128 accumulator registers leave the pass bodies ~76 VGPRs
// short.  Here the 32 column blocks of L^-T are dealt to the four waves of a workgroup (8 each: 64 accumulator
// registers), every kernel value is computed exactly once per tile - the waves take turns, one k-block in four - and
// travels to the other waves through a double-buffered 2 KB slot in LDS.  No second pass, no cache, no recomputation,
// no spills; n <= 512 only (larger models keep the windowed form).
//
//   columns   round rho = j / 4 of column block j belongs to slot rho of every wave; within a round the four column
//             blocks are dealt boustrophedon (wave w owns 4 rho + w in even rounds, 4 rho + 3 - w in odd ones), which
//             balances the triangular work exactly: 528 variance MFMAs per wave at n = 512.
//   k-blocks  in groups of four (group g = round g).  While a wave consumes group g (its slots rho >= g; the diagonal
//             slot rho = g only for the k-blocks not above its column block) it produces the kernel values of k-block
//             4 (g + 1) + w for the next group in micro-steps between its MFMAs, adds their share of the mean, and
//             stores them; one workgroup barrier per group.
//   operands  each wave streams its own slice of L^-T, packed in consumption order (bbh_pack_coop_kernel), through the
//             8-deep register ring of the windowed form.  The slice has the same shape for every wave (the inactive
//             part of the diagonal slot is stored as zeros and skipped by a wave-uniform branch), so ring slots and
//             accumulators stay compile-time indices.
//   smaller n the model occupies the LAST rounds: groups [8 - nb / 4, 8) run, entered by wave-uniform branches.
#pragma once
#define BBH_CANDREG 1
#include "bbh_fused.h"

#define BBH_COOP_ROUNDS 8  // 8 rounds x 4 column blocks x 16 = 512 training points
#define BBH_COOP_KV_TILE (2 * 4 * 256)  // doubles of kernel-value buffer per candidate tile: [2 buffers][4 k-blocks][4 x 64]
#ifndef BBH_COOP_WAVES
#define BBH_COOP_WAVES 2  // waves per SIMD the register budget is set for (workgroups per CU)
#endif
// The operand loads of the main loop are inline assembly in the scalar-base form
// (global_load_dwordx{2,4} v, v_lane_offset, s[base:base+1] offset:imm) with counted s_waitcnt by hand.  The compiler
// addresses the same loads through 64-bit VGPR pointers - 2 VALU instructions per 4 KB window and wave, 18 % of the
// kernel's non-MFMA VALU work on a pipe that VALU and MFMA share - and cannot be talked into the scalar form
// (5.05 -> 4.89 ms).  The hand counts rely on every VMEM load of the loop being issued here (operand ring + training
// fragments), in program order, with sched_barrier fences between the MFMA slots.
#ifndef BBH_COOP_PAIRS
#define BBH_COOP_PAIRS 4  // operand ring: register pairs (two fragments each) in flight per wave
#endif
#ifndef BBH_COOP_ABLATE_BARRIER
#define BBH_COOP_ABLATE_BARRIER 0
#endif
#ifndef BBH_COOP_ABLATE_MEM
#define BBH_COOP_ABLATE_MEM 0  // timing experiment: every operand load hits the same 4 KB (L1-resident)
#endif
#ifndef BBH_COOP_ABLATE_PRO
#define BBH_COOP_ABLATE_PRO 0  // timing experiments on the per-tile set-up: 1 no alpha fill, 2 no candidate-row loads, 4 no first production, 8 no epilogue
#endif
#ifndef BBH_COOP_ABLATE_LOADS
#define BBH_COOP_ABLATE_LOADS 0  // timing experiment: no operand loads in the main loop at all
#endif
#ifndef BBH_COOP_ABLATE_CHAIN
#define BBH_COOP_ABLATE_CHAIN 0
#endif
#ifndef BBH_COOP_ABLATE_KV
#define BBH_COOP_ABLATE_KV 0
#endif

typedef double d2 __attribute__((ext_vector_type(2)));
// two consecutive fragments of the slice with one instruction: the slice stores fragment pairs lane-interleaved
// (pair p, lane l: [fragment 2p | fragment 2p + 1] at byte 1024 p + 16 l).  Every vector-memory instruction costs the
// MFMA stream issue bandwidth whatever it hits (measured: the same loads from one L1-resident window cost the same
// 0.5 ms per launch as the real ones); pairs halve their number.
template <int IMM>
__device__ __forceinline__ void coop_gload2(d2& dst, const double* sbase, unsigned voff) {
  static_assert(IMM >= 0 && IMM + 16 <= 4096, "13-bit signed immediate offset");
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM));
}
template <int N>
__device__ __forceinline__ void coop_vmwait2(d2& slot) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot) : "n"(N));
}
template <int IMM>
__device__ __forceinline__ void coop_gload(double& dst, const double* sbase, unsigned voff) {
  static_assert(IMM >= 0 && IMM < 4096, "13-bit signed immediate offset");
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM));
}
// at most N vector-memory loads may still be in flight afterwards; tied to the register that is about to be consumed
template <int N>
__device__ __forceinline__ void coop_vmwait(double& slot) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot) : "n"(N));
}

struct CoopArgs {
  FusedArgs f;
  const double* rstream;  // [4 waves][frags][64] operand slices
  int64_t frags;          // fragments per wave
  int g0;                 // first group that exists: 8 - nb / 4
};

// fragments of the groups before G (in the full 8-round numbering)
__host__ __device__ constexpr int coop_frags_before(int G) { return 16 * (G * BBH_COOP_ROUNDS - (G * (G - 1)) / 2); }

// NT = candidate tiles (of 16) per workgroup.  NT = 2: every operand fragment that comes back from memory feeds two
// MFMAs (one per tile), which halves the vector-memory traffic per MFMA, at the price of a second set of accumulators
// (2 x 64 registers: 256 VGPRs in all, no spills).  Measured: not faster than NT = 1 (4.72 vs 4.68 ms) - see bbh_panel.hip.
// OPEN: the operand slice continues behind this group with at least BBH_COOP_PAIRS more fragment pairs whatever the model
// size (the two-sweep form, bbh_coop2.h): the ring is always refilled and the waits never shorten.
template <int G, int KD, int KVF, bool PRODUCE, int NT, bool OPEN = false>
__device__ __forceinline__ void coop_group(const WaveCtx (&c)[NT], const double* rs, const double* tfn, const bbh_lds_double* kv_cur,
                                           bbh_lds_double* kv_mine_next, const bbh_lds_double* alpha_next, int tbn, int cw,
                                           d4 (&acc)[NT][BBH_COOP_ROUNDS], d2 (&ring)[BBH_COOP_PAIRS], double (&accm)[NT]) {
  // rs: this wave's operand slice at the start of the group (wave-uniform); tfn: training fragments of k-block tbn
  // (wave-uniform); the lane's slot inside a fragment (pair) is added by the load.  kv_cur / kv_mine_next: tile 0's
  // buffers, tile t's follow at t * BBH_COOP_KV_TILE doubles.
  constexpr int NP = BBH_COOP_PAIRS;        // ring: NP register pairs = 2 NP fragments in flight
  constexpr int CNT = BBH_COOP_ROUNDS - G;  // ring fragments per (k-block, k-step): slots G .. 7
  constexpr int FULL = CNT - 1;             // of which always multiplied
  constexpr int TOT = 16 * CNT;             // fragments of this group (even: pairs never straddle groups)
  constexpr int REM = OPEN ? (1 << 20) : coop_frags_before(BBH_COOP_ROUNDS) - coop_frags_before(G + 1);  // fragments after this group
  constexpr int HOSTS = 12 * FULL;  // MFMA slots of k-blocks 1..3 that carry the micro-steps of the production
  constexpr int STEPS = NT * BBH_KV_STEPS;  // micro-steps of this wave's production: tile 0's, then tile 1's
  const unsigned lane8 = (unsigned)c[0].l * 8u, lane16 = (unsigned)c[0].l * 16u;
  double tfv[KD];
  d4 dsa[NT], dsb[NT];
  KvState<BBH_KV_NU> P;
  double kv[NT][4], kvx[NT][4], kvn[NT][4], alv[4];
  if constexpr (PRODUCE) {
    static_for<0, KD>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      coop_gload<(k % 8) * 512>(tfv[k], tfn + (k / 8) * 512, lane8);
    });
  }
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) kv[t][r] = kv_cur[t * BBH_COOP_KV_TILE + r * 64];
  __builtin_amdgcn_sched_barrier(0);
  static_for<0, 4>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < 3) {  // the next k-block's values are requested one block ahead
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) kvx[t][r] = kv_cur[t * BBH_COOP_KV_TILE + (i + 1) * 256 + r * 64];
    } else if constexpr (PRODUCE) {
#pragma unroll
      for (int r = 0; r < 4; r++) alv[r] = alpha_next[4 * r];
    }
    static_for<0, 4>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      static_for<0, CNT>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        constexpr int f = (i * 4 + r) * CNT + s;  // fragment index inside the group; pair f / 2, half f % 2
        constexpr int pr = f / 2;
        if constexpr (f % 2 == 0 && !BBH_COOP_ABLATE_LOADS) {
          // pair loads younger than pair pr: NP - 1 (fewer at the very end of the slice), plus - for the pairs that
          // were already in flight when this group's training fragments were requested - those KD loads
          constexpr int BEHIND = ((TOT + REM) / 2 - 1 - pr) < (NP - 1) ? ((TOT + REM) / 2 - 1 - pr) : (NP - 1);
          coop_vmwait2<BEHIND + ((PRODUCE && pr < NP) ? KD : 0)>(ring[pr % NP]);
        }
#if BBH_COOP_ABLATE_CHAIN  // timing experiment (wrong results): every MFMA goes to the accumulator (f mod 8)
        static_for<0, NT>([&](auto tc_) __attribute__((always_inline)) {
          constexpr int t = decltype(tc_)::value;
          if constexpr (s == 0) {
            if (cw >= i) acc[t][f % 8] = mfma_f64(kv[t][r], ring[pr % NP][f % 2], acc[t][f % 8]);
          } else {
            acc[t][f % 8] = mfma_f64(kv[t][r], ring[pr % NP][f % 2], acc[t][f % 8]);
          }
        });
#else
        static_for<0, NT>([&](auto tc_) __attribute__((always_inline)) {
          constexpr int t = decltype(tc_)::value;
          if constexpr (s == 0) {  // diagonal slot: column block 4 G + cw, zero (and skipped) for k-blocks above it
            if (cw >= i) acc[t][G] = mfma_f64(kv[t][r], ring[pr % NP][f % 2], acc[t][G]);
          } else {
            acc[t][G + s] = mfma_f64(kv[t][r], ring[pr % NP][f % 2], acc[t][G + s]);
          }
        });
#endif
        if constexpr (f % 2 == 1 && 2 * (pr + NP) < TOT + REM && !BBH_COOP_ABLATE_LOADS) {  // both halves are consumed
          constexpr int np = pr + NP;  // pair to request, relative to the group start; 4 pairs per 4 KB window
          coop_gload2<(np % 4) * 1024>(ring[pr % NP], BBH_COOP_ABLATE_MEM ? rs : rs + (np / 4) * 512, lane16);
        }
        if constexpr (PRODUCE && i >= 1 && s >= 1 && !BBH_COOP_ABLATE_KV) {
          constexpr int m = ((i - 1) * 4 + r) * FULL + (s - 1);
          static_for<(m * STEPS) / HOSTS, ((m + 1) * STEPS) / HOSTS>([&](auto st) __attribute__((always_inline)) {
            constexpr int tile = decltype(st)::value / BBH_KV_STEPS, micro = decltype(st)::value % BBH_KV_STEPS;
            kv_micro<KVF, BBH_KV_NU, micro>(P, c[tile], tbn, 0, dsa[tile], dsb[tile], kvn[tile]);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (PRODUCE && i == 0 && r == 3) {
        coop_vmwait<NP>(tfv[KD - 1]);  // the NP pair loads in flight are all younger than the training fragments
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!BBH_COOP_ABLATE_KV) {
          static_for<0, NT>([&](auto tc_) __attribute__((always_inline)) {
            constexpr int t = decltype(tc_)::value;
            kvp_dist<KD>(c[t], tfv, dsa[t], dsb[t]);
          });
        }
      }
    });
    if constexpr (i < 3) {
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) kv[t][r] = kvx[t][r];
    }
  });
  if constexpr (PRODUCE) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if constexpr (BBH_COOP_ABLATE_KV) kvn[t][r] = tfv[r];
        kv_mine_next[t * BBH_COOP_KV_TILE + r * 64] = kvn[t][r];
        accm[t] = fma(kvn[t][r], alv[r], accm[t]);
      }
  }
}

// GMIN: the first round that can exist.  GMIN = 4 (n <= 256) instantiates only rounds 4 .. 7: four accumulator blocks instead of
// eight, 115 - 123 VGPRs instead of 153 - 193, i.e. four workgroups per CU instead of two or three - the small models BayBE
// campaigns live in have little MFMA work per tile to hide the per-tile set-up and the kernel-value chains behind.  GMIN = 6
// (n <= 128): two accumulator blocks, <= 102 VGPRs, five workgroups per CU.
template <int KD, int KVF, int NT, int GMIN = 0>
__global__ __launch_bounds__(256, (GMIN >= 6 ? 5 : GMIN >= 4 ? 4 : BBH_COOP_WAVES)) void bbh_coop_posterior_kernel(const CoopArgs ca) {
  const FusedArgs& a = ca.f;
  extern __shared__ __attribute__((aligned(16))) double s_mem[];  // alpha [16 nb] | kv [NT][2][4][256] | red [NT][2][4][16]
  const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnd = l & 15, q = l >> 4;
  double* s_alpha = s_mem;
  double* s_kv = s_alpha + 16 * a.nb;
  double* s_red = s_kv + NT * BBH_COOP_KV_TILE;
  const int64_t tile0 = (int64_t)blockIdx.x * (16 * NT);

  WaveCtx c[NT];
  int tcs[NT];
  // candidate fragments: b = x * scl + ofs, augmented with [1, |b|^2]; every wave builds the tiles' fragments itself.
  // Loads first, all of them unconditional (clamped index) and independent: written as `if (dim < dn) v = fma(xr[numcol[dim]],
  // ...)` every k-step became its own divergent block - column index, wait, row value, wait - twelve dependent memory round
  // trips per tile before the first kernel value could be computed.
  int xcol[KD];
  double xval[NT][KD], xscl[KD], xofs[KD];
  const double* xr[NT];
  // the training fragments of this wave's first k-block are requested before anything else: their round trip passes under
  // the candidate-row loads instead of following them (the per-tile set-up is a chain of memory round trips: at n = 128 a tile
  // is 8 us of which the MFMA work is 5)
  double tfv0[KD];
  kvp_load<KD>(a.trainfrag + l, w, tfv0);
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int64_t row = (tile0 + 16 * t + cnd < a.N) ? tile0 + 16 * t + cnd : a.N - 1;
    xr[t] = a.X + row * a.ldx;
  }
  if (a.numcol_identity) {  // the numerical columns are the leading comp-rep columns, in order: no index round trip
#pragma unroll
    for (int k = 0; k < KD; k++) xcol[k] = (4 * k + q < a.dn) ? 4 * k + q : a.dn - 1;
  } else {
#pragma unroll
    for (int k = 0; k < KD; k++) xcol[k] = a.numcol[(4 * k + q < a.dn) ? 4 * k + q : a.dn - 1];
  }
#pragma unroll
  for (int k = 0; k < KD; k++) {
    const int dimc = (4 * k + q < a.dn) ? 4 * k + q : a.dn - 1;
#pragma unroll
    for (int t = 0; t < NT; t++) xval[t][k] = xr[t][xcol[k]];
    xscl[k] = a.scl[dimc];
    xofs[k] = a.ofs[dimc];
  }
  // alpha -> LDS: requested after the candidate-row loads, so that its round trip passes under theirs
  if (!(BBH_COOP_ABLATE_PRO & 1))
    for (int s = threadIdx.x; s < 16 * a.nb; s += 256) s_alpha[s] = a.meanB[(int64_t)s * 16];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    double nbsum = 0.0;
#pragma unroll
    for (int k = 0; k < KD; k++) {
      const int dim = 4 * k + q;
      double v = 0.0;
      if (dim < a.dn) {
        v = (BBH_COOP_ABLATE_PRO & 2) ? 0.01 * (double)(dim + cnd) : fma(xval[t][k], xscl[k], xofs[k]);
        nbsum = fma(v, v, nbsum);
      }
      c[t].cf[k] = v;
    }
    nbsum += __shfl_xor(nbsum, 16, 64);
    nbsum += __shfl_xor(nbsum, 32, 64);
#pragma unroll
    for (int k = 0; k < KD; k++) {
      if (4 * k + q == a.dn) c[t].cf[k] = 1.0;
      if (4 * k + q == a.dn + 1) c[t].cf[k] = nbsum;
    }
    int tc = 0;
    if constexpr ((KVF & 1) != 0) {  // task / outputscale table: the candidate's own task selects the table row
      if (a.task_col >= 0) {
        tc = (int)xr[t][a.task_col];
        tc = tc < 0 ? 0 : (tc >= a.T ? a.T - 1 : tc);
      }
    }
    tcs[t] = tc;
    c[t].tf = a.trainfrag + l;
    c[t].candl = nullptr;
    c[t].mb = nullptr;
    c[t].tbl = a.tasktbl;
    c[t].taskext = a.taskext;
    c[t].kvc = nullptr;
    c[t].kvl = (bbh_lds_double*)nullptr;
    c[t].nl = 0;
    c[t].ncache = 0;
    c[t].al = (const bbh_lds_double*)nullptr;
    c[t].kd = KD;
    c[t].kind = a.kind;
    c[t].T = a.T;
    c[t].tc = tc;
    c[t].q = q;
    c[t].l = l;
    c[t].dn = a.dn;
  }

  bbh_lds_double* kvb = (bbh_lds_double*)(s_kv + l);  // [tile][buffer][k-block of the group][4 values x 64 lanes]
  const bbh_lds_double* alq = (const bbh_lds_double*)(s_alpha + q);  // alpha[16 tb + 4 r + q]
  const int g0 = ca.g0;
  double accm[NT];
  d4 acc[NT][BBH_COOP_ROUNDS];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    accm[t] = 0.0;
#pragma unroll
    for (int s = 0; s < BBH_COOP_ROUNDS; s++) acc[t][s] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  // wave-uniform stream pointer + lane index: scalar base / 32-bit lane offset addressing (no 64-bit VALU pointer
  // arithmetic).  The first fragments are requested before the first kernel values are computed: their L2 latency
  // passes under that work (the compiler's own wait for the training fragments below also covers these older loads).
  const double* rs = ca.rstream + (int64_t)w * ca.frags * 64;
  d2 ring[BBH_COOP_PAIRS];
  static_for<0, BBH_COOP_PAIRS>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    coop_gload2<(i % 4) * 1024>(ring[i], rs + (i / 4) * 512, (unsigned)l * 16u);
  });
  {  // the first group's kernel values: wave w produces k-block w, not overlapped with anything
    double tfv[KD], kv0[NT][4];
    if constexpr (BBH_COOP_ABLATE_PRO & 4) {
      for (int t = 0; t < NT; t++)
        for (int r = 0; r < 4; r++) kv0[t][r] = 0.5 + 0.001 * (double)(l + r);
    } else {
#pragma unroll
      for (int k = 0; k < KD; k++) tfv[k] = tfv0[k];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        d4 dsa, dsb;
        kvp_dist<KD>(c[t], tfv, dsa, dsb);
        kv_all<KVF>(c[t], w, dsa, dsb, kv0[t]);
      }
    }
    __syncthreads();  // alpha is in LDS
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        kvb[t * BBH_COOP_KV_TILE + ((g0 & 1) * 4 + w) * 256 + r * 64] = kv0[t][r];
        accm[t] = fma(kv0[t][r], alq[16 * w + 4 * r], accm[t]);
      }
  }
  __syncthreads();  // group g0 is complete in LDS

  static_for<GMIN, BBH_COOP_ROUNDS>([&](auto gc) __attribute__((always_inline)) {
    constexpr int G = decltype(gc)::value;
    if (G >= g0) {  // wave-uniform: a smaller model occupies the last rounds only
      const int cw = (G & 1) ? 3 - w : w;
      const int tbn = 4 * (G + 1 - g0) + w;  // real k-block this wave produces for the next group
      constexpr bool PRODUCE = G + 1 < BBH_COOP_ROUNDS;
      coop_group<G, KD, KVF, PRODUCE, NT>(c, rs, a.trainfrag + (int64_t)tbn * KD * 64, kvb + (G & 1) * 4 * 256,
                                          kvb + (((G + 1) & 1) * 4 + w) * 256, alq + 16 * tbn, tbn, cw, acc, ring, accm);
      rs += (int64_t)16 * (BBH_COOP_ROUNDS - G) * 64;
#if !BBH_COOP_ABLATE_BARRIER  // (timing experiment only: wrong results without the barrier)
      if constexpr (PRODUCE) __syncthreads();
#endif
    }
  });

  // ---- ||v||^2 over this wave's column blocks, then over the 16 columns of a block (lanes), then over the waves ----
  if constexpr (BBH_COOP_ABLATE_PRO & 8) {  // no reductions: one value per accumulator row keeps the work alive
    double keep = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      keep += accm[t];
#pragma unroll
      for (int s = 0; s < BBH_COOP_ROUNDS; s++) keep += (acc[t][s][0] + acc[t][s][1]) + (acc[t][s][2] + acc[t][s][3]);
    }
    if (l == 0 && w == 0) a.var[tile0] = keep;
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < BBH_COOP_ROUNDS; s++)
#pragma unroll
      for (int r = 0; r < 4; r++) ss[r] = fma(acc[t][s][r], acc[t][s][r], ss[r]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double v = ss[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      ss[r] = v;  // candidate q + 4 r
    }
    double mp = accm[t];  // lane (q, cnd): a quarter of this wave's share of candidate cnd's mean
    mp += __shfl_xor(mp, 16, 64);
    mp += __shfl_xor(mp, 32, 64);
    double* red_v = s_red + t * 128;  // [4 waves][16 candidates]
    double* red_m = red_v + 64;
    if (cnd == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) red_v[w * 16 + q + 4 * r] = ss[r];
    }
    if (q == 0) red_m[w * 16 + cnd] = mp;
  }
  __syncthreads();
  if (threadIdx.x < 16 * NT) {  // wave 0: lane m + 16 t finishes candidate m of tile t
    const int m = threadIdx.x & 15, t = threadIdx.x >> 4;
    const int64_t gi = tile0 + 16 * t + m;
    const double* red_v = s_red + t * 128;
    const double* red_m = red_v + 64;
    const double sv = (red_v[m] + red_v[16 + m]) + (red_v[32 + m] + red_v[48 + m]);
    const double sm = (red_m[m] + red_m[16 + m]) + (red_m[32 + m] + red_m[48 + m]);
    double pv = a.prior_scale, mc = a.mean_const;
    if constexpr ((KVF & 1) != 0) {
      // lane m of wave 0 holds candidate m's task of every tile (cnd = m, q = 0)
      int tcm = __shfl(tcs[0], m, 64);
      if constexpr (NT > 1) {
        const int tcm1 = __shfl(tcs[NT - 1], m, 64);
        tcm = t ? tcm1 : tcm;
      }
      pv = a.tasktbl[tcm * a.T + tcm];
      if (a.taskmean) mc = a.taskmean[tcm];
    }
    if (gi < a.N) {
      if (a.mean) a.mean[gi] = a.ybar + a.ysd * (mc + sm);
      if (a.var) a.var[gi] = a.ysd * a.ysd * (pv - sv);
    }
  }
}

// Instantiations (bbh_fused_coop_a.hip: 2, 4, 6 k-steps of the distance GEMM; _b: 8, 12, 16): Matérn-5/2 with and
// without the task / outputscale table, RBF with and without, Matérn-3/2 without.
// false: no instantiation for this model; grid.x == 0 only asks.
bool bbh_coop_launch(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);
bool bbh_coop_launch_a(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);
bool bbh_coop_launch_b(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);
// n <= 256 (g0 >= 4): the four-round instantiations (Matérn-5/2 with and without table; 2, 4, 6, 8 k-steps)
bool bbh_coop_launch_small(int kd, int kind, bool has_tbl, dim3 grid, size_t lds, hipStream_t s, const CoopArgs& a);

#define BBH_COOP_DISPATCH_KD(KDV, NTV)                                                                                        \
  if (kd == KDV) {                                                                                                       \
    const bool m52 = kind == BBH_KERNEL_MATERN52, rbf = kind == BBH_KERNEL_RBF, plain = kind == BBH_KERNEL_MATERN32 && !has_tbl; \
    if (!m52 && !rbf && !plain) return false;                                                                            \
    if (grid.x == 0) return true;                                                                                        \
    if (rbf && has_tbl) /* multi-task HVARFNER / BOTORCH presets */                                                      \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 3, NTV>), grid, dim3(256), lds, s, a);                               \
    else if (rbf)                                                                                                        \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 2, NTV>), grid, dim3(256), lds, s, a);                               \
    else if (kind == BBH_KERNEL_MATERN32)                                                                                \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 4, NTV>), grid, dim3(256), lds, s, a);                               \
    else if (has_tbl)                                                                                                    \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 1, NTV>), grid, dim3(256), lds, s, a);                               \
    else                                                                                                                 \
      hipLaunchKernelGGL((bbh_coop_posterior_kernel<KDV, 0, NTV>), grid, dim3(256), lds, s, a);                               \
    return true;                                                                                                         \
  }
