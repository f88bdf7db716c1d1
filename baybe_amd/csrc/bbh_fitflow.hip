// One objective evaluation of the hyper-parameter fit as ONE launch for 64 < np <= 2048: Gram tiles -> Cholesky factor and its
// inverse -> K^-1 -> alpha -> criterion value -> gradient sums, as a dataflow of 64 x 64 tile roles handed to persistent
// workgroups by ticket.  theta is read from, and (value, gradient, Cholesky flag) are written to, host-mapped pinned memory:
// one launch and one stream synchronisation per evaluation, no copies, no memsets.
// Reference: the objective botorch.fit.fit_gpytorch_mll minimises, call site baybe/surrogates/gaussian_process/core.py:272-341
// (criterion: components/fit_criterion.py:31-41 - ExactMarginalLogLikelihood for one task, LeaveOneOutPseudoLikelihood for ICM).
//
// Launch by launch the same evaluation was (profiles/r04_cfg3_bench_kernel_stats.csv, n = 512): H2D theta, Gram 15 us, the
// tile-dataflow factorisation 197, residual 5, two mat-vecs 16, X^T X in four slices + their sum 17, memset, value 5, gradient
// pairs 59 + reduction 7, two D2H - 353 us, every kernel waiting for the previous one to drain.  Here:
//   RH(I)    row head: Gram tiles (I, I) and (I, I-1) generated in LDS, left-looking updates, L_{I,I-1}, factor + invert -> D_I
//   L(I,K)   K <= I-2: Gram tile, updates, L_IK = A_IK D_K^T
//   XT(I,J)  I > J:   X_IJ = -D_I sum_K L_IK X_KJ                           (X = L^-1)
//   MT(I,J)  I >= J:  M_IJ = sum_{K >= I} X_KI^T X_KJ  (M = K^-1), and the two 64-vectors M_IJ r_J, M_IJ^T r_I of alpha = M r
//   VEC      alpha, and for the leave-one-out criterion d = diag M, u, w
//   QV(I), QT(I,J)  (LOO only)  q = M w,  Q = M diag(u) M
//   GT(I,J)  I >= J:  the tile's pairs: G_ab times the kernel derivatives, summed per hyper-parameter slot; the last one to
//            finish adds the tiles' partial rows in a fixed order and writes the results.
// Roles are numbered in an order in which every role only waits for lower numbers (column by column of the factorisation,
// then MT, VEC, QV, QT, GT) and are taken by ticket (one atomic add per role), so a waiting workgroup always waits for a role
// that a running workgroup holds: no co-residency requirement, unlike bbh_potrf_tiles_kernel's fixed workgroup <-> tile map.
// Flags are epoch-stamped, counters cumulative (the host passes their base values): nothing is reset between launches.
// Same arithmetic as the launch path (bbh_gram_kernel, bbh_value_kernel, bbh_grad_pair_kernel) with sums in a different fixed
// order: value 1e-11, gradient 1e-8 relative between the two (tests/test_gpu_parity.py).
// Covers: one kernel (any kind but the periodic one), <= 32 numerical columns, <= 4 tasks (ICM), scalar noise / mean.
#include <math.h>

#include "bbh_common.h"
#include "bbh_tiles.h"

#define FF_MAXD 32
#define FF_MAXT 4
#define FF_STRIDE PD_MT_STRIDE  // flag arrays are [32][32]
#define FF_MAXBLK 32             // block rows the dataflow forms take (np <= 2048); beyond 16 only the one-launch form exists (bbh_potrf_tiles_kernel
                                // needs every tile co-resident: 16 block rows on 256 CUs)
#define FF_FLAGS (FF_STRIDE * FF_STRIDE)
#define FF_MISC_LOGDET 0                  // misc: four per-block-row tables of FF_STRIDE doubles
#define FF_MISC_QSUM FF_STRIDE
#define FF_MISC_VALUE (2 * FF_STRIDE)
#define FF_MISC_MEANG (3 * FF_STRIDE)
#define FF_COUNTERS (4 * FF_FLAGS + 2 * FF_STRIDE)  // offset of the counters in the flag block: L | X | M | Q | V [2][stride] | counters [8]

enum { FF_RH = 0, FF_L = 1, FF_XT = 2, FF_MT = 3, FF_VEC = 4, FF_QV = 5, FF_QT = 6, FF_GT = 7 };

struct FlowArgs {
  const double* xnT;    // [dn][np]
  const int* task;      // [np]
  const double* nmask;  // [np]
  const double* ystd;   // [np]
  const double* theta;  // [tl] device or host-mapped; null: thv
  double thv[PD_GRAM_MAXTHV];  // theta as kernel arguments
  int n, np, nbk, dn, T, tl, criterion;
  bbh_kern_spec ks;
  double* A;      // [np][np] L tiles (lower)
  double* D;      // [nbk][64][64]
  double* X;      // [np][np] strictly lower tiles of L^-1
  double* M;      // [np][np] K^-1 (lower tiles; LOO: both triangles)
  double* Q;      // [np][np] LOO: M diag(u) M (lower tiles)
  double* alpha;  // [np]
  double* u;      // [np] LOO
  double* w;      // [np] LOO
  double* q;      // [np] LOO
  double* apart;  // [32*32][2][64] alpha partials of the M-tiles
  double* gpart;  // [nG][nsl] gradient partial rows of the G-tiles
  double* misc;   // FF_MISC_LOGDET + I: log-determinant partial of row head I, _QSUM: LOO sum of q over block I, _VALUE: value share of block I, _MEANG: mean-gradient share (MLL)
  int* flagsL;    // [32][32]
  int* flagsX;
  int* flagsM;
  int* flagsQ;
  int* flagsV;    // [I] VEC(I), [FF_STRIDE + I] QV(I)
  int* counters;  // [0] tickets, [1] finished M-tiles, [2] finished G-roles, [3] abort word (tail form), [4] finished VEC roles   (cumulative over launches)
  int ticket_base, doneM_base, doneG_base, doneV_base;
  const int* roles;  // [nroles] type | I << 4 | J << 9 | row quarter << 14 | partial-row index << 16   (this launch's part of the table)
  int nroles, nM, nG;
  int role_lo;       // index of roles[0] in the whole table (trace slots; split form: the second launch starts behind the first one's roles)
  int epoch, spin;
  int tail_only;   // the factor and its inverse are already in A / D / X (bbh_potrf_trtri ran before): roles MT ... GT only, no waits on them
  int* info;       // device: Cholesky flag (0 ok, > 0 failing pivot + 1, -7 a wait gave up)
  int* abort;      // the word the waits watch for -7: info itself in the full form; a word of its own after a separate factorisation
                   // (whose own give-up is reported through info and must not stop the roles here)
  double* out;     // host-mapped [1 + tl]
  int* info_out;   // host-mapped
  long long* dbg;  // BBH_FLOW_TRACE=1: [nroles][8] wall_clock64 stamps: 0 role start, 1 D arrived (row heads), 2 end, 3 workgroup, 7 D loaded, 4 before / 5 after the factorisation, 6 D published
};

namespace {

typedef double (*tile_t)[PD_LD];

// c (+)= sign * a^T b, 64 x 64 x 64 (a, b stored [k][.])
template <bool ACCUM>
__device__ __forceinline__ void ff_gemm64_tn(tile_t c, const double (*a)[PD_LD], const double (*b)[PD_LD], double sign) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int sb = w; sb < 16; sb += 4) {
    const int mb = sb & 3, nb = sb >> 2;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
        acc = mfma_f64(a[16 * kb + 4 * ks + (l >> 4)][16 * mb + (l & 15)], b[16 * kb + 4 * ks + (l >> 4)][16 * nb + (l & 15)], acc);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double* out = &c[16 * mb + (l >> 4) + 4 * r][16 * nb + (l & 15)];
      *out = ACCUM ? *out + sign * acc[r] : sign * acc[r];
    }
  }
}

// bbh_kbase, inlined: a call from these loops has to save the ~130 live registers of the caller around it (scratch traffic:
// the Gram tile took 30 us, the gradient tile 80 us with the out-of-line function)
__device__ __forceinline__ double ff_kbase(int kind, double r2, int jb, double alpha) {
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

// kernel value and g = -(dk/dr) / r together: the Matern / RBF / RQ kinds share their exponential (bbh_kbase + bbh_gfun evaluate it twice)
__device__ __forceinline__ void ff_kv_pair(int kind, double r2, int jb, double alpha, double& kb, double& g) {
  if (kind == BBH_KERNEL_MATERN52) {
    const double r = sqrt(r2), e = exp(-BBH_SQRT5 * r);
    kb = (1.0 + BBH_SQRT5 * r + (5.0 / 3.0) * r2) * e;
    g = (5.0 / 3.0) * (1.0 + BBH_SQRT5 * r) * e;
  } else if (kind == BBH_KERNEL_MATERN32) {
    const double r = sqrt(r2), e = exp(-BBH_SQRT3 * r);
    kb = (1.0 + BBH_SQRT3 * r) * e;
    g = 3.0 * e;
  } else if (kind == BBH_KERNEL_RBF) {
    kb = g = exp(-0.5 * r2);
  } else {
    kb = ff_kbase(kind, r2, jb, alpha);
    g = bbh_gfun(kind, r2, jb, alpha);
  }
}

// ---- 64 x 64 x 64 products with the result in MFMA accumulators: wave w owns the sub-blocks sb = w, w + 4, w + 8, w + 12
//      (mb = sb & 3, nb = sb >> 2); the operands come from two LDS tiles, nothing is written back between the k-steps ----
__device__ __forceinline__ void ff_mma_tn(d4 (&acc)[4], const double (*a)[PD_LD], const double (*b)[PD_LD]) {  // acc += a^T b
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
__device__ __forceinline__ void ff_mma_nn(d4 (&acc)[4], const double (*a)[PD_LD], const double (*b)[PD_LD]) {  // acc += a b
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int sb = w + 4 * q, mb = sb & 3, nb = sb >> 2;
#pragma unroll
    for (int kb = 0; kb < 4; kb++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
        acc[q] = mfma_f64(a[16 * mb + (l & 15)][16 * kb + 4 * ks + (l >> 4)], b[16 * kb + 4 * ks + (l >> 4)][16 * nb + (l & 15)], acc[q]);
  }
}
__device__ __forceinline__ void ff_acc_to_tile(tile_t out, const d4 (&acc)[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int sb = w + 4 * q, mb = sb & 3, nb = sb >> 2;
#pragma unroll
    for (int r = 0; r < 4; r++) out[16 * mb + (l >> 4) + 4 * r][16 * nb + (l & 15)] = acc[q][r];
  }
}
// a tile through registers: the loads of the next k-step are in flight while the MFMAs of this one run
__device__ __forceinline__ void ff_ld_regs(pd_d2 (&r)[8], const double* src, int64_t ld) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int e = threadIdx.x + 256 * k;
    r[k] = *(const pd_d2*)(src + (int64_t)(e >> 5) * ld + 2 * (e & 31));
  }
}
__device__ __forceinline__ void ff_st_regs(tile_t dst, const pd_d2 (&r)[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int e = threadIdx.x + 256 * k;
    *(pd_d2*)&dst[e >> 5][2 * (e & 31)] = r[k];
  }
}
// sum over the 16 lanes of a DPP row (row_shr 8, 4, 2, 1 with zero fill): the total arrives in the row's lane 15
template <int CTRL>
__device__ __forceinline__ double ff_dpp_shr(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ff_row16_sum(double v) {
  v += ff_dpp_shr<0x118>(v);
  v += ff_dpp_shr<0x114>(v);
  v += ff_dpp_shr<0x112>(v);
  v += ff_dpp_shr<0x111>(v);
  return v;
}

struct FlowShared {
  pd_gram_lds g;          // theta, inverse lengthscales, the tile's noise mask and task ids (rows / columns)
  double vr[64], vc[64];  // per-row / per-column vectors of the current tile (r, alpha, ...)
  double vr2[64], vc2[64];
  double red[4][64 + 8];
  int ticket;
  int last;
};

// rows of block I of the transposed inputs -> flat [dn][64]
__device__ __forceinline__ void ff_stage_x(double* dst, const FlowArgs& fa, int I) {
  for (int e = threadIdx.x; e < fa.dn * 64; e += 256) dst[e] = fa.xnT[(int64_t)(e >> 6) * fa.np + I * 64 + (e & 63)];
}

__device__ __forceinline__ void ff_stage_meta(FlowShared& sh, const FlowArgs& fa, int I, int K) {
  const int t = threadIdx.x;
  if (t < 64) {
    sh.g.tr[t] = fa.task[I * 64 + t];
    sh.g.nm[t] = fa.nmask[I * 64 + t];
  } else if (t < 128) {
    sh.g.tc[t - 64] = fa.task[K * 64 + (t - 64)];
  }
}

// Gram tile (I, K) into out: K[a][b] = os B[ta][tb] k(r_ab) + (noise nmask_a) [a == b]; identity on the padding.
// Thread t owns row i = t >> 2 and the columns j = (t & 3) + 4 m: the row's inputs sit in registers for all 16 entries, the columns'
// come from LDS (four distinct addresses per wave), the dimension loop is unrolled so that the LDS reads are in flight together
// (a run-time loop with three LDS reads per dimension took 30 us per tile: latency-bound).
// (the tile arithmetic of the tile-dataflow factorisation: pd_gram_tile, four entries in flight per thread, kind as a template constant)
__device__ __forceinline__ void ff_gram_tile(tile_t out, const double* xr, const double* xc, int I, int K, const FlowArgs& fa,
                                             const FlowShared& sh) {
  pd_gram_src gs;
  gs.xnT = fa.xnT;
  gs.task = fa.task;
  gs.nmask = fa.nmask;
  gs.theta = nullptr;
  gs.n = fa.n;
  gs.np = fa.np;
  gs.dn = fa.dn;
  gs.T = fa.T;
  gs.tl = fa.tl;
  gs.kind = fa.ks.kind[0];
  gs.use_os = fa.ks.use_os;
  gs.jb = fa.ks.jb;
  gs.alpha_off = fa.ks.alpha_off;
  gs.d_sc1 = 0;
  gs.dbg = nullptr;
  pd_gram_tile(out, xr, xc, I, K, gs, sh.g);
}

__device__ __forceinline__ bool ff_wait(const int* flag, const FlowArgs& fa, bool urgent = false, bool lazy = false) {
  return pd_wait_n(flag, fa.epoch, fa.abort, fa.spin, urgent, true, nullptr, lazy);
}

// wait until a cumulative counter reaches target
__device__ __forceinline__ bool ff_wait_count(const int* counter, int target, const FlowArgs& fa, bool lazy = false) {
  __shared__ int s_ok2;
  if (threadIdx.x == 0) {
    int ok = 1, it = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
      if (++it > fa.spin || ((it & 15) == 0 && __hip_atomic_load(fa.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == -7)) {
        __hip_atomic_store(fa.abort, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      if (lazy)
        __builtin_amdgcn_s_sleep(32);
      else
        __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok2 = ok;
  }
  __syncthreads();
  const bool ok = s_ok2 != 0;
  __syncthreads();
  return ok;
}

__device__ __forceinline__ double ff_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool ff_role_rowhead(const FlowArgs& fa, FlowShared& sh, tile_t a, tile_t b, tile_t c, tile_t al, int I) {
  const int nbk = fa.nbk;
  double* Aii = fa.A + (int64_t)(I * 64) * fa.np + I * 64;
  double* Ail = Aii - 64;
  ff_stage_x((double*)b, fa, I);
  if (I > 0) ff_stage_x((double*)c, fa, I - 1);
  ff_stage_meta(sh, fa, I, I);
  __syncthreads();
  ff_gram_tile(a, (const double*)b, (const double*)b, I, I, fa, sh);
  if (I > 0) {
    __syncthreads();
    ff_stage_meta(sh, fa, I, I - 1);
    __syncthreads();
    ff_gram_tile(al, (const double*)b, (const double*)c, I, I - 1, fa, sh);
  }
  __syncthreads();
  for (int J = 0; J + 1 < I; J++) {  // (order and the early publication below: as in bbh_potrf_tiles_kernel)
    if (!ff_wait(&fa.flagsL[I * FF_STRIDE + J], fa)) return false;
    pd_load_tile(b, fa.A + (int64_t)(I * 64) * fa.np + J * 64, fa.np);
    __syncthreads();
    pd_gemm64<true, true, PD_OUT_LOWER>(a, b, b, -1.0);
    if (!ff_wait(&fa.flagsL[(I - 1) * FF_STRIDE + J], fa)) return false;
    pd_load_tile(c, fa.A + (int64_t)((I - 1) * 64) * fa.np + J * 64, fa.np);
    __syncthreads();
    pd_gemm64<true, true, PD_FULL>(al, b, c, -1.0);
    __syncthreads();
  }
  if (I > 0) {
    if (!ff_wait(&fa.flagsL[(I - 1) * FF_STRIDE + (I - 1)], fa, true)) return false;
    if (fa.dbg && threadIdx.x == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 1] = wall_clock64();
    pd_load_tile(b, fa.D + (int64_t)(I - 1) * 4096, 64);
    __syncthreads();
    if (fa.dbg && threadIdx.x == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 7] = wall_clock64();
    pd_gemm64<true, false, PD_B_LOWER>(c, al, b, 1.0);  // L_{I,I-1} = A_{I,I-1} D_{I-1}^T
    __syncthreads();
    pd_store_tile_wt(Ail, fa.np, c, 1.0);
    pd_gemm64<true, true, PD_OUT_LOWER>(a, c, c, -1.0);
    pd_publish_wt(&fa.flagsL[I * FF_STRIDE + (I - 1)], fa.epoch);  // (before the factorisation: the next row head's last update waits for it)
  }
  if (fa.dbg && threadIdx.x == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 4] = wall_clock64();
  pd_factor_block4(a, b, al, (int64_t)I * 64, fa.info, nullptr, fa.epoch);
  if (fa.dbg && threadIdx.x == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 5] = wall_clock64();
  pd_store_tile_wt(fa.D + (int64_t)I * 4096, 64, b, 1.0);
  pd_publish_wt(&fa.flagsL[I * FF_STRIDE + I], fa.epoch);  // D_I first: the next row head waits for it
  if (fa.dbg && threadIdx.x == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 6] = wall_clock64();
  if (threadIdx.x < 64) {  // log-determinant partial (the padding's diagonal is 1)
    const double v = ff_wave_sum(log(a[threadIdx.x][threadIdx.x]));
    if (threadIdx.x == 0) fa.misc[FF_MISC_LOGDET + I] = v;
  }
  pd_store_tile(Aii, fa.np, a, 1.0);
  (void)nbk;
  return true;
}

__device__ __forceinline__ bool ff_role_ltile(const FlowArgs& fa, FlowShared& sh, tile_t a, tile_t b, tile_t c, int I, int K) {
  double* Aik = fa.A + (int64_t)(I * 64) * fa.np + K * 64;
  ff_stage_x((double*)b, fa, I);
  ff_stage_x((double*)c, fa, K);
  ff_stage_meta(sh, fa, I, K);
  __syncthreads();
  ff_gram_tile(a, (const double*)b, (const double*)c, I, K, fa, sh);
  __syncthreads();
  for (int J = 0; J < K; J++) {
    if (!ff_wait(&fa.flagsL[I * FF_STRIDE + J], fa) || !ff_wait(&fa.flagsL[K * FF_STRIDE + J], fa)) return false;
    pd_load_tile(b, fa.A + (int64_t)(I * 64) * fa.np + J * 64, fa.np);
    pd_load_tile(c, fa.A + (int64_t)(K * 64) * fa.np + J * 64, fa.np);
    __syncthreads();
    pd_gemm64<true, true, PD_FULL>(a, b, c, -1.0);
    __syncthreads();
  }
  if (!ff_wait(&fa.flagsL[K * FF_STRIDE + K], fa)) return false;
  pd_load_tile(b, fa.D + (int64_t)K * 4096, 64);
  __syncthreads();
  pd_gemm64<true, false, PD_B_LOWER>(c, a, b, 1.0);  // L_IK = A_IK D_K^T
  __syncthreads();
  pd_store_tile_wt(Aik, fa.np, c, 1.0);
  pd_publish_wt(&fa.flagsL[I * FF_STRIDE + K], fa.epoch);
  return true;
}

__device__ __forceinline__ bool ff_role_xtile(const FlowArgs& fa, tile_t a, tile_t b, tile_t c, int I, int J) {
  for (int e = threadIdx.x; e < 4096; e += 256) a[e >> 6][e & 63] = 0.0;
  __syncthreads();
  for (int K = J; K < I; K++) {
    if (!ff_wait(&fa.flagsL[I * FF_STRIDE + K], fa)) return false;
    if (!ff_wait(K == J ? &fa.flagsL[J * FF_STRIDE + J] : &fa.flagsX[K * FF_STRIDE + J], fa)) return false;
    pd_load_tile(b, fa.A + (int64_t)(I * 64) * fa.np + K * 64, fa.np);
    if (K == J)
      pd_load_tile(c, fa.D + (int64_t)J * 4096, 64);
    else
      pd_load_tile(c, fa.X + (int64_t)(K * 64) * fa.np + J * 64, fa.np);
    __syncthreads();
    pd_gemm64<false, true, PD_FULL>(a, b, c, 1.0);  // acc += L_IK X_KJ
    __syncthreads();
  }
  if (!ff_wait(&fa.flagsL[I * FF_STRIDE + I], fa)) return false;
  pd_load_tile(b, fa.D + (int64_t)I * 4096, 64);
  __syncthreads();
  pd_gemm64<false, false, PD_A_LOWER>(c, b, a, -1.0);  // X_IJ = -D_I acc
  __syncthreads();
  pd_store_tile_wt(fa.X + (int64_t)(I * 64) * fa.np + J * 64, fa.np, c, 1.0);
  pd_publish_wt(&fa.flagsX[I * FF_STRIDE + J], fa.epoch);
  return true;
}

// M_IJ = sum_{K >= I} X_KI^T X_KJ (X_KK = D_K), its alpha partials, published; I >= J.  Two LDS tiles (t0, t1) for the operands,
// the sum in MFMA accumulators, the next k-step's operands on their way into registers while this one's MFMAs run.
__device__ __forceinline__ bool ff_role_mtile(const FlowArgs& fa, FlowShared& sh, tile_t t0, tile_t t1, int I, int J) {
  const int t = threadIdx.x;
  pd_mt_args ma;
  ma.nM = 0;
  ma.M = fa.M;
  ma.ld = fa.np;
  ma.apart = fa.apart;
  ma.ystd = fa.ystd;
  ma.theta = nullptr;
  ma.cmean = sh.g.th[1];
  ma.n = fa.n;
  ma.loo = fa.criterion == BBH_CRITERION_LOO;
  ma.flagsM = fa.flagsM;
  ma.doneM = &fa.counters[1];
  ma.flow_epoch = fa.epoch;
  auto wait = [&](int K) -> bool {
    if (fa.tail_only) return true;
    // (single-launch form: these roles poll from the first microsecond on - lazily, see pd_wait_n)
    if (!ff_wait(K == I ? &fa.flagsL[I * FF_STRIDE + I] : &fa.flagsX[K * FF_STRIDE + I], fa, false, true)) return false;
    return I == J || ff_wait(&fa.flagsX[K * FF_STRIDE + J], fa, false, true);  // (K >= I > J)
  };
  if (!pd_mtile_core(t0, t1, sh.vr, sh.vc, I, J, fa.nbk, fa.D, fa.X, fa.np, ma, sh.g.th[1], wait)) return false;
  pd_publish(&fa.flagsM[I * FF_STRIDE + J], fa.epoch);
  if (t == 0) atomicAdd(&fa.counters[1], 1);  // (thread 0's release fence in pd_publish precedes it)
  return true;
}

// alpha_I = (M r)_I for block row I from the tiles' partials (fixed order; four threads per entry); MLL: the block's share of the
// data-fit sum (with the log-determinant term after a separate factorisation) and of the mean gradient; LOO: d, u, w of the block.
// One role per block row: a single role for all rows was 7-17 us between the last M-tile and the first gradient role.
__device__ __forceinline__ bool ff_role_vec(const FlowArgs& fa, FlowShared& sh, int I) {
  if (!ff_wait_count(&fa.counters[1], fa.doneM_base + fa.nM, fa, !fa.tail_only)) return false;
  const int t = threadIdx.x, i = t >> 2, part = t & 3, g = I * 64 + i;
  double al = 0.0;
  for (int B2 = part; B2 < fa.nbk; B2 += 4)  // (the order of the sum for nbk <= 16 is what it was with the four unrolled steps)
    al += (B2 <= I) ? fa.apart[(int64_t)(I * FF_STRIDE + B2) * 128 + i] : fa.apart[(int64_t)(B2 * FF_STRIDE + I) * 128 + 64 + i];
  al += __shfl_xor(al, 1, 64);
  al += __shfl_xor(al, 2, 64);
  double v = 0.0, gm = 0.0;
  if (part == 0) {
    fa.alpha[g] = al;
    if (g < fa.n) {
      if (fa.criterion == BBH_CRITERION_MLL) {
        v = -0.5 * (fa.ystd[g] - sh.g.th[1]) * al;
        if (fa.tail_only) v -= log(fa.A[(int64_t)g * fa.np + g]);  // (the row heads of the one-launch form leave these sums in misc[2 + I])
        gm = al;
      } else {
        const double d = fa.M[(int64_t)g * fa.np + g];
        fa.u[g] = 0.5 / d + 0.5 * al * al / (d * d);
        fa.w[g] = al / d;
        v = 0.5 * log(d) - 0.5 * al * al / d;
      }
    } else if (fa.criterion != BBH_CRITERION_MLL) {
      fa.u[g] = 0.0;
      fa.w[g] = 0.0;
    }
  }
  v = ff_wave_sum(v);
  gm = ff_wave_sum(gm);
  if ((t & 63) == 0) {
    sh.red[t >> 6][0] = v;
    sh.red[t >> 6][1] = gm;
  }
  __syncthreads();
  if (t == 0) {
    fa.misc[FF_MISC_VALUE + I] = (sh.red[0][0] + sh.red[1][0]) + (sh.red[2][0] + sh.red[3][0]);
    fa.misc[FF_MISC_MEANG + I] = (sh.red[0][1] + sh.red[1][1]) + (sh.red[2][1] + sh.red[3][1]);
    if (fa.tail_only) fa.misc[FF_MISC_LOGDET + I] = 0.0;
  }
  pd_publish(&fa.flagsV[I], fa.epoch);
  if (t == 0) atomicAdd(&fa.counters[4], 1);
  return true;
}

// LOO: q_I = (M w)_I and its sum (the mean gradient)
__device__ __forceinline__ bool ff_role_qvec(const FlowArgs& fa, FlowShared& sh, int I) {
  if (!ff_wait_count(&fa.counters[4], fa.doneV_base + fa.nbk, fa, !fa.tail_only)) return false;
  const int t = threadIdx.x, row = t >> 2, part = t & 3;
  const double* mr = fa.M + (int64_t)(I * 64 + row) * fa.np;
  double acc = 0.0;
  for (int j = part; j < fa.np; j += 4) acc = fma(mr[j], fa.w[j], acc);
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  const int g = I * 64 + row;
  if (part == 0) {
    fa.q[g] = acc;
    sh.vr[row] = g < fa.n ? acc : 0.0;
  }
  __syncthreads();
  if (t < 64) {
    const double s = ff_wave_sum(sh.vr[t]);
    if (t == 0) fa.misc[FF_MISC_QSUM + I] = s;
  }
  pd_publish(&fa.flagsV[FF_STRIDE + I], fa.epoch);
  return true;
}

// LOO: Q_IJ = sum_K M_IK diag(u_K) M_KJ, I >= J (accumulators in registers, operands prefetched through registers)
__device__ __forceinline__ bool ff_role_qtile(const FlowArgs& fa, tile_t t0, tile_t t1, int I, int J) {
  if (!ff_wait_count(&fa.counters[4], fa.doneV_base + fa.nbk, fa, !fa.tail_only)) return false;
  d4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
  pd_d2 rb[8], rc[8];
  auto fetch = [&](int K) {
    ff_ld_regs(rb, fa.M + (int64_t)(I * 64) * fa.np + K * 64, fa.np);
    ff_ld_regs(rc, fa.M + (int64_t)(K * 64) * fa.np + J * 64, fa.np);
#pragma unroll
    for (int k = 0; k < 8; k++) {  // columns of M_IK scaled by u_K
      const int col = 2 * ((threadIdx.x + 256 * k) & 31);
      rb[k].x *= fa.u[K * 64 + col];
      rb[k].y *= fa.u[K * 64 + col + 1];
    }
  };
  fetch(0);
  for (int K = 0; K < fa.nbk; K++) {
    ff_st_regs(t0, rb);
    ff_st_regs(t1, rc);
    __syncthreads();
    if (K + 1 < fa.nbk) fetch(K + 1);
    ff_mma_nn(acc, t0, t1);
    __syncthreads();
  }
  ff_acc_to_tile(t0, acc);
  __syncthreads();
  pd_store_tile(fa.Q + (int64_t)(I * 64) * fa.np + J * 64, fa.np, t0, 1.0);
  pd_publish(&fa.flagsQ[I * FF_STRIDE + J], fa.epoch);
  return true;
}

// Gradient sums of a quarter (16 rows) of tile (I, J), I >= J, over its pairs (a in block I, b in block J):
//   G_ab (MLL: 0.5 (alpha_a alpha_b - M_ab); LOO: -Q_ab + 0.5 (alpha_a q_b + alpha_b q_a)) contributes
//   d/dl_j G g(r) scale Delta_j^2 / l_j^3,  d/dnoise G [a == b],  d/doutputscale G k B,  d/dB[ta][tb] G k outputscale,  d/dalpha.
// Every term but the task-covariance slot is symmetric in (a, b): an off-diagonal tile counts twice, and feeds B[ta][tb] and
// B[tb][ta].  Thread t: row i = 16 qd + (t >> 4) - the 16 lanes of a DPP row share a matrix row -, columns j = (t & 15) + 16 m.
// Pass 1: per pair the metric, the kernel value and the pair's weight Gg = w G g(r) os B, plus the scalar slots; pass 2: per
// dimension sum_m Gg_m Delta_c^2 in SCALED coordinates x / l (d/dl_c of (Delta_c / l_c)^2 is -2 (Delta_c / l_c)^2 / l_c: one factor
// 1 / l_c at the end).  A slot's 16 partials of a matrix row are added by DPP row shifts, the role's 16 rows through a small LDS
// table in row order.  The last role to finish adds all roles' rows in role order and writes the results.
// LDS: t0 = [rows of M / Q: 16 x PD_LD | table tl x 16], t1 = the two blocks' inputs [dn][64] x 2.
__device__ __forceinline__ bool ff_role_gtile(const FlowArgs& fa, FlowShared& sh, tile_t t0, tile_t t1, int I, int J, int qd, int gidx) {
  const int t = threadIdx.x;
  const bool loo = fa.criterion != BBH_CRITERION_MLL;
  double* xr = (double*)t1;
  double* xc = (double*)t1 + 2048;
  double* tab = (double*)t0 + 16 * PD_LD;  // [tl][16]
  const int kind = fa.ks.kind[0], dn = fa.dn, T = fa.T;
  // what does not depend on the other roles first: the two blocks' inputs (scaled), task ids, noise mask - their loads overlap the wait
  {
    double xv[2 * ((FF_MAXD * 64 + 255) / 256)];
#pragma unroll
    for (int k = 0; k < (FF_MAXD * 64 + 255) / 256; k++) {
      const int e = t + 256 * k;
      const bool in = e < dn * 64;
      xv[2 * k] = in ? fa.xnT[(int64_t)(e >> 6) * fa.np + I * 64 + (e & 63)] : 0.0;
      xv[2 * k + 1] = in ? fa.xnT[(int64_t)(e >> 6) * fa.np + J * 64 + (e & 63)] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < (FF_MAXD * 64 + 255) / 256; k++) {
      const int e = t + 256 * k;
      if (e < dn * 64) {
        const double il = sh.g.inv[e >> 6];
        xr[e] = xv[2 * k] * il;
        xc[e] = xv[2 * k + 1] * il;
      }
    }
  }
  ff_stage_meta(sh, fa, I, J);
  const bool lz = !fa.tail_only;
  if (!ff_wait(&fa.flagsV[I], fa, false, lz) || (I != J && !ff_wait(&fa.flagsV[J], fa, false, lz))) return false;
  if (loo && (!ff_wait(&fa.flagsQ[I * FF_STRIDE + J], fa, false, lz) || !ff_wait(&fa.flagsV[FF_STRIDE + I], fa, false, lz) || !ff_wait(&fa.flagsV[FF_STRIDE + J], fa, false, lz))) return false;
  if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 1] = wall_clock64();
  {
    const double* src = (loo ? fa.Q : fa.M) + (int64_t)(I * 64 + 16 * qd) * fa.np + J * 64;
    for (int e = t; e < 512; e += 256) *(pd_d2*)&t0[e >> 5][2 * (e & 31)] = *(const pd_d2*)(src + (int64_t)(e >> 5) * fa.np + 2 * (e & 31));
  }
  if (t < 64) {
    sh.vr[t] = fa.alpha[I * 64 + t];
    sh.vr2[t] = loo ? fa.q[I * 64 + t] : 0.0;
  } else if (t < 128) {
    sh.vc[t - 64] = fa.alpha[J * 64 + (t - 64)];
    sh.vc2[t - 64] = loo ? fa.q[J * 64 + (t - 64)] : 0.0;
  }
  __syncthreads();
  if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 4] = wall_clock64();
  const bool dot = BBH_KIND_IS_DOT(kind);
  const double os = fa.ks.use_os ? sh.g.th[2] : 1.0;
  const double kalpha = fa.ks.alpha_off >= 0 ? sh.g.th[fa.ks.alpha_off] : 1.0;
  const double wsym = (I == J) ? 1.0 : 2.0;
  const int il_ = t >> 4, i = 16 * qd + il_, part = t & 15, ga = I * 64 + i;
  double Gg[4];
  double g_noise = 0.0, g_os = 0.0, g_al = 0.0, g_B[FF_MAXT * FF_MAXT];
#pragma unroll
  for (int cc = 0; cc < FF_MAXT * FF_MAXT; cc++) g_B[cc] = 0.0;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int j = part + 16 * m, gb = J * 64 + j;
    Gg[m] = 0.0;
    if (ga < fa.n && gb < fa.n) {
      const double mij = t0[il_][j];
      const double G = loo ? -mij + 0.5 * (sh.vr[i] * sh.vc2[j] + sh.vc[j] * sh.vr2[i]) : 0.5 * (sh.vr[i] * sh.vc[j] - mij);
      double r2a = 0.0, r2b = 0.0;
      int cc = 0;
      for (; cc + 1 < dn; cc += 2) {
        const double xa0 = xr[cc * 64 + i], xb0 = xc[cc * 64 + j], xa1 = xr[(cc + 1) * 64 + i], xb1 = xc[(cc + 1) * 64 + j];
        if (dot) {
          r2a = fma(xa0, xb0, r2a);
          r2b = fma(xa1, xb1, r2b);
        } else {
          r2a = fma(xa0 - xb0, xa0 - xb0, r2a);
          r2b = fma(xa1 - xb1, xa1 - xb1, r2b);
        }
      }
      if (cc < dn) {
        const double xa0 = xr[cc * 64 + i], xb0 = xc[cc * 64 + j];
        r2a = dot ? fma(xa0, xb0, r2a) : fma(xa0 - xb0, xa0 - xb0, r2a);
      }
      const double r2 = r2a + r2b;
      double kb, gf;
      ff_kv_pair(kind, r2, fa.ks.jb, kalpha, kb, gf);
      const double Bab = (T > 1) ? sh.g.th[3 + dn + sh.g.tr[i] * T + sh.g.tc[j]] : 1.0;
      if (ga == gb) g_noise += G * sh.g.nm[i];
      g_os += wsym * G * kb * Bab;
      Gg[m] = wsym * G * gf * os * Bab;
      if (fa.ks.alpha_off >= 0) {
        if (kind == BBH_KERNEL_RQ) {
          const double uu = r2 / (2.0 * kalpha);
          g_al += wsym * G * os * Bab * kb * (uu / (1.0 + uu) - log1p(uu));
        } else if (BBH_KIND_IS_POLY(kind)) {
          const int pw = kind - BBH_KERNEL_POLY1 + 1;
          g_al += wsym * G * os * Bab * (double)pw * bbh_powi(r2 + kalpha, pw - 1);
        }
      }
      if (T > 1) {
        const double gk = G * kb * os;
        const int s1 = sh.g.tr[i] * T + sh.g.tc[j], s2 = sh.g.tc[j] * T + sh.g.tr[i];
#pragma unroll
        for (int q2 = 0; q2 < FF_MAXT * FF_MAXT; q2++) {
          g_B[q2] += (q2 == s1) ? gk : 0.0;
          if (I != J) g_B[q2] += (q2 == s2) ? gk : 0.0;
        }
      }
    }
  }
  if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 5] = wall_clock64();
  // a slot's partials of one matrix row: DPP row sum (lane 15 of the row holds it) -> tab[slot][row]
  auto put = [&](int slot, double v) {
    v = ff_row16_sum(v);
    if (part == 15) tab[slot * 16 + il_] = v;
  };
  put(0, g_noise);
  put(1, 0.0);
  put(2, fa.ks.use_os ? g_os : 0.0);
  if (fa.ks.alpha_off >= 0) put(fa.ks.alpha_off, g_al);
  if (T > 1) {
#pragma unroll
    for (int q2 = 0; q2 < FF_MAXT * FF_MAXT; q2++)
      if (q2 < T * T) put(3 + dn + q2, g_B[q2]);
  }
  for (int cc = 0; cc < dn; cc++) {
    const double xa = xr[cc * 64 + i];
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m += 2) {
      const double xb0 = xc[cc * 64 + part + 16 * m], xb1 = xc[cc * 64 + part + 16 * (m + 1)];
      acc0 = fma(Gg[m], dot ? xa * xb0 : (xa - xb0) * (xa - xb0), acc0);
      acc1 = fma(Gg[m + 1], dot ? xa * xb1 : (xa - xb1) * (xa - xb1), acc1);
    }
    put(3 + cc, (acc0 + acc1) * sh.g.inv[cc]);
  }
  __syncthreads();
  if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 6] = wall_clock64();
  double* row = fa.gpart + (int64_t)gidx * fa.tl;
  if (t < fa.tl) {
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < 16; e++) acc += tab[t * 16 + e];
    // write-through (sc1) store: the row is at the memory side once this wave's stores have drained - no release fence (buffer_wbl2 of
    // the whole L2) before the arrival count; the last arriver's acquire + plain loads read it (cdna guide, Guideline 16 R1)
    __hip_atomic_store(&row[t], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- last role: the sums over all roles' rows, in role order (eight row groups per slot, combined in group order) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every storing wave drains)
  __syncthreads();
  if (t == 0) {
    const int old = atomicAdd(&fa.counters[2], 1);
    sh.last = (old == fa.doneG_base + fa.nG - 1) ? 1 : 0;
    if (sh.last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 7] = wall_clock64();
  if (!sh.last) return true;
  {
    const int grp = t >> 5, sl = t & 31;  // 8 groups x 32 slots per sweep
    for (int s0 = 0; s0 < fa.tl; s0 += 32) {
      const int slot = s0 + sl;
      double acc = 0.0;
      if (slot < fa.tl)
        for (int g = grp; g < fa.nG; g += 8) acc += __builtin_nontemporal_load(&fa.gpart[(int64_t)g * fa.tl + slot]);
      __syncthreads();
      tab[grp * 32 + sl] = acc;
      __syncthreads();
      if (t < 32 && slot < fa.tl) {
        double tot = 0.0;
#pragma unroll
        for (int g2 = 0; g2 < 8; g2++) tot += tab[g2 * 32 + t];
        if (slot == 1) {  // constant mean: sum of alpha (MLL) / of q (LOO)
          tot = 0.0;
          if (loo)
            for (int B2 = 0; B2 < fa.nbk; B2++) tot += fa.misc[FF_MISC_QSUM + B2];
          else
            for (int B2 = 0; B2 < fa.nbk; B2++) tot += fa.misc[FF_MISC_MEANG + B2];
        }
        fa.out[1 + slot] = tot;
      }
    }
  }
  if (t == 0) {
    double v = 0.0;
    for (int B2 = 0; B2 < fa.nbk; B2++) v += fa.misc[FF_MISC_VALUE + B2];
    if (!loo)
      for (int B2 = 0; B2 < fa.nbk; B2++) v -= fa.misc[FF_MISC_LOGDET + B2];
    fa.out[0] = v - 0.5 * (double)fa.n * 1.8378770664093453;  // log(2 pi)
    const int inf = __atomic_load_n(fa.info, __ATOMIC_RELAXED);
    *fa.info_out = inf;
    __atomic_store_n(fa.info, 0, __ATOMIC_RELAXED);  // (left clean for the next launch)
  }
  return true;
}

}  // namespace

// TAIL: only the roles after the factorisation (MT, VEC, QV, QT, GT) - two tile buffers and at most 256 registers, so that two
// workgroups share a CU; the full form carries the row heads' four buffers and register budget.
// RS (role set): 0 everything; 1 the roles after the factorisation (MT ... GT: the tail of a tile-dataflow factorisation launch); 2 the
// factorisation and M = X^T X (RH, L, XT, MT); 3 what follows M (VEC, QV, QT, GT).  2 + 3 back to back = the split form: K^-1's tiles are
// built while the factorisation runs, and the heavy gradient roles - whose registers made the all-in-one kernel's row heads 3 us per
// step slower (502 VGPRs) - stay out of the first kernel.
template <int RS>
__device__ __forceinline__ void ff_kernel_body(const FlowArgs& fa) {
  constexpr bool FACT = RS == 0 || RS == 2, MTS = RS != 3, POST = RS != 2;
  extern __shared__ __attribute__((aligned(16))) double s_ff[];
  tile_t a = (tile_t)s_ff;
  tile_t b = (tile_t)(s_ff + 64 * PD_LD);
  tile_t c = (tile_t)(s_ff + 2 * 64 * PD_LD);
  tile_t al = (tile_t)(s_ff + 3 * 64 * PD_LD);
  __shared__ FlowShared sh;
  const int t = threadIdx.x;
  if (t < fa.tl) sh.g.th[t] = fa.theta ? fa.theta[t] : fa.thv[t < PD_GRAM_MAXTHV ? t : 0];
  __syncthreads();
  if (t < fa.dn) sh.g.inv[t] = 1.0 / sh.g.th[3 + t];
  __syncthreads();
  for (;;) {
    if (t == 0) sh.ticket = atomicAdd(&fa.counters[0], 1) - fa.ticket_base;
    __syncthreads();
    const int ticket = sh.ticket;
    if (ticket >= fa.nroles) return;
    if (fa.dbg && t == 0) {
      fa.dbg[8 * (fa.role_lo + sh.ticket) + 0] = wall_clock64();
      fa.dbg[8 * (fa.role_lo + sh.ticket) + 1] = 0;
      fa.dbg[8 * (fa.role_lo + sh.ticket) + 3] = blockIdx.x;
    }
    const int role = fa.roles[ticket];
    const int type = role & 15, I = (role >> 4) & 31, J = (role >> 9) & 31, qd = (role >> 14) & 3, gidx = (role >> 16) & 0xfff;
    bool ok = true;
    switch (type) {
      case FF_RH:
        if constexpr (FACT) ok = ff_role_rowhead(fa, sh, a, b, c, al, I); else ok = false;
        break;
      case FF_L:
        if constexpr (FACT) ok = ff_role_ltile(fa, sh, a, b, c, I, J); else ok = false;
        break;
      case FF_XT:
        if constexpr (FACT) ok = ff_role_xtile(fa, a, b, c, I, J); else ok = false;
        break;
      case FF_MT:
        if constexpr (MTS) ok = ff_role_mtile(fa, sh, a, b, I, J); else ok = false;
        break;
      case FF_VEC:
        if constexpr (POST) ok = ff_role_vec(fa, sh, I); else ok = false;
        break;
      case FF_QV:
        if constexpr (POST) ok = ff_role_qvec(fa, sh, I); else ok = false;
        break;
      case FF_QT:
        if constexpr (POST) ok = ff_role_qtile(fa, a, b, I, J); else ok = false;
        break;
      default:
        if constexpr (POST) ok = ff_role_gtile(fa, sh, a, b, I, J, qd, gidx); else ok = false;
        break;
    }
    if (!ok) return;
    if (fa.dbg && t == 0) fa.dbg[8 * (fa.role_lo + sh.ticket) + 2] = wall_clock64();
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bbh_fit_flow_kernel(const FlowArgs fa) { ff_kernel_body<0>(fa); }
__global__ __launch_bounds__(256, 2) void bbh_fit_tail_kernel(const FlowArgs fa) { ff_kernel_body<1>(fa); }
__global__ __launch_bounds__(256) void bbh_fit_factor_kernel(const FlowArgs fa) { ff_kernel_body<2>(fa); }
__global__ __launch_bounds__(256, 2) void bbh_fit_post_kernel(const FlowArgs fa) { ff_kernel_body<3>(fa); }
// The same roles without the 256-register cap (60 spilled VGPRs, 244 B of scratch per lane in the two-per-CU form): one workgroup per CU,
// for launches whose roles all find a slot that way (n <= 640: 8 ... 10 VEC roles + 4 G-roles per tile <= 256).
__global__ __launch_bounds__(256) void bbh_fit_post1_kernel(const FlowArgs fa) { ff_kernel_body<3>(fa); }
__global__ __launch_bounds__(256) void bbh_fit_tail1_kernel(const FlowArgs fa) { ff_kernel_body<1>(fa); }  // (default since round 6; BBH_FIT_TAIL1=0: the form above)

// ---- host side ---------------------------------------------------------------------------------------------------------
struct bbh_flow_state {
  int np = 0, criterion = -1;
  int nroles = 0, nM = 0, nG = 0, grid = 0;
  int* d_roles = nullptr;
  int* d_flags = nullptr;     // 4 x 256 tile flags + 32 vector flags + 4 counters
  double* d_apart = nullptr;  // [FF_FLAGS][128]
  double* d_gpart = nullptr;  // [nG][tl]
  double* d_misc = nullptr;   // [4][FF_STRIDE]
  long long* d_dbg = nullptr; // BBH_FLOW_TRACE=1
  int epoch = 0, ticket_base = 0, doneM_base = 0, doneG_base = 0, doneV_base = 0;
  bool failed = false, tail_only = false, split = false;
  int nA = 0, gridA = 0, gridB = 0;  // split form: roles / workgroups of the first launch (RH, L, XT, MT), workgroups of the second
};

void bbh_flow_destroy(bbh_handle* h) {
  bbh_flow_state* st = (bbh_flow_state*)h->flow_state;
  if (!st) return;
  if (st->d_roles) hipFree(st->d_roles);
  if (st->d_flags) hipFree(st->d_flags);
  if (st->d_apart) hipFree(st->d_apart);
  if (st->d_gpart) hipFree(st->d_gpart);
  if (st->d_misc) hipFree(st->d_misc);
  if (st->d_dbg) hipFree(st->d_dbg);
  delete st;
  h->flow_state = nullptr;
}

void bbh_flow_mark_failed(bbh_handle* h) {
  bbh_flow_state* st = (bbh_flow_state*)h->flow_state;
  if (st) st->failed = true;
  h->fit_flow = false;
}

// models the dataflow forms take (the launch itself can still fail on resources: then the handle stops using them)
bool bbh_fit_flow_eligible(bbh_handle* h) {
  const int64_t np = h->np;
  const bbh_flow_state* st = (const bbh_flow_state*)h->flow_state;
  return h->fit_flow && np > 64 && np <= 64 * FF_MAXBLK && h->F <= 1 && !h->hadamard && h->dn <= FF_MAXD && h->T <= FF_MAXT && bbh_theta_len(h) <= 49 &&
         h->desc.kernel_kind != BBH_KERNEL_PERIODIC && !h->fit_graph_mode && !(h->fit_stream && h->stream == h->fit_stream) && !(st && st->failed);
}

// true: the evaluation is on the stream (theta_dev / out_dev / info_dev are the device views of the pinned staging buffers)
// skip_mt (tail form): the first skip_mt tiles of K^-1 were built by the factorisation launch (bbh_fit_flow_mt_args) - the tail starts
// at role skip_mt (all of them: only the roles behind the M-tiles run, in the kernel without their code).
// prepare_only: create the state (roles, flags, counters) for this model and form, launch nothing.
bool bbh_fit_flow_launch(bbh_handle* h, const double* theta_dev, double* out_dev, int* info_dev, bool tail_only, const double* theta_host, bool split,
                         int skip_mt, bool prepare_only) {
  if (tail_only) split = false;
  if (!tail_only) skip_mt = 0;
  const int64_t np = h->np;
  const int nbk = (int)(np / 64);
  const int64_t tl = bbh_theta_len(h);
  if (!bbh_fit_flow_eligible(h) || nbk > FF_MAXBLK || (tail_only && nbk > 16)) return false;
  const size_t lds = sizeof(double) * (tail_only ? 2 : 4) * 64 * PD_LD;  // (the roles after the factorisation need two tile buffers: two workgroups per CU)
  bbh_flow_state* st = (bbh_flow_state*)h->flow_state;
  if (st && st->failed) return false;
  const bool loo = h->desc.criterion == BBH_CRITERION_LOO;
  if (!st || st->np != np || st->criterion != h->desc.criterion || st->tail_only != tail_only || st->split != split) {
    if (st) bbh_flow_destroy(h);
    st = new bbh_flow_state();
    h->flow_state = st;
    st->np = (int)np;
    st->criterion = h->desc.criterion;
    st->tail_only = tail_only;
    st->split = split;
    // roles in dependency order: column K of the factorisation (row head, its X-tiles, the L-tiles below it), then M, VEC, (QV, QT), G
    std::vector<int> roles;
    auto add = [&](int type, int I, int J, int qd = 0, int g = 0) { roles.push_back(type | (I << 4) | (J << 9) | (qd << 14) | (g << 16)); };
    for (int K = 0; K < nbk && !tail_only; K++) {
      add(FF_RH, K, K);
      for (int I = K + 2; I < nbk; I++) add(FF_L, I, K);
      for (int J = 0; J < K; J++) add(FF_XT, K, J);
    }
    if (tail_only) {
      // block row 0 first - the tiles with the most terms -, in the order the factorisation launch numbers the M-tiles it builds itself
      // (bbh_potrf_tiles_kernel: m = I (I + 1) / 2 + J): when only its first k are built there, the tail starts at role k
      for (int I = 0; I < nbk; I++)
        for (int J = 0; J <= I; J++) add(FF_MT, I, J);
    } else {
      for (int I = nbk - 1; I >= 0; I--)  // (the tiles of the last block rows have the fewest terms and can finish first)
        for (int J = 0; J <= I; J++) add(FF_MT, I, J);
    }
    st->nM = nbk * (nbk + 1) / 2;
    st->nA = (int)roles.size();  // (split form: the first launch ends here)
    for (int I = 0; I < nbk; I++) add(FF_VEC, I, 0);
    if (loo) {
      for (int I = 0; I < nbk; I++) add(FF_QV, I, 0);
      for (int I = 0; I < nbk; I++)
        for (int J = 0; J <= I; J++) add(FF_QT, I, J);
    }
    int g = 0;
    for (int I = 0; I < nbk; I++)
      for (int J = 0; J <= I; J++)
        for (int qd = 0; qd < 4; qd++) add(FF_GT, I, J, qd, g++);  // four roles per tile: 16 rows x 64 columns each
    st->nG = g;
    st->nroles = (int)roles.size();
    int per_cu = 0, per_cu_b = 0;
    const void* kfn = tail_only ? (const void*)bbh_fit_tail_kernel : (split ? (const void*)bbh_fit_factor_kernel : (const void*)bbh_fit_flow_kernel);
    const size_t lds_b = sizeof(double) * 2 * 64 * PD_LD;
    bool ok = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
              hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, lds) == hipSuccess && per_cu >= 1 &&
              (!split || (hipFuncSetAttribute((const void*)bbh_fit_post_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b) == hipSuccess &&
                          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, (const void*)bbh_fit_post_kernel, 256, lds_b) == hipSuccess && per_cu_b >= 1)) &&
              hipMalloc((void**)&st->d_roles, sizeof(int) * roles.size()) == hipSuccess &&
              hipMemcpy(st->d_roles, roles.data(), sizeof(int) * roles.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMalloc((void**)&st->d_flags, sizeof(int) * (FF_COUNTERS + 8)) == hipSuccess &&
              hipMemset(st->d_flags, 0, sizeof(int) * (FF_COUNTERS + 8)) == hipSuccess &&
              hipMalloc((void**)&st->d_apart, sizeof(double) * FF_FLAGS * 128) == hipSuccess &&
              hipMalloc((void**)&st->d_gpart, sizeof(double) * (size_t)st->nG * 64) == hipSuccess /* (nG counts the quarter roles) */ &&
              hipMalloc((void**)&st->d_misc, sizeof(double) * 4 * FF_STRIDE) == hipSuccess &&
              // (tail mode: the factorisation already on the stream owns the flag - clearing it here would erase its verdict)
              (tail_only || hipMemset(h->d_info, 0, sizeof(int)) == hipSuccess);
    if (!ok) {
      (void)hipGetLastError();
      st->failed = true;
      return false;
    }
    const int slots = per_cu * h->num_cu;
    st->grid = st->nroles < slots ? st->nroles : slots;
    if (split) {
      const int nB = st->nroles - st->nA, slots_b = per_cu_b * h->num_cu;
      st->gridA = st->nA < slots ? st->nA : slots;
      st->gridB = nB < slots_b ? nB : slots_b;
    }
  }
  if (prepare_only) return true;
  if (st->ticket_base > (1 << 30)) {  // cumulative counters: start over long before they wrap
    if (hipStreamSynchronize(h->stream) != hipSuccess || hipMemset(st->d_flags + FF_COUNTERS, 0, sizeof(int) * 8) != hipSuccess) {
      st->failed = true;
      return false;
    }
    st->ticket_base = st->doneM_base = st->doneG_base = st->doneV_base = 0;
  }
  FlowArgs fa{};
  fa.xnT = h->d_xnT;
  fa.task = h->d_task;
  fa.nmask = h->d_nmask;
  fa.ystd = h->d_ystd;
  fa.theta = theta_dev;
  if (!theta_dev) {
    if (!theta_host || tl > PD_GRAM_MAXTHV) return false;
    for (int64_t k = 0; k < tl; k++) fa.thv[k] = theta_host[k];
  }
  fa.n = (int)h->n;
  fa.np = (int)np;
  fa.nbk = nbk;
  fa.dn = h->dn;
  fa.T = h->T;
  fa.tl = (int)tl;
  fa.criterion = h->desc.criterion;
  fa.ks = bbh_kern_spec_of(h);
  fa.A = h->d_K;
  fa.D = h->d_D;
  fa.X = h->d_X;
  fa.M = h->d_M;
  fa.Q = h->d_Q;
  fa.alpha = h->d_alpha;
  fa.u = h->d_u;
  fa.w = h->d_w;
  fa.q = h->d_q;
  fa.apart = st->d_apart;
  fa.gpart = st->d_gpart;
  fa.misc = st->d_misc;
  fa.flagsL = st->d_flags;
  fa.flagsX = st->d_flags + FF_FLAGS;
  fa.flagsM = st->d_flags + 2 * FF_FLAGS;
  fa.flagsQ = st->d_flags + 3 * FF_FLAGS;
  fa.flagsV = st->d_flags + 4 * FF_FLAGS;
  fa.counters = st->d_flags + FF_COUNTERS;
  fa.ticket_base = st->ticket_base;
  fa.doneM_base = st->doneM_base;
  fa.doneG_base = st->doneG_base;
  fa.doneV_base = st->doneV_base;
  fa.roles = st->d_roles;
  fa.nroles = st->nroles;
  fa.nM = st->nM;
  fa.nG = st->nG;
  fa.epoch = ++st->epoch;
  fa.spin = h->flow_spin_limit;
  fa.tail_only = tail_only ? 1 : 0;
  fa.info = h->d_info;
  fa.abort = tail_only ? st->d_flags + FF_COUNTERS + 3 : h->d_info;
  fa.out = out_dev;
  fa.info_out = info_dev;
  if (getenv("BBH_FLOW_TRACE")) {
    if (!st->d_dbg && hipMalloc((void**)&st->d_dbg, sizeof(long long) * 8 * 1024) != hipSuccess) st->d_dbg = nullptr;
    fa.dbg = st->d_dbg;
  }
  if (split) {
    // first launch: the factorisation and K^-1's tiles (tickets 0 .. nA); second: VEC ... GT, its own ticket range and its own abort word
    FlowArgs fb = fa;
    fa.nroles = st->nA;
    hipLaunchKernelGGL(bbh_fit_factor_kernel, dim3((unsigned)st->gridA), dim3(256), lds, h->stream, fa);
    fb.roles = st->d_roles + st->nA;
    fb.role_lo = st->nA;
    fb.nroles = st->nroles - st->nA;
    fb.ticket_base = st->ticket_base + st->nA + st->gridA;
    fb.tail_only = 1;
    fb.abort = st->d_flags + FF_COUNTERS + 3;
    hipLaunchKernelGGL(bbh_fit_post_kernel, dim3((unsigned)st->gridB), dim3(256), sizeof(double) * 2 * 64 * PD_LD, h->stream, fb);
    if (hipGetLastError() != hipSuccess) {
      st->failed = true;
      return false;
    }
    st->ticket_base += st->nroles + st->gridA + st->gridB;
    st->doneM_base += st->nM;
    st->doneG_base += st->nG;
    st->doneV_base += nbk;
    return true;
  }
  if (skip_mt > 0 && skip_mt < st->nM) {  // some of K^-1's tiles are done: the tail's table starts behind them
    const int nB = st->nroles - skip_mt, slots = st->grid;  // (st->grid = min(nroles, slots))
    const int grid = nB < slots ? nB : slots;
    fa.roles = st->d_roles + skip_mt;
    fa.role_lo = skip_mt;
    fa.nroles = nB;
    hipLaunchKernelGGL(bbh_fit_tail_kernel, dim3((unsigned)grid), dim3(256), lds, h->stream, fa);
    if (hipGetLastError() != hipSuccess) {
      st->failed = true;
      return false;
    }
    st->ticket_base += nB + grid;
    st->doneM_base += st->nM;
    st->doneG_base += st->nG;
    st->doneV_base += nbk;
    return true;
  }
  if (skip_mt >= st->nM) {  // the table starts with the nM M-tile roles: begin behind them (their count was added by the factorisation's workgroups)
    int per_cu_b = 0;
    const int nB = st->nroles - st->nA;
    static const bool post1_ok = !(getenv("BBH_FIT_POST1") && getenv("BBH_FIT_POST1")[0] == '0');  // (A/B)
    const bool one_per_cu = post1_ok && nB <= h->num_cu;
    const void* pk = one_per_cu ? (const void*)bbh_fit_post1_kernel : (const void*)bbh_fit_post_kernel;
    if (!st->gridB) {
      if (hipFuncSetAttribute(pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, pk, 256, lds) != hipSuccess || per_cu_b < 1) {
        (void)hipGetLastError();
        return false;
      }
      st->gridB = nB < per_cu_b * h->num_cu ? nB : per_cu_b * h->num_cu;
    }
    fa.roles = st->d_roles + st->nA;
    fa.role_lo = st->nA;
    fa.nroles = nB;
    if (one_per_cu)
      hipLaunchKernelGGL(bbh_fit_post1_kernel, dim3((unsigned)st->gridB), dim3(256), lds, h->stream, fa);
    else
      hipLaunchKernelGGL(bbh_fit_post_kernel, dim3((unsigned)st->gridB), dim3(256), lds, h->stream, fa);
    if (hipGetLastError() != hipSuccess) {
      st->failed = true;
      return false;
    }
    st->ticket_base += nB + st->gridB;
    st->doneM_base += st->nM;
    st->doneG_base += st->nG;
    st->doneV_base += nbk;
    return true;
  }
  // one workgroup per CU without the 256-register cap (the two-per-CU form spills 99 VGPRs: 400 B of scratch per lane); measured at
  // n = 1024: 464 -> 450 us per evaluation, ICM / LOO 552 -> 537 us although its 848 roles then share 256 slots instead of 512
  // (profiles/r06_fit_eval_post1_tail1.log).  BBH_FIT_TAIL1=0: the two-per-CU form (A/B)
  static const bool tail1 = !(getenv("BBH_FIT_TAIL1") && getenv("BBH_FIT_TAIL1")[0] == '0');
  if (tail_only && tail1) {
    static bool attr1 = false;
    if (!attr1) {
      hipFuncSetAttribute((const void*)bbh_fit_tail1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr1 = true;
    }
    const int g1 = st->nroles < h->num_cu ? st->nroles : h->num_cu;
    hipLaunchKernelGGL(bbh_fit_tail1_kernel, dim3((unsigned)g1), dim3(256), lds, h->stream, fa);
    if (hipGetLastError() != hipSuccess) {
      st->failed = true;
      return false;
    }
    st->ticket_base += st->nroles + g1;
    st->doneM_base += st->nM;
    st->doneG_base += st->nG;
    st->doneV_base += nbk;
    return true;
  }
  if (tail_only)
    hipLaunchKernelGGL(bbh_fit_tail_kernel, dim3((unsigned)st->grid), dim3(256), lds, h->stream, fa);
  else
    hipLaunchKernelGGL(bbh_fit_flow_kernel, dim3((unsigned)st->grid), dim3(256), lds, h->stream, fa);
  if (hipGetLastError() != hipSuccess) {
    st->failed = true;
    return false;
  }
  st->ticket_base += st->nroles + st->grid;
  st->doneM_base += st->nM;
  st->doneG_base += st->nG;
  st->doneV_base += nbk;
  return true;
}

// after a launch that gave up (or never reported): counters and flags back to a clean state; the handle stops using the form
void bbh_fit_flow_reset(bbh_handle* h) {
  bbh_flow_state* st = (bbh_flow_state*)h->flow_state;
  if (!st) return;
  hipStreamSynchronize(h->stream);
  if (st->d_flags) hipMemset(st->d_flags, 0, sizeof(int) * (FF_COUNTERS + 8));
  hipMemset(h->d_info, 0, sizeof(int));
  st->ticket_base = st->doneM_base = st->doneG_base = st->doneV_base = 0;
  st->failed = true;
  h->fit_flow = false;
}

// The M-tile arguments of the NEXT tail launch of this model (state created if need be): what the factorisation launch needs to build
// K^-1's tiles itself (bbh_potrf_trtri_from_inputs).  out: pd_mt_args.
bool bbh_fit_flow_mt_args(bbh_handle* h, void* out) {
  if (!bbh_fit_flow_launch(h, nullptr, nullptr, nullptr, true, nullptr, false, 0, true)) return false;
  bbh_flow_state* st = (bbh_flow_state*)h->flow_state;
  if (!st || st->failed || !st->tail_only) return false;
  pd_mt_args* ma = (pd_mt_args*)out;
  ma->nM = st->nM;
  ma->M = h->d_M;
  ma->ld = h->np;
  ma->apart = st->d_apart;
  ma->ystd = h->d_ystd;
  ma->theta = nullptr;
  ma->cmean = 0.0;  // (the Gram-building launch has theta in LDS)
  ma->n = (int)h->n;
  ma->loo = h->desc.criterion == BBH_CRITERION_LOO;
  ma->flagsM = st->d_flags + 2 * FF_FLAGS;
  ma->doneM = st->d_flags + FF_COUNTERS + 1;
  ma->flow_epoch = st->epoch + 1;
  return true;
}

// BBH_FLOW_TRACE=1: the last launch's per-role clock stamps [nroles][4] (start, after the last wait of a row head, end, workgroup) and the role table
extern "C" int bbh_flow_trace_read(bbh_handle* h, long long* stamps_host, int* roles_host, int cap) {
  bbh_flow_state* st = h ? (bbh_flow_state*)h->flow_state : nullptr;
  if (!st || !st->d_dbg || !stamps_host || !roles_host || cap < st->nroles) return -1;
  hipStreamSynchronize(h->stream);
  hipMemcpy(stamps_host, st->d_dbg, sizeof(long long) * 8 * st->nroles, hipMemcpyDeviceToHost);
  hipMemcpy(roles_host, st->d_roles, sizeof(int) * st->nroles, hipMemcpyDeviceToHost);
  return st->nroles;
}
