// Dense fp64 linear algebra for the GP fit / factorisation (n <= a few thousand):
//   * bbh_gemm         — LDS-tiled GEMM on v_mfma_f64_16x16x4_f64 (64x64x16 tiles, 4 waves)
//   * bbh_potrf_trtri  — blocked right-looking Cholesky (64-wide panels) + blocked inverse
//                        of the triangular factor, both built from bbh_gemm and one
//                        single-workgroup 64x64 kernel
//   * matvecs
// These replace what linear_operator/LAPACK do under gpytorch's ExactMarginalLogLikelihood
// (reference call site baybe/surrogates/gaussian_process/core.py:340-341).  All matrices are
// padded to multiples of 64 with an identity block, so no kernel needs edge handling.
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "bbh_common.h"
#include "bbh_tiles.h"

#define GBK 16
#define GLD 80  // LDS row pitch (doubles): 160 dwords == 32 mod 64 -> the two k-rows a
                // 32-lane ds_read_b64 group touches land on disjoint bank halves

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bbh_gemm_kernel(int64_t K, double alpha, const double* A,
                                                       int64_t lda, int64_t sA, const double* B,
                                                       int64_t ldb, int64_t sB, double beta, double* C,
                                                       int64_t ldc, int64_t sC) {
  __shared__ double As[GBK * GLD];
  __shared__ double Bs[GBK * GLD];
  const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  A += (int64_t)blockIdx.z * sA;
  B += (int64_t)blockIdx.z * sB;
  C += (int64_t)blockIdx.z * sC;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  for (int64_t k0 = 0; k0 < K; k0 += GBK) {
    // ---- stage the A tile as As[k][m] ----
    if (!TA) {  // A stored [M][K]: 4 consecutive k per thread, transposing store
      const int m = t >> 2, k4 = (t & 3) * 4;
      const d4 v = *(const d4*)(A + (m0 + m) * lda + k0 + k4);
#pragma unroll
      for (int i = 0; i < 4; i++) As[(k4 + i) * GLD + m] = v[i];
    } else {  // A stored [K][M]: 4 consecutive m per thread
      const int k = t >> 4, m4 = (t & 15) * 4;
      const d4 v = *(const d4*)(A + (k0 + k) * lda + m0 + m4);
#pragma unroll
      for (int i = 0; i < 4; i++) As[k * GLD + m4 + i] = v[i];
    }
    // ---- stage the B tile as Bs[k][n] ----
    if (!TB) {  // B stored [K][N]
      const int k = t >> 4, n4 = (t & 15) * 4;
      const d4 v = *(const d4*)(B + (k0 + k) * ldb + n0 + n4);
#pragma unroll
      for (int i = 0; i < 4; i++) Bs[k * GLD + n4 + i] = v[i];
    } else {  // B stored [N][K]
      const int n = t >> 2, k4 = (t & 3) * 4;
      const d4 v = *(const d4*)(B + (n0 + n) * ldb + k0 + k4);
#pragma unroll
      for (int i = 0; i < 4; i++) Bs[(k4 + i) * GLD + n] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GBK / 4; kk++) {
      const int krow = (kk * 4 + (l >> 4)) * GLD;
      const double a0 = As[krow + wm * 32 + (l & 15)];
      const double a1 = As[krow + wm * 32 + 16 + (l & 15)];
      const double b0 = Bs[krow + wn * 32 + (l & 15)];
      const double b1 = Bs[krow + wn * 32 + 16 + (l & 15)];
      acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
      acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
      acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
      acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
    }
    __syncthreads();  // also orders every global read of this tile before the epilogue (C may alias A)
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t row = m0 + wm * 32 + i * 16 + (l >> 4) + 4 * r;
        const int64_t col = n0 + wn * 32 + j * 16 + (l & 15);
        double* c = C + row * ldc + col;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * (*c);
        *c = v;
      }
}

void bbh_gemm(hipStream_t s, bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
              const double* A, int64_t lda, int64_t strideA, const double* B, int64_t ldb, int64_t strideB,
              double beta, double* C, int64_t ldc, int64_t strideC, int batch) {
  if (M <= 0 || N <= 0 || batch <= 0) return;
  dim3 grid((unsigned)(N / 64), (unsigned)(M / 64), (unsigned)batch), block(256);
  if (!transA && !transB)
    hipLaunchKernelGGL((bbh_gemm_kernel<false, false>), grid, block, 0, s, K, alpha, A, lda, strideA, B, ldb,
                       strideB, beta, C, ldc, strideC);
  else if (!transA && transB)
    hipLaunchKernelGGL((bbh_gemm_kernel<false, true>), grid, block, 0, s, K, alpha, A, lda, strideA, B, ldb,
                       strideB, beta, C, ldc, strideC);
  else if (transA && !transB)
    hipLaunchKernelGGL((bbh_gemm_kernel<true, false>), grid, block, 0, s, K, alpha, A, lda, strideA, B, ldb,
                       strideB, beta, C, ldc, strideC);
  else
    hipLaunchKernelGGL((bbh_gemm_kernel<true, true>), grid, block, 0, s, K, alpha, A, lda, strideA, B, ldb,
                       strideB, beta, C, ldc, strideC);
}

// ---- 64x64 diagonal block: Cholesky factor and its inverse, one workgroup ----------------
// In: the (already updated) diagonal block J of A.  Out: L_JJ in place (upper part zeroed),
// its inverse into D[J] and into the diagonal block of X.  info := 64 J + j + 1 at the first
// non-positive pivot (the factor then contains NaNs; the host retries with jitter).
//
// The factorisation is latency-bound (one block, 64 dependent pivots), so it runs out of registers
// in a single wavefront instead of through LDS with three workgroup barriers per column: lane i
// holds row i of the block (64 doubles); pivot and column entries travel by v_readlane, every index
// is a compile-time constant (both loops fully unrolled: 2016 readlane pairs + FMAs).  The inverse
// is a forward substitution per lane (lane c = column c of L^-1) with L read from LDS as broadcasts.
// 92 us -> 50 us per block; this kernel was 53 % of a fit evaluation at n = 512.
__device__ __forceinline__ double bbh_readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void bbh_potrf_diag_kernel(double* A, int64_t lda, int64_t J, double* D,
                                                             double* X, int64_t ldx, int* info) {
  __shared__ double a[64][65];
  __shared__ double x[64][65];
  const int t = threadIdx.x;
  double* Ajj = A + (J * 64) * lda + J * 64;
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    a[i][j] = Ajj[(int64_t)i * lda + j];
  }
  __syncthreads();
  __shared__ double rdiag[64];
  if (t < 64) {  // wave 0
    double row[64];
#pragma unroll
    for (int k = 0; k < 64; k++) row[k] = a[t][k];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) {
      const double djj = bbh_readlane_f64(row[j], j);  // wave-uniform
      bad = (!(djj > 0.0) && bad == 0) ? j + 1 : bad;
      // 1/sqrt(d) from the rsq seed with two Newton steps, sqrt(d) = d * rs: the pivot chain
      // (broadcast -> sqrt -> reciprocal -> scale) is serial, library sqrt + division cost ~300 cycles of it
      double rs = __builtin_amdgcn_rsq(djj);
      rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
      rs = fma(fma(-djj * rs, rs, 1.0), 0.5 * rs, rs);
      const double sj = djj * rs;
      row[j] = (t == j) ? sj : row[j] * rs;  // l_ij for i > j (rows above the pivot are never read)
      if (t == j) rdiag[j] = rs;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = j + 1; k < 64; k++) {
        const double lkj = bbh_readlane_f64(row[j], k);
        row[k] = fma(-row[j], lkj, row[k]);  // meaningful for i >= k
        // pin the update next to its broadcast: otherwise the compiler sinks the FMAs towards the pivot
        // that needs row[k] and keeps every broadcast value alive in SGPRs (3000 SGPR spills)
        asm volatile("" : "+v"(row[k]));
        if (((k - j) & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t == 0 && bad) atomicCAS(info, 0, (int)(J * 64 + bad));
#pragma unroll
    for (int k = 0; k < 64; k++) a[t][k] = (k <= t) ? row[k] : 0.0;
  }
  __syncthreads();
  if (t < 64) {  // column t of L^-1 by forward substitution, L broadcast from LDS
    // (an explicit 16-coefficient prefetch with fences ran slower than the compiler's own schedule:
    //  82 vs 50 us for the kernel; the next step is a 16 x 16-blocked form on MFMA)
    double xc[64];
#pragma unroll
    for (int r = 0; r < 64; r++) {
      double acc = (r == t) ? 1.0 : 0.0, acc1 = 0.0;  // two chains: the FMA latency is the critical path
#pragma unroll
      for (int k = 0; k + 1 < r; k += 2) {
        acc = fma(-a[r][k], xc[k], acc);  // xc[k] == 0 for k < t
        acc1 = fma(-a[r][k + 1], xc[k + 1], acc1);
      }
      if (r & 1) acc = fma(-a[r][r - 1], xc[r - 1], acc);
      xc[r] = (r >= t) ? (acc + acc1) * rdiag[r] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 64; r++) x[r][t] = xc[r];
  }
  __syncthreads();
  double* Xjj = X + (J * 64) * ldx + J * 64;
  double* Dj = D + J * 4096;
  for (int e = t; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ajj[(int64_t)r * lda + c] = a[r][c];
    const double xv = x[r][c];
    Dj[e] = xv;
    Xjj[(int64_t)r * ldx + c] = xv;
  }
}

__global__ __launch_bounds__(256) void bbh_potrf_diag16_kernel(double* A, int64_t lda, int64_t J, double* D, double* X,
                                                               int64_t ldx, int* info) {
  __shared__ double a[64][PD_LD];  // the block: A -> L (lower), upper part zeroed at the end
  __shared__ double x[64][PD_LD];  // L^-1 (lower)
  __shared__ double s[64][PD_LD];  // products awaiting a second multiplication (inverse assembly)
  const int t = threadIdx.x;
  double* Ajj = A + (J * 64) * lda + J * 64;
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    a[i][j] = Ajj[(int64_t)i * lda + j];
    x[i][j] = 0.0;
  }
  __syncthreads();
  pd_factor_block4(a, x, s, J * 64, info);
  double* Xjj = X + (J * 64) * ldx + J * 64;
  double* Dj = D + J * 4096;
  for (int e = t; e < 4096; e += 256) {
    const int r = e >> 6, c = e & 63;
    Ajj[(int64_t)r * lda + c] = a[r][c];
    const double xv = x[r][c];
    Dj[e] = xv;
    Xjj[(int64_t)r * ldx + c] = xv;
  }
}

__device__ int pd_spin_limit;  // polls before a waiting workgroup gives up (set per launch)
__device__ __forceinline__ bool pd_wait(const int* flag, int epoch, int* info, bool urgent = false, bool acquire = true, long long* polls_out = nullptr) {
  return pd_wait_n(flag, epoch, info, pd_spin_limit, urgent, acquire, polls_out);
}
#define PD_FLAG_STRIDE 32  // ints between two flags: one 128-byte line each (polls of different flags go to different lines / channels)

// GRAM: the tiles of K + s2 M are produced here from the training inputs (fit evaluations, pd_gram_tile) instead of being read from A.
// WT: tiles other workgroups wait for are stored write-through and published without a release fence (pd_publish_wt).
template <bool GRAM, bool WT>
__global__ __launch_bounds__(256) void bbh_potrf_tiles_kernel(double* A, int64_t lda, int nbk, double* D, double* X, int64_t ldx,
                                                              int* flagsL, int* flagsX, int epoch, int* info, const pd_gram_src gs,
                                                              const pd_mt_args ma) {
  auto store_pub = [&](double* dst, int64_t ld, const double (*src)[PD_LD]) {
    if (WT)
      pd_store_tile_wt(dst, ld, src, 1.0);
    else
      pd_store_tile(dst, ld, src, 1.0);
  };
  auto publish = [&](int* flag) {
    if (WT)
      pd_publish_wt(flag, epoch);
    else
      pd_publish(flag, epoch);
  };
  __shared__ pd_gram_lds gl;
  auto stamp = [&](int k) {
    if (GRAM && gs.dbg && threadIdx.x == 0) gs.dbg[8 * blockIdx.x + k] = wall_clock64();
  };
  stamp(0);
  if (GRAM) pd_gram_init(gl, gs);
  extern __shared__ __attribute__((aligned(16))) double s_tiles[];
  double(*a)[PD_LD] = (double(*)[PD_LD])s_tiles;
  double(*b)[PD_LD] = (double(*)[PD_LD])(s_tiles + 64 * PD_LD);
  double(*c)[PD_LD] = (double(*)[PD_LD])(s_tiles + 2 * 64 * PD_LD);
  double(*al)[PD_LD] = (double(*)[PD_LD])(s_tiles + 3 * 64 * PD_LD);  // row heads: the tile left of the diagonal one
  const int nOther = (nbk - 1) * (nbk - 2) / 2;
  int id = blockIdx.x;
  {  // ---- M-tile (I, J) of K^-1 = X^T X: the workgroups behind the factorisation's own (fit evaluations, ma.nM > 0)
    const int nfact = nbk + nOther + nbk * (nbk - 1) / 2;
    if (id >= nfact) {
      const int m = id - nfact;
      int I = 0;
      while ((I + 1) * (I + 2) / 2 <= m) I++;
      const int J = m - I * (I + 1) / 2;
      double* vr = (double*)c;
      auto wait = [&](int K) -> bool {  // (lazy polls: these workgroups wait from the first microsecond on)
        if (!pd_wait_n(K == I ? &flagsL[(I * nbk + I) * PD_FLAG_STRIDE] : &flagsX[(K * nbk + I) * PD_FLAG_STRIDE], epoch, info, pd_spin_limit, false, true,
                       nullptr, true))
          return false;
        return I == J || pd_wait_n(&flagsX[(K * nbk + J) * PD_FLAG_STRIDE], epoch, info, pd_spin_limit, false, true, nullptr, true);
      };
      const double cmean = GRAM ? gl.th[1] : (ma.theta ? ma.theta[1] : ma.cmean);
      // (a wait that gave up - info = -7, the launch is lost -: still count the tile, so that the launch behind this one runs through
      // (on garbage) and hands the flag to the host, which then repeats the evaluation on the per-step path)
      (void)pd_mtile_core(a, b, vr, vr + 64, I, J, nbk, D, X, ldx, ma, cmean, wait);
      __syncthreads();  // (consumers run in the next launch: the kernel boundary publishes the tile; the flag and the count only say "done")
      if (threadIdx.x == 0) {
        __hip_atomic_store(&ma.flagsM[I * PD_MT_STRIDE + J], ma.flow_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(ma.doneM, 1);
      }
      return;
    }
  }
  if (id < nbk) {
    // ---- row head I: the diagonal tile (I, I) AND its left neighbour (I, I-1) in one workgroup, so that the critical
    // chain D_{I-1} -> L_{I,I-1} -> update of (I, I) -> factor -> D_I never leaves LDS in between
    const int I = id;
    double* Aii = A + (int64_t)(I * 64) * lda + I * 64;
    double* Ail = Aii - 64;
    if (GRAM) {
      pd_gram_stage((double*)b, gs, I);
      if (I > 0) pd_gram_stage((double*)c, gs, I - 1);
      pd_gram_meta(gl, gs, I, I);
      __syncthreads();
      pd_gram_tile(a, (const double*)b, (const double*)b, I, I, gs, gl);
      if (I > 0) {
        __syncthreads();
        pd_gram_meta(gl, gs, I, I - 1);
        __syncthreads();
        pd_gram_tile(al, (const double*)b, (const double*)c, I, I - 1, gs, gl);
      }
    } else {
      pd_load_tile(a, Aii, lda);
      if (I > 0) pd_load_tile(al, Ail, lda);
    }
    __syncthreads();
    stamp(1);
    for (int J = 0; J + 1 < I; J++) {
      // L_{I,J} (an L-tile's product, early) first and the diagonal tile's update with it; then L_{I-1,J} - for J = I - 2 the tile the
      // previous row head publishes last - and the left neighbour's update.  (Waiting for both before either update put one product
      // and one tile load behind the later flag: the row heads reached their D wait 3.7 us after D had been published.)
      if (!pd_wait(&flagsL[(I * nbk + J) * PD_FLAG_STRIDE], epoch, info)) return;
      pd_load_tile(b, A + (int64_t)(I * 64) * lda + J * 64, lda);
      __syncthreads();
      pd_gemm64<true, true, PD_OUT_LOWER>(a, b, b, -1.0);
      if (!pd_wait(&flagsL[((I - 1) * nbk + J) * PD_FLAG_STRIDE], epoch, info)) return;
      pd_load_tile(c, A + (int64_t)((I - 1) * 64) * lda + J * 64, lda);
      __syncthreads();
      pd_gemm64<true, true, PD_FULL>(al, b, c, -1.0);
      __syncthreads();
    }
    if (I > 0) {
      // (write-through form: D_{I-1} through sc1 loads, no acquire fence on this critical hand-off; BBH_TILE_ACQ=1 keeps the fence)
      const bool noacq = WT && gs.d_sc1 != 0;
      if (!pd_wait(&flagsL[((I - 1) * nbk + (I - 1)) * PD_FLAG_STRIDE], epoch, info, true, !noacq, (GRAM && gs.dbg) ? &gs.dbg[8 * blockIdx.x + 1] : nullptr)) return;
      stamp(2);
      if (noacq)
        pd_load_tile_sc1(b, D + (int64_t)(I - 1) * 4096);
      else
        pd_load_tile(b, D + (int64_t)(I - 1) * 4096, 64);
      __syncthreads();
      stamp(3);
      pd_gemm64<true, false, PD_B_LOWER>(c, al, b, 1.0);  // L_{I,I-1} = A_{I,I-1} D_{I-1}^T
      __syncthreads();
      stamp(4);
      store_pub(Ail, lda, c);
      pd_gemm64<true, true, PD_OUT_LOWER>(a, c, c, -1.0);
      // L_{I,I-1} is published HERE, before the factorisation (its stores were issued a product ago): the next row head's last update
      // waits for it, and from inside the factorisation it came 3 us later than it had to
      publish(&flagsL[(I * nbk + (I - 1)) * PD_FLAG_STRIDE]);  // (contains the barrier the product needs)
    }
    stamp(5);
    pd_factor_block4(a, b, al, (int64_t)I * 64, info, nullptr, epoch, WT);
    stamp(6);
    store_pub(D + (int64_t)I * 4096, 64, b);
    publish(&flagsL[(I * nbk + I) * PD_FLAG_STRIDE]);  // D_I first: the next row head waits for it
    stamp(7);
    pd_store_tile(Aii, lda, a, 1.0);
    pd_store_tile(X + (int64_t)(I * 64) * ldx + I * 64, ldx, b, 1.0);
    return;
  }
  id -= nbk;
  if (id < nOther) {  // ---- L-tile (I, K), K <= I - 2
    int I = 2;
    while ((I - 1) * I / 2 <= id) I++;   // tiles of the rows 2 .. I-1 number (I-1)(I-2)/2
    const int K = id - (I - 1) * (I - 2) / 2;
    double* Aik = A + (int64_t)(I * 64) * lda + K * 64;
    if (GRAM) {
      pd_gram_stage((double*)b, gs, I);
      pd_gram_stage((double*)c, gs, K);
      pd_gram_meta(gl, gs, I, K);
      __syncthreads();
      pd_gram_tile(a, (const double*)b, (const double*)c, I, K, gs, gl);
    } else {
      pd_load_tile(a, Aik, lda);
    }
    __syncthreads();
    for (int J = 0; J < K; J++) {
      if (!pd_wait(&flagsL[(I * nbk + J) * PD_FLAG_STRIDE], epoch, info) || !pd_wait(&flagsL[(K * nbk + J) * PD_FLAG_STRIDE], epoch, info)) return;
      pd_load_tile(b, A + (int64_t)(I * 64) * lda + J * 64, lda);
      pd_load_tile(c, A + (int64_t)(K * 64) * lda + J * 64, lda);
      __syncthreads();
      pd_gemm64<true, true, PD_FULL>(a, b, c, -1.0);
      __syncthreads();
    }
    if (!pd_wait(&flagsL[(K * nbk + K) * PD_FLAG_STRIDE], epoch, info)) return;
    pd_load_tile(b, D + (int64_t)K * 4096, 64);
    __syncthreads();
    pd_gemm64<true, false, PD_B_LOWER>(c, a, b, 1.0);  // L_IK = A_IK D_K^T
    __syncthreads();
    store_pub(Aik, lda, c);
    publish(&flagsL[(I * nbk + K) * PD_FLAG_STRIDE]);
    return;
  }
  id -= nOther;
  // ---- X-tile (I, J), I > J: enumeration of the strict lower triangle
  int I = 1;
  while (I * (I + 1) / 2 <= id) I++;
  const int J = id - I * (I - 1) / 2;
  for (int e = threadIdx.x; e < 4096; e += 256) a[e >> 6][e & 63] = 0.0;
  __syncthreads();
  for (int K = J; K < I; K++) {
    if (!pd_wait(&flagsL[(I * nbk + K) * PD_FLAG_STRIDE], epoch, info)) return;
    if (!pd_wait(K == J ? &flagsL[(J * nbk + J) * PD_FLAG_STRIDE] : &flagsX[(K * nbk + J) * PD_FLAG_STRIDE], epoch, info)) return;
    pd_load_tile(b, A + (int64_t)(I * 64) * lda + K * 64, lda);
    if (K == J)
      pd_load_tile(c, D + (int64_t)J * 4096, 64);
    else
      pd_load_tile(c, X + (int64_t)(K * 64) * ldx + J * 64, ldx);
    __syncthreads();
    pd_gemm64<false, true, PD_FULL>(a, b, c, 1.0);  // acc += L_IK X_KJ
    __syncthreads();
  }
  if (!pd_wait(&flagsL[(I * nbk + I) * PD_FLAG_STRIDE], epoch, info)) return;
  pd_load_tile(b, D + (int64_t)I * 4096, 64);
  __syncthreads();
  pd_gemm64<false, false, PD_A_LOWER>(c, b, a, -1.0);  // X_IJ = -D_I acc
  __syncthreads();
  store_pub(X + (int64_t)(I * 64) * ldx + J * 64, ldx, c);
  publish(&flagsX[(I * nbk + J) * PD_FLAG_STRIDE]);
}

void bbh_ensure_side_stream(bbh_handle* h) {
  if (!h->fit_overlap || h->side_stream) return;
  if (hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking) != hipSuccess) h->side_stream = nullptr;
  for (auto& e : h->side_events)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
}

// one launch for the whole factorisation + inverse (np <= 1024): see bbh_potrf_tiles_kernel
// A launch that gave up (info = -7: its workgroups were not all co-resident - partitioned compute modes, another process
// holding CUs) costs up to pd_spin_limit polls; that outcome is remembered per device for the whole process, so that only
// the first model ever pays it (bbh_model.hip reports it through bbh_potrf_tiles_mark_unusable).
static bool g_tiles_unusable[64];
void bbh_potrf_tiles_mark_unusable(int device) { g_tiles_unusable[device & 63] = true; }

// Residency ledger.  The dataflow needs every tile of a launch resident at once, and its block order is not topological
// (row heads wait on L-tiles with higher indices).  The per-launch check `ntiles <= tiles_per_device` says nothing about launches of
// OTHER handles on other streams (the targets of a CompositeSurrogate fitted side by side, the extended models of qLogNEHVI): two
// half-resident launches would starve each other until the spin limit.  So every launch is entered here with an event recorded
// behind it; a new launch on another stream waits (host side, polling the events) until the tiles still in flight plus its own
// fit the device.  A launch on the SAME stream as an entry is ordered behind it and replaces it - the single-handle case never polls.
namespace {
struct pd_lease {
  hipStream_t stream;
  hipEvent_t evt;
  int tiles;
};
struct pd_ledger {
  std::mutex mu;
  std::vector<pd_lease> live;
  std::vector<hipEvent_t> spare;
};
pd_ledger g_tile_ledger[64];

// true: `tiles` more workgroups fit next to what other streams still have in flight (waits up to ~20 ms for that); the caller
// launches and then calls pd_ledger_commit.  false: give this evaluation to the per-step path.
bool pd_ledger_reserve(int device, hipStream_t s, int tiles, int cap) {
  pd_ledger& L = g_tile_ledger[device & 63];
  for (int spin = 0;; spin++) {
    {
      std::lock_guard<std::mutex> lk(L.mu);
      int in_flight = 0;
      for (size_t k = 0; k < L.live.size();) {
        if (L.live[k].stream == s || (L.live[k].evt && hipEventQuery(L.live[k].evt) == hipSuccess)) {
          if (L.live[k].evt) L.spare.push_back(L.live[k].evt);
          L.live[k] = L.live.back();
          L.live.pop_back();
        } else {
          in_flight += L.live[k].tiles;
          k++;
        }
      }
      (void)hipGetLastError();  // (hipErrorNotReady of the queries)
      if (in_flight + tiles <= cap) {
        // entered BEFORE the launch (without an event yet: counted as in flight by everyone else until committed)
        L.live.push_back({s, nullptr, tiles});
        return true;
      }
    }
    if (spin > 20000) return false;
    std::this_thread::yield();
  }
}

void pd_ledger_commit(int device, hipStream_t s) {
  pd_ledger& L = g_tile_ledger[device & 63];
  std::lock_guard<std::mutex> lk(L.mu);
  for (size_t k = 0; k < L.live.size(); k++)
    if (L.live[k].stream == s && !L.live[k].evt) {
      hipEvent_t e = nullptr;
      if (!L.spare.empty()) {
        e = L.spare.back();
        L.spare.pop_back();
      } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        e = nullptr;
      }
      if (!e || hipEventRecord(e, s) != hipSuccess) {  // cannot track it: wait for it here instead
        (void)hipGetLastError();
        hipStreamSynchronize(s);
        if (e) L.spare.push_back(e);
        L.live[k] = L.live.back();
        L.live.pop_back();
      } else {
        L.live[k].evt = e;
      }
      return;
    }
}
}  // namespace

// gram_theta != nullptr: the kernel builds the tiles of K + s2 M itself (single-kernel models, bbh_fit_flow_eligible) from theta at
// that (device or host-mapped) address - the fit evaluation's form, which needs no Gram launch before it.
static_assert(sizeof(pd_mt_args) <= 128, "bbh_model.hip hands pd_mt_args over in a 128-byte buffer");
static bool bbh_potrf_tiles(bbh_handle* h, const double* gram_theta = nullptr, const double* gram_theta_host = nullptr, const pd_mt_args* mt = nullptr) {
  const bool info_clean = (gram_theta || gram_theta_host) && h->info_clean;
  h->info_clean = false;  // (whatever runs next may leave a failure flag behind; only a finished dataflow tail re-establishes it)
  const int64_t np = h->np;
  const int nbk = (int)(np / 64);
  if (!h->potrf_tiles || nbk > 16 || g_tiles_unusable[h->device & 63]) return false;
  // Never inside a stream capture (BBH_FIT_GRAPH=1): the epoch is a by-value kernel argument, so every replay of the captured
  // launch would wait for the flag value the previous replay already left behind (all waits pass at once, tiles read
  // unfinished blocks), and the one-time set-up below must not run while capturing.  The captured evaluation uses the
  // per-step launches.
  if (h->fit_graph_mode || (h->fit_stream && h->stream == h->fit_stream)) return false;
  const int ntiles = nbk + (nbk - 1) * (nbk - 2) / 2 + nbk * (nbk - 1) / 2;  // row heads, other L-tiles, X-tiles
  static const size_t lds = sizeof(double) * 4 * 64 * PD_LD;  // 135 KB: one workgroup per CU
  if (!h->tiles_ready) {
    int per_cu = 0;
    if (hipFuncSetAttribute((const void*)bbh_potrf_tiles_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)bbh_potrf_tiles_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)bbh_potrf_tiles_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)bbh_potrf_tiles_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)bbh_potrf_tiles_kernel<true, true>, 256, lds) != hipSuccess ||
        hipMalloc((void**)&h->d_tileflags, sizeof(int) * 2 * 16 * 16 * PD_FLAG_STRIDE) != hipSuccess ||
        hipMemset(h->d_tileflags, 0, sizeof(int) * 2 * 16 * 16 * PD_FLAG_STRIDE) != hipSuccess) {
      (void)hipGetLastError();
      h->potrf_tiles = false;
      return false;
    }
    h->tiles_per_device = per_cu * h->num_cu;  // workgroups of this kernel the device can hold at once
    h->tiles_ready = true;
  }
  if (ntiles > h->tiles_per_device) return false;  // the dataflow needs every tile resident: not on this device / partition
  pd_mt_args ma{};
  if (mt && mt->nM > 0 && ntiles < h->tiles_per_device) {  // K^-1's tiles in the same launch - as many as are co-resident too, block row 0 first
    ma = *mt;
    // (a part of them - block row 0 first - is possible and handled by the tail (skip_mt), but measured without gain at n = 1024, where the
    // tail's gradient roles, not K^-1, are the long pole: BBH_TILE_MT=partial)
    if (ntiles + ma.nM > h->tiles_per_device) ma.nM = h->tile_mt_partial ? h->tiles_per_device - ntiles : 0;
  }
  h->tiles_did_mt = ma.nM;
  const int grid_tiles = ntiles + ma.nM;
  if (h->tile_spin_limit != h->tile_spin_limit_set) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(pd_spin_limit), &h->tile_spin_limit, sizeof(int)) != hipSuccess) {
      (void)hipGetLastError();
      h->potrf_tiles = false;
      return false;
    }
    h->tile_spin_limit_set = h->tile_spin_limit;
  }
  hipStream_t s = h->stream;
  if (!pd_ledger_reserve(h->device, s, grid_tiles, h->tiles_per_device)) {  // other streams' launches hold the device: per-step path this time
    h->tiles_did_mt = 0;
    return false;
  }
  if (!h->skip_x_memset) hipMemsetAsync(h->d_X, 0, sizeof(double) * np * np, s);  // the upper tiles of L^-1 (XᵀX reads the full matrix)
  // (fit evaluations through the dataflow tail: its last role leaves the flag at 0 for the next evaluation - no memset between them)
  if (!info_clean) hipMemsetAsync(h->d_info, 0, sizeof(int), s);
  const int epoch = ++h->tile_epoch;
  pd_gram_src gs{};
  gs.d_sc1 = (h->tile_wt && h->tile_d_sc1) ? 1 : 0;
  if (gram_theta || gram_theta_host) {
    const bbh_kern_spec ks = bbh_kern_spec_of(h);
    gs.xnT = h->d_xnT;
    gs.task = h->d_task;
    gs.nmask = h->d_nmask;
    gs.theta = gram_theta;
    if (!gram_theta)
      for (int64_t k = 0; k < bbh_theta_len(h) && k < PD_GRAM_MAXTHV; k++) gs.thv[k] = gram_theta_host[k];
    gs.n = (int)h->n;
    gs.np = (int)np;
    gs.dn = h->dn;
    gs.T = h->T;
    gs.tl = (int)bbh_theta_len(h);
    gs.kind = ks.kind[0];
    gs.use_os = ks.use_os;
    gs.jb = ks.jb;
    gs.alpha_off = ks.alpha_off;
    if (getenv("BBH_TILE_STAMPS") && !h->d_tiledbg && hipMalloc((void**)&h->d_tiledbg, sizeof(long long) * 8 * 512) != hipSuccess) h->d_tiledbg = nullptr;
    gs.dbg = h->d_tiledbg;
    h->tiledbg_n = ntiles;
    if (h->tile_wt)
      hipLaunchKernelGGL((bbh_potrf_tiles_kernel<true, true>), dim3((unsigned)grid_tiles), dim3(256), lds, s, h->d_K, np, nbk, h->d_D, h->d_X, np, h->d_tileflags,
                         h->d_tileflags + 256 * PD_FLAG_STRIDE, epoch, h->d_info, gs, ma);
    else
      hipLaunchKernelGGL((bbh_potrf_tiles_kernel<true, false>), dim3((unsigned)grid_tiles), dim3(256), lds, s, h->d_K, np, nbk, h->d_D, h->d_X, np, h->d_tileflags,
                         h->d_tileflags + 256 * PD_FLAG_STRIDE, epoch, h->d_info, gs, ma);
  } else if (h->tile_wt) {
    hipLaunchKernelGGL((bbh_potrf_tiles_kernel<false, true>), dim3((unsigned)grid_tiles), dim3(256), lds, s, h->d_K, np, nbk, h->d_D, h->d_X, np, h->d_tileflags,
                       h->d_tileflags + 256 * PD_FLAG_STRIDE, epoch, h->d_info, gs, ma);
  } else {
    hipLaunchKernelGGL((bbh_potrf_tiles_kernel<false, false>), dim3((unsigned)grid_tiles), dim3(256), lds, s, h->d_K, np, nbk, h->d_D, h->d_X, np, h->d_tileflags,
                       h->d_tileflags + 256 * PD_FLAG_STRIDE, epoch, h->d_info, gs, ma);
  }
  pd_ledger_commit(h->device, s);
  return true;
}

// =====================================================================================================================
// One objective evaluation of a SMALL model in one workgroup: np = 64 (n <= 64 training points), one task, one kernel (any kind
// but the periodic one), marginal log-likelihood.  BayBE campaigns start here (tens of measurements), and launch by launch such an
// evaluation is ~13 operations of a few microseconds of work each - 0.13 ms, all of it launch latency (profiles/
// r03_small_space_latency.json: fit 4.5 ms of an 8.5 ms recommend()).  Here: Gram matrix -> Cholesky factor and its inverse
// (pd_factor_block) -> alpha, K^-1 = X^T X (MFMA) -> value -> the gradient sums over all pairs, everything in LDS, results
// written in the layout of bbh_fit_enqueue (out[0] value, out[1 + slot] gradient; the Cholesky flag in *info).
// Same arithmetic as the launch-by-launch kernels (bbh_gram_kernel, bbh_value_kernel, bbh_grad_pair_kernel); the sums are
// taken in a different (fixed) order, so values agree to rounding, not bitwise.
// =====================================================================================================================
#define FS_MAXD 32
#define FS_ENT 9  // pairs j <= i < 64 per thread: 2080 / 256, rounded up
__global__ __launch_bounds__(256) void bbh_fit_small_kernel(const double* __restrict__ xnT, const double* __restrict__ ystd,
                                                            const double* __restrict__ nmask, const double* __restrict__ theta,
                                                            int n, int dn, const bbh_kern_spec ks, double jitter, int tl,
                                                            double* __restrict__ out, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double s_fs[];
  double(*a)[PD_LD] = (double(*)[PD_LD])s_fs;          // K -> L -> X^T
  double(*x)[PD_LD] = a + 64;                            // L^-1
  double(*s)[PD_LD] = x + 64;                            // scratch -> M = K^-1
  double* xs = (double*)(s + 64);                        // [dn][64] normalised training inputs
  double* vec = xs + FS_MAXD * 64;                       // r[64] | t[64] | alpha[64] | invls[FS_MAXD] | theta[64] | red[4][FS_MAXD + 8]
  double* rv = vec;
  double* tv = vec + 64;
  double* al = vec + 128;
  double* invls = vec + 192;
  double* th = invls + FS_MAXD;
  double* red = th + 64;
  double* nm = red + 4 * (FS_MAXD + 8);                  // noise mask [64]
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  const int kind = ks.kind[0];
  // theta / out / info may live in pinned HOST memory (bbh_fit_enqueue passes the staging buffers themselves: no copies around the
  // launch): theta is read once, coalesced, into LDS.  Everything the evaluation reads from memory is requested here, at once.
  if (t < tl) th[t] = theta[t];
  for (int e = t; e < dn * 64; e += 256) xs[e] = xnT[e];  // (np = 64: the rows of xnT are 64 long)
  if (t < 64) {
    nm[t] = nmask[t];
    rv[t] = (t < n) ? ystd[t] : 0.0;
  }
  if (t == 0) *info = 0;
  // a = identity, x = identity on the sub-blocks the factorisation skips (rows >= 16 nbk are padding), 0 elsewhere
  const int nbk = (n + 15) >> 4;
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    a[i][j] = (i == j) ? 1.0 : 0.0;
    x[i][j] = (i == j && i >= 16 * nbk) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double noise = th[0], mean = th[1], os = ks.use_os ? th[2] : 1.0;
  const double kalpha = ks.alpha_off >= 0 ? th[ks.alpha_off] : 1.0;
  if (t < dn) invls[t] = 1.0 / th[3 + t];
  __syncthreads();
  // ---- Gram matrix.  Only the n (n + 1) / 2 pairs j <= i < n are evaluated (the matrix is symmetric, the padding is the identity
  //      written above), dealt round-robin to the threads: at most FS_ENT each - 9 for n = 64, 3 for n = 33, 1 for n <= 22 (every
  //      thread walking its 16 entries of the full 64 x 64 square was 16 kernel-function evaluations in a row whatever n is).  A thread
  //      keeps the metric and the kernel value of its pairs for the gradient sums below.
  const int npairs = n * (n + 1) / 2;
  double r2v[FS_ENT], kbv[FS_ENT];
  int pij[FS_ENT];
#pragma unroll
  for (int k = 0; k < FS_ENT; k++) {
    const int idx = t + 256 * k;
    r2v[k] = 0.0;
    kbv[k] = 0.0;
    pij[k] = -1;
    if (idx < npairs) {
      int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (i * (i + 1) / 2 > idx) i--;
      while ((i + 1) * (i + 2) / 2 <= idx) i++;
      const int j = idx - i * (i + 1) / 2;
      double r2 = 0.0;
      for (int c = 0; c < dn; c++) r2 += bbh_metric_term(kind, xs[c * 64 + i], xs[c * 64 + j], invls[c]);
      const double kb = bbh_kbase(kind, r2, ks.jb, kalpha);
      r2v[k] = r2;
      kbv[k] = kb;
      pij[k] = (i << 8) | j;
      double kv = kb * os;
      if (i == j) kv += noise * nm[i] + jitter;
      a[i][j] = kv;
      a[j][i] = kv;
    }
  }
  __syncthreads();
  pd_factor_block(a, x, s, 0, info, nullptr, 0, nbk);  // a = L (lower), x = L^-1; ends with a barrier
  // ---- residual, t = X r, alpha = X^T t, log-determinant
  if (t < n) rv[t] -= mean;
  __syncthreads();
  const int row = t >> 2, part = t & 3;  // four threads per row / column, partial sums combined by two shuffles
  {
    double acc = 0.0;
    for (int j = part; j <= row; j += 4) acc = fma(x[row][j], rv[j], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) tv[row] = acc;
  }
  __syncthreads();
  double ld = 0.0, ra = 0.0, gmean = 0.0;
  {
    double acc = 0.0;
    for (int j = row + part; j < 64; j += 4) acc = fma(x[j][row], tv[j], acc);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) {
      al[row] = acc;
      if (row < n) {
        ld = log(a[row][row]);
        ra = rv[row] * acc;
        gmean = acc;
      }
    }
  }
  __syncthreads();  // (a is read above for the last time)
  // ---- a <- X^T, then M = X^T X = a a^T on the MFMA (into s)
  for (int e = t; e < 4096; e += 256) {
    const int i = e >> 6, j = e & 63;
    a[i][j] = x[j][i];
  }
  __syncthreads();
  pd_gemm64<true, false, PD_FULL>(s, a, a, 1.0);
  __syncthreads();
  // ---- gradient sums over all ordered pairs (a, b), a, b < n:  G = 0.5 (alpha_a alpha_b - M_ab); every term is symmetric in (a, b),
  //      so the thread's pairs j < i count twice
  double g_noise = 0.0, g_os = 0.0, g_al = 0.0, g_ls[FS_MAXD];
#pragma unroll
  for (int c = 0; c < FS_MAXD; c++) g_ls[c] = 0.0;
  const bool dot = BBH_KIND_IS_DOT(kind);
#pragma unroll
  for (int k = 0; k < FS_ENT; k++) {
    if (pij[k] < 0) continue;
    const int i = pij[k] >> 8, j = pij[k] & 255;
    const double G = (i == j ? 0.5 : 1.0) * (al[i] * al[j] - s[i][j]);
    const double r2 = r2v[k], kb = kbv[k];
    if (i == j) g_noise += G * nm[i];
    g_os += G * kb;
    const double Gg = G * bbh_gfun(kind, r2, ks.jb, kalpha) * os;
#pragma unroll
    for (int c = 0; c < FS_MAXD; c++)
      if (c < dn) {
        const double xa = xs[c * 64 + i], xb = xs[c * 64 + j], il = invls[c];
        g_ls[c] += Gg * (dot ? xa * xb : (xa - xb) * (xa - xb)) * (il * il * il);
      }
    if (ks.alpha_off >= 0) {
      if (kind == BBH_KERNEL_RQ) {
        const double u = r2 / (2.0 * kalpha);
        g_al += G * os * kb * (u / (1.0 + u) - log1p(u));
      } else if (BBH_KIND_IS_POLY(kind)) {
        const int pw = kind - BBH_KERNEL_POLY1 + 1;
        g_al += G * os * (double)pw * bbh_powi(r2 + kalpha, pw - 1);
      }
    }
  }
  // ---- block sums in a fixed order: wave shuffles, then the four waves
  auto wave_sum = [](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
  };
  const int RW = FS_MAXD + 8;
  double v0 = wave_sum(ld), v1 = wave_sum(ra), v2 = wave_sum(gmean), v3 = wave_sum(g_noise), v4 = wave_sum(g_os), v5 = wave_sum(g_al);
  if (l == 0) {
    red[w * RW + 0] = v0;
    red[w * RW + 1] = v1;
    red[w * RW + 2] = v2;
    red[w * RW + 3] = v3;
    red[w * RW + 4] = v4;
    red[w * RW + 5] = v5;
  }
#pragma unroll
  for (int c = 0; c < FS_MAXD; c++)
    if (c < dn) {
      const double v = wave_sum(g_ls[c]);
      if (l == 0) red[w * RW + 8 + c] = v;
    }
  __syncthreads();
  auto tot = [&](int k) { return (red[k] + red[RW + k]) + (red[2 * RW + k] + red[3 * RW + k]); };
  for (int slot = t; slot < tl; slot += 256) out[1 + slot] = 0.0;
  __syncthreads();
  if (t == 0) {
    out[0] = -0.5 * tot(1) - tot(0) - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    out[1 + 0] = tot(3);                        // noise
    out[1 + 1] = tot(2);                        // constant mean
    out[1 + 2] = ks.use_os ? tot(4) : 0.0;      // outputscale
    if (ks.alpha_off >= 0) out[1 + ks.alpha_off] = tot(5);
  }
  if (t < dn) out[1 + 3 + t] = tot(8 + t);
}

bool bbh_fit_small_launch(bbh_handle* h, double jitter, const double* theta_dev, double* out_dev, int* info_dev) {
  if (!h->fit_small || h->np != 64 || h->T != 1 || h->F > 1 || h->hadamard || h->desc.criterion != BBH_CRITERION_MLL ||
      h->dn > FS_MAXD || h->desc.kernel_kind == BBH_KERNEL_PERIODIC || h->fit_graph_mode ||
      (h->fit_stream && h->stream == h->fit_stream))
    return false;
  static const size_t lds = sizeof(double) * (3 * 64 * PD_LD + FS_MAXD * 64 + 192 + FS_MAXD + 64 + 4 * (FS_MAXD + 8) + 64);
  if (!h->fit_small_ready) {
    if (hipFuncSetAttribute((const void*)bbh_fit_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      h->fit_small = 0;
      return false;
    }
    h->fit_small_ready = true;
  }
  if (bbh_theta_len(h) > 64) return false;
  hipLaunchKernelGGL(bbh_fit_small_kernel, dim3(1), dim3(256), lds, h->stream, h->d_xnT, h->d_ystd, h->d_nmask, theta_dev, (int)h->n,
                     h->dn, bbh_kern_spec_of(h), jitter, (int)bbh_theta_len(h), out_dev, info_dev);
  return true;
}

// Fit evaluations of the models bbh_fit_flow_eligible admits: Gram tiles + factor + inverse in the one tile-dataflow launch (theta read
// at theta_any: device or host-mapped).  false: that launch is not available here - nothing was enqueued.
bool bbh_potrf_trtri_from_inputs(bbh_handle* h, const double* theta_any, const double* theta_host, const void* mt_args) {
  if (!theta_any && (!theta_host || bbh_theta_len(h) > PD_GRAM_MAXTHV)) return false;
  if (h->dn > PD_GRAM_MAXD || bbh_theta_len(h) > PD_GRAM_MAXTH || h->F > 1 || h->hadamard || h->desc.kernel_kind == BBH_KERNEL_PERIODIC ||
      h->desc.kernel_kind == BBH_KERNEL_RFF)
    return false;
  return bbh_potrf_tiles(h, theta_any, theta_any ? nullptr : theta_host, (const pd_mt_args*)mt_args);
}

// BBH_TILE_STAMPS=1: clock stamps [tiles][8] of the row heads of the last Gram-building tile launch (0 entry, 1 own tiles ready, 2 D_{I-1}'s flag
// seen, 3 D_{I-1} in LDS, 4 panel product done, 5 factorisation starts, 6 ends, 7 D_I published); returns the number of tiles
extern "C" int bbh_tiles_trace_read(bbh_handle* h, long long* stamps_host, int cap) {
  if (!h || !h->d_tiledbg || !stamps_host || cap < h->tiledbg_n) return -1;
  hipStreamSynchronize(h->stream);
  hipMemcpy(stamps_host, h->d_tiledbg, sizeof(long long) * 8 * h->tiledbg_n, hipMemcpyDeviceToHost);
  return h->tiledbg_n;
}

void bbh_potrf_trtri(bbh_handle* h) {
  if (bbh_potrf_tiles(h)) return;
  hipStream_t s = h->stream;
  const int64_t np = h->np, nbk = np / 64;
  hipMemsetAsync(h->d_X, 0, sizeof(double) * np * np, s);
  hipMemsetAsync(h->d_info, 0, sizeof(int), s);
  double* A = h->d_K;
  // The inverse X = L^-1 is built block row by block row on a second stream, concurrently with the trailing updates
  // of the factorisation: block row I of X needs the diagonal block I (factor + inverse), the rows of L left of it
  // (final since the panels of the earlier steps) and the earlier rows of X -
  //   X[I][0:I] = -D[I] (L[I][0:I] X[0:I][0:I])
  // - none of which the trailing update of step I touches.  Sequentially (sub-diagonal by sub-diagonal, after the
  // factorisation) these 14 small GEMM launches were 25 % of a fit evaluation at n = 512.
  const bool overlap = h->fit_overlap;
  bbh_ensure_side_stream(h);
  hipStream_t s2 = (overlap && h->side_stream && h->side_events[0] && h->side_events[1]) ? h->side_stream : s;
  for (int64_t J = 0; J < nbk; J++) {
    if (h->potrf_register_form)  // env BBH_POTRF_REG=1: the one-wave register form (A/B)
      hipLaunchKernelGGL(bbh_potrf_diag_kernel, dim3(1), dim3(256), 0, s, A, np, J, h->d_D, h->d_X, np, h->d_info);
    else
      hipLaunchKernelGGL(bbh_potrf_diag16_kernel, dim3(1), dim3(256), 0, s, A, np, J, h->d_D, h->d_X, np, h->d_info);
    if (J > 0) {  // block row J of X (everything it reads precedes this point in stream s)
      if (s2 != s) {
        hipEventRecord(h->side_events[0], s);
        hipStreamWaitEvent(s2, h->side_events[0], 0);
      }
      double* T = h->d_tmp;  // [64, 64 J]
      bbh_gemm(s2, false, false, 64, 64 * J, 64 * J, 1.0, A + (J * 64) * np, np, 0, h->d_X, np, 0, 0.0, T, 64 * J, 0, 1);
      bbh_gemm(s2, false, false, 64, 64 * J, 64, -1.0, h->d_D + J * 4096, 64, 0, T, 64 * J, 0, 0.0, h->d_X + (J * 64) * np, np, 0, 1);
    }
    const int64_t rem = nbk - J - 1;
    if (rem > 0) {
      double* A21 = A + ((J + 1) * 64) * np + J * 64;
      double* A22 = A + ((J + 1) * 64) * np + (J + 1) * 64;
      // panel: A21 <- A21 L_JJ^-T   (in place; each workgroup owns its 64 rows)
      bbh_gemm(s, false, true, rem * 64, 64, 64, 1.0, A21, np, 0, h->d_D + J * 4096, 64, 0, 0.0, A21, np, 0, 1);
      // trailing update: A22 <- A22 - A21 A21^T
      bbh_gemm(s, false, true, rem * 64, rem * 64, 64, -1.0, A21, np, 0, A21, np, 0, 1.0, A22, np, 0, 1);
    }
  }
  if (s2 != s) {  // the main stream continues once the last row of X is there
    hipEventRecord(h->side_events[1], s2);
    hipStreamWaitEvent(s, h->side_events[1], 0);
  }
}

// The same factorisation + inverse on caller-owned buffers (no handle state, no second stream): A [np, np] SPD -> L in its lower part,
// X [np, np] (zeroed here) -> L^-1, D [np / 64, 64, 64] the inverses of the diagonal blocks, tmp [64, np].  Used by the feature-space
// system of the RFF kernel beyond two tiles (bbh_rff.hip).
void bbh_potrf_trtri_buf(hipStream_t s, double* A, int64_t np, double* D, double* X, double* tmp, int* info) {
  const int64_t nbk = np / 64;
  hipMemsetAsync(X, 0, sizeof(double) * np * np, s);
  for (int64_t J = 0; J < nbk; J++) {
    hipLaunchKernelGGL(bbh_potrf_diag16_kernel, dim3(1), dim3(256), 0, s, A, np, J, D, X, np, info);
    if (J > 0) {
      bbh_gemm(s, false, false, 64, 64 * J, 64 * J, 1.0, A + (J * 64) * np, np, 0, X, np, 0, 0.0, tmp, 64 * J, 0, 1);
      bbh_gemm(s, false, false, 64, 64 * J, 64, -1.0, D + J * 4096, 64, 0, tmp, 64 * J, 0, 0.0, X + (J * 64) * np, np, 0, 1);
    }
    const int64_t rem = nbk - J - 1;
    if (rem > 0) {
      double* A21 = A + ((J + 1) * 64) * np + J * 64;
      double* A22 = A + ((J + 1) * 64) * np + (J + 1) * 64;
      bbh_gemm(s, false, true, rem * 64, 64, 64, 1.0, A21, np, 0, D + J * 4096, 64, 0, 0.0, A21, np, 0, 1);
      bbh_gemm(s, false, true, rem * 64, rem * 64, 64, -1.0, A21, np, 0, A21, np, 0, 1.0, A22, np, 0, 1);
    }
  }
}

// ---- matvecs -----------------------------------------------------------------------------
__global__ void bbh_matvec_kernel(const double* __restrict__ A, int64_t lda, int64_t rows, int64_t cols,
                                  const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  double s = 0.0;
  for (int64_t c = lane; c < cols; c += 64) s += A[row * lda + c] * x[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) y[row] = s;
}

// y = A^T x: 64 columns x 16 row groups per workgroup, partial sums combined in a fixed order through LDS
// (deterministic).  One thread per column over all rows took 112 us at 512 x 512 - two workgroups.
__global__ __launch_bounds__(1024) void bbh_matvec_t_kernel(const double* __restrict__ A, int64_t lda, int64_t rows,
                                                            int64_t cols, const double* __restrict__ x,
                                                            double* __restrict__ y) {
  __shared__ double part[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < cols)
    for (int64_t r = rg; r < rows; r += 16) s = fma(A[r * lda + c], x[r], s);
  part[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < 16; g++) tot += part[g][cl];
    y[c] = tot;
  }
}

void bbh_matvec(hipStream_t s, const double* A, int64_t lda, int64_t rows, int64_t cols, const double* x,
                double* y) {
  hipLaunchKernelGGL(bbh_matvec_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, A, lda, rows, cols, x, y);
}
void bbh_matvec_t(hipStream_t s, const double* A, int64_t lda, int64_t rows, int64_t cols, const double* x,
                  double* y) {
  hipLaunchKernelGGL(bbh_matvec_t_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, s, A, lda, rows, cols,
                     x, y);
}
