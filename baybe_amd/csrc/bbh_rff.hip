// Random-Fourier-features kernel (baybe/kernels/basic.py:183-199 -> gpytorch.kernels.RFFKernel): the model in FEATURE space.
//
// gpytorch's kernel (RFFKernel.forward / _featurize): with D = num_samples frequencies W [d, D] ~ N(0, 1) drawn once per model,
//   z(x) = [cos(x (W / l)), sin(x (W / l))]  (2 D features, l = ARD lengthscales),   k(x, x') = z(x) . z(x') / D.
// The Gram matrix has rank <= m = 2 D whatever n is, so nothing here is n x n.  With phi = z / sqrt(D), Phi [n, m] the training
// features, r = y~ - c, noise s2, outputscale os (1 without a ScaleKernel), eps = s2 / os and
//   B = eps I + Phi^T Phi  (m x m, SPD),   b = Phi^T r,   a = B^-1 b,   B = L L^T:
//   K = os Phi Phi^T + s2 I,   K^-1 = (I - Phi B^-1 Phi^T) / s2,   alpha = K^-1 r = (r - Phi a) / s2,   Phi^T alpha = a / os,
//   log|K| = (n - m) log s2 + m log os + log|B|,
//   posterior at x*:  mean = c + phi* . a,   var = s2 phi*^T B^-1 phi* = |sqrt(s2) L^-1 phi*|^2,   cov(x*, x_p) = s2 phi*^T B^-1 phi_p
// (Woodbury on the reference's n x n expressions; gpytorch takes the same route through LowRankRootAddedDiagLinearOperator when
// m < n).  Fit objective = the marginal log-likelihood  -1/2 r^T alpha - 1/2 log|K| - n/2 log(2 pi), gradient by hand:
//   d/dc = sum alpha,   d/ds2 = 1/2 (alpha^T alpha - tr K^-1),  tr K^-1 = (n - m + eps tr B^-1) / s2,
//   d/dos = 1/2 (|a|^2 / os^2 - (m - eps tr B^-1) / os),
//   dL/dPhi = alpha a^T - Phi B^-1   (n x m),   P = Xn (W / l):  dL/dP = (-dPhi_cos o sin P + dPhi_sin o cos P) / sqrt(D),
//   d/dl_i = -(1 / l_i^2) sum_j W_ij (Xn^T dL/dP)_ij.
// Candidates never leave feature space either: one kernel featurises a tile of candidates in registers (the A operand of the fp64
// MFMA), multiplies by sqrt(s2) L^-T (lower-triangular blocks skipped) for the variance and by [a | s2 B^-1 phi_p] for the mean and
// the pending points' cross-covariances: 2 m^2 / 2 + 2 m 16 flops per candidate instead of n^2.
// One task, one kernel, MLL, no latent rows.  D <= 64: everything above in two dedicated kernels (one-workgroup m x m solve, fused
// featurise-and-multiply posterior).  64 < D <= 256 (m up to 512 - round 6): the m x m system goes through the blocked factorisation
// (bbh_potrf_trtri_buf), candidates through a chunked form - features of 8192 candidates materialised, two GEMMs, a row-sum kernel.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "bbh_tiles.h"

#define RFF_TH_NOISE 0
#define RFF_TH_MEAN 1
#define RFF_TH_OS 2
#define RFF_TH_LS 3
#define RFF_MAXDN 64
#define RFF_MAXD 256

void bbh_potrf_trtri_buf(hipStream_t s, double* A, int64_t np, double* D, double* X, double* tmp, int* info);  // bbh_linalg.hip

struct bbh_rff_state {
  int D = 0, Dh = 0, mp = 0;  // frequencies; half of the padded feature count (32 | 64 | 128 | 256); padded feature count 2 Dh
  int64_t nr = 0;             // n rounded up to 64: rows of Phi
  int dn = 0;
  double* d_W = nullptr;      // [dn][Dh] the frequencies as drawn (zero columns beyond D)
  double* d_wS = nullptr;     // [dn][4][Dh / 4] W / l of the current theta, column j = 4 t + g stored at [g][t] (the posterior kernel's order)
  double* d_Phi = nullptr;    // [nr][mp] training features (cos block | sin block), zero rows / columns on the padding
  double* d_B = nullptr;      // [mp][mp]
  double* d_Linv = nullptr;   // [mp][mp] L^-1 (lower)
  double* d_Binv = nullptr;   // [mp][mp]
  double* d_T = nullptr;      // [nr][mp] Phi B^-1, then dL/dP in its first Dh columns
  double* d_vec = nullptr;    // b [mp] | t [mp] | a [mp] | scalars [8]: 0 log|B|, 1 tr B^-1 (real features)
  double* d_alpha = nullptr;  // [nr]
  double* d_gl = nullptr;     // [dn][nr / 64] per-chunk partial sums of the lengthscale gradient
  double* d_Qp = nullptr;     // packed lower block rows of sqrt(s2) L^-T (posterior operand), rff_qp_elems(mp / 16) doubles
  double* d_E = nullptr;      // [mp][16] column 0: a, columns 1..p: s2 B^-1 phi_p of the pending points
  double* d_lo = nullptr;     // [dn] lower scaling bound, [dn] 1 / (hi - lo), then numcol as doubles [dn]
  double* d_Dg = nullptr;     // m > 128: [mp / 64][64][64] inverses of the diagonal blocks (bbh_potrf_trtri_buf)
  double* d_tmp = nullptr;    // m > 128: [64][mp] scratch of the blocked inverse
  int* d_info = nullptr;
  std::vector<double> W_host;  // [dn][D] as handed over
};

void bbh_rff_destroy(bbh_handle* h) {
  auto* st = (bbh_rff_state*)h->rff_state;
  if (!st) return;
  for (double* p : {st->d_W, st->d_wS, st->d_Phi, st->d_B, st->d_Linv, st->d_Binv, st->d_T, st->d_vec, st->d_alpha, st->d_gl, st->d_Qp, st->d_E, st->d_lo, st->d_Dg, st->d_tmp})
    if (p) hipFree(p);
  if (st->d_info) hipFree(st->d_info);
  delete st;
  h->rff_state = nullptr;
}

extern "C" int bbh_set_rff_weights(bbh_handle* h, const double* W_host, int32_t dn, int32_t D) {
  if (!h) return -1;
  if (!W_host || dn < 1 || dn > RFF_MAXDN || D < 1 || D > RFF_MAXD) {
    h->err = "bbh_set_rff_weights: weights [dn, D] with 1 <= dn <= 64 numerical columns and 1 <= D <= 256 frequencies";
    return -1;
  }
  h->rff_w_host.assign(W_host, W_host + (size_t)dn * D);
  h->rff_w_dn = dn;
  h->rff_w_D = D;
  return 0;
}

// packed operand of the variance product: for k-block kb (16 feature rows) the columns o >= 16 kb of Q[k][o] = sqrt(s2) L^-1[o][k];
// row pitch = an odd multiple of 16 doubles, so that the two k-rows a half-wave reads sit 32 banks apart
__host__ __device__ constexpr int rff_qp_pitch(int NB, int kb) { return (NB - kb) * 16 + (((NB - kb) & 1) ? 0 : 16); }
__host__ __device__ constexpr int rff_qp_off(int NB, int kb) {
  int o = 0;
  for (int j = 0; j < kb; j++) o += 16 * rff_qp_pitch(NB, j);
  return o;
}
__host__ __device__ constexpr int rff_qp_elems(int NB) { return rff_qp_off(NB, NB); }

// ---- training side ---------------------------------------------------------------------------------------------------------
__global__ void bbh_rff_scale_kernel(const double* __restrict__ W, const double* __restrict__ theta, int dn, int Dh, double* __restrict__ wS) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= dn * Dh) return;
  const int i = e / Dh, j = e % Dh;
  wS[i * Dh + (j & 3) * (Dh / 4) + (j >> 2)] = W[e] / theta[RFF_TH_LS + i];
}

// Phi[a][j] = cos(p) / sqrt(D), Phi[a][Dh + j] = sin(p) / sqrt(D), p = sum_i xn[a][i] W[i][j] / l_i   (j < D, a < n; 0 elsewhere)
__global__ void bbh_rff_features_kernel(const double* __restrict__ xnT, int64_t ldxn, const double* __restrict__ wS, int n, int64_t nr, int dn,
                                        int D, int Dh, double* __restrict__ Phi) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nr * Dh) return;
  const int64_t a = e / Dh;
  const int j = (int)(e % Dh);
  double c = 0.0, s = 0.0;
  if (a < n && j < D) {
    double p = 0.0;
    for (int i = 0; i < dn; i++) p = fma(xnT[(int64_t)i * ldxn + a], wS[i * Dh + (j & 3) * (Dh / 4) + (j >> 2)], p);
    sincos(p, &s, &c);
    const double sc = 1.0 / sqrt((double)D);
    c *= sc;
    s *= sc;
  }
  Phi[a * (2 * Dh) + j] = c;
  Phi[a * (2 * Dh) + Dh + j] = s;
}

// B (= Phi^T Phi on entry) -> L^-1: one workgroup, the matrix as one or 2 x 2 tiles of 64 x 64 in LDS.
//   B += eps on the diagonal of the real features; padded features (zero rows / columns) get a unit diagonal.
__global__ __launch_bounds__(256) void bbh_rff_solve_kernel(const double* __restrict__ B, int mp, int D, int Dh, const double* __restrict__ theta,
                                                            int use_os, double* __restrict__ Linv, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double s_rff[];
  double(*T0)[PD_LD] = (double(*)[PD_LD])s_rff;
  double(*T1)[PD_LD] = T0 + 64;
  double(*T2)[PD_LD] = T1 + 64;
  double(*T3)[PD_LD] = T2 + 64;
  const int t = threadIdx.x;
  const double eps = theta[RFF_TH_NOISE] / (use_os ? theta[RFF_TH_OS] : 1.0);
  auto load = [&](double(*dst)[PD_LD], int I, int J) {
    for (int e = t; e < 4096; e += 256) {
      const int r = e >> 6, c = e & 63;
      double v = B[(int64_t)(64 * I + r) * mp + 64 * J + c];
      if (I == J && r == c) {
        const int k = 64 * I + r;
        v = ((k % Dh) < D) ? v + eps : 1.0;
      }
      dst[r][c] = v;
    }
  };
  auto zero = [&](double(*dst)[PD_LD]) {
    for (int e = t; e < 4096; e += 256) dst[e >> 6][e & 63] = 0.0;
  };
  auto store = [&](const double(*src)[PD_LD], int I, int J) {
    for (int e = t; e < 4096; e += 256) Linv[(int64_t)(64 * I + (e >> 6)) * mp + 64 * J + (e & 63)] = src[e >> 6][e & 63];
  };
  load(T0, 0, 0);
  zero(T1);
  if (mp == 128) load(T3, 1, 0);
  __syncthreads();
  pd_factor_block(T0, T1, T2, 0, info);  // T0 = L11, T1 = L11^-1
  __syncthreads();
  store(T1, 0, 0);
  if (mp == 64) return;
  pd_gemm64<true, false, PD_B_LOWER>(T2, T3, T1, 1.0);  // L21 = B21 L11^-T
  __syncthreads();
  pd_gemm64<false, false, PD_FULL>(T3, T2, T1, 1.0);  // W = L21 L11^-1
  load(T0, 1, 1);
  __syncthreads();
  pd_gemm64<true, true, PD_OUT_LOWER>(T0, T2, T2, -1.0);  // S = B22 - L21 L21^T (lower sub-blocks)
  zero(T1);
  __syncthreads();
  pd_factor_block(T0, T1, T2, 64, info);  // T1 = L22^-1
  __syncthreads();
  store(T1, 1, 1);
  pd_gemm64<false, false, PD_A_LOWER>(T2, T1, T3, -1.0);  // (L^-1)21 = -L22^-1 W
  zero(T0);
  __syncthreads();
  store(T2, 1, 0);
  store(T0, 0, 1);
}

// t = L^-1 b, a = L^-T t, log|B| and tr B^-1 over the real features   (one workgroup; vec = b | t | a | scalars; mp <= 512)
__global__ __launch_bounds__(256) void bbh_rff_vec_kernel(const double* __restrict__ Linv, int mp, int D, int Dh, double* __restrict__ vec) {
  __shared__ double sb[512], stv[512], s_tr[512], s_ld[512];
  const int t = threadIdx.x;
  for (int r = t; r < mp; r += 256) sb[r] = vec[r];
  __syncthreads();
  for (int r = t; r < mp; r += 256) {
    double acc = 0.0, tr = 0.0;
    const bool real = (r % Dh) < D;
    for (int k = 0; k <= r; k++) {
      const double v = Linv[(int64_t)r * mp + k];
      acc = fma(v, sb[k], acc);
      if (real) tr = fma(v, v, tr);  // (rows of padded features are unit vectors; real rows have zeros in padded columns)
    }
    stv[r] = acc;
    vec[mp + r] = acc;
    s_tr[r] = tr;
    s_ld[r] = real ? -2.0 * log(Linv[(int64_t)r * mp + r]) : 0.0;
  }
  __syncthreads();
  for (int r = t; r < mp; r += 256) {
    double acc = 0.0;
    for (int o = r; o < mp; o++) acc = fma(Linv[(int64_t)o * mp + r], stv[o], acc);
    vec[2 * mp + r] = acc;
  }
  if (t == 0) {  // fixed-order sums
    double a = 0.0, b = 0.0;
    for (int k = 0; k < mp; k++) {
      a += s_ld[k];
      b += s_tr[k];
    }
    vec[3 * mp + 0] = a;
    vec[3 * mp + 1] = b;
  }
}

// m > 128: the diagonal of B = Phi^T Phi gets eps on the real features and 1 on the padded ones (what bbh_rff_solve_kernel does while loading)
__global__ void bbh_rff_diag_kernel(double* __restrict__ B, int mp, int D, int Dh, const double* __restrict__ theta, int use_os) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= mp) return;
  const double eps = theta[RFF_TH_NOISE] / (use_os ? theta[RFF_TH_OS] : 1.0);
  B[(int64_t)k * mp + k] = ((k % Dh) < D) ? B[(int64_t)k * mp + k] + eps : 1.0;
}

// alpha = (r - Phi a) / s2   (one wave per training row)
__global__ __launch_bounds__(256) void bbh_rff_alpha_kernel(const double* __restrict__ Phi, int mp, const double* __restrict__ avec,
                                                            const double* __restrict__ r, const double* __restrict__ theta, int n, int64_t nr,
                                                            double* __restrict__ alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nr) return;
  double s = 0.0;
  for (int k = lane; k < mp; k += 64) s = fma(Phi[row * mp + k], avec[k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) alpha[row] = row < n ? (r[row] - s) / theta[RFF_TH_NOISE] : 0.0;
}

// T[a][j] (= (Phi B^-1)[a][j] on entry) -> dL/dP[a][j] for j < Dh (in place, first Dh columns)
__global__ void bbh_rff_dp_kernel(const double* __restrict__ Phi, const double* __restrict__ avec, const double* __restrict__ alpha, int n,
                                  int64_t nr, int D, int Dh, double* __restrict__ T) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nr * Dh) return;
  const int64_t a = e / Dh;
  const int j = (int)(e % Dh);
  const int mp = 2 * Dh;
  double out = 0.0;
  if (a < n && j < D) {
    const double dc = alpha[a] * avec[j] - T[a * mp + j], ds = alpha[a] * avec[Dh + j] - T[a * mp + Dh + j];
    // Phi = [cos P, sin P] / sqrt(D):  d cos / dP = -sin, d sin / dP = cos
    out = -dc * Phi[a * mp + Dh + j] + ds * Phi[a * mp + j];
  }
  T[a * mp + j] = out;
}

// glp[i][c] = sum_{a in chunk c (64 rows), j < D} xn[a][i] W[i][j] dP[a][j]   (one workgroup per numerical column and row chunk; the value
// kernel adds the chunks in a fixed order: gl[i] = -(sum_c glp[i][c]) / l_i^2).  One workgroup per column over all rows was 42 us.
__global__ __launch_bounds__(256) void bbh_rff_gls_kernel(const double* __restrict__ xnT, int64_t ldxn, const double* __restrict__ W,
                                                          const double* __restrict__ dP, int n, int D, int Dh, int nchunks,
                                                          double* __restrict__ glp) {
  __shared__ double sm[4];
  const int i = blockIdx.x, c = blockIdx.y, t = threadIdx.x;
  const int mp = 2 * Dh;
  double acc = 0.0;
  for (int e = t; e < 64 * Dh; e += 256) {
    const int64_t a = (int64_t)c * 64 + e / Dh;
    const int j = e % Dh;
    if (a < n && j < D) acc = fma(xnT[(int64_t)i * ldxn + a] * W[i * Dh + j], dP[a * mp + j], acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((t & 63) == 0) sm[t >> 6] = acc;
  __syncthreads();
  if (t == 0) glp[i * nchunks + c] = sm[0] + sm[1] + sm[2] + sm[3];
}

// out[0] = value, out[1 + slot] = gradient   (one workgroup)
__global__ __launch_bounds__(256) void bbh_rff_value_kernel(const double* __restrict__ r, const double* __restrict__ alpha, const double* __restrict__ vec,
                                                            const double* __restrict__ glp, int nchunks, const double* __restrict__ theta, int use_os,
                                                            int n, int mp, int D, int dn, double* __restrict__ out) {
  __shared__ double sm[4];
  const int t = threadIdx.x;
  double ra = 0.0, aa = 0.0, sa = 0.0;
  for (int i = t; i < n; i += 256) {
    const double al = alpha[i];
    ra = fma(r[i], al, ra);
    aa = fma(al, al, aa);
    sa += al;
  }
  auto block_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((t & 63) == 0) sm[t >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
  };
  ra = block_sum(ra);
  aa = block_sum(aa);
  sa = block_sum(sa);
  if (t == 0) {
    const double s2 = theta[RFF_TH_NOISE], os = use_os ? theta[RFF_TH_OS] : 1.0, eps = s2 / os;
    const double m = 2.0 * (double)D;
    const double logdetB = vec[3 * mp + 0], trBinv = vec[3 * mp + 1];
    const double logdetK = ((double)n - m) * log(s2) + m * log(os) + logdetB;
    out[0] = -0.5 * ra - 0.5 * logdetK - 0.5 * (double)n * 1.8378770664093453;
    out[1 + RFF_TH_NOISE] = 0.5 * (aa - ((double)n - m + eps * trBinv) / s2);
    out[1 + RFF_TH_MEAN] = sa;
    double a2 = 0.0;
    for (int k = 0; k < mp; k++) a2 = fma(vec[2 * mp + k], vec[2 * mp + k], a2);
    out[1 + RFF_TH_OS] = use_os ? 0.5 * (a2 / (os * os) - (m - eps * trBinv) / os) : 0.0;
  }
  if (t < dn) {
    double g = 0.0;
    for (int c = 0; c < nchunks; c++) g += glp[t * nchunks + c];
    const double l = theta[RFF_TH_LS + t];
    out[1 + RFF_TH_LS + t] = -g / (l * l);
  }
}

static int rff_posterior_lds_attr(bbh_handle* h, int Dh);  // (defined with the kernel)

// ---- set-up / fit / factorise ------------------------------------------------------------------------------------------------
int bbh_rff_setup(bbh_handle* h) {
  if (h->T != 1 || h->F != 1 || h->desc.criterion != BBH_CRITERION_MLL || h->desc.task_col >= 0) {
    h->err = "bbh_set_model: the RFF kernel is available for one task, as the single kernel of the model, with the marginal log-likelihood";
    return -1;
  }
  if (h->rff_w_dn != h->dn || h->rff_w_D < 1) {
    h->err = "bbh_set_model: call bbh_set_rff_weights with the frequencies [dn, D] of the RFF kernel first";
    return -1;
  }
  for (int64_t i = 0; i < h->n; i++)
    if (h->nmask_host[i] == 0.0) {
      h->err = "bbh_set_model: latent (noise-free) rows are not available with the RFF kernel";
      return -1;
    }
  bbh_rff_destroy(h);
  auto* st = new bbh_rff_state;
  h->rff_state = st;
  st->D = h->rff_w_D;
  st->Dh = st->D <= 32 ? 32 : st->D <= 64 ? 64 : st->D <= 128 ? 128 : 256;
  st->mp = 2 * st->Dh;
  st->nr = bbh_round_up(h->n, 64);
  st->dn = h->dn;
  st->W_host = h->rff_w_host;
  const int dn = st->dn, Dh = st->Dh, mp = st->mp;
  const int64_t nr = st->nr;
#define RFF_ALLOC(ptr, count) BBH_HIP_TRY(h, hipMalloc((void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)))
  RFF_ALLOC(st->d_W, dn * Dh);
  RFF_ALLOC(st->d_wS, dn * Dh);
  RFF_ALLOC(st->d_Phi, nr * mp);
  RFF_ALLOC(st->d_B, mp * mp);
  RFF_ALLOC(st->d_Linv, mp * mp);
  RFF_ALLOC(st->d_Binv, mp * mp);
  RFF_ALLOC(st->d_T, nr * mp);
  RFF_ALLOC(st->d_vec, 3 * mp + 8);
  RFF_ALLOC(st->d_alpha, nr);
  RFF_ALLOC(st->d_gl, (size_t)dn * (size_t)(nr / 64));
  RFF_ALLOC(st->d_Qp, Dh <= 64 ? rff_qp_elems(mp / 16) : 16);  // (the fused posterior kernel's operand: D <= 64 only)
  if (Dh > 64) {
    RFF_ALLOC(st->d_Dg, (size_t)(mp / 64) * 4096);
    RFF_ALLOC(st->d_tmp, (size_t)64 * mp);
  }
  RFF_ALLOC(st->d_E, mp * 16);
  RFF_ALLOC(st->d_lo, 3 * dn);
  RFF_ALLOC(st->d_info, 1);
  std::vector<double> Wp((size_t)dn * Dh, 0.0), lo3((size_t)3 * dn);
  for (int i = 0; i < dn; i++)
    for (int j = 0; j < st->D; j++) Wp[(size_t)i * Dh + j] = st->W_host[(size_t)i * st->D + j];
  for (int i = 0; i < dn; i++) {
    lo3[i] = h->lo[i];
    lo3[dn + i] = 1.0 / (h->hi[i] - h->lo[i]);
    lo3[2 * dn + i] = (double)h->numcol[i];
  }
  BBH_HIP_TRY(h, hipMemcpy(st->d_W, Wp.data(), sizeof(double) * Wp.size(), hipMemcpyHostToDevice));
  BBH_HIP_TRY(h, hipMemcpy(st->d_lo, lo3.data(), sizeof(double) * lo3.size(), hipMemcpyHostToDevice));
  BBH_HIP_TRY(h, hipMemset(st->d_E, 0, sizeof(double) * mp * 16));
  // (per device and cheap: set whenever a model is set up, not once per process)
  BBH_HIP_TRY(h, hipFuncSetAttribute((const void*)bbh_rff_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * 4 * 64 * PD_LD)));
  return st->Dh <= 64 ? rff_posterior_lds_attr(h, st->Dh) : 0;
}

__global__ void bbh_rff_resid_kernel(const double* __restrict__ ystd, const double* __restrict__ theta, int n, int64_t nr, double* __restrict__ r) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nr) r[i] = i < n ? ystd[i] - theta[RFF_TH_MEAN] : 0.0;
}

// theta (already in h->d_theta) -> Phi, r, B, L^-1, b, t, a, scalars, B^-1
static void bbh_rff_core(bbh_handle* h, bbh_rff_state* st) {
  hipStream_t s = h->stream;
  const int dn = st->dn, Dh = st->Dh, mp = st->mp, D = st->D;
  const int64_t nr = st->nr, n = h->n;
  hipLaunchKernelGGL(bbh_rff_scale_kernel, dim3((unsigned)((dn * Dh + 255) / 256)), dim3(256), 0, s, st->d_W, h->d_theta, dn, Dh, st->d_wS);
  hipLaunchKernelGGL(bbh_rff_features_kernel, dim3((unsigned)((nr * Dh + 255) / 256)), dim3(256), 0, s, h->d_xnT, h->np, st->d_wS, (int)n, nr, dn, D, Dh,
                     st->d_Phi);
  hipLaunchKernelGGL(bbh_rff_resid_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, s, h->d_ystd, h->d_theta, (int)n, nr, h->d_r);  // (nr == np)
  bbh_gemm(s, true, false, mp, mp, nr, 1.0, st->d_Phi, mp, 0, st->d_Phi, mp, 0, 0.0, st->d_B, mp, 0, 1);  // Phi^T Phi
  bbh_matvec_t(s, st->d_Phi, mp, n, mp, h->d_r, st->d_vec);                                              // b = Phi^T r
  hipMemsetAsync(st->d_info, 0, sizeof(int), s);
  if (mp <= 128) {
    hipLaunchKernelGGL(bbh_rff_solve_kernel, dim3(1), dim3(256), sizeof(double) * 4 * 64 * PD_LD, s, st->d_B, mp, D, Dh, h->d_theta,
                       h->desc.use_outputscale, st->d_Linv, st->d_info);
  } else {  // 4 x 4 or 8 x 8 tiles: the blocked factorisation + inverse on the model's own buffers
    hipLaunchKernelGGL(bbh_rff_diag_kernel, dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, s, st->d_B, mp, D, Dh, h->d_theta, h->desc.use_outputscale);
    bbh_potrf_trtri_buf(s, st->d_B, mp, st->d_Dg, st->d_Linv, st->d_tmp, st->d_info);
  }
  hipLaunchKernelGGL(bbh_rff_vec_kernel, dim3(1), dim3(256), 0, s, st->d_Linv, mp, D, Dh, st->d_vec);
  bbh_gemm(s, true, false, mp, mp, mp, 1.0, st->d_Linv, mp, 0, st->d_Linv, mp, 0, 0.0, st->d_Binv, mp, 0, 1);  // B^-1 = L^-T L^-1
}

// One evaluation of the fit objective on h->stream (theta in h->pin_theta; results into h->pin_out / h->pin_info).
int bbh_rff_fit_enqueue(bbh_handle* h) {
  auto* st = (bbh_rff_state*)h->rff_state;
  hipStream_t s = h->stream;
  const int dn = st->dn, Dh = st->Dh, mp = st->mp, D = st->D;
  const int64_t nr = st->nr, n = h->n;
  const int64_t tl = bbh_theta_len(h);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_theta, h->pin_theta, sizeof(double) * tl, hipMemcpyHostToDevice, s));
  bbh_rff_core(h, st);
  double* r = h->d_r;
  hipLaunchKernelGGL(bbh_rff_alpha_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, s, st->d_Phi, mp, st->d_vec + 2 * mp, r, h->d_theta, (int)n, nr,
                     st->d_alpha);
  bbh_gemm(s, false, false, nr, mp, mp, 1.0, st->d_Phi, mp, 0, st->d_Binv, mp, 0, 0.0, st->d_T, mp, 0, 1);  // Phi B^-1
  hipLaunchKernelGGL(bbh_rff_dp_kernel, dim3((unsigned)((nr * Dh + 255) / 256)), dim3(256), 0, s, st->d_Phi, st->d_vec + 2 * mp, st->d_alpha, (int)n, nr, D,
                     Dh, st->d_T);
  const int nchunks = (int)(nr / 64);
  hipLaunchKernelGGL(bbh_rff_gls_kernel, dim3((unsigned)dn, (unsigned)nchunks), dim3(256), 0, s, h->d_xnT, h->np, st->d_W, st->d_T, (int)n, D, Dh, nchunks,
                     st->d_gl);
  hipLaunchKernelGGL(bbh_rff_value_kernel, dim3(1), dim3(256), 0, s, r, st->d_alpha, st->d_vec, st->d_gl, nchunks, h->d_theta, h->desc.use_outputscale, (int)n,
                     mp, D, dn, h->d_out);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->pin_out, h->d_out, sizeof(double) * (1 + tl), hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(h->pin_info, st->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  return 0;
}

__global__ void bbh_rff_pack_kernel(const double* __restrict__ Linv, const double* __restrict__ avec, const double* __restrict__ theta, int mp,
                                    double* __restrict__ Qp, double* __restrict__ E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= mp * mp) return;
  const int k = e / mp, o = e % mp, NB = mp / 16, kb = k / 16;
  if (o == 0) E[k * 16] = avec[k];
  if (o < 16 * kb) return;
  Qp[rff_qp_off(NB, kb) + (k % 16) * rff_qp_pitch(NB, kb) + (o - 16 * kb)] = sqrt(theta[RFF_TH_NOISE]) * Linv[(int64_t)o * mp + k];
}

__global__ void bbh_rff_ecol0_kernel(const double* __restrict__ avec, int mp, double* __restrict__ E) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < mp) E[k * 16] = avec[k];
}

// bbh_factorize for the RFF model: theta (host, already validated) -> every posterior operand
int bbh_rff_factorize(bbh_handle* h) {
  auto* st = (bbh_rff_state*)h->rff_state;
  hipStream_t s = h->stream;
  bbh_rff_core(h, st);
  const int mp = st->mp;
  BBH_HIP_TRY(h, hipMemsetAsync(st->d_E, 0, sizeof(double) * mp * 16, s));
  if (st->Dh <= 64) {
    BBH_HIP_TRY(h, hipMemsetAsync(st->d_Qp, 0, sizeof(double) * rff_qp_elems(mp / 16), s));
    hipLaunchKernelGGL(bbh_rff_pack_kernel, dim3((unsigned)((mp * mp + 255) / 256)), dim3(256), 0, s, st->d_Linv, st->d_vec + 2 * mp, h->d_theta, mp, st->d_Qp,
                       st->d_E);
  } else {  // (the chunked posterior reads L^-1 itself; only the mean column of E is needed)
    hipLaunchKernelGGL(bbh_rff_ecol0_kernel, dim3((unsigned)((mp + 255) / 256)), dim3(256), 0, s, st->d_vec + 2 * mp, mp, st->d_E);
  }
  int info = 0;
  BBH_HIP_TRY(h, hipMemcpyAsync(&info, st->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  BBH_HIP_TRY(h, hipGetLastError());
  if (info != 0) {
    h->err = "bbh_factorize: the RFF model's feature-space matrix is not positive definite";
    return -4;
  }
  return 0;
}

// ---- candidates --------------------------------------------------------------------------------------------------------------
struct RffArgs {
  const double* X;
  int64_t N, ldx;
  const double* wS;   // [dn][4][DH / 4]
  const double* lo;   // lo [dn] | 1 / (hi - lo) [dn] | numcol [dn]
  const double* Qp;
  const double* E;
  double* mean;
  double* var;
  double* cross;
  int dn, D, p;
  double ybar, ysd, cmean;
};

// One wave: 32 candidates per pass (two 16-row MFMA tiles).  Lane l = (g = l >> 4, c = l & 15) featurises candidate c of each tile for the
// feature columns k = 4 ks + g - exactly its share of the A operand -, then walks the output blocks: variance = sum_o u_o^2 with
// u = sqrt(s2) L^-1 z (k-blocks above the diagonal block skipped), mean / cross-covariances = z . [a | s2 B^-1 phi_p].
// (DH = 64: 64 feature values per lane and tile pair - four waves per workgroup, one per SIMD, so that they stay in registers)
template <int DH>
__global__ __launch_bounds__(DH == 64 ? 256 : 512) __attribute__((amdgpu_waves_per_eu(1, 2))) void bbh_rff_posterior_kernel(const RffArgs a) {
  constexpr int NB = DH / 8, KS = DH / 2, TQ = DH / 4, MP = 2 * DH, NT = DH == 64 ? 256 : 512, CPW = NT / 2;  // CPW: candidates per workgroup pass
  extern __shared__ __attribute__((aligned(16))) double s_post[];
  double* sQ = s_post;
  double* sE = sQ + rff_qp_elems(NB);
  double* sW = sE + MP * 16;
  double* sL = sW + a.dn * DH;
  const int t = threadIdx.x;
  for (int e = t; e < rff_qp_elems(NB); e += NT) sQ[e] = a.Qp[e];
  for (int e = t; e < MP * 16; e += NT) sE[e] = a.E[e];
  for (int e = t; e < a.dn * DH; e += NT) sW[e] = a.wS[e];
  for (int e = t; e < 3 * a.dn; e += NT) sL[e] = a.lo[e];
  __syncthreads();
  const int l = t & 63, w = t >> 6, g = l >> 4, c16 = l & 15;
  const double rsD = 1.0 / sqrt((double)a.D);
  const double ysd2 = a.ysd * a.ysd;
  for (int64_t tile = blockIdx.x; tile * CPW < a.N; tile += gridDim.x) {
    const int64_t base = tile * CPW + w * 32;
    if (base >= a.N) continue;
    double z[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
      int64_t cand = base + 16 * mt + c16;
      cand = cand < a.N ? cand : a.N - 1;
      const double* xr = a.X + cand * a.ldx;
      double pj[TQ];
#pragma unroll
      for (int q = 0; q < TQ; q++) pj[q] = 0.0;
      for (int i = 0; i < a.dn; i++) {
        const double xi = (xr[(int)sL[2 * a.dn + i]] - sL[i]) * sL[a.dn + i];
        const double* wr = sW + (i * 4 + g) * TQ;
#pragma unroll
        for (int q = 0; q < TQ; q++) pj[q] = fma(xi, wr[q], pj[q]);
      }
#pragma unroll
      for (int q = 0; q < TQ; q++) {
        double sn, cs;
        sincos(pj[q], &sn, &cs);
        const double sc = (4 * q + g) < a.D ? rsD : 0.0;
        z[mt][q] = cs * sc;
        z[mt][TQ + q] = sn * sc;
      }
    }
    double vp[2][4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    if (a.var) {
#pragma unroll
      for (int ob = 0; ob < NB; ob++) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4 * (ob + 1); ks++) {
          const int kb = ks >> 2;
          const double bq = sQ[rff_qp_off(NB, kb) + (4 * (ks & 3) + g) * rff_qp_pitch(NB, kb) + 16 * (ob - kb) + c16];
          acc0 = mfma_f64(z[0][ks], bq, acc0);
          acc1 = mfma_f64(z[1][ks], bq, acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          vp[0][r] = fma(acc0[r], acc0[r], vp[0][r]);
          vp[1][r] = fma(acc1[r], acc1[r], vp[1][r]);
        }
      }
    }
    d4 e0 = {0.0, 0.0, 0.0, 0.0}, e1 = {0.0, 0.0, 0.0, 0.0};
    if (a.mean || a.cross) {
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        const double be = sE[(4 * ks + g) * 16 + c16];
        e0 = mfma_f64(z[0][ks], be, e0);
        e1 = mfma_f64(z[1][ks], be, e1);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t cand = base + 16 * mt + g + 4 * r;
        double v = vp[mt][r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (cand >= a.N) continue;
        const double ev = mt == 0 ? e0[r] : e1[r];
        if (c16 == 0) {
          if (a.var) a.var[cand] = ysd2 * v;
          if (a.mean) a.mean[cand] = a.ybar + a.ysd * (a.cmean + ev);
        } else if (a.cross && c16 <= a.p) {
          a.cross[cand * a.p + (c16 - 1)] = ysd2 * ev;
        }
      }
  }
}

static int rff_posterior_lds_attr(bbh_handle* h, int Dh) {
  const void* fn = Dh == 64 ? (const void*)bbh_rff_posterior_kernel<64> : (const void*)bbh_rff_posterior_kernel<32>;
  BBH_HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(sizeof(double) * ((size_t)rff_qp_elems(Dh / 8) + 2 * Dh * 16 + RFF_MAXDN * (Dh + 3)))));
  return 0;
}

static size_t rff_post_lds(int Dh, int dn) { return sizeof(double) * ((size_t)rff_qp_elems(Dh / 8) + (size_t)2 * Dh * 16 + (size_t)dn * Dh + 3 * (size_t)dn); }

__global__ void bbh_rff_features_raw_kernel(const double* __restrict__ X, int64_t ldx, int64_t q, int64_t qpad, const double* __restrict__ lo3,
                                            const double* __restrict__ wS, int dn, int D, int Dh, double* __restrict__ Fq);

// ---- candidates, 64 < D <= 256: chunked form ---------------------------------------------------------------------------------------
// var[i] = scale * sum_k V[i][k]^2   (one wave per row, fixed summation order)
__global__ __launch_bounds__(256) void bbh_rff_rowsq_kernel(const double* __restrict__ V, int mp, int64_t rows, double scale, double* __restrict__ var) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  double s = 0.0;
  for (int k = lane; k < mp; k += 64) {
    const double v = V[row * mp + k];
    s = fma(v, v, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) var[row] = scale * s;
}

// mean[i] = ybar + ysd (c + Z[i] . E[:, 0]),  cross[i][c - 1] = ysd^2 Z[i] . E[:, c]  for c = 1 .. p   (one wave per row)
__global__ __launch_bounds__(256) void bbh_rff_me_kernel(const double* __restrict__ Z, const double* __restrict__ E, int mp, int64_t rows, int p,
                                                         double ybar, double ysd, double cmean, double* __restrict__ mean, double* __restrict__ cross) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c0 = mean ? 0 : 1, c1 = cross ? p : 0;
  for (int c = c0; c <= c1; c++) {
    double s = 0.0;
    for (int k = lane; k < mp; k += 64) s = fma(Z[row * mp + k], E[k * 16 + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) {
      if (c == 0)
        mean[row] = ybar + ysd * (cmean + s);
      else
        cross[row * p + (c - 1)] = ysd * ysd * s;
    }
  }
}

static int bbh_rff_posterior_chunked(bbh_handle* h, bbh_rff_state* st, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev,
                                     double* cross_dev) {
  // features of a chunk of candidates in global memory, then two GEMM-shaped products: 2 m^2 flops per candidate either way; what the
  // fused kernel keeps in registers (64 feature values per lane at D = 64) would be 256 per lane at D = 256
  hipStream_t s = h->stream;
  const int mp = st->mp, Dh = st->Dh;
  const int64_t chunk = 8192;
  int rc = bbh_ensure_ws(h, sizeof(double) * 2 * (size_t)chunk * mp);
  if (rc) return rc;
  double* Z = h->d_ws;
  double* V = Z + chunk * mp;
  const double s2 = h->theta[RFF_TH_NOISE];
  bbh_timed_scope scope(h, cross_dev ? BBH_TIMED_CROSS : BBH_TIMED_POSTERIOR);
  for (int64_t off = 0; off < N; off += chunk) {
    const int64_t cn = N - off < chunk ? N - off : chunk, cpad = bbh_round_up(cn, 64);
    hipLaunchKernelGGL(bbh_rff_features_raw_kernel, dim3((unsigned)((cpad * Dh + 255) / 256)), dim3(256), 0, s, X_dev + off * ldx, ldx, cn, cpad, st->d_lo,
                       st->d_wS, st->dn, st->D, Dh, Z);
    if (var_dev) {
      bbh_gemm(s, false, true, cpad, mp, mp, 1.0, Z, mp, 0, st->d_Linv, mp, 0, 0.0, V, mp, 0, 1);  // V = Z L^-T
      hipLaunchKernelGGL(bbh_rff_rowsq_kernel, dim3((unsigned)((cn + 3) / 4)), dim3(256), 0, s, V, mp, cn, h->ysd * h->ysd * s2, var_dev + off);
    }
    if (mean_dev || cross_dev)
      hipLaunchKernelGGL(bbh_rff_me_kernel, dim3((unsigned)((cn + 3) / 4)), dim3(256), 0, s, Z, st->d_E, mp, cn, h->p, h->ybar, h->ysd,
                         h->theta[RFF_TH_MEAN], mean_dev ? mean_dev + off : nullptr, cross_dev ? cross_dev + off * h->p : nullptr);
  }
  BBH_HIP_TRY(h, hipGetLastError());
  h->last_form = 6;
  return 0;
}

int bbh_rff_posterior_launch(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev, double* cross_dev) {
  if (N <= 0) return 0;
  auto* st = (bbh_rff_state*)h->rff_state;
  if (!st) {
    h->err = "RFF model state missing";
    return -1;
  }
  if (st->Dh > 64) return bbh_rff_posterior_chunked(h, st, X_dev, N, ldx, mean_dev, var_dev, cross_dev);
  RffArgs a;
  a.X = X_dev;
  a.N = N;
  a.ldx = ldx;
  a.wS = st->d_wS;
  a.lo = st->d_lo;
  a.Qp = st->d_Qp;
  a.E = st->d_E;
  a.mean = mean_dev;
  a.var = var_dev;
  a.cross = cross_dev;
  a.dn = st->dn;
  a.D = st->D;
  a.p = h->p;
  a.ybar = h->ybar;
  a.ysd = h->ysd;
  a.cmean = h->theta[RFF_TH_MEAN];
  const size_t lds = rff_post_lds(st->Dh, st->dn);
  const int which = st->Dh == 64 ? 1 : 0;
  const int threads = which ? 256 : 512, cpw = threads / 2;
  const int64_t tiles = (N + cpw - 1) / cpw;
  const unsigned grid = (unsigned)(tiles < h->num_cu ? tiles : h->num_cu);
  bbh_timed_scope scope(h, cross_dev ? BBH_TIMED_CROSS : BBH_TIMED_POSTERIOR);
  if (which)
    hipLaunchKernelGGL(bbh_rff_posterior_kernel<64>, dim3(grid), dim3(256), lds, h->stream, a);
  else
    hipLaunchKernelGGL(bbh_rff_posterior_kernel<32>, dim3(grid), dim3(512), lds, h->stream, a);
  BBH_HIP_TRY(h, hipGetLastError());
  h->last_form = 6;
  return 0;
}

// ---- pending points / joint posterior of a small point set ---------------------------------------------------------------------
// features of q raw rows (zero rows beyond q): Fq [qpad][mp]
__global__ void bbh_rff_features_raw_kernel(const double* __restrict__ X, int64_t ldx, int64_t q, int64_t qpad, const double* __restrict__ lo3,
                                            const double* __restrict__ wS, int dn, int D, int Dh, double* __restrict__ Fq) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= qpad * Dh) return;
  const int64_t a = e / Dh;
  const int j = (int)(e % Dh);
  double c = 0.0, s = 0.0;
  if (a < q && j < D) {
    double p = 0.0;
    for (int i = 0; i < dn; i++) {
      const double xi = (X[a * ldx + (int)lo3[2 * dn + i]] - lo3[i]) * lo3[dn + i];
      p = fma(xi, wS[i * Dh + (j & 3) * (Dh / 4) + (j >> 2)], p);
    }
    sincos(p, &s, &c);
    const double sc = 1.0 / sqrt((double)D);
    c *= sc;
    s *= sc;
  }
  Fq[a * (2 * Dh) + j] = c;
  Fq[a * (2 * Dh) + Dh + j] = s;
}

// mean [qpad] (target scale), V = Fq L^-T [qpad][mp], cov = ysd^2 s2 V V^T [qpad][qpad]; optionally G = s2 B^-1 Fq^T into E's columns 1..q
static int bbh_rff_small_posterior(bbh_handle* h, const double* Xq_host, int64_t q, std::vector<double>& mean, std::vector<double>& cov, bool set_E) {
  auto* st = (bbh_rff_state*)h->rff_state;
  hipStream_t s = h->stream;
  const int64_t d = h->desc.d, qpad = bbh_round_up(q, 64);
  const int mp = st->mp, Dh = st->Dh;
  const size_t need = sizeof(double) * ((size_t)qpad * d + 2 * (size_t)qpad * mp + (size_t)qpad * qpad + qpad);
  int rc = bbh_ensure_ws(h, need);
  if (rc) return rc;
  double* dX = h->d_ws;
  double* Fq = dX + qpad * d;
  double* V = Fq + qpad * mp;
  double* C = V + qpad * mp;
  double* dm = C + qpad * qpad;
  BBH_HIP_TRY(h, hipMemcpyAsync(dX, Xq_host, sizeof(double) * q * d, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(bbh_rff_features_raw_kernel, dim3((unsigned)((qpad * Dh + 255) / 256)), dim3(256), 0, s, dX, d, q, qpad, st->d_lo, st->d_wS, st->dn, st->D,
                     Dh, Fq);
  bbh_gemm(s, false, true, qpad, mp, mp, 1.0, Fq, mp, 0, st->d_Linv, mp, 0, 0.0, V, mp, 0, 1);  // V = Fq L^-T
  bbh_gemm(s, false, true, qpad, qpad, mp, 1.0, V, mp, 0, V, mp, 0, 0.0, C, qpad, 0, 1);          // V V^T
  bbh_matvec(s, Fq, mp, qpad, mp, st->d_vec + 2 * mp, dm);                                        // Fq a
  std::vector<double> hc((size_t)qpad * qpad), hm(qpad), hg;
  BBH_HIP_TRY(h, hipMemcpyAsync(hc.data(), C, sizeof(double) * qpad * qpad, hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(hm.data(), dm, sizeof(double) * qpad, hipMemcpyDeviceToHost, s));
  if (set_E) {  // G = s2 B^-1 Fq^T = s2 L^-T V^T: [mp][qpad] = (V L^-1)^T; computed as V L^-1 [qpad][mp] and scattered on the host (q <= 15)
    bbh_gemm(s, false, false, qpad, mp, mp, 1.0, V, mp, 0, st->d_Linv, mp, 0, 0.0, Fq, mp, 0, 1);
    hg.resize((size_t)qpad * mp);
    BBH_HIP_TRY(h, hipMemcpyAsync(hg.data(), Fq, sizeof(double) * qpad * mp, hipMemcpyDeviceToHost, s));
  }
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  const double s2 = h->theta[RFF_TH_NOISE], y2 = h->ysd * h->ysd;
  mean.resize(q);
  cov.resize((size_t)q * q);
  for (int64_t i = 0; i < q; i++) {
    mean[i] = h->ybar + h->ysd * (h->theta[RFF_TH_MEAN] + hm[i]);
    for (int64_t j = 0; j < q; j++) cov[i * q + j] = y2 * s2 * 0.5 * (hc[i * qpad + j] + hc[j * qpad + i]);
  }
  if (set_E) {
    std::vector<double> E((size_t)mp * 16, 0.0), a0(mp);
    BBH_HIP_TRY(h, hipMemcpy(a0.data(), st->d_vec + 2 * mp, sizeof(double) * mp, hipMemcpyDeviceToHost));
    for (int k = 0; k < mp; k++) {
      E[(size_t)k * 16] = a0[k];
      for (int64_t c = 0; c < q; c++) E[(size_t)k * 16 + 1 + c] = s2 * hg[(size_t)c * mp + k];
    }
    BBH_HIP_TRY(h, hipMemcpy(st->d_E, E.data(), sizeof(double) * E.size(), hipMemcpyHostToDevice));
  }
  return 0;
}

int bbh_rff_pending_set(bbh_handle* h, const double* Xpend_host, int64_t p, double* mean_p_host, double* cov_pp_host) {
  auto* st = (bbh_rff_state*)h->rff_state;
  const int d = h->desc.d;
  h->pend_host.assign(Xpend_host, Xpend_host + p * d);
  h->p = (int)p;
  h->pend_mean.clear();
  h->pend_cov.clear();
  if (p == 0) {  // columns 1.. of E are ignored with p = 0
    (void)st;
    return 0;
  }
  int rc = bbh_rff_small_posterior(h, Xpend_host, p, h->pend_mean, h->pend_cov, true);
  if (rc) return rc;
  if (mean_p_host) memcpy(mean_p_host, h->pend_mean.data(), sizeof(double) * p);
  if (cov_pp_host) memcpy(cov_pp_host, h->pend_cov.data(), sizeof(double) * p * p);
  return 0;
}

int bbh_rff_posterior_joint(bbh_handle* h, const double* Xq_host, int64_t q, double* mean_host, double* cov_host) {
  std::vector<double> m, c;
  int rc = bbh_rff_small_posterior(h, Xq_host, q, m, c, false);
  if (rc) return rc;
  memcpy(mean_host, m.data(), sizeof(double) * q);
  memcpy(cov_host, c.data(), sizeof(double) * q * q);
  return 0;
}
