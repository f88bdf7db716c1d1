// Internal declarations shared by the translation units of libbaybe_hip.so.
// gfx950 only: 64-lane wavefronts and the fp64 MFMA v_mfma_f64_16x16x4_f64 are assumed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/baybe_hip.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define BBH_WAVE 64
#define BBH_TB 16          // training points per MFMA block (M/N of the 16x16x4 tile)
#define BBH_PAD 64         // n is padded to a multiple of this (GEMM tile edge)
#define BBH_MEANCOLS 16    // columns of the mean/cross operand: [alpha | -beta_1 .. -beta_15]
#define BBH_SQRT3 1.7320508075688772
#define BBH_SQRT5 2.23606797749979

// ---- fragment layout of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md §3) ----------
//   A (16x4): lane l holds A[l & 15][l >> 4]
//   B (4x16): lane l holds B[l >> 4][l & 15]
//   C/D (16x16): lane l, reg r holds C[(l >> 4) + 4 r][l & 15]
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

#define BBH_HIP_TRY(h, expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)

// How the elementwise kernels (Gram matrix, gradient pairs, K*, K_qq) assemble one kernel value from theta; by value.
#define BBH_MAX_FACTORS 4
struct bbh_kern_spec {
  int F;                        // factors; 1 = the plain single-kernel model
  int combine;                  // 0 product, 1 sum, 2 sum of products (grp)
  int grp[BBH_MAX_FACTORS];     // term of the sum each factor multiplies into: k = sum_g prod_{f in g} u_f  (product: all 0; sum: f)
  int kind[BBH_MAX_FACTORS];    // enum bbh_kernel_kind
  int ls_off[BBH_MAX_FACTORS];  // theta offset of the factor's dn lengthscales
  int fos_off;                  // theta offset of the F per-factor outputscales (-1: F == 1)
  int use_os;                   // outer outputscale theta[2]
  int jb;                       // floor(dn / 2) + 1 (piecewise-polynomial kernels)
  int alpha_off;                // theta offset of the F alpha parameters (-1: no RQ factor)
  int per_off;                  // theta offset of the F * dn period lengths (-1: no periodic factor)
};
// gpytorch PiecewisePolynomialKernel: k = (1 - r)_+^(j + q) P_q(r) with j = jb + q, jb = floor(dn / 2) + 1, and
// g = -(dk/dr)/r = (1 - r)_+^(j + q - 1) Q_q(r) in closed form (no cancellation at r -> 0 for q >= 1)
__host__ __device__ inline double bbh_powi(double x, int n) {
  double acc = 1.0;
  for (int i = 0; i < n; i++) acc *= x;
  return acc;
}
__host__ __device__ inline double bbh_piecewise(int q, int jb, double r2, bool want_g) {
  const double r = sqrt(r2 > 1e-30 ? r2 : 1e-30);  // gpytorch's distance: sqrt(clamp_min(r^2, 1e-30))
  if (!(r < 1.0)) return 0.0;
  const double j = (double)(jb + q), u = 1.0 - r;
  if (!want_g) {
    double P = 1.0;
    if (q == 1) P = fma(j + 1.0, r, 1.0);
    if (q == 2) P = fma(fma((j * j + 4.0 * j + 3.0) / 3.0, r, j + 2.0), r, 1.0);
    if (q == 3)
      P = fma(fma(fma((j * j * j + 9.0 * j * j + 23.0 * j + 15.0) / 15.0, r, (6.0 * j * j + 36.0 * j + 45.0) / 15.0), r, j + 3.0), r, 1.0);
    return bbh_powi(u, jb + 2 * q) * P;
  }
  const double m = bbh_powi(u, jb + 2 * q - 1);
  if (q == 0) return r2 > 1e-30 ? j * m / r : 0.0;  // not differentiable at r = 0 (as Matern-1/2)
  if (q == 1) return (j + 1.0) * (j + 2.0) * m;
  if (q == 2) return (j + 3.0) * (j + 4.0) * fma(j + 1.0, r, 1.0) / 3.0 * m;
  return (j + 5.0) * (j + 6.0) * fma(fma(j * j + 4.0 * j + 3.0, r, 3.0 * j + 6.0), r, 3.0) / 15.0 * m;
}
// base kernel value as a function of the scaled squared distance (jb: only the piecewise-polynomial family needs it;
// alpha: only the RQ kernel)
// (not inlined: nine kinds with their libm calls, called once per factor and entry - inlined into the unrolled K* kernel it
// made 23 000 instructions, far beyond the instruction cache)
// dot-product kinds: the per-factor metric is s = sum_j (x_j / w_j)(x'_j / w_j) instead of a scaled squared distance
#define BBH_KIND_IS_DOT(kind) ((kind) >= BBH_KERNEL_LINEAR && (kind) <= BBH_KERNEL_POLY4)
#define BBH_KIND_IS_POLY(kind) ((kind) >= BBH_KERNEL_POLY1 && (kind) <= BBH_KERNEL_POLY4)
#define BBH_KIND_HAS_ALPHA(kind) ((kind) == BBH_KERNEL_RQ || BBH_KIND_IS_POLY(kind))  // RQ alpha / polynomial offset slot
// one dimension's contribution to a factor's metric
__host__ __device__ __forceinline__ double bbh_metric_term(int kind, double xa, double xb, double invw) {
  if (BBH_KIND_IS_DOT(kind)) return (xa * invw) * (xb * invw);
  const double df = (xa - xb) * invw;
  return df * df;
}
// the same for factor f of a model: periodic factors read their period from theta's period block (invw = 1 / l_j there)
__host__ __device__ __forceinline__ double bbh_metric_term_f(const bbh_kern_spec& ks, const double* theta, int f, int j, int dn,
                                                             double xa, double xb, double invw) {
  if (ks.kind[f] == BBH_KERNEL_PERIODIC) {
    const double sn = sin(M_PI * (xa - xb) / theta[ks.per_off + f * dn + j]);
    return sn * sn * invw;
  }
  return bbh_metric_term(ks.kind[f], xa, xb, invw);
}
__host__ __device__ __attribute__((noinline)) inline double bbh_kbase(int kind, double r2, int jb, double alpha = 1.0) {
  if (kind == BBH_KERNEL_LINEAR) return r2;  // (the ARD variances are the weights of the metric)
  if (BBH_KIND_IS_POLY(kind)) return bbh_powi(r2 + alpha, kind - BBH_KERNEL_POLY1 + 1);
  if (kind == BBH_KERNEL_PERIODIC) return exp(-2.0 * r2);  // the metric is sum_j sin^2(pi Delta_j / p_j) / l_j
  if (kind == BBH_KERNEL_RBF) return exp(-0.5 * r2);
  if (kind == BBH_KERNEL_RQ) return exp(-alpha * log1p(r2 / (2.0 * alpha)));
  if (kind >= BBH_KERNEL_PIECEWISE0 && kind <= BBH_KERNEL_PIECEWISE3) return bbh_piecewise(kind - BBH_KERNEL_PIECEWISE0, jb, r2, false);
  const double r = sqrt(r2);
  if (kind == BBH_KERNEL_MATERN52) return (1.0 + BBH_SQRT5 * r + (5.0 / 3.0) * r2) * exp(-BBH_SQRT5 * r);
  if (kind == BBH_KERNEL_MATERN32) return (1.0 + BBH_SQRT3 * r) * exp(-BBH_SQRT3 * r);
  return exp(-r);
}
// composite value from the per-factor squared distances: (prod | sum)_f os_f k_f(r2_f), without the outer scale.
// (Reference to a fixed-size array and compile-time indices: with a pointer and a run-time loop bound the callers' r2
// arrays went to scratch memory, and the K* kernel ran at 5 TFLOP/s.)
// ---- stationary kernels as functions of the scaled squared distance -----------------------
// g(r) = -(dk/dr)/r, so that dk/dl_j = g(r) * Delta_j^2 / l_j^3
__device__ __forceinline__ double bbh_gfun(int kind, double r2, int jb, double alpha) {
  // dot-product kinds, k = f(s) with s = sum_j x_j x'_j / w_j^2: dk/dw_j = -2 f'(s) x_j x'_j / w_j^3, i.e. g = -2 f'(s) with the
  // product x_j x'_j in the place of Delta_j^2 (the Linear kernel's ARD variances are v_j = 1 / w_j^2)
  if (kind == BBH_KERNEL_LINEAR) return -2.0;
  if (BBH_KIND_IS_POLY(kind)) return -2.0 * (double)(kind - BBH_KERNEL_POLY1 + 1) * bbh_powi(r2 + alpha, kind - BBH_KERNEL_POLY1);
  // periodic, k = exp(-2 sum_j sin^2(u_j) / l_j): dk/dl_j = 2 k sin^2(u_j) / l_j^2 - g = 2 k, the slot's term is assembled at the call
  if (kind == BBH_KERNEL_PERIODIC) return 2.0 * exp(-2.0 * r2);
  if (kind == BBH_KERNEL_RBF) return exp(-0.5 * r2);
  if (kind == BBH_KERNEL_RQ) return exp(-(alpha + 1.0) * log1p(r2 / (2.0 * alpha)));  // (1 + u)^-(alpha + 1), u = r^2 / (2 alpha)
  if (kind >= BBH_KERNEL_PIECEWISE0) return bbh_piecewise(kind - BBH_KERNEL_PIECEWISE0, jb, r2, true);
  const double r = sqrt(r2);
  if (kind == BBH_KERNEL_MATERN52) return (5.0 / 3.0) * (1.0 + BBH_SQRT5 * r) * exp(-BBH_SQRT5 * r);
  if (kind == BBH_KERNEL_MATERN32) return 3.0 * exp(-BBH_SQRT3 * r);
  return r > 0.0 ? exp(-r) / r : 0.0;
}

// theta layout: [noise, mean, outputscale, ls[dn], B[T*T], (hadamard: noise_t[T], mean_t[T])]
// hoff = offset of noise_t (mean_t follows at hoff + T), -1 = the scalar slots are in use

// k = sum over the terms g of prod_{f in g} u[f] (ProductKernel: one term; AdditiveKernel: one factor per term; a sum whose members
// are products or single kernels: kernels/composite.py:60-91 nested).  u[f] = os_f k_f(x, x').  Every array index below is a
// compile-time constant after unrolling: indexing term[grp[f]] with the run-time group would put the local arrays into scratch
// memory (measured: the two-factor fused kernel 6.7 -> 11.6 ms per 1e6 candidates); the group tests are wave-uniform.
template <class GRP>
__host__ __device__ __forceinline__ double bbh_combine(int F, int combine, const GRP& grp, const double* u) {
  if (combine == 0) {
    double acc = 1.0;
#pragma unroll
    for (int f = 0; f < BBH_MAX_FACTORS; f++)
      if (f < F) acc *= u[f];
    return acc;
  }
  if (combine == 1) {
    double acc = 0.0;
#pragma unroll
    for (int f = 0; f < BBH_MAX_FACTORS; f++)
      if (f < F) acc += u[f];
    return acc;
  }
  double acc = 0.0;
#pragma unroll
  for (int g = 0; g < BBH_MAX_FACTORS; g++) {
    double term = 1.0;
    bool used = false;
#pragma unroll
    for (int f = 0; f < BBH_MAX_FACTORS; f++)
      if (f < F && grp[f] == g) {
        term *= u[f];
        used = true;
      }
    if (used) acc += term;
  }
  return acc;
}
// d k / d u_f = prod of the other factors of f's term
template <class GRP>
__host__ __device__ __forceinline__ double bbh_combine_weight(int F, int combine, const GRP& grp, const double* u, int f) {
  if (combine == 1) return 1.0;
  double w = 1.0;
#pragma unroll
  for (int g = 0; g < BBH_MAX_FACTORS; g++)
    if (g < F && g != f && (combine == 0 || grp[g] == grp[f])) w *= u[g];
  return w;
}
__device__ __forceinline__ double bbh_kcomp(const bbh_kern_spec& ks, const double* __restrict__ theta,
                                            const double (&r2)[BBH_MAX_FACTORS]) {
  if (ks.F <= 1) return bbh_kbase(ks.kind[0], r2[0], ks.jb, ks.alpha_off >= 0 ? theta[ks.alpha_off] : 1.0);
  double u[BBH_MAX_FACTORS] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
  for (int f = 0; f < BBH_MAX_FACTORS; f++)
    if (f < ks.F)
      u[f] = theta[ks.fos_off + f] * bbh_kbase(ks.kind[f], r2[f], ks.jb, ks.alpha_off >= 0 ? theta[ks.alpha_off + f] : 1.0);
  return bbh_combine(ks.F, ks.combine, ks.grp, u);
}

struct bbh_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // ---- model description ----
  bbh_model_desc desc{};
  std::string model_sig;  // shape signature of the current model: a refit with the same one keeps every device buffer
  bool have_model = false;
  bool factorized = false;
  int64_t n = 0;      // training points
  int64_t np = 0;     // padded to BBH_PAD
  int64_t nb = 0;     // np / 16
  int dn = 0;         // numerical columns
  int kd = 0;         // k-steps of the augmented distance GEMM: ceil((dn+2)/4)
  int T = 1;          // tasks
  bool hadamard = false;  // per-task noise and mean (theta tail), see bbh_model_desc
  int F = 1;              // kernel factors (composite kernels: 2..4), see bbh_model_desc
  std::vector<int> numcol;        // numerical column -> comp-rep column
  std::vector<double> lo, hi;     // per numerical column
  double ybar = 0.0, ysd = 1.0;
  std::vector<double> theta;      // last factorised theta
  std::vector<double> ystd_host;  // [n]
  std::vector<double> xn_host;    // [n, dn] normalised numerical columns
  std::vector<int> task_host;     // [n]
  std::vector<double> xcenter;    // [dn] column means of xn (centring as gpytorch)

  // ---- device state (all fp64 unless noted) ----
  double* d_xnT = nullptr;     // [dn, np]  normalised training inputs, transposed
  int* d_task = nullptr;       // [np]      task ids (0 for padding)
  double* d_nmask = nullptr;   // [np]      noise mask (1 = noisy observation, 0 = latent value / padding)
  double* d_pendT = nullptr;   // [dn, 16]  normalised pending points, transposed (composite-kernel path)
  double* d_colA = nullptr;    // [np, spad] alpha columns of bbh_set_mean_columns (composite-kernel path)
  int64_t colA_elems = 0;
  std::vector<double> nmask_host;
  double* d_ystd = nullptr;    // [np]
  double* d_theta = nullptr;   // [theta_len]
  double* d_K = nullptr;       // [np, np]  K + s2 I  -> L (lower, in place)
  double* d_X = nullptr;       // [np, np]  L^-1
  double* d_M = nullptr;       // [np, np]  (K + s2 I)^-1
  double* d_Mpart = nullptr;   // [4, np, np] split-K partial products of X^T X (256 <= np <= 1024)
  double* d_Q = nullptr;       // [np, np]  scratch (LOO: M diag(u) M; Msc)
  double* d_Q2 = nullptr;      // [np, np]  scratch
  double* d_D = nullptr;       // [np/64, 64, 64] inverses of the diagonal Cholesky blocks
  double* d_tmp = nullptr;     // [np/64, 64, 64] trtri scratch
  double* d_r = nullptr;       // [np] y~ - c
  double* d_t = nullptr;       // [np] scratch vector
  double* d_alpha = nullptr;   // [np]
  double* d_u = nullptr;       // [np] LOO u
  double* d_w = nullptr;       // [np] LOO w
  double* d_q = nullptr;       // [np] LOO M w
  double* d_partial = nullptr; // [np, nslots] gradient partials
  double* d_out = nullptr;     // [1 + theta_len] value + gradient
  int* d_info = nullptr;       // Cholesky failure flag
  // fused-posterior operands (built by bbh_factorize / bbh_pending_set)
  double* d_trainfrag = nullptr;  // [nb_ext, kd, 64] augmented training fragments (A of the distance GEMM)
  double* d_rfrag = nullptr;      // packed L^-T fragments (B of the variance GEMM)
  double* d_meanB = nullptr;      // [nb_ext*4, 64] fragments of [alpha | -beta] (B of the mean/cross GEMM)
  double* d_sclofs = nullptr;     // [2, dn] per-column scale/offset for candidates
  int* d_numcol = nullptr;        // [dn]
  double* d_tasktbl = nullptr;    // [T, T] outputscale * B (or [1] = outputscale)
  int* d_taskext = nullptr;       // [np_ext] task id per (training | pending) point
  int64_t rfrag_elems = 0;
  bool use_pipeline = true;       // software-pipelined fused kernel (env BBH_PIPELINE=0 -> plain form, for A/B)
  int64_t* d_pass_off = nullptr;  // [npass] element offsets of the passes in d_rfrag
  int* d_pass_w = nullptr;        // [npass] pass widths (16-column blocks)
  int npass = 0;
  int pass_w_last = 0;            // width of the last pass (its first column block = cached k-blocks)
  double* d_kvcache = nullptr;    // kernel-value cache of the multi-pass fused kernel (grow-only)
  size_t kvcache_bytes = 0;
  int* d_slab_flags = nullptr;    // claim flags of the cache slabs (zero = free)
  bool coopg_cross_on = true;     // env BBH_COOPG_CROSS=0: composite models' mean-only / cross passes through the materialised path (A/B)
  bool use_mean_valu = true;      // env BBH_MEAN_VALU=0: mean contraction through the MFMA form (A/B)
  bool use_kvcache = true;        // env BBH_KVCACHE=0: recompute kernel values in every pass (A/B)
  size_t lds_per_block = 65536;   // LDS a workgroup may use (device property; 160 KB on gfx950)
  bool pending_lds_form = false;  // env BBH_PENDING_LDS=1: generic LDS form of the pending qLogEI kernel (A/B)
  int kv_global_mode = -1;        // env BBH_KV_GLOBAL: 0 never use global slabs, 1 always, unset: by size
  int kv_lds_blocks = -1;         // env BBH_KV_LDS: cap on LDS-cached k-blocks per wave (-1 = as many as fit)
  int num_cu = 256;               // compute units of the device (sizes the slab pool of the kernel-value cache)
  int wmax = 16;                  // column blocks per pass of the windowed fused kernel (two waves per SIMD)
  bool fit_overlap = true;        // env BBH_FIT_OVERLAP=0: the inverse of the factor strictly after the factorisation (A/B)
  hipStream_t side_stream = nullptr;  // second stream of the fit (rows of L^-1 next to the trailing updates)
  hipStream_t fit_stream = nullptr;   // stream the captured evaluation graph is replayed on
  int fit_small = 1;  // env BBH_FIT_SMALL=0: evaluations of small models (np = 64, one task, one kernel, MLL) launch by launch instead of the fused one-workgroup kernel
  bool fit_small_ready = false;
  bool potrf_tiles = true, tiles_ready = false;  // env BBH_POTRF_TILES=0: per-step launches instead of the one-launch tile-dataflow factorisation (np <= 1024)
  int tile_spin_limit = 1 << 17, tile_spin_limit_set = -1;  // env BBH_TILE_SPIN: polls (~1 us each) before a waiting tile gives up (a whole factorisation takes ~200 us)
  int tiles_per_device = 0;           // workgroups of the tile kernel the device holds at once (occupancy x CUs)
  int* d_tileflags = nullptr;         // [2][16][16] publish flags of the L- and X-tiles (epoch-stamped)
  int tile_epoch = 0;
  hipGraphExec_t fit_exec = nullptr;  // one evaluation of the fit objective, captured per model (bbh_fit_value_grad)
  bool fit_graph_mode = false, fit_graph_failed = false;  // env BBH_FIT_GRAPH=1: replay the captured graph (slower, see bbh_model.hip)
  double *pin_theta = nullptr, *pin_out = nullptr;  // pinned staging of the evaluation (theta in, value + gradient out)
  int* pin_info = nullptr;
  hipEvent_t side_events[2] = {nullptr, nullptr};
  bool potrf_register_form = false;  // env BBH_POTRF_REG=1: 64x64 diagonal blocks by the one-wave register kernel (A/B)
  int coop_mode = 1;              // env BBH_COOP: 0 never use the cooperative form, 1 where it pays (default), 2 wherever instantiated
  bool coop_ready = false;        // operand slices of the cooperative form are packed for the current factorisation
  bool coopg_ready = false;       // ... of the cooperative form with the generic production (composite / RQ / piecewise models, bbh_coopg.h)
  double* d_trainfrag_f = nullptr;  // [F][nb + 1][kd][64] per-factor training fragments (coopg)
  double* d_sclofs_f = nullptr;     // [F][2][dn] per-factor candidate scale / offset (coopg)
  int64_t tf_f_elems = 0;
  bool coop2_ready = false;       // ... of the two-sweep cooperative form (512 < n <= 1024; shares d_rstream / coop_g0)
  double* d_rstream = nullptr;    // [4 waves][rstream_frags][64]
  int64_t rstream_frags = 0;
  int coop_g0 = 0;
  int last_form = -1;             // bbh_last_posterior_form
  int64_t nb_ext = 0;             // blocks incl. pending points (mean/cross pass)
  // pending state
  int p = 0;
  std::vector<double> pend_host;  // [p, d]
  std::vector<double> pend_mean;  // [p]    posterior mean of the pending points (target scale)
  std::vector<double> pend_cov;   // [p, p] posterior covariance of the pending points
  std::vector<double> xraw_host;  // [n, d] raw training rows (for bbh_train_posterior_mean)
  double* d_beta = nullptr;       // [np, 16] columns: alpha, beta_1..beta_p (dense, for packing)
  // alternative target columns (qLogNEHVI): alpha columns in fragment order
  double* d_colfrag = nullptr;    // [S/128, np/4 (+ext), 8, 64]
  int64_t colfrag_elems = 0;
  int64_t ncols = 0;              // S (padded to 128 internally)
  // fused qLogEI epilogue request (valid during one bbh_score_qlogei call)
  const double* fuse_qz = nullptr;
  int fuse_S = 0;
  double fuse_best_f = 0.0, fuse_sign = 1.0;
  const uint8_t* fuse_alive = nullptr;
  double* fuse_scores = nullptr;
  // generic workspaces
  double* d_ws = nullptr;
  size_t ws_bytes = 0;
  double* d_z = nullptr;
  size_t z_bytes = 0;
  double* h_zstage = nullptr;     // pinned staging copy of the last upload (the async copy reads it later)
  size_t zstage_bytes = 0;
  hipEvent_t z_evt = nullptr;     // recorded after the staged copy: the staging buffer may be rewritten once it fired
  double* d_red = nullptr;        // argmax partials
  int64_t* d_redi = nullptr;
  void* comm_state = nullptr;     // RCCL communicator + exchange buffers (bbh_comm.hip), null until bbh_comm_init
  double* d_rsmall = nullptr;     // register-resident small-model form (bbh_small.h): fragments of the lower triangle of L^-T
  int small_nb = 0;               // its training blocks ceil(n / 16) <= 4 once the operands are packed (0: form not available)
  bool small_on = true;           // env BBH_SMALL=0: keep the cooperative form for n <= 64 (A/B)
  int64_t slice_rows = 0;         // bbh_set_slice_rows: row count the sample-slice heuristics use instead of the local N (0: local)
  void* flow_state = nullptr;     // roles, flags and scratch of the one-launch fit evaluation (bbh_fitflow.hip)
  int flow_spin_limit = 1 << 17;  // env BBH_FLOW_SPIN: polls before a waiting role of the dataflow fit evaluation gives up
  bool skip_x_memset = false;     // bbh_potrf_trtri: leave the upper tiles of L^-1 alone (the caller reads lower tiles only)
  bool flow_in_flight = false;    // the evaluation on the stream is the one-launch form (its flag needs the sentinel check)
  long long* d_tiledbg = nullptr; // BBH_TILE_STAMPS=1: clock stamps of the Gram-building tile launch
  int tiledbg_n = 0;
  bool tile_d_sc1 = false;        // env BBH_TILE_ACQ=0: row heads take D_{I-1} through sc1 loads without an acquire fence (valid after write-through stores; measured: no gain, so the fence form stays the default)
  bool tile_wt = true;            // env BBH_TILE_WT=0: tiles handed between workgroups through plain stores + an agent-scope release fence instead of write-through (sc1) stores (A/B)
  int tiles_did_mt = 0;           // tiles of K^-1 the last tile-dataflow launch built itself (bbh_potrf_trtri_from_inputs with mt_args; block row 0 first)
  bool tile_mt_partial = false;   // env BBH_TILE_MT=partial
  bool tile_mt = true;            // env BBH_TILE_MT=0: K^-1's tiles stay in the dataflow tail (A/B)
  bool info_clean = false;        // the Cholesky flag on the device is known to be 0 (the dataflow tail's last role resets it)
  bool tile_gram = true;          // env BBH_TILE_GRAM=0: fit evaluations launch bbh_gram_kernel before the factorisation instead of building the tiles inside it (A/B)
  int fit_flow = 1;               // env BBH_FIT_FLOW: 0 fit evaluations for 64 < np <= 1024 launch by launch, 1 (default) Gram + factorisation launches, then ONE dataflow launch for K^-1, alpha, value and gradient, 2 the whole evaluation as one dataflow launch
  void* rff_state = nullptr;      // feature-space model of the RFF kernel (bbh_rff.hip), null for every other kernel
  std::vector<double> rff_w_host; // bbh_set_rff_weights: the frequencies [dn, D] the next bbh_set_model with BBH_KERNEL_RFF takes
  int rff_w_dn = 0, rff_w_D = 0;
  void* nehvi_state = nullptr;    // device-resident box decompositions + their scratch (bbh_nehvi.hip), null until bbh_cells_build_dev
  void* select_state = nullptr;   // chunk keys, result block and base-sample tables of the selection kernels (bbh_select.hip)
  void* sobol_state = nullptr;    // staging of the device-side base-sample draw (bbh_sobol.hip), null until bbh_sobol_normal_dev
  bool q1_sliced = true;          // env BBH_Q1_SLICED=0: q' = 1 qLogEI as one thread per candidate (A/B)
  bool select_on = true;          // env BBH_SELECT=0: top-k / argmax by k rounds of workgroup argmax (A/B)
  // timing
  int timing = 0;  // 0 off, 1 every kernel family, otherwise 2 x (bit mask of the families that record events)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double timed_ms[BBH_TIMED_FAMILIES] = {};
  int64_t timed_launches[BBH_TIMED_FAMILIES] = {};
  struct TimedSpan {
    hipEvent_t e0, e1;
    int family;
  };
  std::vector<TimedSpan> pending_events;
};

// Brackets the launches of one kernel family with HIP events on the handle's stream while timing is enabled.
struct bbh_timed_scope {
  bbh_handle* h;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int family;
  bbh_timed_scope(bbh_handle* h_, int family_, bool wanted = true) : h(h_), family(family_) {
    if ((h->timing == 1 || ((h->timing >> 1) >> family_ & 1)) && wanted) {
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, h->stream);
    }
  }
  ~bbh_timed_scope() {
    if (e0) {
      hipEventRecord(e1, h->stream);
      h->pending_events.push_back({e0, e1, family});
    }
  }
};

inline int64_t bbh_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Device-resident box decompositions of one qLogNEHVI selection step (bbh_nehvi.hip: bbh_cells_build_dev writes them,
// bbh_qlognehvi_cells in bbh_acq.hip scores against them).
struct bbh_nehvi_state {
  int64_t S = 0, total = 0, cap = 0;  // samples, cells in all samples, cell capacity per sample
  int m = 0;
  double* d_slots = nullptr;  // [S][cap][2][m] per-sample cells as the decomposition kernel leaves them (lower bound, side length)
  size_t slot_bytes = 0;
  double* d_pack = nullptr;   // [off: S + 1 (int64) | lo: S cap m | ll: S cap m | len: S cap m], the first `total` cells of each in use
  size_t pack_bytes = 0;
  int* d_cnt = nullptr;       // [S] cells per sample, [S] = overflow flag
  int64_t cnt_cap = 0;
  double* d_ref = nullptr;    // [BBH_MAX_OBJECTIVES]
  int64_t* h_status = nullptr;  // pinned: [0] total cells, [1] samples whose bound list overflowed
  const int64_t* off() const { return (const int64_t*)d_pack; }
  const double* lo() const { return d_pack + (S + 1); }
  const double* ll() const { return lo() + S * cap * m; }
  const double* len() const { return ll() + S * cap * m; }
};

// ---- linalg (bbh_linalg.hip) ------------------------------------------------------------
// C[M,N] = alpha * op(A) op(B) + beta * C, fp64 MFMA, M,N multiples of 64, K multiple of 16.
// transA: A stored [K,M]; transB: B stored [N,K].  Batched over `batch` with element strides.
void bbh_gemm(hipStream_t s, bool transA, bool transB, int64_t M, int64_t N, int64_t K, double alpha,
              const double* A, int64_t lda, int64_t strideA, const double* B, int64_t ldb,
              int64_t strideB, double beta, double* C, int64_t ldc, int64_t strideC, int batch);
// In-place blocked Cholesky of the np x np matrix K (lower), with X = L^-1; info!=0 on failure.
void bbh_potrf_trtri(bbh_handle* h);
bool bbh_potrf_trtri_from_inputs(bbh_handle* h, const double* theta_any, const double* theta_host, const void* mt_args = nullptr);  // mt_args: pd_mt_args (bbh_tiles.h) - K^-1's tiles built by extra workgroups of the same launch; h->tiles_did_mt says whether they were  // theta_any null: theta_host travels as kernel arguments  // Gram tiles built inside the tile-dataflow launch (fit evaluations); false: not available, nothing enqueued
void bbh_potrf_tiles_mark_unusable(int device);  // a tile-dataflow launch gave up: per-step launches from now on, process-wide
void bbh_ensure_side_stream(bbh_handle* h);  // creates the fit's second stream and its events (not during a capture)
void bbh_matvec(hipStream_t s, const double* A, int64_t lda, int64_t rows, int64_t cols,
                const double* x, double* y);   // y = A x   (row-major A)
void bbh_matvec_t(hipStream_t s, const double* A, int64_t lda, int64_t rows, int64_t cols,
                  const double* x, double* y); // y = A^T x

// ---- model (bbh_model.hip) --------------------------------------------------------------
int bbh_upload_theta(bbh_handle* h, const double* theta_host);
void bbh_launch_gram(bbh_handle* h, double jitter, double jitter_latent);  // jitter_latent: rows with noise mask 0
bool bbh_fit_small_launch(bbh_handle* h, double jitter, const double* theta_dev, double* out_dev, int* info_dev);  // whole objective evaluation of a small model in one workgroup; false: not eligible
bbh_kern_spec bbh_kern_spec_of(const bbh_handle* h);
int bbh_hadamard_offset(const bbh_handle* h);
double bbh_prior_base(const bbh_handle* h);
bool bbh_materialised_only(const bbh_handle* h);
bool bbh_has_dot_kind(const bbh_handle* h);  // a Linear / Polynomial kernel somewhere: k(x, x) is not constant  // composite / piecewise-polynomial models: no fused kernel form  // k(x, x) without the task factor
int bbh_launch_unfused_ext(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev,
                           double* cross_dev);  // composite kernels: every posterior output through the materialised K*  // per-task noise block in theta (means follow at + T), -1 = none

// ---- fused posterior (bbh_panel.hip) ----------------------------------------------------
int bbh_pack_operands(bbh_handle* h);   // trainfrag, rfrag, meanB, tables (after factorize / pending_set)
int bbh_launch_fused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev,
                     double* var_dev, double* cross_dev, bool with_var);
int bbh_launch_unfused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev,
                       double* var_dev);

int bbh_ensure_ws(bbh_handle* h, size_t bytes);
void* bbh_stage_pinned(bbh_handle* h, size_t bytes);  // the handle's pinned staging buffer, free to be rewritten (bbh_acq.hip); null on failure
int bbh_stage_done(bbh_handle* h);                   // the copies enqueued from it on h->stream are the last readers
int bbh_upload_z(bbh_handle* h, const double* z_host, size_t count);  // host doubles -> h->d_z through the handle's pinned staging buffer (bbh_acq.hip)
void bbh_select_destroy(bbh_handle* h);  // bbh_select.hip
void bbh_nehvi_destroy(bbh_handle* h);   // bbh_nehvi.hip
void bbh_sobol_destroy(bbh_handle* h);   // bbh_sobol.hip
void bbh_flow_destroy(bbh_handle* h);    // bbh_fitflow.hip
bool bbh_fit_flow_launch(bbh_handle* h, const double* theta_dev, double* out_dev, int* info_dev, bool tail_only, const double* theta_host = nullptr,
                         bool split = false, int skip_mt = 0, bool prepare_only = false);  // split: two launches - the factorisation with K^-1's tiles, then everything behind them; skip_mt: tail form without the M-tile roles
bool bbh_fit_flow_mt_args(bbh_handle* h, void* pd_mt_args_out);  // arguments for K^-1's tiles inside the factorisation launch, matching the next tail launch  // 64 < np <= 1024: the whole evaluation as one dataflow launch, or (tail_only) everything after the factorisation; false: not eligible
bool bbh_fit_flow_eligible(bbh_handle* h);
void bbh_fit_flow_reset(bbh_handle* h);  // after a launch that gave up: clean state, the handle stops using the form
void bbh_free_model_public(bbh_handle* h);
// ---- RFF kernel: the model in feature space (bbh_rff.hip) ----
inline bool bbh_is_rff(const bbh_handle* h) { return h->desc.kernel_kind == BBH_KERNEL_RFF; }
int bbh_rff_setup(bbh_handle* h);          // after the generic part of bbh_set_model_ex
int bbh_rff_fit_enqueue(bbh_handle* h);    // one evaluation of the fit objective on h->stream
int bbh_rff_factorize(bbh_handle* h);      // theta in h->d_theta -> posterior operands
int bbh_rff_posterior_launch(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* mean_dev, double* var_dev, double* cross_dev);
int bbh_rff_pending_set(bbh_handle* h, const double* Xpend_host, int64_t p, double* mean_p_host, double* cov_pp_host);
int bbh_rff_posterior_joint(bbh_handle* h, const double* Xq_host, int64_t q, double* mean_host, double* cov_host);
void bbh_rff_destroy(bbh_handle* h);
