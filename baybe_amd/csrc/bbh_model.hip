// GP model state, Gram assembly, fit objective (value + analytic gradient) and factorisation.
//
// Follows (see oracle/gp_oracle.py for the maths and the reference citations):
//   model assembly          baybe/surrogates/gaussian_process/core.py:272-341
//   kernel / priors         baybe/surrogates/gaussian_process/presets/baybe.py:56-144
//   ICM task covariance     baybe/surrogates/gaussian_process/components/kernel.py:298-337
//   MLL / LOO criterion     baybe/surrogates/gaussian_process/components/fit_criterion.py:31-41
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "bbh_common.h"

// Up to four byte ranges of a (host-mapped, pinned) staging buffer into their device arrays, and one range cleared: blockIdx.y selects
// the range, 8 bytes per thread (every range is a multiple of 4 bytes and 8-byte aligned at both ends; the tail is copied as words).
struct bbh_scatter_args {
  const unsigned char* src;
  unsigned char* dst[4];
  size_t off[4], bytes[4];
  unsigned char* zero;
  size_t zero_bytes;
  int* flag;       // diagnostics (BBH_SETMODEL_TRACE): host-mapped word the kernel stamps when it starts executing
  int flag_value;
};
__global__ __launch_bounds__(256) void bbh_scatter_kernel(const bbh_scatter_args a) {
  const int r = blockIdx.y;
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (a.flag && r == 0 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.flag, a.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (r == 4) {
    if (i + 8 <= a.zero_bytes) *(unsigned long long*)(a.zero + i) = 0ull;
    return;
  }
  if (i + 8 <= a.bytes[r]) {
    *(unsigned long long*)(a.dst[r] + i) = *(const unsigned long long*)(a.src + a.off[r] + i);
  } else if (i + 4 <= a.bytes[r]) {
    *(unsigned int*)(a.dst[r] + i) = *(const unsigned int*)(a.src + a.off[r] + i);
  }
}

// (bbh_gfun: bbh_common.h)
#define TH_NOISE 0
#define TH_MEAN 1
#define TH_OS 2
#define TH_LS 3

int bbh_ensure_ws(bbh_handle* h, size_t bytes) {
  if (bytes <= h->ws_bytes) return 0;
  if (h->d_ws) hipFree(h->d_ws);
  h->d_ws = nullptr;
  h->ws_bytes = 0;
  BBH_HIP_TRY(h, hipMalloc((void**)&h->d_ws, bytes));
  h->ws_bytes = bytes;
  return 0;
}

int bbh_upload_theta(bbh_handle* h, const double* theta_host) {
  const int64_t len = bbh_theta_len(h);
  for (int64_t i = 0; i < len; i++)
    if (!(theta_host[i] == theta_host[i])) {
      h->err = "theta contains NaN";
      return -3;
    }
  h->theta.assign(theta_host, theta_host + len);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_theta, theta_host, sizeof(double) * len, hipMemcpyHostToDevice, h->stream));
  return 0;
}

// K[a][b] = scale * k(r_ab) + (s2 + jitter) [a==b]; identity on the padding.  Rows with noise mask 0 (latent values: the baseline
// rows of an extended qLogNEHVI model) take jitter_latent instead of jitter.
__global__ __launch_bounds__(256) void bbh_gram_kernel(const double* __restrict__ xnT, const int* __restrict__ task,
                                                       const double* __restrict__ nmask,
                                                       const double* __restrict__ theta, int n, int np, int dn,
                                                       const bbh_kern_spec ks, int T, int hoff, double jitter, double jitter_latent,
                                                       double* __restrict__ K) {
  extern __shared__ double s_invls[];  // [F][dn]
  for (int e = threadIdx.x; e < ks.F * dn; e += blockDim.x) s_invls[e] = 1.0 / theta[ks.ls_off[e / dn] + e % dn];
  __syncthreads();
  const int a = blockIdx.y;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= np) return;
  if (a >= n || b >= n) {
    K[(int64_t)a * np + b] = (a == b) ? 1.0 : 0.0;
    return;
  }
  double r2[BBH_MAX_FACTORS] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < dn; j++) {
    const double xa = xnT[(int64_t)j * np + a], xb = xnT[(int64_t)j * np + b];
    for (int f = 0; f < ks.F; f++) r2[f] += bbh_metric_term_f(ks, theta, f, j, dn, xa, xb, s_invls[f * dn + j]);
  }
  double k = bbh_kcomp(ks, theta, r2);
  if (ks.use_os) k *= theta[TH_OS];
  if (T > 1) k *= theta[TH_LS + dn + task[a] * T + task[b]];
  if (a == b) k += (hoff >= 0 ? theta[hoff + task[a]] : theta[TH_NOISE]) * nmask[a] + (nmask[a] != 0.0 ? jitter : jitter_latent);
  K[(int64_t)a * np + b] = k;
}

void bbh_launch_gram(bbh_handle* h, double jitter, double jitter_latent) {
  dim3 grid((unsigned)((h->np + 255) / 256), (unsigned)h->np), block(256);
  hipLaunchKernelGGL(bbh_gram_kernel, grid, block, sizeof(double) * h->dn * h->F, h->stream, h->d_xnT, h->d_task, h->d_nmask,
                     h->d_theta, (int)h->n, (int)h->np, h->dn, bbh_kern_spec_of(h), h->T, bbh_hadamard_offset(h), jitter,
                     jitter_latent, h->d_K);
}

__global__ void bbh_resid_kernel(const double* __restrict__ ystd, const double* __restrict__ theta,
                                 const int* __restrict__ task, int T, int hoff, int n, int np, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np) r[i] = (i < n) ? ystd[i] - (hoff >= 0 ? theta[hoff + T + task[i]] : theta[TH_MEAN]) : 0.0;
}

// LOO helper vectors: d = diag(M), u = 0.5/d + 0.5 alpha^2/d^2, w = alpha/d (0 on padding)
__global__ void bbh_loo_vec_kernel(const double* __restrict__ M, const double* __restrict__ alpha, int n, int np,
                                   double* __restrict__ u, double* __restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  if (i >= n) {
    u[i] = 0.0;
    w[i] = 0.0;
    return;
  }
  const double d = M[(int64_t)i * np + i], a = alpha[i];
  u[i] = 0.5 / d + 0.5 * a * a / (d * d);
  w[i] = a / d;
}

// out = p0 + p1 + p2 + p3 (partial products of the split-K X^T X), fixed order
__global__ void bbh_sum4_kernel(const double* __restrict__ P, int64_t n, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = (P[e] + P[n + e]) + (P[2 * n + e] + P[3 * n + e]);
}

// Msc[a][b] = M[a][b] * u[b]
__global__ void bbh_colscale_kernel(const double* __restrict__ M, const double* __restrict__ u, int np,
                                    double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < (int64_t)np * np) out[e] = M[e] * u[e % np];
}

__device__ __forceinline__ double bbh_block_sum_256(double v, double* sm) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return r;
}

// out[0] = data-term value, out[1 + TH_MEAN] = d/dc   (single workgroup)
__global__ __launch_bounds__(256) void bbh_value_kernel(const double* __restrict__ L, const double* __restrict__ M,
                                                        const double* __restrict__ r,
                                                        const double* __restrict__ alpha,
                                                        const double* __restrict__ q, const int* __restrict__ task,
                                                        int T, int hoff, int n, int np, int criterion,
                                                        double* __restrict__ out) {
  __shared__ double sm[4];
  double v = 0.0, gm = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    if (criterion == BBH_CRITERION_MLL) {
      v += -0.5 * r[i] * alpha[i] - log(L[(int64_t)i * np + i]);
      gm += alpha[i];
    } else {
      const double d = M[(int64_t)i * np + i];
      v += 0.5 * log(d) - 0.5 * alpha[i] * alpha[i] / d;
      gm += q[i];
    }
  }
  v = bbh_block_sum_256(v, sm);
  gm = bbh_block_sum_256(gm, sm);
  if (threadIdx.x == 0) {
    out[0] = v - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
    out[1 + TH_MEAN] = hoff >= 0 ? 0.0 : gm;
  }
  if (hoff < 0) return;
  for (int t = 0; t < T; t++) {  // per-task constant means: d/dc_t = the same sum over the task's points
    double gt = 0.0;
    for (int i = threadIdx.x; i < n; i += 256)
      if (task[i] == t) gt += (criterion == BBH_CRITERION_MLL) ? alpha[i] : q[i];
    gt = bbh_block_sum_256(gt, sm);
    if (threadIdx.x == 0) out[1 + hoff + T + t] = gt;
  }
}

// Pair kernel: for every ordered pair (a,b), a,b < n:
//   G_ab  (MLL: 0.5 (alpha_a alpha_b - M_ab);  LOO: -Q_ab + 0.5 (alpha_a q_b + alpha_b q_a))
// contributes to   d/dl_j       G g(r) scale Delta_j^2 / l_j^3
//                  d/dnoise     G [a==b]
//                  d/doutscale  G k scale_B
//                  d/dB[ta][tb] G k outputscale
// One wave-level partial row per (a, b-chunk, wave): partial[row][slot].
__global__ __launch_bounds__(256) void bbh_grad_pair_kernel(
    const double* __restrict__ xnT, const int* __restrict__ task, const double* __restrict__ nmask,
    const double* __restrict__ theta, const double* __restrict__ M, const double* __restrict__ Q, const double* __restrict__ alpha,
    const double* __restrict__ q, int n, int np, int dn, const bbh_kern_spec ks, int T, int hoff, int criterion, int nslots,
    double* __restrict__ partial) {
  const int a = blockIdx.x;
  const int b = blockIdx.y * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool act = (b < n);
  const int bb = act ? b : a;
  double G;
  if (criterion == BBH_CRITERION_MLL)
    G = 0.5 * (alpha[a] * alpha[bb] - M[(int64_t)a * np + bb]);
  else
    G = -Q[(int64_t)a * np + bb] + 0.5 * (alpha[a] * q[bb] + alpha[bb] * q[a]);
  if (!act) G = 0.0;
  const int ta = (T > 1) ? task[a] : 0, tb = (T > 1) ? task[bb] : 0;
  double r2[BBH_MAX_FACTORS] = {0.0, 0.0, 0.0, 0.0};
  for (int j = 0; j < dn; j++) {
    const double xa = xnT[(int64_t)j * np + a], xb = xnT[(int64_t)j * np + bb];
    for (int f = 0; f < ks.F; f++) r2[f] += bbh_metric_term_f(ks, theta, f, j, dn, xa, xb, 1.0 / theta[ks.ls_off[f] + j]);
  }
  const double os = ks.use_os ? theta[TH_OS] : 1.0;
  const double Bab = (T > 1) ? theta[TH_LS + dn + ta * T + tb] : 1.0;
  // per factor: value u_f = os_f k_f, and the weight W_f of its derivative in the composite:
  //   product: d/dk_f = os_f prod_{g != f} u_g,   sum: d/dk_f = os_f
  double kf[BBH_MAX_FACTORS], wf[BBH_MAX_FACTORS];
  double al[BBH_MAX_FACTORS];
  for (int f = 0; f < ks.F; f++) al[f] = ks.alpha_off >= 0 ? theta[ks.alpha_off + f] : 1.0;
  for (int f = 0; f < ks.F; f++) kf[f] = bbh_kbase(ks.kind[f], r2[f], ks.jb, al[f]);
  {
    double uf[BBH_MAX_FACTORS] = {1.0, 1.0, 1.0, 1.0};
    for (int f = 0; f < ks.F; f++) uf[f] = (ks.F > 1 ? theta[ks.fos_off + f] : 1.0) * kf[f];
    for (int f = 0; f < ks.F; f++) wf[f] = ks.F > 1 ? bbh_combine_weight(ks.F, ks.combine, ks.grp, uf, f) : 1.0;  // without the factor's own outputscale
  }
  const double kb = bbh_kcomp(ks, theta, r2);
  double* prow = partial + ((int64_t)(a * gridDim.y + blockIdx.y) * 4 + wave) * nslots;
  // slot layout = gradient layout of theta: [noise, mean(unused), outputscale, ls.., B.., (noise_t, mean_t), (ls_f.., os_f..)]
  {
    double v = (a == bb) ? G * nmask[a] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) {
      prow[TH_NOISE] = hoff >= 0 ? 0.0 : v;
      if (hoff >= 0)  // per-task noise: the diagonal term of row a belongs to the task of a (wave-uniform)
        for (int t = 0; t < T; t++) {
          prow[hoff + t] = (t == ta) ? v : 0.0;
          prow[hoff + T + t] = 0.0;  // mean slots: written by the value kernel
        }
    }
  }
  {
    double v = ks.use_os ? G * kb * Bab : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) {
      prow[TH_OS] = v;
      prow[TH_MEAN] = 0.0;
    }
  }
  for (int f = 0; f < ks.F; f++) {
    const double fos = ks.F > 1 ? theta[ks.fos_off + f] : 1.0;
    const double Gg = G * bbh_gfun(ks.kind[f], r2[f], ks.jb, al[f]) * os * Bab * wf[f] * fos;
    for (int j = 0; j < dn; j++) {
      const double l = theta[ks.ls_off[f] + j];
      const double xa = xnT[(int64_t)j * np + a], xb = xnT[(int64_t)j * np + bb];
      double v, vp = 0.0;
      if (ks.kind[f] == BBH_KERNEL_PERIODIC) {  // dk/dl_j = 2 k sin^2(u) / l^2,  dk/dp_j = (2 k / l) sin(2 u) pi Delta / p^2,  u = pi Delta / p
        const double pl = theta[ks.per_off + f * dn + j], u = M_PI * (xa - xb) / pl, sn = sin(u);
        v = Gg * sn * sn / (l * l);
        vp = Gg * sin(2.0 * u) * M_PI * (xa - xb) / (l * pl * pl);
      } else {
        v = Gg * (BBH_KIND_IS_DOT(ks.kind[f]) ? xa * xb : (xa - xb) * (xa - xb)) / (l * l * l);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) prow[ks.ls_off[f] + j] = v;
      if (ks.per_off >= 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vp += __shfl_down(vp, o, 64);
        if (lane == 0) prow[ks.per_off + f * dn + j] = vp;
      }
    }
    if (ks.F > 1) {  // d/d os_f = W_f k_f
      double v = G * os * Bab * wf[f] * kf[f];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) prow[ks.fos_off + f] = v;
    }
    if (ks.alpha_off >= 0) {  // RQ: dk/dalpha = k (u / (1 + u) - log(1 + u)), u = r^2 / (2 alpha); 0 for the other kinds
      double v = 0.0;
      if (ks.kind[f] == BBH_KERNEL_RQ) {
        const double u = r2[f] / (2.0 * al[f]);
        v = G * os * Bab * wf[f] * fos * kf[f] * (u / (1.0 + u) - log1p(u));
      } else if (BBH_KIND_IS_POLY(ks.kind[f])) {  // d/d offset (s + offset)^p = p (s + offset)^(p - 1)
        const int pw = ks.kind[f] - BBH_KERNEL_POLY1 + 1;
        v = G * os * Bab * wf[f] * fos * (double)pw * bbh_powi(r2[f] + al[f], pw - 1);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) prow[ks.alpha_off + f] = v;
    }
  }
  if (T > 1) {
    const double gk = G * kb * os;
    for (int c = 0; c < T * T; c++) {
      double v = (c == ta * T + tb) ? gk : 0.0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) prow[TH_LS + dn + c] = v;
    }
  }
}

// out[1 + slot] = sum over partial rows (fixed order -> deterministic); mean slot is skipped
__global__ __launch_bounds__(256) void bbh_grad_reduce_kernel(const double* __restrict__ partial, int64_t rows,
                                                              int nslots, int mean_lo, int mean_hi,
                                                              double* __restrict__ out) {
  __shared__ double sm[4];
  const int slot = blockIdx.x;
  if (slot == TH_MEAN || (slot >= mean_lo && slot < mean_hi)) return;  // written by the value kernel
  double v = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) v += partial[r * nslots + slot];
  v = bbh_block_sum_256(v, sm);
  if (threadIdx.x == 0) out[1 + slot] = v;
}

// -----------------------------------------------------------------------------------------
// any factor (or the single kernel) a rational-quadratic kernel: theta carries F alpha slots at its end
static bool bbh_has_rq(const bbh_handle* h) {  // (RQ alpha or polynomial offset: one slot per factor)
  if (h->F <= 1) return BBH_KIND_HAS_ALPHA(h->desc.kernel_kind);
  for (int f = 0; f < h->F; f++)
    if (BBH_KIND_HAS_ALPHA(h->desc.factor_kind[f])) return true;
  return false;
}

bool bbh_has_dot_kind(const bbh_handle* h) {
  if (h->F <= 1) return BBH_KIND_IS_DOT(h->desc.kernel_kind);
  for (int f = 0; f < h->F; f++)
    if (BBH_KIND_IS_DOT(h->desc.factor_kind[f])) return true;
  return false;
}

static bool bbh_has_periodic(const bbh_handle* h) {
  if (h->F <= 1) return h->desc.kernel_kind == BBH_KERNEL_PERIODIC;
  for (int f = 0; f < h->F; f++)
    if (h->desc.factor_kind[f] == BBH_KERNEL_PERIODIC) return true;
  return false;
}

static int64_t bbh_theta_len_of(const bbh_handle* h) {
  return 3 + h->dn + (h->T > 1 ? (int64_t)h->T * h->T : 0) + (h->hadamard ? 2 * (int64_t)h->T : 0) +
         (h->F > 1 ? (int64_t)(h->F - 1) * h->dn + h->F : 0) + (bbh_has_rq(h) ? h->F : 0) +
         (bbh_has_periodic(h) ? (int64_t)h->F * h->dn : 0);
}

extern "C" int64_t bbh_theta_len(bbh_handle* h) {
  if (!h || !h->have_model) return -1;
  return bbh_theta_len_of(h);
}

// how the elementwise kernels find the factors of the kernel in theta
bbh_kern_spec bbh_kern_spec_of(const bbh_handle* h) {
  bbh_kern_spec ks{};
  ks.F = h->F;
  ks.combine = h->desc.combine;
  for (int f = 0; f < BBH_MAX_FACTORS; f++) {
    const int g = h->desc.combine == 0 ? 0 : h->desc.combine == 1 ? f : h->desc.factor_group[f];
    ks.grp[f] = (g < 0 || g >= BBH_MAX_FACTORS) ? 0 : g;
  }
  ks.use_os = h->desc.use_outputscale;
  ks.jb = h->dn / 2 + 1;
  const int base = 3 + h->dn + (h->T > 1 ? h->T * h->T : 0) + (h->hadamard ? 2 * h->T : 0);
  for (int f = 0; f < BBH_MAX_FACTORS; f++) {
    ks.kind[f] = (f == 0 || h->F <= 1) ? h->desc.kernel_kind : h->desc.factor_kind[f];
    ks.ls_off[f] = (f == 0 || f >= h->F) ? 3 : base + (f - 1) * h->dn;
  }
  ks.fos_off = h->F > 1 ? base + (h->F - 1) * h->dn : -1;
  ks.alpha_off = bbh_has_rq(h) ? base + (h->F > 1 ? (h->F - 1) * h->dn + h->F : 0) : -1;
  ks.per_off = bbh_has_periodic(h) ? base + (h->F > 1 ? (h->F - 1) * h->dn + h->F : 0) + (bbh_has_rq(h) ? h->F : 0) : -1;
  return ks;
}

// offset of the per-task noise block in theta (the per-task means follow), -1 without one
int bbh_hadamard_offset(const bbh_handle* h) { return h->hadamard ? 3 + h->dn + h->T * h->T : -1; }

// what a new model of the SAME shape invalidates: flags and derived state, not the buffers
static void bbh_reset_model_state(bbh_handle* h) {
  h->coopg_ready = false;
  h->coop_ready = false;
  h->coop2_ready = false;
  h->small_nb = 0;
  if (h->fit_exec) hipGraphExecDestroy(h->fit_exec);
  h->fit_exec = nullptr;
  h->fit_graph_failed = false;
  // (the installed weight columns are INVALIDATED, not freed: the shape signature is unchanged, so the buffers fit the next
  // bbh_install_columns - hipFree synchronises the device)
  h->ncols = 0;
  h->have_model = false;
  h->factorized = false;
}

static void bbh_free_model(bbh_handle* h) {
  void* ptrs[] = {h->d_xnT,   h->d_task,    h->d_ystd,      h->d_theta, h->d_K,     h->d_X,       h->d_M,
                  h->d_Q,     h->d_Q2,      h->d_D,         h->d_tmp,   h->d_r,     h->d_t,       h->d_alpha,
                  h->d_u,     h->d_w,       h->d_q,         h->d_partial, h->d_out, h->d_info,    h->d_trainfrag,
                  h->d_rfrag, h->d_meanB,   h->d_sclofs,    h->d_numcol, h->d_tasktbl, h->d_taskext, h->d_beta, h->d_pass_off, h->d_pass_w, h->d_nmask, h->d_colfrag, h->d_pendT, h->d_colA, h->d_Mpart, h->d_trainfrag_f, h->d_sclofs_f};
  for (void* p : ptrs)
    if (p) hipFree(p);
  bbh_rff_destroy(h);
  h->d_xnT = h->d_ystd = h->d_theta = h->d_K = h->d_X = h->d_M = h->d_Q = h->d_Q2 = h->d_D = h->d_tmp = nullptr;
  h->d_r = h->d_t = h->d_alpha = h->d_u = h->d_w = h->d_q = h->d_partial = h->d_out = nullptr;
  h->d_trainfrag = h->d_rfrag = h->d_meanB = h->d_sclofs = h->d_tasktbl = h->d_beta = nullptr;
  h->d_task = h->d_info = h->d_numcol = h->d_taskext = h->d_pass_w = nullptr;
  h->d_pass_off = nullptr;
  h->d_nmask = nullptr;
  h->d_pendT = h->d_colA = h->d_Mpart = nullptr;
  h->d_trainfrag_f = h->d_sclofs_f = nullptr;
  h->tf_f_elems = 0;
  h->coopg_ready = false;
  h->colA_elems = 0;
  if (h->fit_exec) hipGraphExecDestroy(h->fit_exec);
  h->fit_exec = nullptr;
  h->fit_graph_failed = false;
  if (h->pin_theta) hipHostFree(h->pin_theta);
  if (h->pin_out) hipHostFree(h->pin_out);
  if (h->pin_info) hipHostFree(h->pin_info);
  h->pin_theta = h->pin_out = nullptr;
  h->pin_info = nullptr;
  h->d_colfrag = nullptr;
  h->colfrag_elems = 0;
  h->ncols = 0;
  h->rfrag_elems = 0;
  h->have_model = false;
  h->factorized = false;
  h->model_sig.clear();
}

void bbh_free_model_public(bbh_handle* h) { bbh_free_model(h); }

extern "C" int bbh_set_model(bbh_handle* h, const bbh_model_desc* desc, int64_t n, const double* X_train_host,
                             const double* y_train_host, const double* lo_host, const double* hi_host) {
  return bbh_set_model_ex(h, desc, n, X_train_host, y_train_host, lo_host, hi_host, nullptr, 0, 0.0, 1.0);
}

extern "C" int bbh_set_model_ex(bbh_handle* h, const bbh_model_desc* desc, int64_t n, const double* X_train_host,
                                const double* y_train_host, const double* lo_host, const double* hi_host,
                                const uint8_t* noise_mask_host, int use_given_std, double ybar_in, double ysd_in) {
  if (!h) return -1;
  if (!desc || n < 1 || !X_train_host || !y_train_host || !lo_host || !hi_host) {
    h->err = "bbh_set_model: bad arguments";
    return -1;
  }
  if (desc->kernel_kind < 0 || desc->kernel_kind > BBH_KERNEL_RFF || desc->d < 1 || desc->n_tasks < 1 ||
      (desc->n_tasks > 1 && (desc->task_col < 0 || desc->task_col >= desc->d)) ||
      (desc->criterion != BBH_CRITERION_MLL && desc->criterion != BBH_CRITERION_LOO)) {
    h->err = "bbh_set_model: invalid model description";
    return -1;
  }
  if (desc->n_factors > 1) {
    bool ok = desc->n_factors <= BBH_MAX_FACTORS && (desc->combine == 0 || desc->combine == 1 || desc->combine == 2) &&
              desc->factor_kind[0] == desc->kernel_kind;
    for (int f = 0; ok && desc->combine == 2 && f < desc->n_factors; f++) ok = desc->factor_group[f] >= 0 && desc->factor_group[f] < BBH_MAX_FACTORS;
    for (int f = 0; ok && f < desc->n_factors; f++) ok = desc->factor_kind[f] >= 0 && desc->factor_kind[f] <= BBH_KERNEL_PERIODIC;
    if (!ok) {
      h->err = "bbh_set_model: invalid composite kernel (2..4 factors, combine 0 | 1 | 2 with factor_group in 0..3, factor_kind[0] == kernel_kind)";
      return -1;
    }
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  // BBH_SETMODEL_TRACE=1: wall-clock stamps of the stages below on stderr (scripts/gpu_set_model_probe.py)
  // (=2: the stamps are kept and printed in one piece when the call returns - a write per stage shifts the timing of what follows:
  // with per-stage prints the call measured 0.2 ms, without them 18 ms, profiles/r06_small_space_latency*.log)
  static const char* sm_env = getenv("BBH_SETMODEL_TRACE");
  static const bool sm_trace = sm_env != nullptr, sm_quiet = sm_env && sm_env[0] == '2';
  const auto sm_t0 = std::chrono::steady_clock::now();
  struct sm_rec { const char* what; double us; };
  sm_rec sm_log[16];
  int sm_n = 0;
  auto sm_stamp = [&](const char* what) {
    if (!sm_trace) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - sm_t0).count();
    if (sm_quiet) {
      if (sm_n < 16) sm_log[sm_n++] = {what, us};
    } else {
      fprintf(stderr, "bbh_set_model %-10s %8.1f us\n", what, us);
    }
  };
  struct sm_flush_t {
    sm_rec* log; int* n;
    ~sm_flush_t() {
      if (*n == 0) return;
      char buf[1024]; int o = 0;
      for (int k = 0; k < *n && o < 900; k++) o += snprintf(buf + o, sizeof(buf) - o, " %s %.1f", log[k].what, log[k].us);
      fprintf(stderr, "bbh_set_model stages (us):%s\n", buf);
    }
  } sm_flush{sm_log, &sm_n};
  // A campaign re-fits after every batch of measurements: the model description is the same and the padded size np changes only
  // every 64 measurements.  Then every device buffer keeps its size - freeing and re-allocating ~40 of them (hipFree synchronises
  // the device) was 1 ms of a 6.8 ms small-space recommend().  Buffers are kept when the shape signature is unchanged.
  {
    int dn_new = 0;
    const int tc0 = (desc->n_tasks > 1 || desc->task_col >= 0) ? desc->task_col : -1;
    for (int c = 0; c < desc->d; c++) dn_new += (c != tc0);
    char sig[256];
    snprintf(sig, sizeof(sig), "%lld/%d/%d/%d/%d/%d/%d/%d/%d/%d/%d:%d,%d,%d,%d:%d,%d,%d,%d", (long long)bbh_round_up(n, BBH_PAD), desc->d, dn_new,
             desc->criterion, desc->kernel_kind, desc->task_col, desc->n_tasks, desc->use_outputscale, desc->hadamard, desc->n_factors, desc->combine,
             desc->factor_kind[0], desc->factor_kind[1], desc->factor_kind[2], desc->factor_kind[3], desc->factor_scaled[0],
             desc->factor_scaled[1], desc->factor_scaled[2], desc->factor_scaled[3]);
    if (desc->kernel_kind == BBH_KERNEL_RFF) snprintf(sig + strlen(sig), sizeof(sig) - strlen(sig), ":rff%d", h->rff_w_D);
    if (desc->combine == 2)
      snprintf(sig + strlen(sig), sizeof(sig) - strlen(sig), ":g%d%d%d%d", desc->factor_group[0], desc->factor_group[1], desc->factor_group[2],
               desc->factor_group[3]);
    if (h->have_model && h->model_sig == sig) {
      bbh_reset_model_state(h);
    } else {
      bbh_free_model(h);
      h->model_sig = sig;
    }
  }
  sm_stamp("reset");
  h->desc = *desc;
  h->n = n;
  h->np = bbh_round_up(n, BBH_PAD);
  h->nb = h->np / BBH_TB;
  h->T = desc->n_tasks;
  h->hadamard = (desc->hadamard != 0 && desc->n_tasks > 1);
  h->F = desc->n_factors > 1 ? desc->n_factors : 1;
  const int d = desc->d;
  const int tc = (desc->n_tasks > 1 || desc->task_col >= 0) ? desc->task_col : -1;
  h->numcol.clear();
  h->lo.clear();
  h->hi.clear();
  for (int c = 0; c < d; c++)
    if (c != tc) {
      h->numcol.push_back(c);
      h->lo.push_back(lo_host[c]);
      h->hi.push_back(hi_host[c]);
      if (!(hi_host[c] > lo_host[c])) {
        h->err = "bbh_set_model: scaling bounds need hi > lo for every numerical column";
        return -1;
      }
    }
  h->dn = (int)h->numcol.size();
  if (h->dn < 1) {
    h->err = "bbh_set_model: no numerical column";
    return -1;
  }
  h->kd = (h->dn + 2 + 3) / 4;
  // round up to a k-step count the software-pipelined kernel is instantiated for (zero-padded dims)
  if (h->kd <= 2)
    h->kd = 2;
  else if (h->kd <= 4)
    h->kd = 4;
  else if (h->kd <= 6)
    h->kd = 6;
  else if (h->kd <= 8)
    h->kd = 8;
  else if (h->kd <= 12)
    h->kd = 12;
  else if (h->kd <= 16)
    h->kd = 16;
  // Standardize(m=1): Bessel-corrected std, < 1e-8 (or undefined) -> 1
  double ybar = 0.0;
  for (int64_t i = 0; i < n; i++) ybar += y_train_host[i];
  ybar /= (double)n;
  double ss = 0.0;
  for (int64_t i = 0; i < n; i++) ss += (y_train_host[i] - ybar) * (y_train_host[i] - ybar);
  double sd = (n > 1) ? sqrt(ss / (double)(n - 1)) : NAN;
  if (!(sd >= 1e-8)) sd = 1.0;
  if (use_given_std) {
    ybar = ybar_in;
    sd = ysd_in;
    if (!(sd > 0.0)) {
      h->err = "bbh_set_model_ex: ysd must be positive";
      return -1;
    }
  }
  h->ybar = ybar;
  h->ysd = sd;
  h->ystd_host.resize(n);
  for (int64_t i = 0; i < n; i++) h->ystd_host[i] = (y_train_host[i] - ybar) / sd;
  // Normalize numerical columns; keep the task ids
  h->xn_host.assign((size_t)n * h->dn, 0.0);
  h->task_host.assign(n, 0);
  h->xcenter.assign(h->dn, 0.0);
  for (int64_t i = 0; i < n; i++) {
    for (int j = 0; j < h->dn; j++) {
      const double v = (X_train_host[i * d + h->numcol[j]] - h->lo[j]) / (h->hi[j] - h->lo[j]);
      h->xn_host[(size_t)i * h->dn + j] = v;
      h->xcenter[j] += v;
    }
    if (tc >= 0) {
      const int t = (int)X_train_host[i * d + tc];
      if (t < 0 || t >= h->T) {
        h->err = "bbh_set_model: task id out of range";
        return -1;
      }
      h->task_host[i] = t;
    }
  }
  for (int j = 0; j < h->dn; j++) h->xcenter[j] /= (double)n;

  const int64_t np = h->np;
  const int64_t tl = bbh_theta_len_of(h);
  std::vector<double> xnT((size_t)h->dn * np, 0.0), ypad(np, 0.0);
  std::vector<int> tpad(np, 0);
  h->nmask_host.assign(np, 0.0);
  for (int64_t i = 0; i < n; i++) h->nmask_host[i] = (noise_mask_host && !noise_mask_host[i]) ? 0.0 : 1.0;
  for (int64_t i = 0; i < n; i++) {
    for (int j = 0; j < h->dn; j++) xnT[(size_t)j * np + i] = h->xn_host[(size_t)i * h->dn + j];
    ypad[i] = h->ystd_host[i];
    tpad[i] = h->task_host[i];
  }
  sm_stamp("host-prep");
#define BBH_ALLOC(ptr, count) \
  if (!(ptr)) BBH_HIP_TRY(h, hipMalloc((void**)&(ptr), sizeof(*(ptr)) * (size_t)(count)))  /* (kept from the previous model of the same shape) */
  BBH_ALLOC(h->d_xnT, h->dn * np);
  BBH_ALLOC(h->d_task, np);
  BBH_ALLOC(h->d_nmask, np);
  BBH_ALLOC(h->d_ystd, np);
  BBH_ALLOC(h->d_theta, tl);
  BBH_ALLOC(h->d_K, np * np);
  BBH_ALLOC(h->d_X, np * np);
  BBH_ALLOC(h->d_M, np * np);
  BBH_ALLOC(h->d_Q, np * np);
  BBH_ALLOC(h->d_Q2, np * np);
  BBH_ALLOC(h->d_D, np * 64);
  BBH_ALLOC(h->d_tmp, np * 64);
  BBH_ALLOC(h->d_r, np);
  BBH_ALLOC(h->d_t, np);
  BBH_ALLOC(h->d_alpha, np);
  BBH_ALLOC(h->d_u, np);
  BBH_ALLOC(h->d_w, np);
  BBH_ALLOC(h->d_q, np);
  const int64_t nchunks = (np + 255) / 256;  // (sized by the padded count: the buffer outlives refits with more rows)
  BBH_ALLOC(h->d_partial, np * nchunks * 4 * tl);
  BBH_ALLOC(h->d_out, 1 + tl);
  BBH_ALLOC(h->d_info, 1);
  BBH_ALLOC(h->d_beta, np * BBH_MEANCOLS);
  if (np >= 256 && np <= 1024) BBH_ALLOC(h->d_Mpart, 4 * np * np);
  BBH_ALLOC(h->d_pendT, (int64_t)h->dn * 16);
  sm_stamp("alloc");
  // The model's arrays go up through the handle's pinned staging buffer as stream-ordered copies: everything that reads them is
  // enqueued on h->stream behind them.  (They were four synchronous copies from pageable memory - 13 - 26 ms of a 1e5-row campaign
  // step, profiles/r06_set_model_probe.log; VERDICT r5 item 3 measured it as "8 ms of set_model".)
  {
    const size_t b_x = sizeof(double) * xnT.size(), b_t = sizeof(int) * (size_t)np, b_m = sizeof(double) * (size_t)np, b_y = sizeof(double) * (size_t)np;
    const size_t o_x = 0, o_m = o_x + b_x, o_y = o_m + b_m, o_t = o_y + b_y;  // (doubles first: the int block needs no padding behind them)
    sm_stamp("packed");
    unsigned char* stage = (unsigned char*)bbh_stage_pinned(h, o_t + b_t);
    sm_stamp("staged");
    if (!stage) {
      (void)hipGetLastError();
      h->err = "bbh_set_model: no pinned staging buffer";
      return -2;
    }
    memcpy(stage + o_x, xnT.data(), b_x);
    memcpy(stage + o_m, h->nmask_host.data(), b_m);
    memcpy(stage + o_y, ypad.data(), b_y);
    memcpy(stage + o_t, tpad.data(), b_t);
    hipStream_t s = h->stream;
    // BBH_SETMODEL_UPLOAD=copy: four hipMemcpyAsync from the staging buffer (copy engine); default: ONE kernel reads the staging buffer
    // through its device mapping and scatters the four arrays (and clears d_pendT) - no copy-engine command on this path
    static const char* up_env = getenv("BBH_SETMODEL_UPLOAD");
    static const char* sync_env = getenv("BBH_SETMODEL_SYNC");
    void* stage_dev = nullptr;
    const bool by_kernel = !(up_env && up_env[0] == 'c') && hipHostGetDevicePointer(&stage_dev, stage, 0) == hipSuccess;
    if (by_kernel) {
      bbh_scatter_args sa{};
      sa.src = (const unsigned char*)stage_dev;
      sa.dst[0] = (unsigned char*)h->d_xnT;   sa.off[0] = o_x; sa.bytes[0] = b_x;
      sa.dst[1] = (unsigned char*)h->d_nmask; sa.off[1] = o_m; sa.bytes[1] = b_m;
      sa.dst[2] = (unsigned char*)h->d_ystd;  sa.off[2] = o_y; sa.bytes[2] = b_y;
      sa.dst[3] = (unsigned char*)h->d_task;  sa.off[3] = o_t; sa.bytes[3] = b_t;
      sa.zero = (unsigned char*)h->d_pendT;   sa.zero_bytes = sizeof(double) * (size_t)h->dn * 16;
      const size_t longest = b_x > b_m ? b_x : b_m;
      static int sm_seq = 0;
      volatile int* flag_host = nullptr;
      if (sm_trace && h->pin_info) {  // (the Cholesky flag's pinned word is idle here: borrowed as the kernel's "I am running" stamp)
        void* fd = nullptr;
        if (hipHostGetDevicePointer(&fd, h->pin_info, 0) == hipSuccess) {
          sa.flag = (int*)fd;
          sa.flag_value = 0x5a000000 | (++sm_seq & 0xffffff);
          flag_host = h->pin_info;
        }
      }
      hipLaunchKernelGGL(bbh_scatter_kernel, dim3((unsigned)((longest / 8 + 255) / 256), 5), dim3(256), 0, s, sa);
      sm_stamp("memset");
      if (flag_host) {  // busy-poll plain memory: when does the GPU actually start the kernel, independent of the runtime's waits?
        const auto t_poll = std::chrono::steady_clock::now();
        while (*flag_host != sa.flag_value && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_poll).count() < 0.2) {}
        sm_stamp("kernel-ran");
      }
    } else {
      (void)hipGetLastError();
      BBH_HIP_TRY(h, hipMemsetAsync(h->d_pendT, 0, sizeof(double) * h->dn * 16, s));
      sm_stamp("memset");
      BBH_HIP_TRY(h, hipMemcpyAsync(h->d_xnT, stage + o_x, b_x, hipMemcpyHostToDevice, s));
      BBH_HIP_TRY(h, hipMemcpyAsync(h->d_nmask, stage + o_m, b_m, hipMemcpyHostToDevice, s));
      BBH_HIP_TRY(h, hipMemcpyAsync(h->d_ystd, stage + o_y, b_y, hipMemcpyHostToDevice, s));
      BBH_HIP_TRY(h, hipMemcpyAsync(h->d_task, stage + o_t, b_t, hipMemcpyHostToDevice, s));
    }
    int rc_stage = bbh_stage_done(h);
    if (rc_stage) return rc_stage;
    // (complete before the call returns, as the synchronous copies were: the handle's stream may be changed before the model is used -
    // bbh_set_stream, the captured-graph stream of BBH_FIT_GRAPH - and the side streams are not ordered behind h->stream)
    sm_stamp("enqueued");
    if (sync_env && sync_env[0] == 'p') {  // poll
      hipError_t q;
      while ((q = hipStreamQuery(s)) == hipErrorNotReady) {}
      (void)hipGetLastError();
    } else if (!(sync_env && sync_env[0] == '0')) {
      BBH_HIP_TRY(h, hipStreamSynchronize(s));
    }
    sm_stamp("synced");
  }
  sm_stamp("copies");
  h->xraw_host.assign(X_train_host, X_train_host + n * d);
  h->p = 0;
  h->pend_host.clear();
  h->factorized = false;
  h->info_clean = false;  // (d_info may be a fresh allocation)
  if (desc->kernel_kind == BBH_KERNEL_RFF) {
    int rc = bbh_rff_setup(h);
    if (rc) return rc;
  } else {
    bbh_rff_destroy(h);
  }
  h->have_model = true;
  return 0;
}

extern "C" int bbh_get_standardization(bbh_handle* h, double* ybar, double* ysd) {
  if (!h || !h->have_model) return -1;
  if (ybar) *ybar = h->ybar;
  if (ysd) *ysd = h->ysd;
  return 0;
}

// K -> L, X = L^-1, r, alpha.  Returns the Cholesky info flag (0 ok) via *info_out.
// info_out == nullptr: nothing is read back here (the caller fetches d_info together with its own results)
static int bbh_chol_and_alpha(bbh_handle* h, double jitter, int* info_out, double jitter_latent = -1.0) {
  hipStream_t s = h->stream;
  const int64_t np = h->np;
  bbh_launch_gram(h, jitter, jitter_latent < 0.0 ? jitter : jitter_latent);
  bbh_potrf_trtri(h);
  hipLaunchKernelGGL(bbh_resid_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, h->d_ystd, h->d_theta,
                     h->d_task, h->T, bbh_hadamard_offset(h), (int)h->n, (int)np, h->d_r);
  bbh_matvec(s, h->d_X, np, np, np, h->d_r, h->d_t);        // t = L^-1 r
  bbh_matvec_t(s, h->d_X, np, np, np, h->d_t, h->d_alpha);  // alpha = L^-T t
  if (!info_out) return 0;
  int info = 0;
  BBH_HIP_TRY(h, hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  if (info == -7 && h->potrf_tiles) {  // tile-dataflow launch gave up: redo with the per-step path
    if (getenv("BBH_TILE_TRACE")) fprintf(stderr, "bbh_chol_and_alpha: tile-dataflow launch gave up (np = %lld, spin limit %d)\n", (long long)h->np, h->tile_spin_limit);
    h->potrf_tiles = false;
    if (h->tile_spin_limit >= 1024) bbh_potrf_tiles_mark_unusable(h->device);
    return bbh_chol_and_alpha(h, jitter, info_out, jitter_latent);
  }
  *info_out = info;
  return 0;
}

// Everything one evaluation of the fit objective puts on h->stream: theta from the pinned staging buffer, Gram matrix,
// factorisation, inverse, the criterion's value and gradient slots, results into the pinned result buffers.
static int bbh_fit_enqueue(bbh_handle* h) {
  hipStream_t s = h->stream;
  const int64_t np = h->np, n = h->n;
  const int64_t tl = bbh_theta_len_of(h);
  if (bbh_is_rff(h)) return bbh_rff_fit_enqueue(h);  // the RFF kernel's model lives in feature space (bbh_rff.hip)
  {  // small models: the whole evaluation in one workgroup (bbh_linalg.hip), reading theta from and writing the results to the
     // pinned staging buffers themselves - one launch, no copies
    void *th_dev = nullptr, *out_dev = nullptr, *info_dev = nullptr;
    if (h->fit_small && h->np == 64 && hipHostGetDevicePointer(&th_dev, h->pin_theta, 0) == hipSuccess &&
        hipHostGetDevicePointer(&out_dev, h->pin_out, 0) == hipSuccess && hipHostGetDevicePointer(&info_dev, h->pin_info, 0) == hipSuccess &&
        bbh_fit_small_launch(h, 0.0, (const double*)th_dev, (double*)out_dev, (int*)info_dev))
      return 0;
    (void)hipGetLastError();
  }
  {  // 64 < np <= 1024: the whole evaluation as one dataflow launch (bbh_fitflow.hip), the same zero-copy staging
    void *th_dev = nullptr, *out_dev = nullptr, *info_dev = nullptr;
    // BBH_FIT_FLOW=2 / 3: this form for every eligible size (A/B).  Default (1): for 1024 < np <= 2048 only - the tile-dataflow
    // factorisation of the two-launch default needs all its tiles co-resident, which ends at 16 block rows on 256 CUs; the ticketed
    // roles of the one-launch form have no such requirement (launch by launch an evaluation at np = 1088 was 1.36 ms against 0.46 ms
    // at np = 1024: profiles/r06_fit_eval_beyond_1024.log)
    const bool one_launch = (h->fit_flow == 2 || h->fit_flow == 3) ? h->np <= 1024 : (h->fit_flow == 1 && h->np > 1024 && bbh_fit_flow_eligible(h));
    if (one_launch && h->np > 64 && hipHostGetDevicePointer(&th_dev, h->pin_theta, 0) == hipSuccess &&
        hipHostGetDevicePointer(&out_dev, h->pin_out, 0) == hipSuccess && hipHostGetDevicePointer(&info_dev, h->pin_info, 0) == hipSuccess) {
      *h->pin_info = -99;  // (sentinel: the kernel's last role writes the flag; a launch that gave up never does)
      // (theta as kernel arguments when it fits: ~250 workgroups fetching it from the host-mapped buffer is the slower way)
      if (bbh_fit_flow_launch(h, tl <= 52 ? nullptr : (const double*)th_dev, (double*)out_dev, (int*)info_dev, false, h->pin_theta, h->fit_flow == 3 && h->np <= 1024)) {
        h->flow_in_flight = true;
        return 0;
      }
    }
    (void)hipGetLastError();
  }
  {  // default for 64 < np <= 1024, zero copies: the tile-dataflow factorisation builds its Gram tiles itself, the dataflow tail reads the
     // factor - theta from, results into the host-mapped staging buffers (two launches)
    void *th_dev = nullptr, *out_dev = nullptr, *info_dev = nullptr;
    if (h->fit_flow == 1 && h->np <= 1024 && h->tile_gram && bbh_fit_flow_eligible(h) && hipHostGetDevicePointer(&th_dev, h->pin_theta, 0) == hipSuccess &&
        hipHostGetDevicePointer(&out_dev, h->pin_out, 0) == hipSuccess && hipHostGetDevicePointer(&info_dev, h->pin_info, 0) == hipSuccess) {
      // (theta through one H2D copy, not read from the host-mapped buffer by every workgroup: ~100 workgroups fetching it over the
      // host link at the same moment took 30 us - profiles/r05_tile_gram.log)
      // theta travels as kernel arguments of the two launches (<= 49 doubles): no copy in the stream, and no workgroup reads it from
      // the host-mapped buffer (BBH_TILE_GRAM_THETA=copy: one H2D copy and device reads, the A/B form)
      const char* th_env = getenv("BBH_TILE_GRAM_THETA");
      const bool by_value = !(th_env && th_env[0] == 'c') && tl <= 52;
      const double* th_src = by_value ? nullptr : h->d_theta;
      if (!by_value) BBH_HIP_TRY(h, hipMemcpyAsync(h->d_theta, h->pin_theta, sizeof(double) * tl, hipMemcpyHostToDevice, s));
      // K^-1's tiles ride in the factorisation launch where its workgroups and theirs are co-resident (n <= 832): they follow the rows of
      // L^-1 as these appear, and the launch behind it starts at alpha
      alignas(16) unsigned char mt_buf[128];
      static_assert(sizeof(mt_buf) >= 96, "pd_mt_args");
      const bool want_mt = h->tile_mt && bbh_fit_flow_mt_args(h, mt_buf);
      h->skip_x_memset = true;
      const bool tiles = bbh_potrf_trtri_from_inputs(h, th_src, h->pin_theta, want_mt ? mt_buf : nullptr);
      h->skip_x_memset = false;
      (void)th_dev;
      if (tiles) {
        *h->pin_info = -99;
        if (bbh_fit_flow_launch(h, th_src, (double*)out_dev, (int*)info_dev, true, h->pin_theta, false, h->tiles_did_mt)) {
          h->flow_in_flight = true;
          return 0;
        }
        (void)hipGetLastError();  // (the tail is unavailable: the evaluation starts over below, launch by launch)
        if (h->tiles_did_mt) bbh_fit_flow_reset(h);  // (the M-tile count of a tail that never ran must not be credited to a later one)
      }
    }
    (void)hipGetLastError();
  }
  BBH_HIP_TRY(h, hipMemcpyAsync(h->d_theta, h->pin_theta, sizeof(double) * tl, hipMemcpyHostToDevice, s));
  // One host synchronisation per evaluation: the Cholesky flag is fetched with the results at the end (after a failed
  // factorisation the remaining kernels run on NaNs, harmlessly, and the outcome is discarded).
  {  // default for 64 < np <= 1024: Gram + tile-dataflow factorisation as launches, then everything after the factor - K^-1, alpha,
     // (LOO: q, Q), value, gradient pairs, their sums - as ONE dataflow launch writing the results into the pinned buffers
     // (7-10 kernels, a memset and two copies before)
    void *out_dev = nullptr, *info_dev = nullptr;
    if (h->fit_flow == 1 && h->np <= 1024 && bbh_fit_flow_eligible(h) && hipHostGetDevicePointer(&out_dev, h->pin_out, 0) == hipSuccess &&
        hipHostGetDevicePointer(&info_dev, h->pin_info, 0) == hipSuccess) {
      bbh_launch_gram(h, 0.0, 0.0);
      h->skip_x_memset = true;  // (the tail reads the lower tiles of L^-1 only)
      bbh_potrf_trtri(h);
      h->skip_x_memset = false;
      *h->pin_info = -99;
      if (bbh_fit_flow_launch(h, h->d_theta, (double*)out_dev, (int*)info_dev, true)) {
        h->flow_in_flight = true;
        return 0;
      }
      (void)hipGetLastError();  // (resources: the handle has stopped using the form; this evaluation starts over on the launch path)
    }
    int rc0 = bbh_chol_and_alpha(h, 0.0, nullptr);
    if (rc0) return rc0;
  }
  int rc = 0;
  // M = X^T X.  One 64 x 64 output tile per workgroup means np / 64 squared workgroups walking all of K: 35 us at np = 512
  // on a quarter of the CUs.  Up to np = 1024 the product is split four ways along K (batched launch into four partial
  // matrices) and summed in a fixed order.
  if (h->d_Mpart) {
    bbh_gemm(s, true, false, np, np, np / 4, 1.0, h->d_X, np, (np / 4) * np, h->d_X, np, (np / 4) * np, 0.0, h->d_Mpart, np,
             np * np, 4);
    hipLaunchKernelGGL(bbh_sum4_kernel, dim3((unsigned)((np * np + 255) / 256)), dim3(256), 0, s, h->d_Mpart, np * np, h->d_M);
  } else {
    bbh_gemm(s, true, false, np, np, np, 1.0, h->d_X, np, 0, h->d_X, np, 0, 0.0, h->d_M, np, 0, 1);
  }
  const int crit = h->desc.criterion;
  if (crit == BBH_CRITERION_LOO) {
    hipLaunchKernelGGL(bbh_loo_vec_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, h->d_M, h->d_alpha,
                       (int)n, (int)np, h->d_u, h->d_w);
    bbh_matvec(s, h->d_M, np, np, np, h->d_w, h->d_q);  // q = M w
    hipLaunchKernelGGL(bbh_colscale_kernel, dim3((unsigned)((np * np + 255) / 256)), dim3(256), 0, s, h->d_M, h->d_u,
                       (int)np, h->d_Q2);
    bbh_gemm(s, false, false, np, np, np, 1.0, h->d_Q2, np, 0, h->d_M, np, 0, 0.0, h->d_Q, np, 0, 1);
  }
  hipMemsetAsync(h->d_out, 0, sizeof(double) * (1 + tl), s);
  hipLaunchKernelGGL(bbh_value_kernel, dim3(1), dim3(256), 0, s, h->d_K, h->d_M, h->d_r, h->d_alpha, h->d_q, h->d_task,
                     h->T, bbh_hadamard_offset(h), (int)n, (int)np, crit, h->d_out);
  const int nchunks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(bbh_grad_pair_kernel, dim3((unsigned)n, (unsigned)nchunks), dim3(256), 0, s, h->d_xnT, h->d_task,
                     h->d_nmask, h->d_theta, h->d_M, h->d_Q, h->d_alpha, h->d_q, (int)n, (int)np, h->dn, bbh_kern_spec_of(h),
                     h->T, bbh_hadamard_offset(h), crit, (int)tl, h->d_partial);
  const int hoff = bbh_hadamard_offset(h);
  hipLaunchKernelGGL(bbh_grad_reduce_kernel, dim3((unsigned)tl), dim3(256), 0, s, h->d_partial,
                     (int64_t)n * nchunks * 4, (int)tl, hoff >= 0 ? hoff + h->T : -1, hoff >= 0 ? hoff + 2 * h->T : -1,
                     h->d_out);
  BBH_HIP_TRY(h, hipMemcpyAsync(h->pin_out, h->d_out, sizeof(double) * (1 + tl), hipMemcpyDeviceToHost, s));
  BBH_HIP_TRY(h, hipMemcpyAsync(h->pin_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  return 0;
}

// The evaluation is ~60 short launches (45 kernels of 5 - 30 us, memsets, event edges between the two streams).  Enqueued
// one by one the host spends 586 us in launch calls per evaluation at n = 512 and then waits 10 us for the device
// (BBH_FIT_TRACE=1), which looks launch-bound - but the device's own dependency chain is as long: captured once per model
// into a hipGraph (both streams; the side stream joins the capture through its event edges; theta and the results
// travel through pinned staging buffers) and replayed, the same evaluation takes 147 us of hipGraphLaunch + 503 us on
// the device = 0.66 ms against 0.59 ms launch by launch, where device execution overlaps the enqueueing.  The chain
// (8 x [diagonal block 28 us -> panel 9 us -> trailing update 12 us] + inverse product, gradient) is what a faster fit
// has to shorten; the graph path stays available for that work (BBH_FIT_GRAPH=1) and is off by default.
static int bbh_fit_graph_build(bbh_handle* h) {
  bbh_ensure_side_stream(h);
  if (!h->fit_stream && hipStreamCreateWithFlags(&h->fit_stream, hipStreamNonBlocking) != hipSuccess) {
    h->fit_stream = nullptr;
    return -1;
  }
  hipStream_t saved = h->stream;
  h->stream = h->fit_stream;
  int rc = -1;
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(h->fit_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
    rc = bbh_fit_enqueue(h);
    const hipError_t e = hipStreamEndCapture(h->fit_stream, &graph);
    if (e != hipSuccess || !graph) rc = -1;
  }
  h->stream = saved;
  if (rc == 0 && hipGraphInstantiate(&h->fit_exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    h->fit_exec = nullptr;
    rc = -1;
  }
  if (graph) hipGraphDestroy(graph);
  (void)hipGetLastError();
  return rc;
}

extern "C" int bbh_fit_value_grad(bbh_handle* h, const double* theta_host, double* value_host, double* grad_host) {
  if (!h) return -1;
  if (!h->have_model || !theta_host || !value_host || !grad_host) {
    h->err = "bbh_fit_value_grad: no model / bad arguments";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  const auto t_begin = std::chrono::steady_clock::now();
  const int64_t tl = bbh_theta_len(h);
  for (int64_t i = 0; i < tl; i++)
    if (!(theta_host[i] == theta_host[i])) {
      h->err = "theta contains NaN";
      return -3;
    }
  h->theta.assign(theta_host, theta_host + tl);
  h->factorized = false;
  if (!h->pin_theta) {  // pinned staging: theta in, [value, gradient] and the Cholesky flag out
    BBH_HIP_TRY(h, hipHostMalloc((void**)&h->pin_theta, sizeof(double) * tl, hipHostMallocDefault));
    BBH_HIP_TRY(h, hipHostMalloc((void**)&h->pin_out, sizeof(double) * (1 + tl), hipHostMallocDefault));
    BBH_HIP_TRY(h, hipHostMalloc((void**)&h->pin_info, sizeof(int), hipHostMallocDefault));
  }
  memcpy(h->pin_theta, theta_host, sizeof(double) * tl);
  if (h->fit_graph_mode && !h->fit_exec && !h->fit_graph_failed && bbh_fit_graph_build(h) != 0) h->fit_graph_failed = true;
  hipStream_t s = h->stream;
  if (h->fit_exec) {
    BBH_HIP_TRY(h, hipStreamSynchronize(h->stream));  // earlier work on the handle's stream (normally idle here)
    s = h->fit_stream;
    BBH_HIP_TRY(h, hipGraphLaunch(h->fit_exec, s));
  } else {
    int rc = bbh_fit_enqueue(h);
    if (rc) return rc;
  }
  const auto t_enq = std::chrono::steady_clock::now();
  BBH_HIP_TRY(h, hipStreamSynchronize(s));
  if (getenv("BBH_FIT_TRACE")) {  // host time spent enqueueing vs waiting for the device
    const auto t_end = std::chrono::steady_clock::now();
    fprintf(stderr, "bbh_fit_value_grad (%s): enqueue %.1f us, wait %.1f us\n", h->fit_exec ? "graph" : "launches",
            std::chrono::duration<double, std::micro>(t_enq - t_begin).count(),
            std::chrono::duration<double, std::micro>(t_end - t_enq).count());
  }
  if (h->flow_in_flight) {
    h->flow_in_flight = false;
    h->info_clean = *h->pin_info != -99;  // (the tail's last role copied the Cholesky flag out and reset it)
    if (*h->pin_info == -99) {  // the dataflow launch never reported: one of its waits ran out of polls - launch path from now on
      // (a reported -7 is the tile-dataflow FACTORISATION in front of it giving up: handled below, the dataflow tail stays in use)
      if (getenv("BBH_TILE_TRACE")) fprintf(stderr, "bbh_fit_value_grad: one-launch evaluation gave up (np = %lld, flag %d)\n", (long long)h->np, *h->pin_info);
      bbh_fit_flow_reset(h);
      return bbh_fit_value_grad(h, theta_host, value_host, grad_host);
    }
  }
  if (*h->pin_info == -7 && h->potrf_tiles) {  // the tile-dataflow launch gave up (workgroups not co-resident): per-step path
    if (getenv("BBH_TILE_TRACE")) fprintf(stderr, "bbh_fit_value_grad: tile-dataflow launch gave up (np = %lld, spin limit %d)\n", (long long)h->np, h->tile_spin_limit);
    h->potrf_tiles = false;
    if (h->tile_spin_limit >= 1024) bbh_potrf_tiles_mark_unusable(h->device);  // (not when a test forced the give-up with a tiny poll budget)
    if (h->fit_exec) {
      hipGraphExecDestroy(h->fit_exec);
      h->fit_exec = nullptr;
    }
    return bbh_fit_value_grad(h, theta_host, value_host, grad_host);
  }
  if (*h->pin_info != 0) {
    *value_host = -INFINITY;
    for (int64_t i = 0; i < tl; i++) grad_host[i] = 0.0;
    return 1;
  }
  *value_host = h->pin_out[0];
  for (int64_t i = 0; i < tl; i++) grad_host[i] = h->pin_out[1 + i];
  return 0;
}

extern "C" int bbh_factorize(bbh_handle* h, const double* theta_host, double* jitter_used) {
  if (!h) return -1;
  if (!h->have_model || !theta_host) {
    h->err = "bbh_factorize: no model / bad arguments";
    return -1;
  }
  BBH_HIP_TRY(h, hipSetDevice(h->device));
  int rc = bbh_upload_theta(h, theta_host);
  if (rc) return rc;
  h->factorized = false;
  if (bbh_is_rff(h)) {
    rc = bbh_rff_factorize(h);
    if (rc) return rc;
    if (jitter_used) *jitter_used = 0.0;
    h->p = 0;
    h->pend_host.clear();
    h->factorized = true;
    return 0;
  }
  // gpytorch psd_safe_cholesky: plain attempt, then jitter 1e-8 * 10^i, i = 0..2
  // A model with latent rows (noise mask 0: the baseline rows of an extended qLogNEHVI model) whose factorisation fails INSIDE
  // the latent block: BoTorch factorises the baseline's joint posterior covariance - the Schur complement of that block, in the
  // target's original scale - with the same ladder, which is this matrix with the jitter on the latent rows' diagonal only
  // (1e-8 * 10^i / ysd^2 in standardised units).  A failure inside the noisy block keeps the whole-diagonal ladder.
  bool latent = false;
  for (int64_t i = 0; i < h->n && !latent; i++) latent = h->nmask_host[i] == 0.0;
  double jitter = 0.0, jitter_latent = 0.0;
  int info = 0;
  for (int attempt = 0; attempt < 4; attempt++) {
    rc = bbh_chol_and_alpha(h, jitter, &info, jitter_latent);
    if (rc) return rc;
    if (info == 0) break;
    const double step = 1e-8 * pow(10.0, attempt);
    if (latent && info > 0 && h->nmask_host[info - 1 < h->n ? info - 1 : h->n - 1] == 0.0 && jitter == 0.0) {
      jitter_latent = step / (h->ysd * h->ysd);
    } else {
      jitter = step;
      jitter_latent = step;
    }
  }
  if (info != 0) {
    h->err = "bbh_factorize: train covariance not positive definite (even with jitter 1e-6)";
    return -4;
  }
  if (jitter_used) *jitter_used = jitter > 0.0 ? jitter : jitter_latent;
  h->p = 0;
  h->pend_host.clear();
  h->factorized = true;
  rc = bbh_pack_operands(h);
  if (rc) {
    h->factorized = false;
    return rc;
  }
  return 0;
}
