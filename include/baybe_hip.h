/*
 * baybe_hip.h — C-ABI of libbaybe_hip.so, the MI355X (gfx950) implementation of
 * BayBE's GP-surrogate recommend() hot path.
 *
 * The reference (emdgroup/baybe) is pure Python; it has no FFI of its own.  Each
 * entry point below replaces the *third-party call* BayBE makes at the cited
 * line (botorch / gpytorch, CPU), so that a ctypes stub at that call site is
 * the whole integration:
 *
 *   bbh_set_model            botorch.models.SingleTaskGP(...)            baybe/surrogates/gaussian_process/core.py:331-339
 *                            Normalize / Standardize                     baybe/surrogates/gaussian_process/core.py:301-306
 *   bbh_fit_value_grad       one closure call of fit_gpytorch_mll        baybe/surrogates/gaussian_process/core.py:340-341
 *                            (ExactMLL / LOO-PL value + gradient)        baybe/surrogates/gaussian_process/components/fit_criterion.py:31-41
 *   bbh_factorize            prediction-strategy caches (L, alpha, L^-T) baybe/surrogates/gaussian_process/core.py:268-269 (first posterior call)
 *   bbh_posterior            model.posterior(X) for N q=1 t-batches      baybe/surrogates/base.py:249-272, 308-384
 *   bbh_train_posterior_mean posterior mean at the training inputs       baybe/acquisition/_builder.py:141-161 (best_f, 256-265)
 *   bbh_qlogei_q1            qLogExpectedImprovement.forward, q'=1       baybe/acquisition/acqfs.py:219-223; baybe/acquisition/base.py:112-159
 *   bbh_pending_set /        set_X_pending + joint q'-batch posterior    baybe/acquisition/_builder.py:326-334
 *   bbh_cross_cov /
 *   bbh_qlogei_pending       qLogEI forward with pending points
 *   bbh_argmax               the argmax of optimize_acqf_discrete        baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126
 *   bbh_set_model_ex /       ModelListGP member conditioned on sampled   baybe/surrogates/composite.py:125-134;
 *   bbh_posterior_joint /    baseline values (joint draw of f(X) with    baybe/acquisition/_builder.py:319-324 (X_baseline)
 *   bbh_set_mean_columns /   f(X_baseline), cached Cholesky root)
 *   bbh_posterior_columns(_sm)
 *   bbh_qlognehvi(_sm)       qLogNoisyExpectedHypervolumeImprovement     baybe/acquisition/acqfs.py:477-484
 *
 * Conventions
 *  - extern "C"; every function returns 0 on success, <0 on error;
 *    bbh_last_error(h) returns a message owned by the library (valid until the
 *    next call on that handle).
 *  - "_dev" arguments are DEVICE pointers (row-major, contiguous, fp64 unless
 *    noted) owned by the caller (e.g. torch.Tensor.data_ptr() of a ROCm tensor
 *    used purely as a container).  "_host" arguments are HOST pointers; they are
 *    O(n*d) model data or O(S*q) base samples and are copied by the call.
 *  - All work is enqueued on the handle's stream (bbh_set_stream; default: the
 *    legacy null stream).  Calls that return host data synchronise that stream.
 *  - One handle = one single-output GP on one device.  Not thread-safe per
 *    handle; distinct handles are independent.
 *  - No CPU fallback exists: without a HIP device bbh_create fails.
 */
#ifndef BAYBE_HIP_H
#define BAYBE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbh_handle bbh_handle;

enum bbh_kernel_kind {
  BBH_KERNEL_MATERN12 = 0,
  BBH_KERNEL_MATERN32 = 1,
  BBH_KERNEL_MATERN52 = 2, /* BayBE default: presets/baybe.py:100-107 */
  BBH_KERNEL_RBF = 3,
  /* gpytorch PiecewisePolynomialKernel(q), baybe/kernels/basic.py:114-131: (1 - r)_+^(j + q) P_q(r), j = floor(dn / 2) + q + 1.
   * Compact support; evaluated through the materialised-K* posterior path (like composite kernels). */
  BBH_KERNEL_PIECEWISE0 = 4,
  BBH_KERNEL_PIECEWISE1 = 5,
  BBH_KERNEL_PIECEWISE2 = 6,
  BBH_KERNEL_PIECEWISE3 = 7,
  /* gpytorch RQKernel, baybe/kernels/basic.py:202-216: (1 + r^2 / (2 alpha))^-alpha with a learnable alpha per kernel
   * (theta: one slot per factor at the very end, present when any factor is an RQ kernel). */
  BBH_KERNEL_RQ = 8,
  /* Dot-product kernels (baybe/kernels/basic.py:20-46, 135-163): functions of s = sum_j x_j x'_j / w_j^2 over the normalised
   * inputs instead of a scaled squared distance; the lengthscale slots of theta carry the weights w_j (huge for columns the
   * kernel does not act on), and bbh_fit_value_grad returns d/dw_j in those slots.
   *   gpytorch LinearKernel with ARD variances v_j: k = sum_j v_j x_j x'_j, i.e. w_j = v_j^-1/2;
   *   gpytorch PolynomialKernel(power p): (x . x' + offset)^p, p = 1 .. 4, w_j = 1, offset in the factor's alpha slot
   *   (where an RQ kernel keeps its alpha).
   * k(x, x) is not constant: these kinds are evaluated through the materialised-K* posterior path only. */
  BBH_KERNEL_LINEAR = 9,
  BBH_KERNEL_POLY1 = 10,
  BBH_KERNEL_POLY2 = 11,
  BBH_KERNEL_POLY3 = 12,
  BBH_KERNEL_POLY4 = 13,
  /* gpytorch PeriodicKernel (baybe/kernels/basic.py:73-112): exp(-2 sum_j sin^2(pi (x_j - x'_j) / p_j) / l_j) with one lengthscale
   * l_j (lengthscale slots; note: not squared) and one period p_j per column.  theta: a block of F * dn periods at the very end
   * (after the alpha slots), present when any factor is periodic; slots of the other factors are ignored.  Materialised-K* path. */
  BBH_KERNEL_PERIODIC = 14,
  /* gpytorch.kernels.RFFKernel (baybe/kernels/basic.py:183-199): k(x, x') = z(x) . z(x') / D with z = [cos(x (W / l)), sin(x (W / l))], D =
   * num_samples frequencies W [dn, D] drawn once per model (bbh_set_rff_weights BEFORE bbh_set_model); the model is held in feature space:
   * fit and posterior cost O(n D^2 + D^3) and O(D^2) per candidate.  One task, single kernel, MLL, D <= 64; theta as for RBF. */
  BBH_KERNEL_RFF = 15
};

enum bbh_criterion {
  BBH_CRITERION_MLL = 0, /* gpytorch.ExactMarginalLogLikelihood   (1 task)  */
  BBH_CRITERION_LOO = 1  /* gpytorch.mlls.LeaveOneOutPseudoLikelihood (>1)  */
};

/* Architecture of the GP (what BayBE's component factories decide). */
typedef struct bbh_model_desc {
  int32_t kernel_kind;     /* enum bbh_kernel_kind                                   */
  int32_t d;               /* comp-rep columns, incl. the task column if any          */
  int32_t task_col;        /* column index of the INT-coded task parameter, -1 = none */
  int32_t n_tasks;         /* 1 = single task                                         */
  int32_t use_outputscale; /* 1 = ScaleKernel wrapper (user kernels)                   */
  int32_t criterion;       /* enum bbh_criterion                                      */
  int32_t hadamard;        /* 1 (n_tasks > 1 only) = one noise variance and one constant mean PER TASK:
                              HadamardGaussianLikelihood + HadamardConstantMean of the multi-task HVARFNER /
                              BOTORCH presets (surrogates/gaussian_process/components/_gpytorch.py:15-75)     */
  /* Composite kernels (baybe/kernels/composite.py:60-91): n_factors in 2..4 stationary factors, each with its own ARD
   * lengthscales over the numerical columns, combined as a ProductKernel (combine 0) or an AdditiveKernel (combine 1);
   * factor f may sit in its own ScaleKernel (factor_scaled[f]).  n_factors 0 / 1 = the single kernel_kind above;
   * factor_kind[0] must equal kernel_kind otherwise.  Evaluated through the materialised-K* posterior path (the fused
   * kernels are specialised for one factor). */
  int32_t n_factors;
  int32_t combine;
  int32_t factor_kind[4];
  int32_t factor_scaled[4];
  /* combine 2 = a sum whose members are products or single kernels - e.g. (Matern * Matern) + (Matern + Matern), the nested entry of the
   * reference's kernel matrix (tests/test_iterations.py:294-296): factor f multiplies into term factor_group[f] in 0..3,
   * k = sum_g prod_{f in g} os_f k_f.  Ignored for combine 0 (one term) and 1 (one factor per term). */
  int32_t factor_group[4];
} bbh_model_desc;

/*
 * Natural hyper-parameter vector "theta" (doubles), length bbh_theta_len(h):
 *   [0]            noise variance s2 (standardised target scale)
 *   [1]            constant mean c
 *   [2]            outputscale (ignored unless use_outputscale)
 *   [3 .. 3+dn)    ARD lengthscales of the dn = d - (task_col>=0) numerical columns,
 *                  in column order (normalised input scale)
 *   [3+dn .. +T*T) task covariance B[t][t'] row-major (only when n_tasks > 1)
 *   [.. +T) [.. +T) per-task noise variances, then per-task constant means (only with desc.hadamard;
 *                  slots [0] and [1] are then ignored and their gradients are 0)
 *   [.. +(F-1)*dn)  lengthscales of the factors 1 .. F-1 of a composite kernel (factor 0 uses [3 .. 3+dn))
 *   [.. +F)         per-factor outputscales (1 for an unscaled factor; its gradient slot is still filled)
 *                  k = outputscale * (prod_f | sum_f) os_f k_f(r_f) * B[t][t']
 *   [.. +F)         alpha of every factor (only when at least one factor is BBH_KERNEL_RQ; 1 for the other factors)
 * Gradients are returned in the same layout (for B: dL/dB[t][t'], accumulated
 * over ordered pairs, i.e. the matrix S with dL = sum_tt' S[t][t'] dB[t][t']).
 * Constraint transforms (softplus) and prior terms are O(d) scalar work and stay
 * with the host driver (baybe_amd/gp_spec.py, baybe_amd/engine.py).
 */

/* ---- lifecycle --------------------------------------------------------------------- */
int bbh_create(int device_id, bbh_handle** out);
int bbh_destroy(bbh_handle* h);
const char* bbh_last_error(bbh_handle* h);
int bbh_set_stream(bbh_handle* h, void* hip_stream);
/* library/ABI version: major*10000 + minor*100 + patch */
int bbh_version(void);
/* device self-test of the fp64 MFMA fragment layout the kernels rely on (0 = ok) */
int bbh_selftest(bbh_handle* h);

/* ---- model ------------------------------------------------------------------------- */
/* The random frequencies of a BBH_KERNEL_RFF model: W_host [dn, D] (numerical columns in comp-rep order x num_samples), as gpytorch's
 * RFFKernel._init_weights draws them (torch.randn(d, D)); kept on the handle for the next bbh_set_model / bbh_set_model_ex. */
int bbh_set_rff_weights(bbh_handle* h, const double* W_host, int32_t dn, int32_t D);
/* X_train_host [n,d] raw comp-rep rows; y_train_host [n] raw targets;
 * lo_host/hi_host [d] scaling bounds of every column (task column ignored).
 * Normalises the numerical columns, standardises y (Bessel std, <1e-8 -> 1). */
int bbh_set_model(bbh_handle* h, const bbh_model_desc* desc, int64_t n,
                  const double* X_train_host, const double* y_train_host,
                  const double* lo_host, const double* hi_host);
/* Same, with (a) an optional per-point noise mask [n] (0 = noise-free observation: the point is a
 * latent function value, as the sampled baseline values of qLogNEHVI are) and (b) an optional fixed
 * standardisation (use_given_std != 0: ybar/ysd are taken as given instead of computed from y). */
int bbh_set_model_ex(bbh_handle* h, const bbh_model_desc* desc, int64_t n, const double* X_train_host,
                     const double* y_train_host, const double* lo_host, const double* hi_host,
                     const uint8_t* noise_mask_host, int use_given_std, double ybar, double ysd);
int64_t bbh_theta_len(bbh_handle* h);
/* standardisation constants chosen by bbh_set_model */
int bbh_get_standardization(bbh_handle* h, double* ybar, double* ysd);

/* Data term of the fit objective and its gradient w.r.t. theta:
 *   MLL: log N(y~ | c 1, K_theta + s2 I);  LOO: sum_i log N(y~_i | mu_-i, s2_-i).
 * Returns 1 (and value=-inf) when K_theta + s2 I is not positive definite. */
int bbh_fit_value_grad(bbh_handle* h, const double* theta_host, double* value_host,
                       double* grad_host);

/* Build the prediction caches for theta: L = chol(K + s2 I) (jitter 1e-8*10^i on
 * failure, as gpytorch psd_safe_cholesky), alpha = (K + s2 I)^-1 (y~ - c), the
 * packed L^-T operand of the variance contraction.  jitter_used may be NULL. */
int bbh_factorize(bbh_handle* h, const double* theta_host, double* jitter_used);

/* ---- posterior --------------------------------------------------------------------- */
/* Marginal posterior (original target scale, no observation noise) of N candidates.
 * X_dev [N, ldx>=d] raw comp-rep rows on the device; mean_dev / var_dev [N]. */
int bbh_posterior(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx,
                  double* mean_dev, double* var_dev);
/* Same through the unfused verification path (materialised K(X*,X), generic GEMM). */
int bbh_posterior_unfused(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx,
                          double* mean_dev, double* var_dev);
/* Joint posterior (original scale, no observation noise) of q <= 4096 points given on the host:
 * mean_host [q], cov_host [q,q]. */
int bbh_posterior_joint(bbh_handle* h, const double* Xq_host, int64_t q, double* mean_host, double* cov_host);
/* Alternative target columns for the mean contraction: Y_host [n, S] (original target scale, point
 * major).  Computes alpha_s = (K + s2 M)^-1 (y~_s - c) for every column on the device.  With the
 * extended model of bbh_set_model_ex this yields, per MC sample s, the posterior mean of f(x)
 * conditioned on the sampled baseline values. */
int bbh_set_mean_columns(bbh_handle* h, const double* Y_host, int64_t S);
/* tmat_dev [N, S]: posterior mean of every candidate under each target column (original scale). */
int bbh_posterior_columns(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev);
/* The same in SAMPLE-MAJOR layout, tmat_dev [S, N]: what bbh_qlognehvi_sm reads.  One thread of the scoring kernel owns one
 * candidate and walks the samples; with the candidate-major [N, S] layout the 64 lanes of a wave touch 64 different cache lines
 * per load (17 GB of L2 <-> fabric traffic per 1e5 x 512 x 3 pass instead of the 1.2 GB the values occupy), sample-major they
 * read 512 contiguous bytes. */
int bbh_posterior_columns_sm(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* tmat_dev);
/* Posterior mean at the n training inputs -> host (for best_f). */
int bbh_train_posterior_mean(bbh_handle* h, double* mean_host);

/* ---- qLogEI ------------------------------------------------------------------------ */
/* q'=1 scores: scores[i] = logmeanexp_s log_fatplus(sign*(mean_i + sd_i z_s) - best_f; 1e-6).
 * z_host [S] Sobol-normal base samples; alive_dev [N] uint8 or NULL (0 -> score = -inf). */
int bbh_qlogei_q1(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N,
                  const double* z_host, int64_t S, double best_f, double sign,
                  const uint8_t* alive_dev, double* scores_dev);

/* The same scores AND their k best (descending, ties -> lower index first; (-inf, -1) beyond the number of scored candidates) in
 * one call: the whole tail of a selection step of optimize_acqf_discrete (baybe/recommenders/pure/bayesian/botorch/
 * discrete.py:120-126) - scores kernel, one selection kernel whose results land in host-mapped memory, one synchronisation. */
int bbh_qlogei_q1_topk(bbh_handle* h, const double* mean_dev, const double* var_dev, int64_t N,
                       const double* z_host, int64_t S, double best_f, double sign,
                       const uint8_t* alive_dev, double* scores_dev, int64_t k, double* vals_host, int64_t* idx_host);

/* Fused scoring pass (the hot call of optimize_acqf_discrete's first greedy step): posterior AND
 * q'=1 qLogEI in one kernel.  mean_dev / var_dev may be NULL (scores only). */
int bbh_score_qlogei(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, const double* z_host,
                     int64_t S, double best_f, double sign, const uint8_t* alive_dev, double* mean_dev,
                     double* var_dev, double* scores_dev);

/* Pending points (base pending + greedy picks), p <= BBH_MAX_PENDING.  Computes and
 * caches beta_j = (K+s2I)^-1 k(X, P_j), the pending posterior mean [p] and covariance
 * [p,p] (returned to the host if the pointers are non-NULL). */
#define BBH_MAX_PENDING 15
int bbh_pending_set(bbh_handle* h, const double* Xpend_host, int64_t p, double* mean_p_host,
                    double* cov_pp_host);
/* Posterior cross-covariance of every candidate with the pending points: cross_dev [N,p]. */
int bbh_cross_cov(bbh_handle* h, const double* X_dev, int64_t N, int64_t ldx, double* cross_dev);
/* q' = 1+p scores of the t-batches [x_i ; pending]; z_host [S, 1+p]. */
int bbh_qlogei_pending(bbh_handle* h, const double* mean_dev, const double* var_dev,
                       const double* cross_dev, int64_t N, const double* z_host, int64_t S,
                       double best_f, double sign, const uint8_t* alive_dev, double* scores_dev);

/* ---- the other acquisition functions of acqfs.py:161-290 on the same inputs ----------- */
enum bbh_acq_kind {
  BBH_ACQ_QLOGEI = 0, BBH_ACQ_QEI = 1, BBH_ACQ_QPI = 2, BBH_ACQ_QSR = 3, BBH_ACQ_QUCB = 4, BBH_ACQ_QPSTD = 5,
  BBH_ACQ_PM = 10, BBH_ACQ_PSTD = 11, BBH_ACQ_UCB = 12, BBH_ACQ_EI = 13, BBH_ACQ_LOGEI = 14, BBH_ACQ_PI = 15
};
/* MC family, q'=1 and q'=1+p (mean_s max_j u(obj_sj); qPI tau = 1e-3; qUCB/qPSTD use the sample mean). */
int bbh_mc_acq_q1(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev, int64_t N,
                  const double* z_host, int64_t S, double best_f, double sign, double beta,
                  const uint8_t* alive_dev, double* scores_dev);
int bbh_mc_acq_pending(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev,
                       const double* cross_dev, int64_t N, const double* z_host, int64_t S, double best_f,
                       double sign, double beta, const uint8_t* alive_dev, double* scores_dev);
/* qLogEI of N t-batches [x_i ; pending] with the pending statistics GIVEN by the caller, 1 <= p <= 63 pending points - in
 * particular more than 15 (the reference's optimize_acqf_discrete has no cap on batch_size + pending experiments,
 * baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126); p <= 15 runs the kernels of bbh_qlogei_pending (a greedy loop
 * that already holds the statistics of a superset of its picks saves the bbh_pending_set round trip per step).  The caller
 * supplies what bbh_pending_set keeps for p <= 15: cross_dev [N, p] (columns from bbh_cross_cov over chunks of <= 15 pending
 * points - a column does not depend on the other pending points), mean_p_host [p] and cov_pp_host [p, p] (bbh_posterior_joint of
 * the pending points), z_host [S, 1 + p].  The per-candidate Cholesky factors live in a global workspace of
 * (p + 1)(p + 2) / 2 x N doubles. */
int bbh_qlogei_pending_big(bbh_handle* h, const double* mean_dev, const double* var_dev, const double* cross_dev,
                           int64_t N, int64_t p, const double* mean_p_host, const double* cov_pp_host,
                           const double* z_host, int64_t S, double best_f, double sign, const uint8_t* alive_dev,
                           double* scores_dev);
/* Analytic family (q = 1): PM, PSTD(+-), UCB(beta), EI, LogEI, PI; sigma^2 clamped at 1e-12. */
int bbh_analytic_acq(bbh_handle* h, int32_t kind, const double* mean_dev, const double* var_dev, int64_t N,
                     double best_f, double sign, double beta, int32_t maximize, const uint8_t* alive_dev,
                     double* scores_dev);

/* ---- qLogNEHVI ---------------------------------------------------------------------- */
#define BBH_MAX_OBJECTIVES 4
/* q'=1 scores over m <= 4 independent outputs.  Per output o: tmat_dev[o] [N,S] conditional means
 * per MC sample (bbh_posterior_columns of the extended model), var_dev[o] [N] conditional variance;
 * sample f_o(x)_s = tmat[o][i][s] + sqrt(var[o][i]) zx_host[s*m+o], oriented by sign_host[o].
 * Cells of sample s: cell_off_host[s] .. cell_off_host[s+1] (prefix offsets, [S+1]); cell c stores
 * cell_lo_host[c*m+o] (lower bound) and cell_loglen_host[c*m+o] = log(min(upper,1e10) - lower).
 * score = logmeanexp_s logsumexp_c sum_o fatmin(log_fatplus(f_o - lo; 1e-6), loglen; 1e-2). */
int bbh_qlognehvi(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                  const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                  const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                  const uint8_t* alive_dev, double* scores_dev);
/* bbh_qlognehvi with the conditional means in sample-major layout, tmat_dev[o] [S, N] (bbh_posterior_columns_sm). */
int bbh_qlognehvi_sm(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                  const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                  const int64_t* cell_off_host, const double* cell_lo_host, const double* cell_loglen_host,
                  const uint8_t* alive_dev, double* scores_dev);

/* Baseline pruning support (prune_inferior_points_multi_objective): obj_host [S, n, m] oriented
 * objective samples; counts_host[i] = number of samples in which point i is non-dominated and above
 * ref_host [m] in every objective. */
int bbh_pareto_frequency(bbh_handle* h, const double* obj_host, int64_t S, int64_t n, int32_t m,
                         const double* ref_host, int64_t* counts_host);

/* bbh_pareto_frequency with the objective samples already on the device (obj_dev [S, n, m], as bbh_nehvi_samples writes them). */
int bbh_pareto_frequency_dev(bbh_handle* h, const double* obj_dev, int64_t S, int64_t n, int32_t m,
                             const double* ref_host, int64_t* counts_host);

/* qLogNEHVI set-up of one target on the device (replaces: baseline joint posterior -> host Cholesky -> host samples -> an
 * (n + nb) x S host array for bbh_set_mean_columns).  h holds the target's model extended by nb baseline rows as noise-free
 * observations (bbh_set_model_ex with a noise mask: the last nb rows; their target values are not read) and is factorised.
 * BoTorch's joint draw of the baseline values through the cached Cholesky root (qLogNoisyExpectedHypervolumeImprovement with
 * cache_root, built at baybe/acquisition/_builder.py:319-324) is the generative form of that factor,
 * y_ext,s = c + L_ext [t; z_s] with t = L^-1 (y - c) on the training rows and z_s = z_host[s, :] the sample's base samples:
 * Fb_dev[(s * nb + b) * m + o] = sign * (baseline row b of y_ext,s, original target scale); with want_columns != 0 the S weight
 * columns L_ext^-T [t; z_s] of the model conditioned on each sample are installed for bbh_posterior_columns(_sm).
 * Asynchronous on the handle's stream. */
int bbh_nehvi_samples(bbh_handle* h, const double* z_host, int64_t S, int64_t nb, double sign, int32_t o, int32_t m,
                      double* Fb_dev, int32_t want_columns);
/* The same with the base samples already on the device (bbh_sobol_normal_dev): sample s, baseline row b reads
 * z_dev[s * ld + cols_dev[b]] (cols_dev [nb] int32: where the row's base sample sits in a draw over all baseline rows and
 * targets - repeated baseline rows are skipped by the caller). */
int bbh_nehvi_samples_dev(bbh_handle* h, const double* z_dev, int64_t ld, const int32_t* cols_dev, int64_t S, int64_t nb, double sign,
                          int32_t o, int32_t m, double* Fb_dev, int32_t want_columns);

/* Box decompositions on the device: one wavefront per MC sample over Fb_dev [S, nb, m] (oriented objective samples, as
 * bbh_nehvi_samples writes them) - the algorithm and visiting order of bbh_cells_create, the cell lists stay on the handle
 * for bbh_qlognehvi_cells.  *total_out = cells over all samples; *overflow_out = samples whose list of local upper bounds
 * exceeded the kernel's capacity (then, or for nb > 512, the caller takes the host form).  bbh_cells_read_dev copies the
 * lists out in the layout of bbh_cells_get (tests, statistics). */
int bbh_cells_build_dev(bbh_handle* h, const double* Fb_dev, int64_t S, int64_t nb, int32_t m, const double* ref_host,
                        int64_t* total_out, int64_t* overflow_out);
int bbh_cells_read_dev(bbh_handle* h, int64_t* off_host, double* lo_host, double* loglen_host);
/* bbh_qlognehvi_sm against the handle's device-resident cell lists (bbh_cells_build_dev with the same S and m). */
int bbh_qlognehvi_cells(bbh_handle* h, int32_t m, int64_t N, const double* const* tmat_dev,
                        const double* const* var_dev, const double* sign_host, const double* zx_host, int64_t S,
                        const uint8_t* alive_dev, double* scores_dev);

/* Box decomposition of the non-dominated region, one per MC sample (host code, no device work; BoTorch's
 * FastNondominatedPartitioning inside qLogNoisyExpectedHypervolumeImprovement, built at
 * baybe/acquisition/_builder.py:319-324).  obj_host [S, n, m] oriented baseline objective samples, ref_host [m].
 * bbh_cells_create computes every sample's disjoint boxes and returns an opaque object and the total box count;
 * bbh_cells_get copies them out in the layout bbh_qlognehvi takes (off_host [S+1], lo_host / loglen_host
 * [total, m], loglen = log(min(upper, 1e10) - lower)); bbh_cells_destroy frees the object. */
int bbh_cells_create(const double* obj_host, int64_t S, int64_t n, int32_t m, const double* ref_host,
                     void** cells_out, int64_t* total_out);
int bbh_cells_get(void* cells, int64_t* off_host, double* lo_host, double* loglen_host);
int bbh_cells_destroy(void* cells);

/* Scrambled Sobol points bitwise as torch.quasirandom.SobolEngine(dimension, scramble=True, seed) draws them (host code, integer
 * arithmetic; the engine botorch's SobolQMCNormalSampler uses).  state [dim, 30]: the engine's unscrambled direction numbers,
 * scrambled in place with the engine's random lower-triangular bit matrices ltm [dim, 30, 30] (0 / 1 entries as drawn, before
 * tril);  bbh_sobol_draw: out [n, dim] = the first n points for the scrambled state and the integer shift [dim]. */
int bbh_sobol_scramble(int64_t* state, const int64_t* ltm, int64_t dim);
int bbh_sobol_draw(const int64_t* state, const int64_t* shift, int64_t n, int64_t dim, double* out);
/* The sampler's complete base-sample draw, sqrt(2) erfinv(2 v - 1) of the engine's points for a seed (SobolQMCNormalSampler, built
 * through BoTorch's acquisition constructors at baybe/acquisition/_builder.py:195-334): state0 [dim, 30] = the engine's UNSCRAMBLED
 * direction numbers; the scrambling bits come from an MT19937 restating torch's CPU generator.  bbh_sobol_normal: host code,
 * out_host [n, dim];  bbh_sobol_normal_dev: generator and scrambling on the host, points and transform on the device into
 * out_dev [n, dim], asynchronous on the handle's stream. */
int bbh_sobol_normal(const int64_t* state0, uint64_t seed, int64_t n, int64_t dim, double* out_host);
int bbh_sobol_normal_dev(bbh_handle* h, const int64_t* state0, uint64_t seed, int64_t n, int64_t dim, double* out_dev);

/* 64-bit content key of host buffers (host code, std::threads): multiply-fold hash over 4 MB pieces, the pieces' digests folded in
 * order.  Keys the device-resident copy of the discrete subspace's computational representation on its content
 * (baybe/searchspace/discrete.py:704-735 hands the frame out on every call; botorch/discrete.py:123 converts it per call). */
uint64_t bbh_content_key(const void* const* bufs, const int64_t* lens, int32_t nbuf, int32_t threads);

/* ---- selection --------------------------------------------------------------------- */
/* First-index argmax of scores_dev [N] (NaN never wins) -> host. */
int bbh_argmax(bbh_handle* h, const double* scores_dev, int64_t N, double* best_val_host,
               int64_t* best_idx_host);
/* k best scores, descending, ties -> lower index first; -> host arrays [k]. */
int bbh_topk(bbh_handle* h, const double* scores_dev, int64_t N, int64_t k, double* vals_host,
             int64_t* idx_host);

/* The joint q'-batch and qLogNEHVI kernels split the MC samples into slices when a candidate set alone would not fill the chip; the
 * slice count - and with it the order in which a candidate's partial sums are added - follows the number of candidate rows.
 * rows > 0 fixes the row count the heuristic sees (a row shard passes the GLOBAL count: every rank then adds in the order the
 * unsharded pass uses, so scores are bit-identical across shard layouts); 0 = the rows of each call (default). */
int bbh_set_slice_rows(bbh_handle* h, int64_t rows);

/* Release what an idle handle holds above keep_bytes per buffer: the scratch workspace (bbh_qlogei_pending_big, the materialised
 * K* path), the global kernel-value cache, and the model's matrices (6 np^2 doubles; the next bbh_set_model re-creates them).  No
 * reference counterpart (BayBE holds no device state); called by the host side when a handle goes back to its pool. */
int bbh_trim(bbh_handle* h, int64_t keep_bytes);

/* ---- row-sharded selection over the GPUs of one node (RCCL over xGMI) ---------------------------
 * No reference call site (BayBE is single-process): these belong to the argmax / top-k of
 * optimize_acqf_discrete (baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126) once the candidate rows are
 * sharded contiguously over the ranks.  One collective per selection step; ties resolve to the lowest GLOBAL row index,
 * which follows shard order, i.e. exactly as on a single device.  RCCL is bound at run time (dlopen): without a
 * communicator nothing here is needed. */
/* rank 0: a fresh ncclUniqueId into id_out (bytes >= 128); returns its size, < 0 on error.  Ship it to the other
 * ranks by any means (file, MPI, torch.distributed broadcast). */
int bbh_comm_unique_id(void* id_out, int64_t bytes);
int bbh_comm_init(bbh_handle* h, int32_t rank, int32_t world, const void* unique_id, int64_t bytes);
int bbh_comm_destroy(bbh_handle* h);
/* Global top-k (k <= 64) of the scores of all shards: this rank holds scores_dev [N] for the global rows
 * [row_offset, row_offset + N).  Payload k x (score, global index) built on the device, one ncclAllGather, one
 * read-back; every rank returns the same vals_host / idx_host [k] (descending; (-inf, -1) beyond the total count). */
int bbh_allgather_topk(bbh_handle* h, const double* scores_dev, int64_t N, int64_t row_offset, int64_t k,
                       double* vals_host, int64_t* idx_host);
/* One greedy step: the global first-index argmax and the winner's comp-rep row (row_host [d]; every rank appends it
 * to its pending points).  X_dev [N, ldx] are this rank's rows; an empty shard (N = 0) still takes part. */
int bbh_allgather_argmax(bbh_handle* h, const double* scores_dev, int64_t N, int64_t row_offset, const double* X_dev,
                         int64_t ldx, double* val_host, int64_t* gidx_host, double* row_host);

/* ---- instrumentation --------------------------------------------------------------- */
/* Duration (ms) and launch count of the fused posterior kernel accumulated since the
 * last reset, measured with HIP events on the handle's stream when enabled.
 * enable: 0 = off, 1 = every kernel family, 2 * m = only the families whose bit is set in m (1 << BBH_TIMED_POSTERIOR ...): an event
 * between two back-to-back kernels costs the stream ~5 us, so a timed region brackets only the kernel it reports. */
int bbh_timing_enable(bbh_handle* h, int enable);
int bbh_timing_read(bbh_handle* h, double* fused_ms_total, int64_t* fused_launches, int reset);
/* The same per kernel family: BBH_TIMED_POSTERIOR = variance passes of the fused posterior kernel (what
 * bbh_timing_read reports), BBH_TIMED_CROSS = its mean-only passes (bbh_cross_cov),
 * BBH_TIMED_PENDING = the joint q'-batch acquisition kernels (bbh_qlogei_pending, bbh_mc_acq_pending). */
/* Form of the fused posterior kernel the last variance pass of this handle ran as: 0 = windowed (one wave per 16
 * candidates, bbh_fused_posterior_kernel), 1 = cooperative (one workgroup per 16 candidates, bbh_coop_posterior_kernel),
 * 2 = materialised K* (verification path; models outside the fused forms), 3 = two-sweep cooperative (512 < n <= 1024,
 * bbh_coop2_posterior_kernel), 4 = cooperative with the generic kernel-value production (composite, RQ, piecewise, Linear,
 * Polynomial, Periodic: bbh_coopg_posterior_kernel), 5 = register- / LDS-resident (n <= 128, bbh_small_posterior_kernel),
 * 6 = feature space (BBH_KERNEL_RFF: bbh_rff_posterior_kernel), -1 = none yet. */
int bbh_last_posterior_form(bbh_handle* h);
enum bbh_timed_family {
  BBH_TIMED_POSTERIOR = 0, BBH_TIMED_CROSS = 1, BBH_TIMED_PENDING = 2,
  BBH_TIMED_COLUMNS = 3,  /* bbh_posterior_columns: conditional means under S target columns (qLogNEHVI) */
  BBH_TIMED_NEHVI = 4,    /* bbh_qlognehvi: the scoring kernel over the cells of the box decompositions */
  BBH_TIMED_Q1 = 5,       /* q' = 1 acquisition kernels (bbh_qlogei_q1, bbh_mc_acq_q1) */
  BBH_TIMED_SELECT = 6,   /* chunk keys + the selection kernel (bbh_topk, bbh_argmax, bbh_qlogei_q1_topk) */
  BBH_TIMED_FAMILIES = 7
};
int bbh_timing_read_family(bbh_handle* h, int32_t family, double* ms_total, int64_t* launches, int reset);

/* Diagnostics of the dataflow fit evaluation (declared because the library exports them; scripts/gpu_flow_trace.py,
 * scripts/gpu_tile_stamps.py).  With BBH_FLOW_TRACE=1 / BBH_TILE_STAMPS=1 in the environment the last launch leaves device clock
 * stamps behind: bbh_flow_trace_read copies the per-role stamps [nroles][8] and the role table and returns nroles;
 * bbh_tiles_trace_read the row heads' stamps [tiles][8] of the tile-dataflow factorisation and returns the tile count.  -1 when no
 * trace exists or cap is too small.  No reference counterpart (baybe has no native code). */
int bbh_flow_trace_read(bbh_handle* h, long long* stamps_host, int* roles_host, int cap);
int bbh_tiles_trace_read(bbh_handle* h, long long* stamps_host, int cap);

#ifdef __cplusplus
}
#endif
#endif /* BAYBE_HIP_H */
