"""CPU oracle: numpy/scipy fp64 restatement of BayBE's GP recommend() hot path.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  PARITY UNPINNED (no botorch /
gpytorch importable; no golden vectors in the reference's tests).  Partial independent pin: the GP
algebra (Matérn / RBF ARD kernels with outputscale, exact posterior mean, latent variance and joint
covariance, log marginal likelihood) agrees with scikit-learn's GaussianProcessRegressor to 1e-9
(``tests/test_oracle_cpu.py::test_gp_algebra_is_pinned_against_scikit_learn``); the BoTorch-specific
pieces (qLogEI smoothing constants, sampler seeding, NEHVI) remain restated from memory.

What is restated, and the reference call site it follows
---------------------------------------------------------
* model assembly (Normalize over the numerical columns with the search-space
  bounds, Standardize(m=1), ConstantMean, Matérn-5/2 ARD *without* outputscale,
  Gamma(3, 2/e^{sqrt2-3}/sqrt(d)) lengthscale prior with box constraint
  l >= 0.025, Gamma(2, e^5) noise prior with box constraint s2 >= 1e-4, MLL for one
  task / LOO pseudo-likelihood for several):
  ``baybe/surrogates/gaussian_process/core.py:272-341``,
  ``baybe/surrogates/gaussian_process/presets/baybe.py:56-144,269-281``.
* ICM task kernel  K((x,t),(x',t')) = k(x,x') * B[t,t'],  B = W W^T + diag(v):
  ``presets/baybe.py:203-230``, ``components/kernel.py:298-337``,
  ``baybe/kernels/basic.py:239-248``.
* fit = scipy L-BFGS-B on -(objective/n) over the raw parameters with box bounds
  for the un-transformed constraints (botorch.fit.fit_gpytorch_mll, call site
  ``gaussian_process/core.py:340-341``).
* exact Cholesky posterior, no observation noise, q-batch joint covariance
  (botorch Model.posterior via ``surrogates/base.py:249-272``).
* best_f = max_i objective(posterior mean at training x_i):
  ``baybe/acquisition/_builder.py:141-161,256-265``.
* qLogEI (fat=True, tau_relu=1e-6, tau_max=1e-2) with Sobol-normal base samples
  shared by all candidates; minimisation = factor -1 on the samples:
  ``baybe/acquisition/acqfs.py:219-223``, ``baybe/objectives/base.py:99-150``,
  ``baybe/objectives/single.py:65-74``.
* sequential-greedy optimize_acqf_discrete (chunks of 2048, first-index argmax,
  unique=True, pending points appended after the candidate):
  ``baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126``.

[UPSTREAM] marks semantics recalled from botorch 0.16 / gpytorch 1.14 sources.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
from scipy import linalg as sla
from scipy import optimize as sopt

SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)
TAU_RELU = 1e-6  # [UPSTREAM] botorch.acquisition.logei.TAU_RELU
TAU_MAX = 1e-2  # [UPSTREAM] botorch.acquisition.logei.TAU_MAX
FAT_ALPHA_PLUS = 0.1  # [UPSTREAM] botorch.utils.safe_math.fatplus alpha
FAT_ALPHA_MAX = 2.0  # [UPSTREAM] botorch.utils.safe_math.fatmax alpha
MIN_INFERRED_NOISE_LEVEL = 0.0001  # [UPSTREAM] botorch.models.utils.gpytorch_modules.MIN_INFERRED_NOISE_LEVEL
MAX_BATCH_SIZE = 2048  # [UPSTREAM] optimize_acqf_discrete(max_batch_size=2048)

KERNELS = ("matern12", "matern32", "matern52", "rbf", "piecewise0", "piecewise1", "piecewise2", "piecewise3", "rq",
           "linear", "poly1", "poly2", "poly3", "poly4", "periodic")
# "periodic" (baybe/kernels/basic.py:73-112 -> gpytorch PeriodicKernel [UPSTREAM]): exp(-2 sum_j sin^2(pi |x_j - x'_j| / p_j) / l_j)
# with a lengthscale AND a period length per active column (``KernelTerm.period`` / ``GPSpec.period``; ``GPParams.period``).
# Dot-product kernels (baybe/kernels/basic.py:20-46, 135-163 -> gpytorch LinearKernel / PolynomialKernel [UPSTREAM]):
#   "linear": k = (x sqrt(v)) . (x' sqrt(v)) with one variance v_j per active column (``ard_num_dims`` is always passed,
#             kernels/base.py:218-239); the term's ``lengthscale`` Hyper / parameter array IS that variance here;
#   "polyP":  k = (x . x' + offset)^P; no per-column parameter, the offset (``KernelTerm.offset`` / ``GPSpec.offset``) is kept
#             where an RQ kernel keeps its alpha (``GPParams.rq_alpha``).
DOT_KERNELS = ("linear", "poly1", "poly2", "poly3", "poly4")
SCALAR_KERNELS = ("rq", "poly1", "poly2", "poly3", "poly4")  # kernels with one extra Positive() scalar (alpha / offset)


# --------------------------------------------------------------------------------------
# model specification (what BayBE's factories decide) and parameters (what the fit finds)
# --------------------------------------------------------------------------------------
@dataclass
class Hyper:
    """One positive hyper-parameter as gpytorch sees it: constraint, prior, start value.

    ``transformed=False`` is ``GreaterThan(lower, transform=None)`` (BayBE / BoTorch presets: the raw parameter
    IS the value and ``lower`` becomes an optimiser bound); ``transformed=True`` is ``value = lower +
    softplus(raw)`` (gpytorch's ``Positive()`` with ``lower = 0``, its default ``GreaterThan(1e-4)`` noise
    constraint).  ``prior`` is ``("gamma", concentration, rate)`` / ``("lognormal", loc, scale)`` / ``None``;
    ``init=None`` means "raw parameter 0" (gpytorch's default initialisation)."""

    lower: float = 0.0
    transformed: bool = True
    prior: tuple | None = None
    init: float | None = None

    def start(self) -> float:
        if self.init is not None:
            return float(self.init)
        if not self.transformed:
            raise ValueError("an un-transformed constraint needs an explicit start value")
        return self.lower + math.log(2.0)  # softplus(0)


@dataclass
class KernelTerm:
    """One base kernel of a user ``ProductKernel`` / ``AdditiveKernel`` (baybe/kernels/composite.py:60-91): a stationary
    ARD kernel over all numerical columns; ``outputscale`` is set when the term sits in its own ``ScaleKernel``."""

    kernel: str = "matern52"
    lengthscale: Hyper = field(default_factory=Hyper)
    outputscale: "Hyper | None" = None
    active_dims: "np.ndarray | None" = None  # positions among the numerical columns the kernel acts on (gpytorch active_dims
    # from ``BasicKernel.parameter_names``, baybe/kernels/base.py:198-240); its lengthscale has len(active_dims) entries
    offset: "Hyper | None" = None  # polynomial kernels: the offset (PolynomialKernel.offset_prior / offset_initial_value)
    period: "Hyper | None" = None  # periodic kernels: the period lengths (period_length_prior / period_length_initial_value)


@dataclass
class GPSpec:
    """Architecture + priors + constraints of one single-output GP (``baybe_default`` = presets/baybe.py)."""

    d: int  # comp-rep columns (incl. the task column if any)
    num_idx: np.ndarray  # numerical columns = kernel active dims = Normalize indices
    lo: np.ndarray  # scaling bounds of the numerical columns (searchspace.scaling_bounds)
    hi: np.ndarray
    kernel: str = "matern52"
    task_idx: int | None = None
    n_tasks: int = 1
    use_outputscale: bool = bool(0)  # ScaleKernel wrapper
    lengthscale: Hyper = field(default_factory=Hyper)
    noise: Hyper = field(default_factory=lambda: Hyper(lower=MIN_INFERRED_NOISE_LEVEL))
    outputscale: Hyper = field(default_factory=Hyper)
    criterion: str = "mll"  # "mll" (ExactMarginalLogLikelihood) | "loo" (LeaveOneOutPseudoLikelihood)
    # multi-task HVARFNER / BOTORCH presets (presets/hvarfner.py:72-137, presets/botorch.py:80-92):
    task_model: str = "shared"  # "per_task": HadamardGaussianLikelihood + HadamardConstantMean (one noise, one mean per task)
    index_kernel_scaling: str = "none"  # "target": botorch PositiveIndexKernel default, covariance / its [0, 0] entry
    task_rank: "int | None" = None  # columns of the index kernel's covariance factor (None: n_tasks)
    task_factor_transformed: bool = True  # False: gpytorch IndexKernel (free covar_factor) instead of botorch's PositiveIndexKernel
    correlation_prior: tuple | None = None  # ("beta", 2.5, 1.5): BetaPrior on the lower-triangle task correlations
    members: "list[KernelTerm] | None" = None  # base kernels of a ProductKernel / AdditiveKernel (replaces `kernel`)
    composition: str = "product"  # "product" | "sum" | "nested": an AdditiveKernel whose members are ProductKernels or base kernels
    member_terms: "list[int] | None" = None  # "nested": the summand each member is a factor of (baybe/kernels/composite.py:60-91 nested)
    active_dims: "np.ndarray | None" = None  # single kernel on a parameter subset (see KernelTerm.active_dims)
    offset: "Hyper | None" = None  # single polynomial kernel: its offset (see KernelTerm.offset)
    period: "Hyper | None" = None  # single periodic kernel: its period lengths (see KernelTerm.period)
    frequencies: "np.ndarray | None" = None  # kernel "rff": gpytorch RFFKernel.randn_weights [active columns, num_samples]

    def period_of(self, m: int | None = None) -> Hyper:
        h = self.period if (m is None or not self.members) else self.members[m].period
        return h if h is not None else Hyper()

    def offset_of(self, m: int | None = None) -> Hyper:
        h = self.offset if (m is None or not self.members) else self.members[m].offset
        return h if h is not None else Hyper()

    def dims_of(self, m: int | None = None) -> np.ndarray:
        """Positions (among the numerical columns) base kernel ``m`` - or the single kernel - acts on."""
        a = self.active_dims if (m is None or not self.members) else self.members[m].active_dims
        return np.arange(len(self.num_idx)) if a is None else np.asarray(a, dtype=np.int64)

    @property
    def dn(self) -> int:
        return len(self.num_idx)

    # flat views used by oracle/fit_objective.py
    ls_constraint = property(lambda self: "softplus" if self.lengthscale.transformed else "box")
    ls_lower = property(lambda self: self.lengthscale.lower)
    ls_prior = property(lambda self: self.lengthscale.prior)
    noise_constraint = property(lambda self: "softplus" if self.noise.transformed else "box")
    noise_lower = property(lambda self: self.noise.lower)
    noise_prior = property(lambda self: self.noise.prior)
    outputscale_prior = property(lambda self: self.outputscale.prior)

    @classmethod
    def baybe_default(cls, d, lo, hi, task_idx=None, n_tasks=1, kernel="matern52"):
        """BayBEKernelFactory / BayBELikelihoodFactory / criterion switch, presets/baybe.py:95-144, 269-281:
        ``lengthscale_prior = GammaPrior(3, 2 / x / sqrt(d))`` with ``x = exp(sqrt(2) - 3)`` (mode ``x sqrt(d)`` =
        the start value), ``GreaterThan(2.5e-2, transform=None)``; ``noise_prior = GammaPrior(2, 1 / exp(-5))``
        (mode ``exp(-5)`` = the start value), ``GreaterThan(1e-4, transform=None)``."""
        numerical = [j for j in range(int(d)) if j != task_idx]
        x = math.exp(math.sqrt(2.0) - 3.0)
        sd = math.sqrt(len(numerical))
        lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        return cls(
            d=int(d),
            num_idx=np.array(numerical, dtype=np.int64),
            lo=lo[numerical] if lo.size == d else lo,
            hi=hi[numerical] if hi.size == d else hi,
            kernel=kernel,
            task_idx=task_idx,
            n_tasks=int(n_tasks),
            lengthscale=Hyper(2.5e-2, False, ("gamma", 3.0, 2.0 / x / sd), x * sd),
            noise=Hyper(MIN_INFERRED_NOISE_LEVEL, False, ("gamma", 2.0, math.exp(5.0)), math.exp(-5.0)),
            criterion="loo" if n_tasks > 1 else "mll",
        )


@dataclass
class GPParams:
    """Natural (constrained) hyper-parameters of the standardised / normalised GP."""

    lengthscale: np.ndarray  # [dn]
    noise: "float | np.ndarray"  # sigma^2 ([T] for task_model == "per_task")
    mean: "float | np.ndarray" = 0.0  # ConstantMean ([T] for task_model == "per_task")
    outputscale: float = 1.0
    task_W: "np.ndarray | None" = None  # PositiveIndexKernel covar_factor [T, rank = T]
    task_v: "np.ndarray | None" = None  # PositiveIndexKernel var [T]
    target_scaled: bool = False  # index_kernel_scaling == "target"
    member_ls: "list[np.ndarray] | None" = None  # per base kernel of a composite: lengthscales [dn] (ALL members)
    member_scale: "np.ndarray | None" = None  # per base kernel: its own outputscale (1 where it has none)
    rq_alpha: "np.ndarray | None" = None  # RQ kernels: alpha per base kernel ([1] for a single kernel; 1 for other kinds)
    period: "list | None" = None  # periodic kernels: period lengths per base kernel (None entries for the other kinds)

    def task_B(self) -> np.ndarray | None:
        if self.task_W is None:
            return None
        full = self.task_W @ self.task_W.T + np.diag(self.task_v)
        return full / full[0, 0] if self.target_scaled else full

    def noise_of(self, tasks: np.ndarray) -> np.ndarray:
        """Noise variance of every row given its task index (the scalar for single-noise models)."""
        return np.broadcast_to(self.noise, tasks.shape) if np.ndim(self.noise) == 0 else np.asarray(self.noise)[tasks]

    def mean_of(self, tasks: np.ndarray) -> np.ndarray:
        return np.broadcast_to(self.mean, tasks.shape) if np.ndim(self.mean) == 0 else np.asarray(self.mean)[tasks]

    def copy(self) -> "GPParams":
        dup = lambda a: None if a is None else np.array(a, dtype=np.float64, copy=True)  # noqa: E731
        scal = lambda a: float(a) if np.ndim(a) == 0 else dup(a)  # noqa: E731
        return GPParams(dup(self.lengthscale), scal(self.noise), scal(self.mean), float(self.outputscale),
                        dup(self.task_W), dup(self.task_v), self.target_scaled,
                        None if self.member_ls is None else [dup(a) for a in self.member_ls], dup(self.member_scale),
                        dup(self.rq_alpha), None if self.period is None else [dup(a) for a in self.period])


def softplus(x):
    return np.logaddexp(0.0, np.asarray(x, dtype=np.float64))


def kernel_names(spec) -> list:
    """Names of the base kernels of the model: the members of a composite, or the single kernel."""
    return [t.kernel for t in spec.members] if spec.members else [spec.kernel]


def initial_params(spec, task_init=1.0):
    """Start of the fit: the values the presets set explicitly (prior modes for the BAYBE preset,
    presets/baybe.py:100-105, 134-142), raw = 0 for everything else [UPSTREAM].

    Task factors: gpytorch draws raw W / v randomly; the oracle (and the HIP path) start deterministically at
    W = task_init / sqrt(T), v = softplus(0) — documented deviation, unpinned."""
    T = int(spec.n_tasks)
    per_task = spec.task_model == "per_task"
    return GPParams(
        lengthscale=np.full(len(spec.dims_of(0 if spec.members else None)), spec.lengthscale.start()),
        noise=np.full(T, spec.noise.start()) if per_task else spec.noise.start(),
        mean=np.zeros(T) if per_task else 0.0,
        outputscale=spec.outputscale.start() if spec.use_outputscale else 1.0,
        task_W=np.full((T, int(spec.task_rank or T)), task_init / math.sqrt(T)) if T > 1 else None,
        task_v=np.full(T, math.log(2.0)) if T > 1 else None,
        target_scaled=spec.index_kernel_scaling == "target",
        member_ls=[np.full(len(spec.dims_of(m)), t.lengthscale.start()) for m, t in enumerate(spec.members)] if spec.members else None,
        member_scale=np.array([1.0 if t.outputscale is None else t.outputscale.start() for t in spec.members])
        if spec.members else None,
        rq_alpha=np.array([math.log(2.0) if k == "rq" else (spec.offset_of(m).start() if k in SCALAR_KERNELS else 1.0)
                           for m, k in enumerate(kernel_names(spec))]) if any(k in SCALAR_KERNELS for k in kernel_names(spec)) else None,
        period=[np.full(len(spec.dims_of(m if spec.members else None)), spec.period_of(m).start()) if k == "periodic" else None
                for m, k in enumerate(kernel_names(spec))] if "periodic" in kernel_names(spec) else None,
    )


# --------------------------------------------------------------------------------------
# transforms
# --------------------------------------------------------------------------------------
def normalize_inputs(spec: GPSpec, X: np.ndarray) -> np.ndarray:
    """botorch Normalize(d, bounds, indices) [UPSTREAM A1]; task column untouched."""
    Xn = np.array(X, dtype=np.float64, copy=True)
    Xn[:, spec.num_idx] = (Xn[:, spec.num_idx] - spec.lo) / (spec.hi - spec.lo)
    return Xn


def standardize_targets(y: np.ndarray) -> tuple[np.ndarray, float, float]:
    """botorch Standardize(m=1) [UPSTREAM A2]: Bessel std, std<1e-8 -> 1."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    ybar = float(y.mean())
    s = float(y.std(ddof=1)) if y.size > 1 else float("nan")
    if not (s >= 1e-8):
        s = 1.0
    return (y - ybar) / s, ybar, s


# --------------------------------------------------------------------------------------
# kernels
# --------------------------------------------------------------------------------------
def _scaled_sqdist(XA: np.ndarray, XB: np.ndarray, ls: np.ndarray) -> np.ndarray:
    A = XA / ls
    B = XB / ls
    # direct differences (most accurate form); r^2 >= 0 by construction
    d2 = np.zeros((A.shape[0], B.shape[0]), dtype=np.float64)
    for j in range(A.shape[1]):
        diff = A[:, j : j + 1] - B[None, :, j]
        d2 += diff * diff
    return d2


def _metric(kernel: str, XA: np.ndarray, XB: np.ndarray, ls: np.ndarray, period=None) -> np.ndarray:
    """What the base kernel is a function of: the scaled squared distance (stationary kernels), x sqrt(v) . x' sqrt(v) (gpytorch
    LinearKernel.forward, ``ls`` = the variances), x . x' (PolynomialKernel.forward) or sum_j sin^2(pi (x_j - x'_j) / p_j) / l_j
    (PeriodicKernel.forward: ``diff.sin().pow(2).div(lengthscale)`` summed over the dimensions)."""
    if kernel == "periodic":
        out = np.zeros((XA.shape[0], XB.shape[0]))
        for j in range(XA.shape[1]):
            out += np.sin(math.pi * (XA[:, j : j + 1] - XB[None, :, j]) / period[j]) ** 2 / ls[j]
        return out
    if kernel == "linear":
        return (XA * np.sqrt(ls)) @ (XB * np.sqrt(ls)).T
    if kernel in DOT_KERNELS:
        return XA @ XB.T
    return _scaled_sqdist(XA, XB, ls)


def _self_metric(kernel: str, XA: np.ndarray, ls: np.ndarray) -> np.ndarray:
    """The metric of every row with itself (0 for the stationary kernels)."""
    if kernel == "linear":
        return (XA * XA * ls).sum(axis=1)
    if kernel in DOT_KERNELS:
        return (XA * XA).sum(axis=1)
    return np.zeros(XA.shape[0])


def _piecewise_terms(q: int, dims: int, r: np.ndarray):
    """gpytorch PiecewisePolynomialKernel [UPSTREAM, piecewise_polynomial_kernel.py]: exponent j = floor(D / 2) + q + 1 and
    the polynomial ``get_cov(r, j, q)`` with its derivative."""
    j = math.floor(dims / 2.0) + q + 1
    if q == 0:
        return j, np.ones_like(r), np.zeros_like(r)
    if q == 1:
        return j, (j + 1) * r + 1, np.full_like(r, j + 1.0)
    if q == 2:
        c2 = (j**2 + 4 * j + 3) / 3.0
        return j, 1 + (j + 2) * r + c2 * r**2, (j + 2) + 2 * c2 * r
    c2, c3 = (6 * j**2 + 36 * j + 45) / 15.0, (j**3 + 9 * j**2 + 23 * j + 15) / 15.0
    return j, 1 + (j + 3) * r + c2 * r**2 + c3 * r**3, (j + 3) + 2 * c2 * r + 3 * c3 * r**2


def base_kernel_from_r2(kernel: str, r2: np.ndarray, dims: int | None = None, alpha: float | None = None) -> np.ndarray:
    """Stationary kernels of gpytorch [UPSTREAM A3] as functions of r^2 (``dims``: input dimension, piecewise family;
    ``alpha``: RQ kernel, ``(1 + r^2 / (2 alpha))^-alpha``)."""
    if kernel == "periodic":  # ``r2`` is the sum of ``_metric`` here: exp_term.mul(-2).exp()
        return np.exp(-2.0 * r2)
    if kernel == "linear":  # ``r2`` is the dot product of ``_metric`` here
        return r2
    if kernel in DOT_KERNELS:  # gpytorch PolynomialKernel: (x1 @ x2^T + offset).pow(power)
        return (r2 + alpha) ** int(kernel[-1])
    if kernel == "rq":
        return (1.0 + r2 / (2.0 * alpha)) ** (-alpha)
    if kernel.startswith("piecewise"):
        q = int(kernel[-1])
        r = np.sqrt(np.maximum(r2, 1e-30))
        j, poly, _ = _piecewise_terms(q, dims, r)
        return np.maximum(0.0, 1.0 - r) ** (j + q) * poly
    if kernel == "rbf":
        return np.exp(-0.5 * r2)
    r = np.sqrt(r2)
    if kernel == "matern52":
        return (1.0 + SQRT5 * r + (5.0 / 3.0) * r2) * np.exp(-SQRT5 * r)
    if kernel == "matern32":
        return (1.0 + SQRT3 * r) * np.exp(-SQRT3 * r)
    if kernel == "matern12":
        return np.exp(-r)
    raise ValueError(kernel)


def rq_alpha_derivative(r2: np.ndarray, alpha: float) -> np.ndarray:
    """d/dalpha (1 + u)^-alpha with u = r^2 / (2 alpha):  k (u / (1 + u) - log(1 + u))."""
    u = r2 / (2.0 * alpha)
    return (1.0 + u) ** (-alpha) * (u / (1.0 + u) - np.log1p(u))


def base_kernel_gfac_from_r2(kernel: str, r2: np.ndarray, dims: int | None = None, alpha: float | None = None) -> np.ndarray:
    """g(r) = -(dk/dr)/r, so that dk/dl_j = g(r) * Delta_j^2 / l_j^3."""
    if kernel == "rq":  # dk/dr = -r (1 + u)^-(alpha + 1)
        return (1.0 + r2 / (2.0 * alpha)) ** (-alpha - 1.0)
    if kernel.startswith("piecewise"):  # product rule on max(0, 1 - r)^(j + q) * poly(r); plain quotient by r
        q = int(kernel[-1])
        r = np.sqrt(np.maximum(r2, 1e-30))
        j, poly, dpoly = _piecewise_terms(q, dims, r)
        base = np.maximum(0.0, 1.0 - r)
        dk = -(j + q) * base ** (j + q - 1) * poly + base ** (j + q) * dpoly
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(r2 > 1e-30, -dk / r, 0.0 if q == 0 else -_second_derivative_at_zero(q, j))
    if kernel == "rbf":
        return np.exp(-0.5 * r2)
    r = np.sqrt(r2)
    if kernel == "matern52":
        return (5.0 / 3.0) * (1.0 + SQRT5 * r) * np.exp(-SQRT5 * r)
    if kernel == "matern32":
        return 3.0 * np.exp(-SQRT3 * r)
    if kernel == "matern12":
        with np.errstate(divide="ignore", invalid="ignore"):
            g = np.where(r > 0, np.exp(-r) / r, 0.0)
        return g
    raise ValueError(kernel)


def _second_derivative_at_zero(q: int, j: int) -> float:
    """k''(0) of the piecewise-polynomial kernel (the limit of k'(r) / r for q >= 1, where k'(0) = 0)."""
    p = j + q
    c1 = {1: j + 1.0, 2: j + 2.0, 3: j + 3.0}[q]
    c2 = {1: 0.0, 2: (j**2 + 4 * j + 3) / 3.0, 3: (6 * j**2 + 36 * j + 45) / 15.0}[q]
    # k = (1 - r)^p (1 + c1 r + c2 r^2 + ...):  k'' (0) = p (p - 1) - 2 p c1 + 2 c2
    return p * (p - 1) - 2 * p * c1 + 2 * c2


def _alpha_of(p: GPParams, m: int):
    return None if p.rq_alpha is None else float(p.rq_alpha[m])


def _period_of(p: GPParams, m: int):
    return None if p.period is None else p.period[m]


def member_grams(spec: GPSpec, p: GPParams, A: np.ndarray, B: np.ndarray) -> list[np.ndarray]:
    """Scaled Gram matrix of every base kernel of a composite (numerical columns only)."""
    out = []
    for m, t in enumerate(spec.members):
        c = spec.dims_of(m)
        out.append(p.member_scale[m] * base_kernel_from_r2(t.kernel, _metric(t.kernel, A[:, c], B[:, c], p.member_ls[m], _period_of(p, m)), len(c), _alpha_of(p, m)))
    return out


def rff_features(W: np.ndarray, X: np.ndarray, ls: np.ndarray) -> np.ndarray:
    """gpytorch ``RFFKernel._featurize`` (baybe/kernels/basic.py:183-199 maps to it): ``x.matmul(randn_weights / lengthscale^T)``, then
    ``cat([cos, sin], -1)``; the kernel is ``z1 z2^T / D`` (RFFKernel.forward)."""
    P = X @ (W / np.asarray(ls, dtype=np.float64).reshape(-1, 1))
    return np.concatenate([np.cos(P), np.sin(P)], axis=1)


def stationary_part(spec: GPSpec, p: GPParams, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """The kernel over the numerical columns without outer outputscale / task factor: the single stationary kernel, or
    the elementwise product / sum of the members' Gram matrices (``reduce(mul | add, ...)``, composite.py:75,91)."""
    if not spec.members and spec.kernel == "rff":
        c = spec.dims_of(None)
        W = np.array(spec.frequencies, dtype=float)
        return rff_features(W, A[:, c], p.lengthscale) @ rff_features(W, B[:, c], p.lengthscale).T / W.shape[1]
    if not spec.members:
        c = spec.dims_of(None)
        return base_kernel_from_r2(spec.kernel, _metric(spec.kernel, A[:, c], B[:, c], p.lengthscale, _period_of(p, 0)), len(c), _alpha_of(p, 0))
    return compose_members(spec, member_grams(spec, p, A, B))


def compose_members(spec: GPSpec, grams: list):
    """``reduce(mul, ...)`` of a ProductKernel, ``reduce(add, ...)`` of an AdditiveKernel (composite.py:75,91), or - an AdditiveKernel with
    ProductKernel members - the sum over its members of the products of theirs."""
    if spec.composition == "product":
        out = grams[0]
        for Km in grams[1:]:
            out = out * Km
        return out + 0.0
    if spec.composition == "sum":
        return sum(grams[1:], grams[0] + 0.0)
    summands: dict = {}
    for m, Km in enumerate(grams):
        t = spec.member_terms[m]
        summands[t] = Km if t not in summands else summands[t] * Km
    return sum(summands.values())


def siblings_product(spec: GPSpec, grams: list, m: int):
    """d(composite) / d(member m's Gram matrix): the product of the other members of m's summand (1 for a plain sum)."""
    out = 1.0
    for k, Kk in enumerate(grams):
        same = spec.composition == "product" or (spec.composition == "nested" and spec.member_terms[k] == spec.member_terms[m])
        if k != m and same:
            out = out * Kk
    return out


def cross_cov(spec: GPSpec, p: GPParams, XAn: np.ndarray, XBn: np.ndarray) -> np.ndarray:
    """K(XA, XB) on *normalised* inputs, incl. outputscale and task factor."""
    K = stationary_part(spec, p, XAn[:, spec.num_idx], XBn[:, spec.num_idx])
    if spec.use_outputscale:
        K = K * p.outputscale
    if spec.task_idx is not None:
        B = p.task_B()
        ta = XAn[:, spec.task_idx].astype(np.int64)
        tb = XBn[:, spec.task_idx].astype(np.int64)
        K = K * B[np.ix_(ta, tb)]
    return K


def task_rows(spec: GPSpec, Xn: np.ndarray) -> np.ndarray:
    """Task index of every row (zeros without a task column)."""
    return np.zeros(Xn.shape[0], dtype=np.int64) if spec.task_idx is None else Xn[:, spec.task_idx].astype(np.int64)


def prior_var(spec: GPSpec, p: GPParams, Xn: np.ndarray) -> np.ndarray:
    """k(x,x) for each row (stationary kernels: 1 * outputscale * B[t,t]; dot-product kernels: a function of the row)."""
    v = np.full(Xn.shape[0], p.outputscale if spec.use_outputscale else 1.0)
    Xs = Xn[:, spec.num_idx]
    if spec.members:  # k_m(x, x) = 1 for every stationary member
        diags = []
        for m, t in enumerate(spec.members):
            c = spec.dims_of(m)
            diags.append(p.member_scale[m] * base_kernel_from_r2(t.kernel, _self_metric(t.kernel, Xs[:, c], p.member_ls[m]), len(c), _alpha_of(p, m)))
        v = v * compose_members(spec, diags)
    elif spec.kernel in DOT_KERNELS:
        c = spec.dims_of(None)
        v = v * base_kernel_from_r2(spec.kernel, _self_metric(spec.kernel, Xs[:, c], p.lengthscale), len(c), _alpha_of(p, 0))
    if spec.task_idx is not None:
        B = p.task_B()
        t = Xn[:, spec.task_idx].astype(np.int64)
        v = v * B[t, t]
    return v


# --------------------------------------------------------------------------------------
# fit objective: data term (this is what the device computes) + priors (host)
# --------------------------------------------------------------------------------------
@dataclass
class DataTerm:
    value: float  # log-likelihood (mll) or LOO sum incl. the -n/2 log(2pi) constant
    g_ls: np.ndarray
    g_noise: "float | np.ndarray"  # [T] for per-task noise
    g_mean: "float | np.ndarray"  # [T] for per-task means
    g_outputscale: float
    g_task_B: np.ndarray | None  # dL/dB[t,t'] (symmetric accumulation S)
    g_member_ls: "list[np.ndarray] | None" = None  # composite kernels: per base kernel
    g_member_scale: "np.ndarray | None" = None
    g_alpha: "np.ndarray | None" = None  # RQ kernels: one entry per base kernel (0 for the other kinds)


def data_term(spec: GPSpec, p: GPParams, Xn: np.ndarray, ystd: np.ndarray) -> DataTerm:
    """Value + analytic gradient of the data-fit term w.r.t. natural parameters.

    mll:  log N(y | c 1, K + s2 I)                      [UPSTREAM A5, gpytorch ExactMLL]
    loo:  sum_i log N(y_i | mu_-i, s2_-i)               [UPSTREAM A5, gpytorch LOO-PL]
    Gradient:  dL/dtheta = sum_ab G_ab dK_ab/dtheta  with
      mll: G = 0.5 (alpha alpha^T - M),  M = (K + s2 I)^-1
      loo: G = sym(-M diag(u) M + alpha (M w)^T),  u = 0.5/d + 0.5 alpha^2/d^2,
           w = alpha/d,  d = diag(M).
    """
    n = Xn.shape[0]
    if spec.active_dims is not None or any(t.active_dims is not None for t in (spec.members or [])):
        raise NotImplementedError("analytic data-term gradients are not restated for kernels on parameter subsets; "
                                  "use fit_objective (autograd)")
    if any(k in DOT_KERNELS or k in ("periodic", "rff") for k in kernel_names(spec)):
        raise NotImplementedError("analytic data-term gradients are not restated for Linear / Polynomial / Periodic / RFF kernels; use fit_objective")
    trow = task_rows(spec, Xn)
    Kf = cross_cov(spec, p, Xn, Xn)
    Ky = Kf + np.diag(p.noise_of(trow))
    L = sla.cholesky(Ky, lower=True)
    r = ystd - p.mean_of(trow)
    alpha = sla.cho_solve((L, True), r)
    Linv = sla.solve_triangular(L, np.eye(n), lower=True)
    M = Linv.T @ Linv
    if spec.criterion == "mll":
        value = -0.5 * float(r @ alpha) - float(np.log(np.diag(L)).sum()) - 0.5 * n * math.log(2 * math.pi)
        G = 0.5 * (np.outer(alpha, alpha) - M)
        g_mean_rows = alpha
    elif spec.criterion == "loo":
        dg = np.diag(M)
        value = float((0.5 * np.log(dg) - 0.5 * alpha**2 / dg).sum()) - 0.5 * n * math.log(2 * math.pi)
        u = 0.5 / dg + 0.5 * alpha**2 / dg**2
        w = alpha / dg
        Mw = M @ w
        G = -(M * u[None, :]) @ M + np.outer(alpha, Mw)
        G = 0.5 * (G + G.T)
        g_mean_rows = M @ w
    else:
        raise ValueError(spec.criterion)

    # kernel-parameter gradients through G
    Xs = Xn[:, spec.num_idx]
    if spec.members:
        return _composite_gradients(spec, p, Xn, Xs, G, Kf, value, g_mean_rows, trow)
    r2 = _scaled_sqdist(Xs, Xs, p.lengthscale)
    gfac = base_kernel_gfac_from_r2(spec.kernel, r2, spec.dn, _alpha_of(p, 0))
    scale = np.full((n, n), p.outputscale if spec.use_outputscale else 1.0)
    Bsel = None
    if spec.task_idx is not None:
        B = p.task_B()
        t = Xn[:, spec.task_idx].astype(np.int64)
        Bsel = B[np.ix_(t, t)]
        scale = scale * Bsel
    GW = G * gfac * scale
    g_ls = np.empty(spec.dn)
    for j in range(spec.dn):
        diff = Xs[:, j : j + 1] - Xs[None, :, j]
        g_ls[j] = float((GW * diff * diff).sum()) / p.lengthscale[j] ** 3
    if spec.task_model == "per_task":  # one slot per task: the rows of that task contribute
        g_noise = np.array([float(np.diag(G)[trow == k].sum()) for k in range(spec.n_tasks)])
        g_mean = np.array([float(g_mean_rows[trow == k].sum()) for k in range(spec.n_tasks)])
    else:
        g_noise, g_mean = float(np.trace(G)), float(g_mean_rows.sum())
    g_os = float((G * Kf).sum()) / p.outputscale if spec.use_outputscale else 0.0
    g_B = None
    if spec.task_idx is not None:
        kb = base_kernel_from_r2(spec.kernel, r2, spec.dn, _alpha_of(p, 0)) * (p.outputscale if spec.use_outputscale else 1.0)
        T = spec.n_tasks
        onehot = np.zeros((n, T))
        onehot[np.arange(n), t] = 1.0
        g_B = onehot.T @ (G * kb) @ onehot
    g_alpha = None
    if spec.kernel == "rq":
        g_alpha = np.array([float((G * scale * rq_alpha_derivative(r2, float(p.rq_alpha[0]))).sum())])
    return DataTerm(value, g_ls, g_noise, g_mean, g_os, g_B, g_alpha=g_alpha)


def _composite_gradients(spec, p, Xn, Xs, G, Kf, value, g_mean_rows, trow) -> DataTerm:
    """dL/d(member lengthscales, member scales, outer scale, task table, noise, mean) for a product / sum of members.
    With S = stationary part, K = os * S * B:  dK/dtheta_m = os * B * dS/dtheta_m, and
      product: dS/dtheta_m = (S / S_m) dS_m/dtheta_m = prod_{g != m} S_g * dS_m/dtheta_m;   sum: dS/dtheta_m = dS_m/dtheta_m."""
    n = Xn.shape[0]
    os = p.outputscale if spec.use_outputscale else 1.0
    Bsel = np.ones((n, n))
    if spec.task_idx is not None:
        Bsel = p.task_B()[np.ix_(trow, trow)]
    grams = member_grams(spec, p, Xs, Xs)
    g_ls, g_sc, g_al = [], np.zeros(len(spec.members)), np.zeros(len(spec.members))
    for m, t in enumerate(spec.members):
        others = np.ones((n, n)) * siblings_product(spec, grams, m)
        r2 = _scaled_sqdist(Xs, Xs, p.member_ls[m])
        front = G * os * Bsel * others * p.member_scale[m] * base_kernel_gfac_from_r2(t.kernel, r2, spec.dn, _alpha_of(p, m))
        gl = np.empty(spec.dn)
        for j in range(spec.dn):
            diff = Xs[:, j : j + 1] - Xs[None, :, j]
            gl[j] = float((front * diff * diff).sum()) / p.member_ls[m][j] ** 3
        g_ls.append(gl)
        g_sc[m] = float((G * os * Bsel * others * base_kernel_from_r2(t.kernel, r2, spec.dn, _alpha_of(p, m))).sum())
        if t.kernel == "rq":
            g_al[m] = float((G * os * Bsel * others * p.member_scale[m] * rq_alpha_derivative(r2, float(p.rq_alpha[m]))).sum())
    S = stationary_part(spec, p, Xs, Xs)
    g_os = float((G * S * Bsel).sum()) if spec.use_outputscale else 0.0
    g_B = None
    if spec.task_idx is not None:
        onehot = np.zeros((n, spec.n_tasks))
        onehot[np.arange(n), trow] = 1.0
        g_B = onehot.T @ (G * S * os) @ onehot
    if spec.task_model == "per_task":
        g_noise = np.array([float(np.diag(G)[trow == k].sum()) for k in range(spec.n_tasks)])
        g_mean = np.array([float(g_mean_rows[trow == k].sum()) for k in range(spec.n_tasks)])
    else:
        g_noise, g_mean = float(np.trace(G)), float(g_mean_rows.sum())
    return DataTerm(value, g_ls[0], g_noise, g_mean, g_os, g_B, g_ls, g_sc, g_al if p.rq_alpha is not None else None)


# ---- the optimiser's view: raw vector <-> natural parameters, objective, bounds -----------------------
# All of it is oracle/fit_objective.py (torch.distributions + autograd); these wrappers only translate between
# GPParams and the gpytorch-named parameter dictionary.
def _natural_dict(spec: GPSpec, p: GPParams) -> dict:
    nat = {"noise": np.atleast_1d(p.noise), "constant": p.mean, "lengthscale": p.lengthscale, "variance": p.lengthscale}
    scalar = lambda k: "alpha" if k == "rq" else "offset"  # noqa: E731  (gpytorch's leaf names: raw_alpha / raw_offset)
    if spec.members:
        for m, t in enumerate(spec.members):
            nat[f"lengthscale.{m}"] = nat[f"variance.{m}"] = p.member_ls[m]
            if t.outputscale is not None:
                nat[f"outputscale.{m}"] = p.member_scale[m]
            if t.kernel in SCALAR_KERNELS:
                nat[f"{scalar(t.kernel)}.{m}"] = [p.rq_alpha[m]]
            if t.kernel == "periodic":
                nat[f"period_length.{m}"] = p.period[m]
    elif spec.kernel in SCALAR_KERNELS:
        nat[scalar(spec.kernel)] = [p.rq_alpha[0]]
    elif spec.kernel == "periodic":
        nat["period_length"] = p.period[0]
    if spec.use_outputscale:
        nat["outputscale"] = p.outputscale
    if spec.n_tasks > 1:
        nat["covar_factor"], nat["var"] = p.task_W, p.task_v
    return nat


def pack_raw(spec, p):
    from oracle import fit_objective as fo

    return fo.natural_to_raw(spec, _natural_dict(spec, p))


def unpack_raw(spec, raw):
    import torch

    from oracle import fit_objective as fo

    nat = {k: v.detach().numpy() for k, v in fo.split_raw(spec, torch.as_tensor(np.asarray(raw, dtype=np.float64))).items()}
    def percol(m):  # the per-column parameter of base kernel m: lengthscales / Linear variances / none (Polynomial: ones)
        kind = kernel_names(spec)[0 if m is None else m]
        suffix = "" if m is None else f".{m}"
        if kind == "linear":
            return nat["variance" + suffix].reshape(-1).copy()
        if kind in DOT_KERNELS:
            return np.ones(len(spec.dims_of(m)))
        return nat["lengthscale" + suffix].reshape(-1).copy()

    mls = [percol(m) for m in range(len(spec.members))] if spec.members else None
    msc = np.array([float(nat[f"outputscale.{m}"]) if t.outputscale is not None else 1.0
                    for m, t in enumerate(spec.members)]) if spec.members else None
    names = kernel_names(spec)
    if any(k in SCALAR_KERNELS for k in names):
        leaf = lambda k: "alpha" if k == "rq" else "offset"  # noqa: E731
        key = (lambda m, k: f"{leaf(k)}.{m}") if spec.members else (lambda m, k: leaf(k))
        alphas = np.array([float(nat[key(m, k)].reshape(-1)[0]) if k in SCALAR_KERNELS else 1.0 for m, k in enumerate(names)])
    else:
        alphas = None
    periods = None
    if "periodic" in names:
        pkey = (lambda m: f"period_length.{m}") if spec.members else (lambda m: "period_length")
        periods = [nat[pkey(m)].reshape(-1).copy() if k == "periodic" else None for m, k in enumerate(names)]
    return GPParams(
        period=periods,
        rq_alpha=alphas,
        lengthscale=mls[0] if spec.members else percol(None),
        member_ls=mls,
        member_scale=msc,
        noise=nat["noise"].reshape(-1).copy() if spec.task_model == "per_task" else float(nat["noise"].reshape(-1)[0]),
        mean=nat["constant"].reshape(-1).copy() if spec.task_model == "per_task" else float(nat["constant"]),
        outputscale=float(nat["outputscale"]) if spec.use_outputscale else 1.0,
        task_W=nat["covar_factor"].copy() if spec.n_tasks > 1 else None,
        task_v=nat["var"].copy() if spec.n_tasks > 1 else None,
        target_scaled=spec.index_kernel_scaling == "target",
    )


def raw_bounds(spec):
    from oracle import fit_objective as fo

    return fo.optimiser_bounds(spec)


def fit_objective(spec: GPSpec, raw: np.ndarray, Xn: np.ndarray, ystd: np.ndarray):
    """-(log-likelihood + log-priors)/n and its autograd gradient w.r.t. the raw vector (gpytorch:
    ``res = output.log_prob(target); res += sum(prior.log_prob); res / n``)."""
    from oracle import fit_objective as fo

    return fo.objective(spec, raw, Xn, ystd)


@dataclass
class FitResult:
    params: GPParams
    fun: float
    nit: int
    nfev: int
    status: int
    message: str


def fit_hyperparameters(spec: GPSpec, Xn: np.ndarray, ystd: np.ndarray, p0: GPParams | None = None,
                        maxiter: int = 15000) -> FitResult:
    """botorch.fit.fit_gpytorch_mll -> scipy L-BFGS-B with scipy defaults [UPSTREAM A6]."""

    def closure(raw):
        try:
            return fit_objective(spec, raw, Xn, ystd)
        except (RuntimeError, ValueError):  # torch: Cholesky of a non-PD matrix / invalid distribution argument
            return np.inf, np.zeros(len(raw))

    res = sopt.minimize(closure, pack_raw(spec, p0 or initial_params(spec)), jac=True, method="L-BFGS-B",
                        bounds=raw_bounds(spec), options={"maxiter": maxiter})
    return FitResult(unpack_raw(spec, res.x), float(res.fun), int(res.nit), int(res.nfev), int(res.status), str(res.message))


# --------------------------------------------------------------------------------------
# fitted model: exact Cholesky posterior
# --------------------------------------------------------------------------------------
@dataclass
class GPModel:
    spec: GPSpec
    params: GPParams
    X_train: np.ndarray  # raw comp-rep rows [n, d]
    y_train: np.ndarray  # raw targets [n]
    # derived
    Xn: np.ndarray = field(init=False)
    ystd: np.ndarray = field(init=False)
    ybar: float = field(init=False)
    ysd: float = field(init=False)
    L: np.ndarray = field(init=False)
    alpha: np.ndarray = field(init=False)
    jitter: float = field(init=False, default=0.0)

    def __post_init__(self):
        self.X_train = np.ascontiguousarray(self.X_train, dtype=np.float64)
        self.Xn = normalize_inputs(self.spec, self.X_train)
        self.ystd, self.ybar, self.ysd = standardize_targets(self.y_train)
        n = self.Xn.shape[0]
        trow = task_rows(self.spec, self.Xn)
        Ky = cross_cov(self.spec, self.params, self.Xn, self.Xn) + np.diag(self.params.noise_of(trow))
        # gpytorch psd_safe_cholesky: retry with jitter 1e-8 * 10^i [UPSTREAM A7]
        jit = 0.0
        for attempt in range(4):
            try:
                self.L = sla.cholesky(Ky + jit * np.eye(n), lower=True)
                break
            except sla.LinAlgError:
                jit = 1e-8 * 10**attempt
        else:
            raise sla.LinAlgError("train covariance not PD even with jitter")
        self.jitter = jit
        self.alpha = sla.cho_solve((self.L, True), self.ystd - self.params.mean_of(trow))

    # posterior of standardised GP at normalised inputs
    def _std_posterior(self, Xc: np.ndarray, joint: bool):
        Xcn = normalize_inputs(self.spec, np.atleast_2d(Xc))
        Ks = cross_cov(self.spec, self.params, Xcn, self.Xn)  # [N, n]
        mu = self.params.mean_of(task_rows(self.spec, Xcn)) + Ks @ self.alpha
        V = sla.solve_triangular(self.L, Ks.T, lower=True)  # [n, N]
        if joint:
            Kss = cross_cov(self.spec, self.params, Xcn, Xcn)
            return mu, Kss - V.T @ V
        return mu, prior_var(self.spec, self.params, Xcn) - (V * V).sum(axis=0)

    def posterior(self, Xc: np.ndarray, chunk: int = 8192) -> tuple[np.ndarray, np.ndarray]:
        """Marginal posterior mean/variance (original target scale), t-batch of q=1."""
        Xc = np.atleast_2d(np.asarray(Xc, dtype=np.float64))
        mus, vs = [], []
        for s in range(0, Xc.shape[0], chunk):
            mu, v = self._std_posterior(Xc[s : s + chunk], joint=False)
            mus.append(self.ybar + self.ysd * mu)
            vs.append(self.ysd**2 * v)
        return np.concatenate(mus), np.concatenate(vs)

    def posterior_joint(self, Xq: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        mu, cov = self._std_posterior(Xq, joint=True)
        return self.ybar + self.ysd * mu, self.ysd**2 * cov


def fit_gp(spec: GPSpec, X_train, y_train, params: GPParams | None = None, p0: GPParams | None = None) -> GPModel:
    """Fit (or, with ``params`` given, just factorise) a GP on raw data."""
    X_train = np.ascontiguousarray(X_train, dtype=np.float64)
    if params is None:
        ystd, _, _ = standardize_targets(y_train)
        params = fit_hyperparameters(spec, normalize_inputs(spec, X_train), ystd, p0).params
    return GPModel(spec, params, X_train, np.asarray(y_train, dtype=np.float64).reshape(-1))


# --------------------------------------------------------------------------------------
# MC acquisition: qLogEI
# --------------------------------------------------------------------------------------
def sobol_normal_base_samples(S, q, seed):
    """botorch SobolQMCNormalSampler base samples [UPSTREAM A8]: scrambled Sobol points u (torch's engine, the one
    BoTorch uses), pulled off the boundary, v = 0.5 + (1 - eps)(u - 0.5), and pushed through the standard normal
    quantile function (``Normal.icdf`` = sqrt(2) erfinv(2 v - 1)).  Returns [S, q] float64."""
    import torch

    engine = torch.quasirandom.SobolEngine(q, scramble=True, seed=int(seed))
    points = engine.draw(int(S), dtype=torch.float64)
    inner = 0.5 + (points - 0.5) * (1.0 - torch.finfo(torch.float64).eps)
    return torch.distributions.Normal(0.0, 1.0).icdf(inner).numpy().copy()


def draw_sampler_seed():
    """MCSampler seed when none is given: torch.randint(0, 1000000, (1,)) [UPSTREAM A8]."""
    import torch

    return torch.randint(0, 1000000, (1,)).item()


def log_fatplus(x: np.ndarray, tau: float = TAU_RELU) -> np.ndarray:
    """log(tau * (softplus(x/tau) + 0.1 / (1 + (x/tau)^2)))  [UPSTREAM A9, safe_math.fatplus]."""
    t = np.asarray(x, dtype=np.float64) / tau
    return math.log(tau) + np.log(softplus(t) + FAT_ALPHA_PLUS / (1.0 + t * t))


def fatmax(x: np.ndarray, tau: float = TAU_MAX, alpha: float = FAT_ALPHA_MAX, axis: int = -1) -> np.ndarray:
    """M + tau log sum_j (alpha / (alpha + (M - x_j)/tau))^alpha  [UPSTREAM A9, safe_math.fatmax]."""
    M = x.max(axis=axis, keepdims=True)
    s = ((alpha / (alpha + (M - x) / tau)) ** alpha).sum(axis=axis)
    return np.squeeze(M, axis=axis) + tau * np.log(s)


def logmeanexp(x: np.ndarray, axis: int = 0) -> np.ndarray:
    M = x.max(axis=axis, keepdims=True)
    return np.squeeze(M, axis=axis) + np.log(np.exp(x - M).mean(axis=axis))


def _safe_sqrt_var(v: np.ndarray) -> np.ndarray:
    """1x1 psd_safe_cholesky: v <= 0 -> add jitter 1e-8, 1e-7, 1e-6 [UPSTREAM A7]."""
    v = np.asarray(v, dtype=np.float64).copy()
    for attempt in range(3):
        bad = ~(v > 0)
        if not bad.any():
            break
        v = np.where(bad, v + 1e-8 * 10**attempt, v)
    return np.sqrt(np.maximum(v, 0.0))


def qlogei_q1(mu: np.ndarray, var: np.ndarray, z: np.ndarray, best_f: float, sign: float = 1.0) -> np.ndarray:
    """qLogEI of N independent q=1 candidates; z [S] shared base samples."""
    sd = _safe_sqrt_var(var)
    obj = sign * (mu[None, :] + sd[None, :] * z.reshape(-1, 1))  # [S, N]
    li = log_fatplus(obj - best_f)
    return logmeanexp(li, axis=0)


def _safe_cholesky(A: np.ndarray) -> np.ndarray:
    jit = 0.0
    for attempt in range(4):
        try:
            return sla.cholesky(A + jit * np.eye(A.shape[0]), lower=True)
        except sla.LinAlgError:
            jit = 1e-8 * 10**attempt
    raise sla.LinAlgError("q-batch covariance not PD")


def qlogei_joint(mean: np.ndarray, cov: np.ndarray, z: np.ndarray, best_f: float, sign: float = 1.0) -> float:
    """qLogEI of one q'-batch; z [S, q'] [UPSTREAM A9]."""
    Lq = _safe_cholesky(cov)
    samples = mean[None, :] + z @ Lq.T  # [S, q']
    li = log_fatplus(sign * samples - best_f)
    return float(logmeanexp(fatmax(li, axis=-1), axis=0))


def qlogei_with_pending(
    model: GPModel, Xc: np.ndarray, pend: np.ndarray, z: np.ndarray, best_f: float, sign: float = 1.0
) -> np.ndarray:
    """Vectorised qLogEI of N t-batches [x_i ; pend] (candidate first, then pending).

    Same maths as ``qlogei_joint`` on ``posterior_joint(vstack([x_i, pend]))``; the
    (1+p)x(1+p) Cholesky is done in block form: l11 = sqrt(var_x), l_P1 = c_xP/l11,
    L_S = chol(Sigma_PP - l_P1 l_P1^T).
    """
    spec, prm = model.spec, model.params
    Xcn = normalize_inputs(spec, np.atleast_2d(Xc))
    Pn = normalize_inputs(spec, np.atleast_2d(pend))
    N, p = Xcn.shape[0], Pn.shape[0]
    Ks = cross_cov(spec, prm, Xcn, model.Xn)
    Kp = cross_cov(spec, prm, Pn, model.Xn)
    Vc = sla.solve_triangular(model.L, Ks.T, lower=True)  # [n, N]
    Vp = sla.solve_triangular(model.L, Kp.T, lower=True)  # [n, p]
    s2 = model.ysd**2
    mu_c = model.ybar + model.ysd * (prm.mean + Ks @ model.alpha)
    mu_p = model.ybar + model.ysd * (prm.mean + Kp @ model.alpha)
    var_c = s2 * (prior_var(spec, prm, Xcn) - (Vc * Vc).sum(axis=0))
    cross = s2 * (cross_cov(spec, prm, Xcn, Pn) - Vc.T @ Vp)  # [N, p]
    cov_pp = s2 * (cross_cov(spec, prm, Pn, Pn) - Vp.T @ Vp)  # [p, p]
    out = np.empty(N)
    l11 = _safe_sqrt_var(var_c)
    ok = var_c > 0
    lp1 = cross / l11[:, None]
    schur = cov_pp[None, :, :] - lp1[:, :, None] * lp1[:, None, :]
    try:
        LS = np.linalg.cholesky(schur)
    except np.linalg.LinAlgError:
        ok[:] = False
        LS = None
    if LS is not None and ok.any():
        z0 = z[:, 0][:, None]  # [S,1]
        y0 = mu_c[None, :] + l11[None, :] * z0  # [S,N]
        yp = mu_p[None, None, :] + lp1[None, :, :] * z0[:, :, None] + np.einsum("nij,sj->sni", LS, z[:, 1:])
        samples = np.concatenate([y0[:, :, None], yp], axis=2)  # [S,N,1+p]
        li = log_fatplus(sign * samples - best_f)
        out[:] = logmeanexp(fatmax(li, axis=-1), axis=0)
    for i in np.nonzero(~ok)[0]:  # jitter path: exact restatement
        m, C = model.posterior_joint(np.vstack([np.atleast_2d(Xc)[i : i + 1], pend]))
        out[i] = qlogei_joint(m, C, z, best_f, sign)
    return out


def best_f_from_model(model: GPModel, sign: float = 1.0) -> float:
    """max_i objective(posterior mean at training x_i)  (_builder.py:141-161,256-265)."""
    mu, _ = model.posterior(model.X_train)
    return float((sign * mu).max())


@dataclass
class GreedyResult:
    indices: list[int]  # positions into the candidate matrix, in selection order
    values: list[float]  # acquisition value of each greedy step
    first_scores: np.ndarray | None = None  # q=1 scores of all candidates (step 0)
    runner_up: list | None = None  # per greedy step: (index, value) of the second-best remaining candidate


def optimize_acqf_discrete_qlogei(
    model: GPModel,
    Xcand: np.ndarray,
    q: int,
    z_by_q: dict[int, np.ndarray] | None = None,
    seed: int | None = None,
    S: int = 512,
    sign: float = 1.0,
    X_pending=None,
    best_f: float | None = None,
    keep_scores: bool = False,
) -> GreedyResult:
    """Sequential greedy over a discrete set [UPSTREAM A10] with qLogEI.

    Each greedy step scores every remaining candidate as its own q=1 t-batch,
    jointly with (base pending + already chosen) points, takes the first-index
    argmax and removes the row.
    """
    Xcand = np.ascontiguousarray(Xcand, dtype=np.float64)
    N = Xcand.shape[0]
    if z_by_q is None and seed is None:
        seed = int(draw_sampler_seed())
    if best_f is None:
        best_f = best_f_from_model(model, sign)
    base_pending = np.zeros((0, Xcand.shape[1])) if X_pending is None else np.atleast_2d(X_pending)
    alive = np.ones(N, dtype=bool)
    chosen: list[int] = []
    values: list[float] = []
    runner_up: list = []
    first_scores = None

    def get_z(qp):
        if z_by_q is not None:
            return z_by_q[qp]
        return sobol_normal_base_samples(S, qp, seed)

    for step in range(q):
        pend = np.vstack([base_pending, Xcand[chosen]]) if chosen else base_pending
        p = pend.shape[0]
        z = get_z(1 + p)
        scores = np.full(N, -np.inf)
        idx_alive = np.nonzero(alive)[0]
        if p == 0:
            for s in range(0, idx_alive.size, MAX_BATCH_SIZE):
                ids = idx_alive[s : s + MAX_BATCH_SIZE]
                mu, var = model.posterior(Xcand[ids])
                scores[ids] = qlogei_q1(mu, var, z[:, 0], best_f, sign)
        else:
            for s in range(0, idx_alive.size, MAX_BATCH_SIZE):
                ids = idx_alive[s : s + MAX_BATCH_SIZE]
                scores[ids] = qlogei_with_pending(model, Xcand[ids], pend, z, best_f, sign)
        if step == 0 and keep_scores:
            first_scores = scores.copy()
        best = int(np.argmax(scores))  # first index on ties (torch.argmax)
        chosen.append(best)
        values.append(float(scores[best]))
        alive[best] = False
        if N > 1:  # how decisive the step was: the best of the rest (a test may need it to judge a differing pick)
            held, scores[best] = scores[best], -np.inf
            second = int(np.argmax(scores))
            runner_up.append((second, float(scores[second])))
            scores[best] = held
    return GreedyResult(chosen, values, first_scores, runner_up)


def topk_first_index(scores: np.ndarray, k: int) -> np.ndarray:
    """k best q=1 scores, descending, ties -> lower index first (stable)."""
    order = np.argsort(-scores, kind="stable")
    return order[:k]


# --------------------------------------------------------------------------------------
# other acquisition functions of baybe/acquisition/acqfs.py:161-290
# --------------------------------------------------------------------------------------
# MC family (BoTorch SampleReducingMCAcquisitionFunction [UPSTREAM]): value = mean_s max_j u(obj_sj)
#   qEI:   u = relu(obj - best_f)                      qSR:  u = obj
#   qPI:   u = sigmoid((obj - best_f) / tau), tau=1e-3  qUCB: u = m + sqrt(beta pi / 2) |obj - m|, m = mean_s obj
#   qPSTD: u = sqrt(pi / 2) |obj - m|
# analytic family (q = 1; u = (mu~ - best_f) / sigma with mu~ = sign * mu):
#   PM: mu~   PSTD: +-sigma   UCB: mu~ + sqrt(beta) sigma   EI: sigma (phi(u) + u Phi(u))   PI: Phi(u)
#   LogEI: log(sigma) + log(phi(u) + u Phi(u)) evaluated with the asymptotic branch for u < -1
MC_KINDS = ("qLogEI", "qEI", "qPI", "qSR", "qUCB", "qPSTD")
ANALYTIC_KINDS = ("PM", "PSTD", "UCB", "EI", "LogEI", "PI")
TAU_PI = 1e-3  # [UPSTREAM] qProbabilityOfImprovement tau


def _mc_utility(kind: str, obj: np.ndarray, best_f: float, beta: float) -> np.ndarray:
    """obj [S, ...] -> per-sample utilities of the same shape."""
    if kind == "qEI":
        return np.maximum(obj - best_f, 0.0)
    if kind == "qPI":
        return 1.0 / (1.0 + np.exp(-(obj - best_f) / TAU_PI))
    if kind == "qSR":
        return obj
    m = obj.mean(axis=0, keepdims=True)
    if kind == "qUCB":
        return m + math.sqrt(beta * math.pi / 2.0) * np.abs(obj - m)
    if kind == "qPSTD":
        return math.sqrt(math.pi / 2.0) * np.abs(obj - m)
    raise ValueError(kind)


def mc_acq_q1(kind: str, mu, var, z, best_f: float = 0.0, sign: float = 1.0, beta: float = 0.2) -> np.ndarray:
    if kind == "qLogEI":
        return qlogei_q1(mu, var, z, best_f, sign)
    sd = _safe_sqrt_var(var)
    obj = sign * (mu[None, :] + sd[None, :] * z.reshape(-1, 1))
    return _mc_utility(kind, obj, best_f, beta).mean(axis=0)


def mc_acq_joint(kind: str, mean, cov, z, best_f: float = 0.0, sign: float = 1.0, beta: float = 0.2) -> float:
    if kind == "qLogEI":
        return qlogei_joint(mean, cov, z, best_f, sign)
    samples = mean[None, :] + z @ _safe_cholesky(cov).T
    return float(_mc_utility(kind, sign * samples, best_f, beta).max(axis=-1).mean())


def _log_h(u: np.ndarray) -> np.ndarray:
    """log(phi(u) + u Phi(u)), stable for very negative u (BoTorch _log_ei_helper [UPSTREAM])."""
    from scipy.special import erfcx, log_ndtr  # noqa: F401
    from scipy.stats import norm

    u = np.asarray(u, dtype=np.float64)
    out = np.empty_like(u)
    hi = u > -1.0
    out[hi] = np.log(norm.pdf(u[hi]) + u[hi] * norm.cdf(u[hi]))
    lo = ~hi
    ul = u[lo]
    logphi = -0.5 * ul * ul - 0.5 * math.log(2 * math.pi)
    w = np.abs(ul) * erfcx(np.abs(ul) / math.sqrt(2.0)) * math.sqrt(math.pi / 2.0)  # |u| Phi(u) / phi(u) < 1
    with np.errstate(divide="ignore"):
        tail = np.where(ul > -1e6, np.log1p(-w), -2.0 * np.log(np.abs(ul)))
    out[lo] = logphi + tail
    return out


def analytic_acq(kind: str, mu, var, best_f: float = 0.0, sign: float = 1.0, beta: float = 0.2, maximize: bool = True):
    from scipy.stats import norm

    mu = np.asarray(mu, dtype=np.float64)
    sd = np.sqrt(np.maximum(np.asarray(var, dtype=np.float64), 1e-12 if kind in ("EI", "LogEI", "PI", "UCB") else 0.0))
    mt = sign * mu
    if kind == "PM":
        return mt
    if kind == "PSTD":
        return sd if maximize else -sd
    if kind == "UCB":
        return mt + math.sqrt(beta) * sd
    u = (mt - best_f) / sd
    if kind == "EI":
        return sd * (norm.pdf(u) + u * norm.cdf(u))
    if kind == "PI":
        return norm.cdf(u)
    if kind == "LogEI":
        return np.log(sd) + _log_h(u)
    raise ValueError(kind)
