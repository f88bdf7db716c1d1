"""CPU oracle for qLogNoisyExpectedHypervolumeImprovement (BayBE default for ParetoObjective).

TEST INFRASTRUCTURE — see oracle/__init__.py.  PARITY UNPINNED, more so than the qLogEI oracle:
BoTorch's implementation (botorch/acquisition/multi_objective/logei.py, .../base.py
NoisyExpectedHypervolumeMixin, utils/multi_objective/box_decompositions) is restated from the
published algorithms, and the following could not be checked against the source: the order in which
sampler seeds are drawn, how base samples are extended when pending points are cached into the
baseline, and the exact cell list of ``FastNondominatedPartitioning`` (the smoothed value depends on
the decomposition at O(tau_max) where an improvement equals a cell length).

Reference call sites: ``baybe/acquisition/acqfs.py:477-484`` (qLogNEHVI, prune_baseline=True),
``baybe/acquisition/_builder.py:301-324`` (ref_point from measured targets, X_baseline = all
training inputs), ``baybe/acquisition/acqfs.py:369-426`` (compute_ref_point),
``baybe/surrogates/composite.py:101-134`` (one independent GP per target -> ModelListGP),
``baybe/objectives/base.py:99-150`` (per-target orientation).

Estimator (q = 1, pending points cached into the baseline) [UPSTREAM A12]:
  for each MC sample s: baseline values F_b,s = mu_b + L_b z_b,s per output (joint over the baseline
  points, independent across outputs), oriented by the objective; cells = box decomposition of the
  region not dominated by the Pareto front of F_b,s above the reference point; the candidate value
  f(x)_s is drawn from the posterior *jointly* with the baseline (same base samples), i.e.
  f(x)_s = mu_x + S_xb S_bb^-1 (F_b,s - mu_b) + sqrt(s_xx - S_xb S_bb^-1 S_bx) z_x,s;
  log_area(cell) = sum_j fatmin( log_fatplus(f_j - l_j; 1e-6), log(min(u_j, 1e10) - l_j); 1e-2 );
  value = logmeanexp_s logsumexp_cells log_area.
"""

from __future__ import annotations

import math

import numpy as np
from scipy import linalg as sla

from oracle import gp_oracle as go

TAU_RELU = 1e-6
TAU_MAX = 1e-2
UPPER_CLAMP = 1e10  # cell_upper_bounds.clamp_max(1e10) for double [UPSTREAM]


# ---- reference point (BayBE side, exact) ----------------------------------------------------------
def compute_ref_point(array, maximize=None, factor=0.1):
    """``_ExpectedHypervolumeImprovement.compute_ref_point`` (acqfs.py:369-426), stated the way its docstring does:
    per target, the reference value lies ``factor`` x (best - worst) beyond the worst observed value."""
    columns = np.asarray(array, dtype=np.float64).T
    upward = [True] * len(columns) if maximize is None else [bool(f) for f in maximize]
    point = []
    for col, up in zip(columns, upward):
        worst, best = (col.min(), col.max()) if up else (col.max(), col.min())
        point.append(worst - factor * (best - worst))
    return np.array(point)


# ---- Pareto front & box decomposition (maximisation) -----------------------------------------------
def pareto_front(Y: np.ndarray) -> np.ndarray:
    """Non-dominated rows of Y (maximisation), duplicates kept once."""
    Y = np.unique(np.asarray(Y, dtype=np.float64), axis=0)
    keep = np.ones(len(Y), bool)
    for i in range(len(Y)):
        if keep[i]:
            dom = (Y >= Y[i]).all(1) & (Y > Y[i]).any(1)
            if dom.any():
                keep[i] = False
    return Y[keep]


def nondominated_cells(Y, ref):
    """Disjoint boxes [l, u) (u may be +inf) tiling {y >= ref : y not dominated by the front of Y}.

    Local upper bounds of the negated (minimisation) problem, incremental algorithm of
    Lacour, Klamroth & Fonseca (2017), Alg. 1, general-position form (one defining point per
    dimension); the search zone of bound u is cut to the box
      z_1 in (-inf, u_1),  z_j in [max_{k<j} z^k(u)_j, u_j)  (j >= 2),
    which tiles the search region.  Returned in maximisation coordinates.
    """
    ref = np.array(ref, dtype=np.float64, copy=True)
    m = ref.shape[0]
    P = pareto_front(Y) if len(Y) else np.zeros((0, m))
    P = P[(P > ref).all(1)] if len(P) else P
    U = [(-ref).copy()]  # local upper bounds (min space), start: the (negated) reference point
    Z0 = np.full((m, m), -np.inf)
    for j in range(m):
        Z0[j, j] = -ref[j]
    Z = [Z0]  # Z[u][k] = defining point of bound u in dimension k
    for p in -P:  # insert the front, one point at a time
        newU, newZ = [], []
        for u, zd in zip(U, Z):
            if not (p < u).all():
                newU.append(u)
                newZ.append(zd)
                continue
            for j in range(m):
                if all(zd[k, j] < p[j] for k in range(m) if k != j):
                    uj = u.copy()
                    uj[j] = p[j]
                    zj = zd.copy()
                    zj[j] = p
                    newU.append(uj)
                    newZ.append(zj)
        U, Z = newU, newZ
    lows, ups = [], []
    for u, zd in zip(U, Z):
        lb = np.full(m, -np.inf)
        for j in range(1, m):
            lb[j] = max(zd[k, j] for k in range(j))
        if (lb < u).all():
            lows.append(-u)  # max space: y > -u
            ups.append(-lb)  # max space: y <= -lb (possibly +inf)
    if not lows:
        return np.zeros((0, m)), np.zeros((0, m))
    return np.array(lows), np.array(ups)


def hypervolume(Y: np.ndarray, ref: np.ndarray) -> float:
    """Exact dominated hypervolume by recursive slicing (independent of the decomposition)."""
    ref = np.array(ref, dtype=np.float64, copy=True)
    P = pareto_front(Y) if len(Y) else np.zeros((0, len(ref)))
    P = P[(P > ref).all(1)] if len(P) else P
    if len(P) == 0:
        return 0.0
    if P.shape[1] == 1:
        return float(P[:, 0].max() - ref[0])
    order = np.argsort(-P[:, 0])
    P = P[order]
    hv, prev = 0.0, None
    xs = list(P[:, 0]) + [ref[0]]
    for i in range(len(P)):
        width = xs[i] - xs[i + 1]
        if width > 0:
            hv += width * hypervolume(P[: i + 1, 1:], ref[1:])
    return hv


def exact_hvi(y: np.ndarray, Y: np.ndarray, ref: np.ndarray) -> float:
    return hypervolume(np.vstack([Y, y[None, :]]), ref) - hypervolume(Y, ref)


def hvi_from_cells(y: np.ndarray, lows: np.ndarray, ups: np.ndarray) -> float:
    if len(lows) == 0:
        return 0.0
    side = np.clip(np.minimum(y[None, :], ups) - lows, 0.0, None)
    return float(side.prod(axis=1).sum())


# ---- smoothed log-HVI --------------------------------------------------------------------------------
def fatmin2(a: np.ndarray, b: np.ndarray, tau: float = TAU_MAX, alpha: float = 2.0) -> np.ndarray:
    """fatmin over the pair (a, b) = -fatmax(-a, -b) [UPSTREAM safe_math.fatmin]."""
    mn = np.minimum(a, b)
    with np.errstate(invalid="ignore"):
        diff = np.abs(a - b)
    diff = np.where(np.isnan(diff), np.inf, diff)  # (-inf) - (-inf)
    return mn - tau * np.log1p((alpha / (alpha + diff / tau)) ** alpha)


def log_hvi_smoothed(f: np.ndarray, lows: np.ndarray, ups: np.ndarray) -> float:
    """log sum_cells prod_j smooth-clamp side lengths for one sample; f [m]."""
    if len(lows) == 0:
        return -np.inf
    li = go.log_fatplus(f[None, :] - lows, TAU_RELU)
    with np.errstate(divide="ignore"):
        ll = np.log(np.minimum(ups, UPPER_CLAMP) - lows)
    la = fatmin2(li, ll).sum(axis=1)
    M = la.max()
    if not np.isfinite(M):
        return -np.inf
    return float(M + np.log(np.exp(la - M).sum()))


def logmeanexp_with_neginf(v: np.ndarray) -> float:
    M = v.max()
    if not np.isfinite(M):
        return -np.inf
    return float(M + np.log(np.exp(v - M).mean()))


def _solve_lower(L: np.ndarray, b: np.ndarray) -> np.ndarray:
    import scipy.linalg as sla

    return sla.solve_triangular(L, b, lower=True)


# ---- the acquisition -----------------------------------------------------------------------------------
class NEHVIOracle:
    """qLogNEHVI over m independent GPs (``go.GPModel``), q = 1 t-batches."""

    def __init__(self, models, signs, X_baseline, ref_point, z: np.ndarray):
        """z: base samples [S, n_b + 1, m] (point-major, output-minor; the last point is the candidate).
        ref_point is given in *objective* space (i.e. after orientation by ``signs``)."""
        self.models, self.signs = models, np.asarray(signs, dtype=np.float64)
        self.Xb = np.atleast_2d(np.asarray(X_baseline, dtype=np.float64))
        self.ref, self.z = np.array(ref_point, dtype=np.float64), z
        S, nb1, m = z.shape
        assert nb1 == len(self.Xb) + 1 and m == len(models)
        self.mu_b, self.L_b, self.Fb = [], [], np.empty((S, len(self.Xb), m))
        for o, mod in enumerate(models):
            mu, cov = mod.posterior_joint(self.Xb)
            L = go._safe_cholesky(cov)
            self.mu_b.append(mu)
            self.L_b.append(L)
            self.Fb[:, :, o] = mu[None, :] + z[:, :-1, o] @ L.T
        self.obj_b = self.Fb * self.signs[None, None, :]
        self.cells = [nondominated_cells(self.obj_b[s], self.ref) for s in range(S)]

    def candidate_samples(self, x: np.ndarray) -> np.ndarray:
        """f(x)_s [S, m]: joint draw with the baseline through the CACHED baseline factor, as BoTorch's
        ``sample_cached_cholesky`` (botorch/utils/low_rank.py) does it [UPSTREAM]: with the joint posterior covariance of
        [X_b; x] partitioned into the baseline block (factor L_b, cached when the acquisition function is built), the
        cross block c and the candidate's variance v,

            l = L_b^-1 c,   s = psd_safe_cholesky(v - l.l)  (1 x 1: jitter 1e-8, 1e-7, 1e-6 when it is not positive),
            f(x) = mu_x + z_b . l + s z_x.

        The jitter therefore only ever touches the candidate's conditional variance - a candidate that coincides with a
        baseline point is drawn as that point's sampled value plus sqrt(1e-8) z_x - and never the baseline factor (an
        earlier version of this oracle re-factorised the whole (n_b + 1) matrix with diagonal jitter, which moves such a
        candidate's tail value: sqrt(2e-8) instead of sqrt(1e-8))."""
        S, _, m = self.z.shape
        out = np.empty((S, m))
        for o, mod in enumerate(self.models):
            mu, cov = mod.posterior_joint(np.vstack([self.Xb, np.atleast_2d(x)]))
            nb = len(self.Xb)
            if nb:
                l = _solve_lower(self.L_b[o], cov[:nb, -1])
            else:
                l = np.zeros(0)
            sd = go._safe_sqrt_var(np.array([cov[-1, -1] - l @ l]))[0]
            out[:, o] = mu[-1] + self.z[:, :nb, o] @ l + sd * self.z[:, nb, o]
        return out

    def value(self, x: np.ndarray) -> float:
        f = self.candidate_samples(x) * self.signs[None, :]
        per_sample = np.array([log_hvi_smoothed(f[s], *self.cells[s]) for s in range(f.shape[0])])
        return logmeanexp_with_neginf(per_sample)

    def values(self, X: np.ndarray) -> np.ndarray:
        return np.array([self.value(x) for x in np.atleast_2d(X)])


def sobol_normal_base_samples_nd(S: int, n_points: int, m: int, seed: int) -> np.ndarray:
    """[S, n_points, m] from one scrambled Sobol draw of dimension n_points * m [UPSTREAM A8]."""
    return go.sobol_normal_base_samples(S, n_points * m, seed).reshape(S, n_points, m)


def prune_baseline(models, signs, X_baseline, ref_point, seed: int, num_samples: int = 2048):
    """``prune_inferior_points_multi_objective`` [UPSTREAM]: keep baseline points that are
    Pareto-optimal and above the reference point in at least one joint posterior sample."""
    Xb = np.atleast_2d(X_baseline)
    nb, m = len(Xb), len(models)
    z = sobol_normal_base_samples_nd(num_samples, nb, m, seed)
    F = np.empty((num_samples, nb, m))
    for o, mod in enumerate(models):
        mu, cov = mod.posterior_joint(Xb)
        F[:, :, o] = (mu[None, :] + z[:, :, o] @ go._safe_cholesky(cov).T) * signs[o]
    keep = np.zeros(nb, bool)
    for s in range(num_samples):
        Y = F[s]
        nd = np.ones(nb, bool)
        for i in range(nb):
            dom = (Y >= Y[i]).all(1) & (Y > Y[i]).any(1)
            nd[i] = not dom.any()
        keep |= nd & (Y > ref_point).all(1)
    return np.nonzero(keep)[0]
