"""TEST INFRASTRUCTURE (oracle): restatement of the hybrid-space enumeration of
``baybe/recommenders/pure/bayesian/botorch/hybrid.py:30-163`` -> ``botorch.optim.optimize_acqf_mixed`` [UPSTREAM, botorch 0.16.1] for
qLogEI.  Only ``tests/`` may import this module; the product's hybrid search (``baybe_amd/recommenders.py::_recommend_hybrid``) is
compared with it by the acquisition value it reaches.  PARITY UNPINNED: botorch is not importable here and the reference's tests hold no
vectors for this path (tests/hypothesis_strategies / test_searchspace only build hybrid spaces) - ``optimize_acqf_mixed``'s control flow is
restated from botorch 0.16.1 as recalled; the acquisition arithmetic is ``oracle/gp_oracle.py``'s (its own header says what pins it).

What the reference does (hybrid.py:95-136): every row of the (possibly subsampled) discrete candidate set becomes one
``fixed_features`` dictionary; ``optimize_acqf_mixed`` then

* for q = 1: runs ``optimize_acqf`` (raw samples -> ``num_restarts`` starts -> L-BFGS-B on the continuous coordinates, the
  discrete ones fixed) once PER DISCRETE ROW and returns the best of the per-row optima;
* for q > 1: sequential greedy - q rounds of the q = 1 search, each earlier winner appended to ``X_pending`` (restored at the end).

The multi-start optimiser is a stochastic approximation of "the maximum of the acquisition function over the continuous box for this
row".  The oracle computes that maximum itself, deterministically, for few continuous dimensions: a dense grid over the box per row,
then scipy's L-BFGS-B (finite-difference gradient, the box as bounds) from the best grid points.  Rows whose grid optimum is far
below the best row's are not polished (they cannot win: the polish moves a value by much less than the margin).
"""

from __future__ import annotations

import itertools

import numpy as np
from scipy import optimize as sopt

from oracle import gp_oracle as go

GRID_POINTS = {1: 513, 2: 65, 3: 25, 4: 13}  # per continuous axis
POLISH_STARTS = 12  # best (row, grid point) pairs overall that get an L-BFGS-B run
ROW_MARGIN = 0.5  # rows whose grid optimum is this far below the best grid value are not polished


def _scores(model, X, pend, z, best_f, sign):
    if len(pend) == 0:
        mu, var = model.posterior(X)
        return go.qlogei_q1(mu, var, z[:, 0], best_f, sign)
    return go.qlogei_with_pending(model, X, pend, z, best_f, sign)


def mixed_step(model, D: np.ndarray, cb: np.ndarray, pend: np.ndarray, z: np.ndarray, best_f: float, sign: float = 1.0):
    """One q = 1 round of ``optimize_acqf_mixed``: (index of the winning discrete row, its continuous optimum, the value, the best
    value of every row on the grid).  ``D`` [Nd, dd] discrete rows (comp rep), ``cb`` [2, dc] the continuous box, ``pend`` [p, dd + dc]
    points already in the batch / pending, ``z`` [S, 1 + p] base samples."""
    Nd, dd = D.shape
    dc = cb.shape[1]
    if dc not in GRID_POINTS:
        raise ValueError(f"the enumeration oracle is dense in the continuous box: 1..{max(GRID_POINTS)} continuous dimensions")
    axes = [np.linspace(cb[0, a], cb[1, a], GRID_POINTS[dc]) for a in range(dc)]
    G = np.array(list(itertools.product(*axes)))  # [g, dc]
    row_best = np.empty(Nd)
    starts = []
    for i in range(Nd):
        X = np.hstack([np.repeat(D[i : i + 1], len(G), axis=0), G])
        s = _scores(model, X, pend, z, best_f, sign)
        order = np.argsort(-s)[:3]
        row_best[i] = s[order[0]]
        starts += [(float(s[j]), i, G[j]) for j in order]
    top = row_best.max()
    starts = sorted((st for st in starts if row_best[st[1]] >= top - ROW_MARGIN), key=lambda st: -st[0])[:POLISH_STARTS]
    best = (-np.inf, -1, None)
    for s0, i, c0 in starts:
        def neg(c, i=i):
            return -float(_scores(model, np.concatenate([D[i], c])[None, :], pend, z, best_f, sign)[0])

        res = sopt.minimize(neg, c0, method="L-BFGS-B", bounds=list(zip(cb[0], cb[1])), options={"ftol": 1e-15, "gtol": 1e-10, "maxiter": 200})
        val, c = (-float(res.fun), res.x) if -res.fun >= s0 else (s0, c0)
        if val > best[0]:
            best = (val, i, np.asarray(c, dtype=np.float64))
    return best[1], best[2], best[0], row_best


def optimize_acqf_mixed_qlogei(model, D, cb, q: int, seed: int, S: int = 512, sign: float = 1.0, X_pending=None, best_f=None):
    """The reference's hybrid recommendation for a batch of q: sequential greedy over ``mixed_step`` (see the module docstring).
    Returns (points [q, dd + dc], per-step values, indices of the discrete rows)."""
    D = np.atleast_2d(np.asarray(D, dtype=np.float64))
    cb = np.asarray(cb, dtype=np.float64)
    d = D.shape[1] + cb.shape[1]
    best_f = go.best_f_from_model(model, sign) if best_f is None else best_f
    base = np.zeros((0, d)) if X_pending is None else np.atleast_2d(X_pending)
    chosen = np.zeros((0, d))
    values, rows = [], []
    for _ in range(q):
        in_batch = np.concatenate([base, chosen], axis=0)  # X_pending of this round: the caller's pending points, then earlier winners
        z = go.sobol_normal_base_samples(S, 1 + len(in_batch), seed)
        i, c, val, _ = mixed_step(model, D, cb, in_batch, z, best_f, sign)
        chosen = np.concatenate([chosen, np.concatenate([D[i], c])[None, :]], axis=0)
        values.append(val)
        rows.append(i)
    return chosen, values, rows
