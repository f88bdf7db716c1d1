"""CPU oracle for the GP-surrogate recommend() hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

PARITY STATUS: **parity unpinned**.  The arithmetic of this path lives in
third-party packages that are neither vendored in ``/root/reference`` nor
installable here (botorch==0.16.1, gpytorch==1.14.3, linear-operator==0.6, see
``/root/reference/uv.lock:559-560,1502-1503,2480-2481``), and the reference's own
tests hold no golden vectors for GP posteriors / MLL / qLogEI values
(SURVEY.md §4, §8c).  The oracle therefore restates the *published* algorithms
of those packages, anchored on BayBE's call sites, and is pinned only by the
reference's property tests (sign symmetry, pending-point exclusion, linear-data
boundary optimum) plus independent derivations (finite differences, dense
linear algebra identities).  See ``oracle/gp_oracle.py`` for per-function
citations.

One part IS pinned against an implementation by other people: the GP core -
stationary ARD kernels (Matern-1/2, -3/2, -5/2, RBF, rational quadratic, scaled
products and sums), the log marginal likelihood with its gradient, and the exact
Cholesky posterior (mean, variance, joint covariance) - agrees with scikit-learn's
``GaussianProcessRegressor`` to rounding (``tests/test_oracle_vs_sklearn_cpu.py``;
scikit-learn is importable here).  Everything BoTorch-specific (prior constants,
constraint transforms, LOO, fat-tailed qLogEI / qLogNEHVI, Sobol base samples,
greedy semantics, index kernels) stays unpinned.
"""
