"""Oracle pins (CPU).  The reference holds no golden vectors for this path, so
the oracle is pinned by (i) independent derivations — finite differences, closed forms, dense
identities —, (ii) the reference's own *property* tests restated on the oracle, and (iii)
regression fixtures under tests/golden/."""

import math
from pathlib import Path

import numpy as np
import pytest

from _problems import fixed_theta, make_problem
from oracle import gp_oracle as go

GOLD = Path(__file__).resolve().parent / "golden"


def _fd_grad(spec, raw, Xn, ystd, eps=1e-6):
    g = np.zeros_like(raw)
    for i in range(len(raw)):
        e = np.zeros_like(raw)
        e[i] = eps
        g[i] = (go.fit_objective(spec, raw + e, Xn, ystd)[0] - go.fit_objective(spec, raw - e, Xn, ystd)[0]) / (2 * eps)
    return g


@pytest.mark.parametrize("criterion", ["mll", "loo"])
@pytest.mark.parametrize("kernel", go.KERNELS)
def test_fit_objective_gradient_matches_finite_differences(criterion, kernel):
    rng = np.random.default_rng(0)
    d, n, T = 4, 30, 3
    X = np.hstack([rng.random((n, d)), rng.integers(0, T, size=(n, 1)).astype(float)])
    y = rng.standard_normal(n)
    spec = go.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=T, kernel=kernel)
    spec.criterion = criterion
    spec.use_outputscale = True
    spec.lengthscale = go.Hyper(0.0, True, spec.lengthscale.prior, spec.lengthscale.init)  # Positive()
    Xn = go.normalize_inputs(spec, X)
    ystd, _, _ = go.standardize_targets(y)
    p0 = go.initial_params(spec)
    p0.task_W = 0.3 + rng.random((T, T))
    p0.mean = 0.2
    if kernel in go.DOT_KERNELS:  # rank <= d kernel matrices: with the preset's start noise the central differences lose digits
        p0.noise = 0.1
    raw = go.pack_raw(spec, p0)
    _, g = go.fit_objective(spec, raw, Xn, ystd)
    assert np.allclose(g, _fd_grad(spec, raw, Xn, ystd), rtol=1e-5, atol=1e-6)


def test_mll_value_matches_dense_gaussian_logpdf():
    from scipy.stats import multivariate_normal

    X, Xt, y = make_problem(300, 4, 25, seed=3)
    spec = go.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
    p = go.initial_params(spec)
    Xn = go.normalize_inputs(spec, Xt)
    ystd, ybar, s = go.standardize_targets(y)
    assert math.isclose(ybar, y.mean()) and math.isclose(s, y.std(ddof=1))
    K = go.cross_cov(spec, p, Xn, Xn) + p.noise * np.eye(25)
    ref = multivariate_normal(mean=np.full(25, p.mean), cov=K).logpdf(ystd)
    assert math.isclose(go.data_term(spec, p, Xn, ystd).value, ref, rel_tol=1e-10)


def test_loo_value_matches_explicit_leave_one_out():
    X, Xt, y = make_problem(300, 3, 18, seed=4)
    spec = go.GPSpec.baybe_default(3, np.zeros(3), np.ones(3))
    spec.criterion = "loo"
    p = go.initial_params(spec)
    Xn = go.normalize_inputs(spec, Xt)
    ystd, _, _ = go.standardize_targets(y)
    n = 18
    K = go.cross_cov(spec, p, Xn, Xn) + p.noise * np.eye(n)
    tot = 0.0
    for i in range(n):
        m = np.arange(n) != i
        Kmm = K[np.ix_(m, m)]
        k = K[m, i]
        sol = np.linalg.solve(Kmm, k)
        mu = p.mean + sol @ (ystd[m] - p.mean)
        s2 = K[i, i] - k @ sol
        tot += -0.5 * math.log(2 * math.pi * s2) - 0.5 * (ystd[i] - mu) ** 2 / s2
    assert math.isclose(go.data_term(spec, p, Xn, ystd).value, tot, rel_tol=1e-9)


def test_default_preset_constants():
    """presets/baybe.py:95-106,134-144: prior modes are the initial values."""
    spec = go.GPSpec.baybe_default(20, np.zeros(20), np.ones(20))
    assert math.isclose(spec.lengthscale.init, math.exp(math.sqrt(2) - 3) * math.sqrt(20))
    assert math.isclose(spec.noise.init, math.exp(-5.0))
    assert not spec.lengthscale.transformed and spec.lengthscale.lower == 2.5e-2
    assert not spec.noise.transformed and spec.noise.lower == 1e-4
    assert spec.ls_prior == ("gamma", 3.0, 2.0 / math.exp(math.sqrt(2) - 3) / math.sqrt(20))
    assert math.isclose(spec.noise_prior[2], math.exp(5.0))
    assert spec.criterion == "mll" and go.GPSpec.baybe_default(5, np.zeros(5), np.ones(5), 4, 3).criterion == "loo"


def test_posterior_interpolates_and_matches_dense_formula():
    X, Xt, y = make_problem(400, 5, 30, seed=5)
    spec = go.GPSpec.baybe_default(5, np.zeros(5), np.ones(5))
    ls, nz, _ = fixed_theta(5)
    m = go.fit_gp(spec, Xt, y, params=go.GPParams(np.full(5, ls), nz, 0.1))
    Xn, Xcn = go.normalize_inputs(spec, Xt), go.normalize_inputs(spec, X[:50])
    K = go.cross_cov(spec, m.params, Xn, Xn) + nz * np.eye(30)
    Ks = go.cross_cov(spec, m.params, Xcn, Xn)
    mu = m.ybar + m.ysd * (0.1 + Ks @ np.linalg.solve(K, m.ystd - 0.1))
    var = m.ysd**2 * (1.0 - np.einsum("ij,ij->i", Ks, np.linalg.solve(K, Ks.T).T))
    pm, pv = m.posterior(X[:50])
    assert np.allclose(pm, mu, rtol=1e-10) and np.allclose(pv, var, rtol=1e-8)
    jm, jc = m.posterior_joint(X[:7])
    assert np.allclose(jm, pm[:7]) and np.allclose(np.diag(jc), pv[:7])


def test_safe_math_closed_forms():
    # fatmax of a single element is the identity; log_fatplus -> log(x) for x >> tau
    x = np.array([[0.3], [-2.0]])
    assert np.allclose(go.fatmax(x, axis=-1), x[:, 0])
    assert math.isclose(float(go.log_fatplus(np.array(0.5))), math.log(0.5), rel_tol=1e-9)
    # far negative: fat tail  log(tau * 0.1 / (1 + t^2))
    t = -0.3 / go.TAU_RELU
    assert math.isclose(float(go.log_fatplus(np.array(-0.3))), math.log(go.TAU_RELU * 0.1 / (1 + t * t)), rel_tol=1e-9)
    # logmeanexp(log f) == log(mean f)
    f = np.random.default_rng(0).random((64, 5)) + 1e-3
    assert np.allclose(go.logmeanexp(np.log(f), axis=0), np.log(f.mean(0)))


def test_sobol_base_samples_are_pinned_and_normal():
    z = go.sobol_normal_base_samples(512, 1, 1234)[:, 0]
    z2 = go.sobol_normal_base_samples(512, 1, 1234)[:, 0]
    assert np.array_equal(z, z2)
    assert abs(z.mean()) < 2e-2 and abs(z.std() - 1.0) < 3e-2
    g = np.load(GOLD / "small_matern52_max.npz")
    assert np.array_equal(z, g["z1"])  # torch SobolEngine stream pinned by the fixture


# ---- the reference's own property tests, restated ----------------------------------------------
def test_minimization_is_maximization_of_negated_target():
    """tests/integration/test_minimization.py:41-78: p_min.mean == -p_max.mean, equal covariance,
    acquisition values equal."""
    X, Xt, y = make_problem(800, 5, 40, seed=6)
    spec = go.GPSpec.baybe_default(5, np.zeros(5), np.ones(5))
    ls, nz, _ = fixed_theta(5)
    prm = go.GPParams(np.full(5, ls), nz, 0.0)
    m_max = go.fit_gp(spec, Xt, y, params=prm)
    m_min = go.fit_gp(spec, Xt, -y, params=prm)
    a, va = m_max.posterior(X)
    b, vb = m_min.posterior(X)
    assert np.allclose(a, -b, rtol=0, atol=1e-12) and np.allclose(va, vb, rtol=1e-12)
    z = go.sobol_normal_base_samples(512, 1, 9)[:, 0]
    s_max = go.qlogei_q1(a, va, z, go.best_f_from_model(m_max, 1.0), 1.0)
    zneg = -z  # the sample y = mu + sd z of the negated model mirrors with -z
    s_min = go.qlogei_q1(b, vb, zneg, go.best_f_from_model(m_min, -1.0), -1.0)
    assert np.allclose(s_max, s_min, rtol=1e-4, atol=0.1)
    assert np.allclose(s_max, s_min, rtol=1e-9, atol=1e-9)


def test_pending_points_are_not_recommended_again():
    """tests/test_pending_experiments.py:100-128: a second batch given the first as pending has
    no overlap with it."""
    X, Xt, y = make_problem(600, 4, 30, seed=7)
    spec = go.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
    m = go.fit_gp(spec, Xt, y)
    r1 = go.optimize_acqf_discrete_qlogei(m, X, 3, seed=1337)
    mask = np.ones(len(X), bool)
    mask[r1.indices] = False  # Campaign excludes pending rows from the candidates (campaign.py:552-566)
    r2 = go.optimize_acqf_discrete_qlogei(m, X[mask], 3, seed=1337, X_pending=X[r1.indices])
    first = {tuple(np.round(X[i], 3)) for i in r1.indices}
    second = {tuple(np.round(X[mask][i], 3)) for i in r2.indices}
    assert not (first & second)
    assert len(set(r1.indices)) == 3


def test_linear_data_recommends_the_boundary():
    """tests/test_objective.py:142-151 (discrete analogue): linear data on [0,1] -> the optimum
    of the acquisition is the correct boundary for max and min targets."""
    grid = np.linspace(0, 1, 101)[:, None]
    Xt = np.linspace(0.1, 0.9, 9)[:, None]
    y = 2.0 * Xt[:, 0]
    spec = go.GPSpec.baybe_default(1, np.zeros(1), np.ones(1))
    m = go.fit_gp(spec, Xt, y)
    r = go.optimize_acqf_discrete_qlogei(m, grid, 1, seed=0, sign=1.0)
    assert np.isclose(grid[r.indices[0], 0], 1.0)
    r = go.optimize_acqf_discrete_qlogei(m, grid, 1, seed=0, sign=-1.0)
    assert np.isclose(grid[r.indices[0], 0], 0.0)


def test_greedy_first_index_ties_and_uniqueness():
    X, Xt, y = make_problem(300, 3, 20, seed=8)
    Xdup = np.vstack([X, X[:50]])  # duplicated rows -> exact score ties
    spec = go.GPSpec.baybe_default(3, np.zeros(3), np.ones(3))
    m = go.fit_gp(spec, Xt, y)
    r = go.optimize_acqf_discrete_qlogei(m, Xdup, 2, seed=3, keep_scores=True)
    assert r.indices[0] == int(np.argmax(r.first_scores))  # np.argmax = first index
    assert len(set(r.indices)) == 2


# ---- regression fixtures ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["cfg1_plumbing", "small_matern52_max", "small_matern52_min", "small_rbf",
                                  "small_matern32", "small_matern12", "mid_320"])
def test_oracle_reproduces_golden_fixture(name):
    g = np.load(GOLD / f"{name}.npz")
    d = g["X"].shape[1]
    spec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), kernel=str(g["kernel"]))
    prm = go.GPParams(g["ls"], float(g["noise"]), float(g["mean_const"]))
    m = go.fit_gp(spec, g["Xt"], g["y"], params=prm)
    mean, var = m.posterior(g["X"])
    assert np.allclose(mean, g["post_mean"], rtol=1e-10, atol=1e-12)
    assert np.allclose(var, g["post_var"], rtol=1e-8)
    sign = float(g["sign"])
    scores = go.qlogei_q1(mean, var, g["z1"], float(g["best_f"]), sign)
    assert np.allclose(scores, g["scores"], rtol=1e-9, atol=1e-9)
    if name != "mid_320":
        r = go.optimize_acqf_discrete_qlogei(m, g["X"], int(g["q"]), seed=4321, sign=sign, X_pending=g["pend"])
        assert r.indices == g["greedy_idx"].tolist()
        assert np.allclose(r.values, g["greedy_val"], rtol=1e-9)


def test_fitted_hyperparameters_of_cfg1_are_reproduced():
    g = np.load(GOLD / "cfg1_plumbing.npz")
    spec = go.GPSpec.baybe_default(3, np.zeros(3), np.ones(3))
    m = go.fit_gp(spec, g["Xt"], g["y"])
    assert np.allclose(m.params.lengthscale, g["ls"], rtol=1e-6)
    assert math.isclose(m.params.noise, float(g["noise"]), rel_tol=1e-6)


@pytest.mark.parametrize("kernel,nu", [("matern52", 2.5), ("matern32", 1.5), ("rbf", None)])
def test_gp_algebra_is_pinned_against_scikit_learn(kernel, nu):
    """Independent pin of the oracle's GP algebra (kernel definition with ARD lengthscales and an
    outputscale, exact posterior mean / latent variance / joint covariance, log marginal likelihood)
    against scikit-learn's GaussianProcessRegressor at fixed hyper-parameters.  botorch/gpytorch are not
    importable here, so the BoTorch-specific pieces (qLogEI smoothing, samplers, NEHVI) stay unpinned;
    this covers the part of the path every other result is built on."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern

    d, n = 5, 40
    X, Xt, y = make_problem(300, d, n, seed=12)
    lo, hi = np.full(d, -0.5), np.full(d, 2.0)  # scaling bounds different from the data range
    spec = go.GPSpec.baybe_default(d, lo, hi, kernel=kernel)
    spec.use_outputscale = True
    rng = np.random.default_rng(0)
    p = go.GPParams(lengthscale=0.3 + rng.random(d), noise=0.037, mean=0.0, outputscale=1.7)
    m = go.GPModel(spec, p, Xt, y)
    base = RBF(length_scale=p.lengthscale) if nu is None else Matern(length_scale=p.lengthscale, nu=nu)
    gpr = GaussianProcessRegressor(kernel=ConstantKernel(p.outputscale) * base, alpha=p.noise, optimizer=None)
    Xn, Xcn = (Xt - lo) / (hi - lo), (X - lo) / (hi - lo)
    ystd = (y - y.mean()) / y.std(ddof=1)
    gpr.fit(Xn, ystd)
    mu_s, cov_s = gpr.predict(Xcn[:64], return_cov=True)
    mu, var = m.posterior(X)
    _, std_all = gpr.predict(Xcn, return_std=True)
    s = y.std(ddof=1)
    assert np.allclose(mu[:64], y.mean() + s * mu_s, rtol=1e-9, atol=1e-11)
    assert np.allclose(var, (s * std_all) ** 2, rtol=1e-7, atol=1e-12)
    _, cov = m.posterior_joint(X[:64])
    assert np.allclose(cov, s * s * cov_s, rtol=1e-7, atol=1e-11)
    lml = gpr.log_marginal_likelihood(gpr.kernel_.theta)
    assert math.isclose(go.data_term(spec, p, m.Xn, m.ystd).value, lml, rel_tol=1e-10)


def test_mc_qlogei_converges_to_the_closed_form_log_ei():
    """Structure pin of the MC acquisition (sign convention, best_f, sample construction): for q = 1 the
    fat-softplus MC estimate over Sobol-normal base samples must approach the closed-form log EI computed
    with scipy (quadrature-free formula sigma * h((mu - f*)/sigma)); QMC error at S = 4096 is ~1e-3 in log
    space as long as EI is not deep in the tail."""
    from scipy.stats import norm

    rng = np.random.default_rng(1)
    mu, var = rng.normal(0.0, 1.0, 200), rng.uniform(0.05, 1.5, 200)
    best_f = 0.3
    z = go.sobol_normal_base_samples(4096, 1, seed=1234)[:, 0]
    for sign in (1.0, -1.0):
        mc = go.qlogei_q1(mu, var, z, best_f * sign if sign > 0 else -best_f, sign)
        sd = np.sqrt(var)
        u = (sign * mu - (best_f if sign > 0 else -best_f)) / sd
        exact = np.log(sd * (norm.pdf(u) + u * norm.cdf(u)))
        keep = exact > -6.0  # beyond that the 4096-point rule has no samples in the improvement region
        assert keep.sum() > 120
        assert np.max(np.abs(mc[keep] - exact[keep])) < 2e-2
        assert np.allclose(go.analytic_acq("LogEI", mu, var, best_f if sign > 0 else -best_f, sign)[keep], exact[keep], atol=1e-9)
