"""An EXTERNAL pin for the GP core of the oracle: scikit-learn's ``GaussianProcessRegressor`` (an implementation written by
other people, importable here, unlike gpytorch / botorch) evaluates the same stationary ARD kernels, the same log marginal
likelihood with its gradient, and the same exact-Cholesky posterior.  The oracle (``oracle/gp_oracle.py``) must agree with
it to rounding on all three - kernel formulas (Matern-1/2, -3/2, -5/2, RBF, rational quadratic, products and sums with
scales; linear, polynomial and periodic kernels), likelihood value and gradient, posterior mean / variance / joint covariance.

What this does NOT pin: everything BoTorch-specific (priors and their constants, constraint transforms, the LOO criterion,
fat-tailed qLogEI / qLogNEHVI, Sobol base samples, greedy semantics, the index kernel) - those parts of the oracle remain
"parity unpinned" (DESIGN.md §2)."""

import math

import numpy as np
import pytest

sk = pytest.importorskip("sklearn.gaussian_process")
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, RationalQuadratic, WhiteKernel  # noqa: E402

from oracle import gp_oracle as go  # noqa: E402


def _problem(n=40, N=300, d=4, seed=0):
    rng = np.random.default_rng(seed)
    Xt, X = rng.random((n, d)), rng.random((N, d))
    y = np.sin(3 * Xt[:, 0]) + Xt[:, 1] ** 2 - 0.5 * Xt[:, 2] + 0.05 * rng.standard_normal(n)
    return Xt, X, y


def _single(kernel, d):
    return go.GPSpec(d=d, num_idx=np.arange(d), lo=np.zeros(d), hi=np.ones(d), kernel=kernel, use_outputscale=True,
                     lengthscale=go.Hyper(0.0, True, None, 0.5), noise=go.Hyper(1e-4, True, None, 0.05),
                     outputscale=go.Hyper(0.0, True, None, 1.0))


CASES = {
    "matern12": lambda ls: Matern(length_scale=ls, nu=0.5),
    "matern32": lambda ls: Matern(length_scale=ls, nu=1.5),
    "matern52": lambda ls: Matern(length_scale=ls, nu=2.5),
    "rbf": lambda ls: RBF(length_scale=ls),
}


@pytest.mark.parametrize("kernel", sorted(CASES))
def test_kernel_likelihood_gradient_and_posterior_against_scikit_learn(kernel):
    d = 4
    Xt, X, y = _problem(d=d)
    rng = np.random.default_rng(1)
    ls = 0.3 + rng.random(d)
    os_, noise, mean = 1.7, 0.03, 0.2
    spec = _single(kernel, d)
    p = go.GPParams(lengthscale=ls, noise=noise, mean=mean, outputscale=os_)
    Xn = go.normalize_inputs(spec, Xt)  # identity here (bounds [0, 1])
    ystd, ybar, ysd = go.standardize_targets(y)

    # --- kernel values
    k_sk = ConstantKernel(os_) * CASES[kernel](ls)
    assert np.allclose(go.cross_cov(spec, p, go.normalize_inputs(spec, X), Xn), k_sk(X, Xt), rtol=1e-12, atol=1e-14)

    # --- log marginal likelihood and its gradient (scikit-learn differentiates w.r.t. log-parameters)
    full = k_sk + WhiteKernel(noise)
    gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None, normalize_y=False).fit(Xn, ystd - mean)
    lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
    dt = go.data_term(spec, p, Xn, ystd)
    assert math.isclose(dt.value, lml, rel_tol=1e-10)
    # theta order of the sklearn kernel: [log constant, log length scales..., log noise]
    ours = np.concatenate([[dt.g_outputscale * os_], dt.g_ls * ls, [dt.g_noise * noise]])
    assert np.allclose(ours, grad_log, rtol=1e-7, atol=1e-9 * np.abs(grad_log).max())
    # the constant mean: d/dc of the likelihood = sum(alpha); check against a central difference of sklearn's value
    e = 1e-5
    up = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xn, ystd - (mean + e)).log_marginal_likelihood(full.theta)
    dn = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xn, ystd - (mean - e)).log_marginal_likelihood(full.theta)
    assert math.isclose(dt.g_mean, (up - dn) / (2 * e), rel_tol=1e-5, abs_tol=1e-6)

    # --- posterior of the latent function (sklearn's predict includes the WhiteKernel in the prior variance: subtract it)
    model = go.GPModel(spec, p, Xt, y)
    mu, var = model.posterior(X)
    m_sk, s_sk = gpr.predict(X, return_std=True)
    assert np.allclose(mu, ybar + ysd * (mean + m_sk), rtol=1e-9, atol=1e-11)
    assert np.allclose(var, ysd**2 * (s_sk**2 - noise), rtol=1e-6, atol=1e-10)
    mj, cj = model.posterior_joint(X[:6])
    _, c_sk = gpr.predict(X[:6], return_cov=True)
    assert np.allclose(cj, ysd**2 * (c_sk - noise * np.eye(6)), rtol=1e-6, atol=1e-10)


def test_rational_quadratic_products_and_sums_against_scikit_learn():
    """RationalQuadratic is isotropic in scikit-learn: equal lengthscales.  Products / sums of scaled kernels as BayBE's
    ProductKernel / AdditiveKernel build them."""
    d = 3
    Xt, X, y = _problem(d=d, seed=3)
    ystd = go.standardize_targets(y)[0]
    l1, alpha, l2 = 0.8, 1.3, np.array([0.4, 0.9, 0.6])
    members = [go.KernelTerm("rq", go.Hyper(), go.Hyper()), go.KernelTerm("matern52", go.Hyper(), None)]
    for composition, combine in (("product", lambda a, b: a * b), ("sum", lambda a, b: a + b)):
        spec = go.GPSpec(d=d, num_idx=np.arange(d), lo=np.zeros(d), hi=np.ones(d), members=members, composition=composition,
                         noise=go.Hyper(1e-4, True, None, 0.05))
        p = go.GPParams(lengthscale=np.full(d, l1), noise=0.04, mean=0.0, member_ls=[np.full(d, l1), l2],
                        member_scale=np.array([0.7, 1.0]), rq_alpha=np.array([alpha, 1.0]))
        k_sk = combine(ConstantKernel(0.7) * RationalQuadratic(length_scale=l1, alpha=alpha), Matern(length_scale=l2, nu=2.5))
        assert np.allclose(go.cross_cov(spec, p, X, Xt), k_sk(X, Xt), rtol=1e-12, atol=1e-14)
        full = k_sk + WhiteKernel(0.04)
        gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xt, ystd)
        lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
        dt = go.data_term(spec, p, Xt, ystd)
        assert math.isclose(dt.value, lml, rel_tol=1e-10)
        # sklearn theta: [log 0.7, log alpha?...]: order follows the kernel tree: constant, (rq: alpha, length_scale), matern ls.., noise
        names = [h.name for h in full.hyperparameters]
        got = {}
        got["k1__k1__constant_value"] = dt.g_member_scale[0] * 0.7
        got["k1__k2__length_scale"] = float(dt.g_member_ls[0].sum()) * l1  # isotropic: the ARD slots add up
        got["k1__k2__alpha"] = dt.g_alpha[0] * alpha
        ref = dict(zip(names, np.split(grad_log, np.cumsum([h.n_elements for h in full.hyperparameters])[:-1])))
        for key, val in ref.items():
            if key.endswith("k1__k1__constant_value"):
                assert np.allclose(val, got["k1__k1__constant_value"], rtol=1e-7)
            elif key.endswith("k1__k2__length_scale") and val.size == 1:
                assert np.allclose(val, got["k1__k2__length_scale"], rtol=1e-7)
            elif key.endswith("alpha"):
                assert np.allclose(val, got["k1__k2__alpha"], rtol=1e-7)
            elif key.endswith("length_scale"):
                assert np.allclose(val, dt.g_member_ls[1] * l2, rtol=1e-7, atol=1e-10)
            elif key.endswith("noise_level"):
                assert np.allclose(val, dt.g_noise * 0.04, rtol=1e-7)
            else:
                raise AssertionError(f"unexpected scikit-learn hyper-parameter {key}")


def _autograd_lml_and_log_gradient(spec, p, Xn, ystd):
    """Log marginal likelihood and its gradient w.r.t. the LOG of every positive hyper-parameter from the oracle's autograd
    objective (oracle/fit_objective.py; priors must be absent): the quantities scikit-learn reports.  Keyed by the oracle's
    parameter names."""
    import torch

    from oracle import fit_objective as fo

    raw = go.pack_raw(spec, p)
    f, g = go.fit_objective(spec, raw, Xn, ystd)
    n = len(ystd)
    out, i = {}, 0
    nat = fo.split_raw(spec, torch.as_tensor(raw))
    for prm in fo.parameter_layout(spec):
        sl = slice(i, i + prm.size)
        i += prm.size
        grad_raw = -n * g[sl]  # the objective is -(log-likelihood) / n
        val = nat[prm.key].detach().numpy().reshape(-1)
        if prm.lower is None:  # the mean constant
            out[prm.key] = grad_raw
        else:  # value = lower + softplus(raw): d/dlog(value) = d/draw * value / sigmoid(raw)
            sig = 1.0 / (1.0 + np.exp(-raw[sl]))
            out[prm.key] = grad_raw / sig * val if prm.transformed else grad_raw * val
    return -n * f, out


def test_linear_polynomial_and_periodic_kernels_against_scikit_learn():
    """The oracle's restatements of gpytorch's LinearKernel / PolynomialKernel / PeriodicKernel (baybe/kernels/basic.py:20-46,
    135-163, 73-112) against scikit-learn's DotProduct, DotProduct ** p and ExpSineSquared: kernel matrices, log marginal
    likelihood, its gradient (the oracle's is autograd), posterior.  Correspondences: Linear with equal variances v = ConstantKernel(v)
    * DotProduct(sigma_0 -> 0); Polynomial(p) with offset c = DotProduct(sigma_0 = sqrt(c)) ** p; Periodic in one dimension with
    gpytorch's lengthscale l (it divides by l, not l^2) = ExpSineSquared(length_scale = sqrt(l), periodicity)."""
    from sklearn.gaussian_process.kernels import DotProduct, ExpSineSquared, Exponentiation

    d = 3
    Xt, X, y = _problem(n=30, N=50, d=d, seed=11)
    ystd, ybar, ysd = go.standardize_targets(y)
    noise, mean = 0.06, 0.1

    def spec_of(kernel, dd=d, **kw):
        return go.GPSpec(d=dd, num_idx=np.arange(dd), lo=np.zeros(dd), hi=np.ones(dd), kernel=kernel, lengthscale=go.Hyper(0.0, True, None, 0.5),
                         noise=go.Hyper(1e-4, True, None, 0.05), outputscale=go.Hyper(0.0, True, None, 1.0), **kw)

    # ---- linear: k = v x . x'
    v = 0.8
    spec = spec_of("linear")
    p = go.GPParams(lengthscale=np.full(d, v), noise=noise, mean=mean)
    k_sk = ConstantKernel(v) * DotProduct(sigma_0=1e-7, sigma_0_bounds="fixed")
    assert np.allclose(go.cross_cov(spec, p, X, Xt), k_sk(X, Xt), rtol=1e-10, atol=1e-12)
    full = k_sk + WhiteKernel(noise)
    gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xt, ystd - mean)
    lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
    ours, g = _autograd_lml_and_log_gradient(spec, p, Xt, ystd)
    assert math.isclose(ours, lml, rel_tol=1e-9)
    # sklearn theta: [log v, log noise]; the ARD variances of the oracle add up to the single variance here
    assert np.allclose([g["variance"].sum(), g["noise"].sum()], grad_log, rtol=1e-6, atol=1e-8 * np.abs(grad_log).max())
    mu, var = go.GPModel(spec, p, Xt, y).posterior(X)
    m_sk, s_sk = gpr.predict(X, return_std=True)
    assert np.allclose(mu, ybar + ysd * (mean + m_sk), rtol=1e-8, atol=1e-10) and np.allclose(var, ysd**2 * (s_sk**2 - noise), rtol=1e-5, atol=1e-9)
    assert np.ptp(go.prior_var(spec, p, X)) > 0 and np.allclose(go.prior_var(spec, p, X), v * (X * X).sum(1), rtol=1e-12)

    # ---- polynomial: k = os (x . x' + c)^p
    for power in (1, 2, 3):
        c, os_ = 0.7, 1.4
        spec = spec_of(f"poly{power}", use_outputscale=True)
        p = go.GPParams(lengthscale=np.ones(d), noise=noise, mean=mean, outputscale=os_, rq_alpha=np.array([c]))
        k_sk = ConstantKernel(os_) * Exponentiation(DotProduct(sigma_0=math.sqrt(c)), power)
        assert np.allclose(go.cross_cov(spec, p, X, Xt), k_sk(X, Xt), rtol=1e-12, atol=1e-14)
        full = k_sk + WhiteKernel(noise)
        gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xt, ystd - mean)
        lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
        ours, g = _autograd_lml_and_log_gradient(spec, p, Xt, ystd)
        assert math.isclose(ours, lml, rel_tol=1e-9)
        # sklearn theta: [log os, log sigma_0, log noise]; c = sigma_0^2: d/dlog(sigma_0) = 2 d/dlog(c)
        assert np.allclose([g["outputscale"].sum(), 2.0 * g["offset"].sum(), g["noise"].sum()], grad_log, rtol=1e-6,
                           atol=1e-8 * np.abs(grad_log).max()), (power, g, grad_log)

    # ---- periodic, one input dimension: k = os exp(-2 sin^2(pi |x - x'| / period) / l)
    Xt1, X1 = Xt[:, :1], X[:, :1]
    l, per, os_ = 0.6, 0.45, 1.2
    spec = spec_of("periodic", dd=1, use_outputscale=True)
    p = go.GPParams(lengthscale=np.array([l]), noise=noise, mean=mean, outputscale=os_, period=[np.array([per])])
    k_sk = ConstantKernel(os_) * ExpSineSquared(length_scale=math.sqrt(l), periodicity=per)
    assert np.allclose(go.cross_cov(spec, p, X1, Xt1), k_sk(X1, Xt1), rtol=1e-12, atol=1e-14)
    full = k_sk + WhiteKernel(noise)
    gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xt1, ystd - mean)
    lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
    ours, g = _autograd_lml_and_log_gradient(spec, p, Xt1, ystd)
    assert math.isclose(ours, lml, rel_tol=1e-9)
    # sklearn theta: [log os, log length_scale, log periodicity, log noise]; l = length_scale^2: d/dlog(length_scale) = 2 d/dlog(l)
    assert np.allclose([g["outputscale"].sum(), 2.0 * g["lengthscale"].sum(), g["period_length"].sum(), g["noise"].sum()], grad_log,
                       rtol=1e-6, atol=1e-8 * np.abs(grad_log).max()), (g, grad_log)
    mu, var = go.GPModel(spec, p, Xt1, y).posterior(X1)
    m_sk, s_sk = gpr.predict(X1, return_std=True)
    assert np.allclose(mu, ybar + ysd * (mean + m_sk), rtol=1e-8, atol=1e-10) and np.allclose(var, ysd**2 * (s_sk**2 - noise), rtol=1e-5, atol=1e-9)


def test_leave_one_out_criterion_against_brute_force_refits_with_scikit_learn():
    """The LOO pseudo-likelihood of the transfer-learning presets (presets/baybe.py:269-281): sum_i log N(y_i | mu_-i,
    s2_-i) with the moments of a GP conditioned on all other points at the same hyper-parameters.  The oracle gets them
    from one inverse; here every point is really left out and scikit-learn does the conditioning."""
    d, n = 3, 18
    Xt, _, y = _problem(n=n, d=d, seed=7)
    ystd = go.standardize_targets(y)[0]
    ls, os_, noise, mean = np.array([0.5, 0.9, 0.7]), 1.3, 0.05, -0.1
    spec = _single("matern52", d)
    spec.criterion = "loo"
    p = go.GPParams(lengthscale=ls, noise=noise, mean=mean, outputscale=os_)
    kern = ConstantKernel(os_) * Matern(length_scale=ls, nu=2.5) + WhiteKernel(noise)
    total = 0.0
    for i in range(n):
        keep = np.arange(n) != i
        gpr = sk.GaussianProcessRegressor(kernel=kern, alpha=0.0, optimizer=None).fit(Xt[keep], ystd[keep] - mean)
        m, s = gpr.predict(Xt[i : i + 1], return_std=True)  # predictive of the OBSERVATION: the WhiteKernel is part of the prior
        total += -0.5 * math.log(2 * math.pi * s[0] ** 2) - 0.5 * (ystd[i] - mean - m[0]) ** 2 / s[0] ** 2
    dt = go.data_term(spec, p, Xt, ystd)
    assert math.isclose(dt.value, total, rel_tol=1e-9)
    # and its gradient by central differences of that brute-force value would cost n fits per parameter; the analytic
    # gradient is covered against autograd (tests/test_host_logic_cpu.py) - here only the lengthscale of dimension 0
    def brute(l0):
        k2 = ConstantKernel(os_) * Matern(length_scale=np.array([l0, ls[1], ls[2]]), nu=2.5) + WhiteKernel(noise)
        t = 0.0
        for i in range(n):
            keep = np.arange(n) != i
            g = sk.GaussianProcessRegressor(kernel=k2, alpha=0.0, optimizer=None).fit(Xt[keep], ystd[keep] - mean)
            m, s = g.predict(Xt[i : i + 1], return_std=True)
            t += -0.5 * math.log(2 * math.pi * s[0] ** 2) - 0.5 * (ystd[i] - mean - m[0]) ** 2 / s[0] ** 2
        return t
    e = 1e-5
    assert math.isclose(dt.g_ls[0], (brute(ls[0] + e) - brute(ls[0] - e)) / (2 * e), rel_tol=1e-5, abs_tol=1e-6)


@pytest.mark.parametrize("q", [0, 1, 2, 3])
def test_piecewise_polynomial_kernel_is_the_textbook_one_and_positive_definite(q):
    """No library here implements gpytorch's PiecewisePolynomialKernel; its definition is Rasmussen & Williams (2006), eq.
    (4.21): k_pp,q(r) = (1 - r)_+^(j + q) P_q(r), j = floor(D / 2) + q + 1, with the polynomials written out below.  Two pins:
    the oracle equals those formulas, and the Gram matrix of random points in D dimensions is positive semi-definite - which
    the family guarantees for exactly this exponent j (a smaller one loses it)."""
    D, n = 5, 120
    rng = np.random.default_rng(q)
    X = rng.random((n, D))
    ls = 0.8 + rng.random(D)
    r = np.sqrt(go._scaled_sqdist(X, X, ls))
    j = D // 2 + q + 1
    textbook = {
        0: lambda r: np.maximum(0, 1 - r) ** j,
        1: lambda r: np.maximum(0, 1 - r) ** (j + 1) * ((j + 1) * r + 1),
        2: lambda r: np.maximum(0, 1 - r) ** (j + 2) * ((j * j + 4 * j + 3) * r**2 + (3 * j + 6) * r + 3) / 3,
        3: lambda r: np.maximum(0, 1 - r) ** (j + 3)
        * ((j**3 + 9 * j * j + 23 * j + 15) * r**3 + (6 * j * j + 36 * j + 45) * r**2 + (15 * j + 45) * r + 15) / 15,
    }[q](r)
    K = go.base_kernel_from_r2(f"piecewise{q}", r * r, D)
    assert np.allclose(K, textbook, rtol=1e-12, atol=1e-15)
    assert 0.05 < (K > 0).mean() < 0.99  # compact support is actually exercised
    assert np.linalg.eigvalsh(K).min() > -1e-10
    # with too small an exponent the same construction stops being a valid kernel in D dimensions
    bad = np.maximum(0, 1 - r) ** 1 if q == 0 else None
    if bad is not None:
        assert np.linalg.eigvalsh(bad).min() < -1e-6


def test_a_sum_with_product_members_against_scikit_learn():
    """AdditiveKernel([ProductKernel([a, b]), c, d]) - the nested entry of the reference's kernel matrix,
    ``(Matern * Matern) + (Matern + Matern)`` (tests/test_iterations.py:294-296; kernels/composite.py:60-91) - as the oracle composes it
    (``composition="nested"``, ``member_terms``) against scikit-learn's own Sum / Product kernel algebra: cross-covariance, log marginal
    likelihood and its gradient with respect to every lengthscale, the member scale and the noise."""
    d = 3
    Xt, X, y = _problem(d=d, seed=5)
    ystd = go.standardize_targets(y)[0]
    la, lb, lc, ld = np.array([0.5, 0.8, 0.7]), np.array([0.9, 0.6, 1.1]), np.array([0.4, 0.9, 0.6]), np.array([1.2, 0.7, 0.8])
    members = [go.KernelTerm("matern52", go.Hyper(), None), go.KernelTerm("rbf", go.Hyper(), None),
               go.KernelTerm("matern32", go.Hyper(), go.Hyper()), go.KernelTerm("rbf", go.Hyper(), None)]
    spec = go.GPSpec(d=d, num_idx=np.arange(d), lo=np.zeros(d), hi=np.ones(d), members=members, composition="nested",
                     member_terms=[0, 0, 1, 2], noise=go.Hyper(1e-4, True, None, 0.05))
    p = go.GPParams(lengthscale=la, noise=0.03, mean=0.0, member_ls=[la, lb, lc, ld], member_scale=np.array([1.0, 1.0, 0.7, 1.0]))
    k_sk = Matern(length_scale=la, nu=2.5) * RBF(length_scale=lb) + ConstantKernel(0.7) * Matern(length_scale=lc, nu=1.5) + RBF(length_scale=ld)
    assert np.allclose(go.cross_cov(spec, p, X, Xt), k_sk(X, Xt), rtol=1e-12, atol=1e-14)
    assert np.allclose(go.prior_var(spec, p, X), np.diag(k_sk(X)), rtol=1e-12)
    full = k_sk + WhiteKernel(0.03)
    gpr = sk.GaussianProcessRegressor(kernel=full, alpha=0.0, optimizer=None).fit(Xt, ystd)
    lml, grad_log = gpr.log_marginal_likelihood(full.theta, eval_gradient=True)
    dt = go.data_term(spec, p, Xt, ystd)
    assert math.isclose(dt.value, lml, rel_tol=1e-10)
    names = [h.name for h in full.hyperparameters]
    ref = dict(zip(names, np.split(grad_log, np.cumsum([h.n_elements for h in full.hyperparameters])[:-1])))
    # scikit-learn differentiates with respect to log(theta): d/dlog(l) = l d/dl
    want = {"k1__k1__k1__k1__length_scale": dt.g_member_ls[0] * la, "k1__k1__k1__k2__length_scale": dt.g_member_ls[1] * lb,
            "k1__k1__k2__k1__constant_value": np.array([dt.g_member_scale[2] * 0.7]), "k1__k1__k2__k2__length_scale": dt.g_member_ls[2] * lc,
            "k1__k2__length_scale": dt.g_member_ls[3] * ld, "k2__noise_level": np.array([dt.g_noise * 0.03])}
    assert set(ref) == set(want), sorted(ref)
    for key, val in ref.items():
        assert np.allclose(val, want[key], rtol=1e-7, atol=1e-10), key
    # and the torch / autograd objective of the fit sees the same function
    raw = go.pack_raw(spec, p)
    f0, g0 = go.fit_objective(spec, raw, Xt, ystd)
    eps, gnum = 1e-6, np.zeros_like(raw)
    for i in range(len(raw)):
        e = np.zeros_like(raw)
        e[i] = eps
        gnum[i] = (go.fit_objective(spec, raw + e, Xt, ystd)[0] - go.fit_objective(spec, raw - e, Xt, ystd)[0]) / (2 * eps)
    assert np.allclose(g0, gnum, rtol=1e-5, atol=1e-8)
