"""The register-resident form of the fused posterior kernel for small models (``csrc/bbh_small.h``: n <= 128, the whole model in a
wave's registers / LDS, persistent waves over 16-candidate tiles) against the oracle's exact Cholesky posterior
(``oracle/gp_oracle.py::GPModel.posterior``: what ``model.posterior(X)`` is for BayBE's GP, surrogates/gaussian_process/core.py:268-269)
and against the cooperative form it replaces for these sizes (``BBH_SMALL=0``)."""

import math
import os

import numpy as np
import pytest

from _problems import fixed_theta, make_problem, make_tl_problem, oracle_params, oracle_spec

pytestmark = pytest.mark.gpu

MEAN_RTOL, VAR_RTOL = 1e-9, 1e-8


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("N,d,n", [(10_000, 3, 20), (100_000, 6, 33), (100_000, 10, 64), (777, 2, 5), (10_000, 14, 48), (5, 4, 16),
                                   (1, 3, 17), (40_003, 22, 64), (3_000, 30, 31), (2_000, 6, 1), (50_000, 15, 128), (20_000, 9, 65), (20_000, 26, 100),
                                   (9_000, 5, 113)])
@pytest.mark.parametrize("kernel", ["matern52", "rbf", "matern32"])
def test_small_form_matches_the_oracle_and_the_cooperative_form(N, d, n, kernel, monkeypatch):
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    if n > 64 and kernel == "matern32":
        pytest.skip("64 < n <= 128 is instantiated for Matern-5/2 and RBF")
    monkeypatch.setenv("BBH_SMALL_FORCE", "1")  # (beyond n = 64 the form is chosen from a candidate count on; here always)

    X, Xt, y = make_problem(max(N, n + 1), d, n, seed=4)
    X = X[:N]
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), kernel=kernel)
    ls, nz, _ = fixed_theta(d)
    p = gp_spec.GPParams(np.full(d, ls) * (0.8 + 0.4 * np.random.default_rng(7).random(d)), nz, 0.05)
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    g.factorize(p)
    m, v = g.posterior(X)
    assert g.posterior_kernel_form() == "register-resident"
    om = go.GPModel(oracle_spec(spec), oracle_params(spec, p), Xt, y)
    mo, vo = om.posterior(X)
    assert np.allclose(_np(m), mo, rtol=MEAN_RTOL, atol=1e-12)
    assert np.allclose(_np(v), vo, rtol=VAR_RTOL, atol=1e-14)
    g.close()
    os.environ["BBH_SMALL"] = "0"
    try:
        c = engine.HipGP(0)
        c.set_model(spec, Xt, y)
        c.factorize(p)
        mc, vc = c.posterior(X)
        assert c.posterior_kernel_form() != "register-resident"
        assert np.allclose(_np(m), _np(mc), rtol=1e-11, atol=1e-13) and np.allclose(_np(v), _np(vc), rtol=1e-9, atol=1e-15)
        c.close()
    finally:
        del os.environ["BBH_SMALL"]


@pytest.mark.parametrize("variant", ["outputscale", "tasks", "tasks_rbf"])
def test_small_form_with_the_scale_and_task_table(variant):
    """``ScaleKernel`` outputscale and the ICM task covariance (PositiveIndexKernel, kernels/basic.py:239-248) enter through the
    per-(candidate task, training task) table; candidates of any task."""
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    d = 5
    if variant == "outputscale":
        X, Xt, y = make_problem(20_000, d, 50, seed=5)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        spec.use_outputscale = True
        p = gp_spec.initial_params(spec)
        p.outputscale = 1.7
        p.lengthscale = p.lengthscale * np.linspace(0.7, 1.4, d)
    else:
        X, Xt, y = make_tl_problem(20_000, d, 20, T=3, seed=6)
        X[:, d] = np.random.default_rng(0).integers(0, 3, len(X))  # candidates of every task
        spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=3,
                                            kernel="rbf" if variant == "tasks_rbf" else "matern52")
        p = gp_spec.initial_params(spec)
        p.task_W = p.task_W * np.array([[1.0, 0.6, 0.3], [0.5, 1.1, 0.2], [0.2, 0.4, 0.9]])
        p.lengthscale = p.lengthscale * np.linspace(0.8, 1.3, d)
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    g.factorize(p)
    m, v = g.posterior(X)
    assert g.posterior_kernel_form() == "register-resident"
    mo, vo = go.GPModel(oracle_spec(spec), oracle_params(spec, p), Xt, y).posterior(X)
    assert np.allclose(_np(m), mo, rtol=MEAN_RTOL, atol=1e-12) and np.allclose(_np(v), vo, rtol=VAR_RTOL, atol=1e-14)
    g.close()


@pytest.mark.parametrize("variant", ["positive_rank2", "free_rank1"])
def test_user_supplied_task_kernels_on_the_device(variant):
    """``ProductKernel([numerical kernel, (Positive)IndexKernel(num_tasks, rank < T)])`` (kernels/basic.py:220-248; what
    ``ICMKernelFactory`` with a custom task kernel returns, components/kernel.py:238-337): data term and its gradient through the
    host's chain rules against the oracle's autograd objective, the posterior against the oracle's, the whole device fit against
    the oracle's objective at the fit's end point."""
    import torch

    from baybe_amd import engine, gp_spec
    from baybe_amd.kernels import GammaPrior, IndexKernel, MaternKernel, PositiveIndexKernel, ProductKernel, RBFKernel, ScaleKernel, apply_kernel_spec
    from oracle import gp_oracle as go

    d, T = 5, 3
    X, Xt, y = make_tl_problem(20_000, d, 22, T=T, seed=16)
    X[:, d] = np.random.default_rng(0).integers(0, T, len(X))

    class Space:
        comp_rep_columns = tuple(f"x{j}" for j in range(d)) + ("task",)
        n_tasks, task_idx = T, d

    spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=T)
    if variant == "positive_rank2":
        kern = ProductKernel([ScaleKernel(MaternKernel(2.5, GammaPrior(3, 1)), GammaPrior(2, 0.5)), PositiveIndexKernel(num_tasks=T, rank=2)])
    else:
        kern = ProductKernel([RBFKernel(GammaPrior(3, 1)), IndexKernel(num_tasks=T, rank=1)])
    apply_kernel_spec(spec, kern, Space())
    ospec = oracle_spec(spec)
    torch.manual_seed(5)
    p = gp_spec.initial_params(spec)
    raw = gp_spec.pack_raw(spec, p) + 0.15 * np.random.default_rng(3).standard_normal(len(gp_spec.pack_raw(spec, p)))
    raw[0] = abs(raw[0]) + 2e-4
    p = gp_spec.unpack_raw(spec, raw)
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    val, gth = g.data_term(p)
    f1, g1 = gp_spec.objective_from_data_term(spec, raw, len(y), val, gth, params=p)
    Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
    f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
    assert math.isclose(f0, f1, rel_tol=1e-10) and np.allclose(g0, g1, rtol=1e-7, atol=1e-9 * np.abs(g0).max())
    g.factorize(p)
    m, v = g.posterior(X)
    mo, vo = go.GPModel(ospec, oracle_params(spec, p), Xt, y).posterior(X)
    assert np.allclose(_np(m), mo, rtol=MEAN_RTOL, atol=1e-12) and np.allclose(_np(v), vo, rtol=VAR_RTOL, atol=1e-14)
    torch.manual_seed(5)
    fi = g.fit()
    f_at, _ = go.fit_objective(ospec, go.pack_raw(ospec, oracle_params(spec, fi.params)), Xn, ys)
    assert math.isclose(f_at, fi.fun, rel_tol=1e-8, abs_tol=1e-10) and fi.fun < f0
    g.close()


def test_greedy_batch_on_a_small_model_equals_the_oracle():
    """The first pass of every selection step runs on the register-resident form, the cross-covariance passes of the later steps on
    the cooperative one: the batch equals the oracle's."""
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    X, Xt, y = make_problem(30_000, 4, 27, seed=8)
    spec = gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    fi = g.fit()
    om = go.GPModel(oracle_spec(spec), oracle_params(spec, fi.params), Xt, y)
    for sign in (1.0, -1.0):
        got = g.greedy_qlogei(X, 4, seed=21, sign=sign)
        ref = go.optimize_acqf_discrete_qlogei(om, X, 4, seed=21, sign=sign)
        assert got.indices == ref.indices and np.allclose(got.values, ref.values, rtol=0, atol=1e-8)
    g.close()
