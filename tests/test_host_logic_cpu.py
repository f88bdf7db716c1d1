"""Host-side fit bookkeeping (baybe_amd/gp_spec.py) against the oracle, using the oracle's data
term in place of the device call (checker role only)."""

import math

import numpy as np
import pytest

from _problems import make_problem, make_tl_problem, oracle_spec
from baybe_amd import gp_spec
from oracle import gp_oracle as go


def _ospec(spec):
    return oracle_spec(spec)


@pytest.mark.parametrize("tl", [False, True])
def test_pack_unpack_and_objective_match_oracle(tl):
    if tl:
        X, Xt, y = make_tl_problem(200, 4, 15, T=3, seed=1)
        spec = gp_spec.GPSpec.baybe_default(5, np.zeros(5), np.ones(5), task_idx=4, n_tasks=3)
        spec.use_outputscale = True
        spec.ls_constraint = "softplus"
        spec.outputscale_prior = ("gamma", 2.0, 0.15)
    else:
        X, Xt, y = make_problem(200, 4, 30, seed=1)
        spec = gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
    ospec = _ospec(spec)
    p = gp_spec.initial_params(spec)
    p.lengthscale = p.lengthscale * np.linspace(0.8, 1.3, spec.dn)
    raw = gp_spec.pack_raw(spec, p)
    p2 = gp_spec.unpack_raw(spec, raw)
    assert np.allclose(p2.lengthscale, p.lengthscale) and np.isclose(p2.noise, p.noise)
    assert gp_spec.raw_bounds(spec) == go.raw_bounds(ospec)
    op = go.unpack_raw(ospec, raw)
    Xn = go.normalize_inputs(ospec, Xt)
    ystd, _, _ = go.standardize_targets(y)
    dt = go.data_term(ospec, op, Xn, ystd)
    grad_theta = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls] + ([dt.g_task_B.reshape(-1)] if tl else []))
    f, g = gp_spec.objective_from_data_term(spec, raw, len(y), dt.value, grad_theta)
    fo, go_ = go.fit_objective(ospec, raw, Xn, ystd)
    assert np.isclose(f, fo, rtol=1e-13) and np.allclose(g, go_, rtol=1e-12, atol=1e-14)
    theta = gp_spec.theta_from_params(spec, p2)
    assert theta.shape[0] == 3 + spec.dn + (9 if tl else 0)
    if tl:
        assert np.allclose(theta[3 + spec.dn:].reshape(3, 3), p2.task_B())


def test_default_spec_matches_reference_preset():
    import math

    spec = gp_spec.GPSpec.baybe_default(15, np.zeros(15), np.ones(15))
    assert spec.kernel == "matern52" and not spec.use_outputscale and spec.ls_constraint == "box"
    assert math.isclose(spec.ls_lower, 2.5e-2) and math.isclose(spec.noise_lower, 1e-4)
    assert math.isclose(spec.ls_init, math.exp(math.sqrt(2) - 3) * math.sqrt(15))


def test_prior_samples_respect_constraints():
    rng = np.random.default_rng(0)
    spec = gp_spec.GPSpec.baybe_default(6, np.zeros(6), np.ones(6))
    for _ in range(50):
        p = gp_spec.sample_params_from_priors(spec, rng)
        assert (p.lengthscale >= spec.ls_lower).all() and p.noise >= spec.noise_lower
        raw = gp_spec.pack_raw(spec, p)
        for v, (lo, hi) in zip(raw, gp_spec.raw_bounds(spec)):
            assert lo is None or v >= lo


# ---- GP presets as data ---------------------------------------------------------
def test_preset_tables_follow_the_reference():
    """Spot values of presets/edbo.py:75-119,147-172, edbo_smoothed.py:60-72, chen.py:43-60 and the
    dimension-scaled BoTorch defaults used by presets/hvarfner.py / botorch.py."""
    import math

    from baybe_amd import gp_spec

    lo, hi = np.zeros(30), np.ones(30)
    e3 = gp_spec.from_preset("EDBO", 3, lo[:3], hi[:3])
    assert e3.use_outputscale and e3.ls_constraint == "softplus" and e3.noise_constraint == "softplus"
    assert e3.ls_prior == ("gamma", 1.2, 1.1) and e3.ls_init == 0.2
    assert e3.outputscale_prior == ("gamma", 5.0, 0.5) and e3.outputscale_init == 8.0
    assert e3.noise_prior == ("gamma", 1.05, 0.5) and e3.noise_init == 0.1
    e30 = gp_spec.from_preset("EDBO", 30, lo, hi)
    assert e30.ls_prior == ("gamma", 3.0, 1.0) and e30.outputscale_init == 20.0 and e30.noise_init == 5.0
    big = gp_spec.from_preset("edbo", 60, np.zeros(60), np.ones(60), edbo_encodings=True)
    assert big.ls_prior == ("gamma", 2.0, 0.2) and big.ls_init == 5.0
    huge = gp_spec.from_preset("EDBO", 120, np.zeros(120), np.ones(120), edbo_encodings=True)
    assert huge.ls_prior == ("gamma", 2.0, 0.1) and huge.outputscale_prior == ("gamma", 2.0, 0.1)
    sm = gp_spec.from_preset("EDBO_SMOOTHED", 8, lo[:8], hi[:8])
    assert sm.ls_prior == ("gamma", 1.2, 1.1) and sm.noise_init == pytest.approx(0.1)
    sm2 = gp_spec.from_preset("EDBO_SMOOTHED", 200, np.zeros(200), np.ones(200))
    assert sm2.ls_prior == ("gamma", 2.5, 0.55) and sm2.outputscale_init == 15.0
    mid = gp_spec.from_preset("EDBO_SMOOTHED", 41, np.zeros(41), np.ones(41))
    assert 1.2 < mid.ls_prior[1] < 2.5 and 0.2 < mid.ls_init < 6.0
    ch = gp_spec.from_preset("CHEN", 16, lo[:16], hi[:16])
    assert ch.ls_init == pytest.approx(5.6) and ch.ls_prior == ("gamma", pytest.approx(11.2), 2.0)
    assert ch.outputscale_prior == ("gamma", pytest.approx(5.6), 1.0)
    # ChenLikelihoodFactory = bare gpytorch GaussianLikelihood(): softplus-transformed GreaterThan(1e-4), no prior, raw 0
    assert ch.noise_constraint == "softplus" and ch.noise_prior is None
    assert ch.noise_init == pytest.approx(1e-4 + math.log(2.0)) and gp_spec.pack_raw(ch, gp_spec.initial_params(ch))[0] == pytest.approx(0.0, abs=1e-12)
    hv = gp_spec.from_preset("HVARFNER", 10, lo[:10], hi[:10])
    assert hv.kernel == "rbf" and not hv.use_outputscale and hv.ls_constraint == "box"
    assert hv.ls_prior == ("lognormal", pytest.approx(math.sqrt(2) + 0.5 * math.log(10)), pytest.approx(math.sqrt(3)))
    assert hv.ls_init == pytest.approx(math.exp(math.sqrt(2) + 0.5 * math.log(10) - 3.0))
    assert hv.noise_init == pytest.approx(math.exp(-5.0)) and hv.criterion == "mll"
    assert gp_spec.from_preset("BOTORCH", 10, lo[:10], hi[:10]).ls_prior == hv.ls_prior
    tl = gp_spec.from_preset("CHEN", 5, lo[:5], hi[:5], task_idx=4, n_tasks=3)
    assert tl.criterion == "loo" and tl.dn == 4
    # multi-task forms (presets/hvarfner.py:72-137, presets/botorch.py:80-92): per-task noise + mean, plain MLL,
    # botorch's own PositiveIndexKernel defaults (scaled to the target task), BetaPrior(2.5, 1.5) for BOTORCH only
    mh = gp_spec.from_preset("HVARFNER", 5, lo[:5], hi[:5], task_idx=4, n_tasks=2)
    assert mh.hadamard and mh.task_unit_scale and mh.task_prior is None and mh.criterion == "mll" and mh.kernel == "rbf"
    assert mh.ls_prior == ("lognormal", pytest.approx(math.sqrt(2) + 0.5 * math.log(4)), pytest.approx(math.sqrt(3)))
    mb = gp_spec.from_preset("BOTORCH", 5, lo[:5], hi[:5], task_idx=4, n_tasks=2)
    assert mb.hadamard and mb.task_prior == ("beta", 2.5, 1.5)
    with pytest.raises(ValueError):
        gp_spec.from_preset("NOPE", 5, lo[:5], hi[:5])


@pytest.mark.parametrize("preset", ["EDBO", "EDBO_SMOOTHED", "CHEN", "HVARFNER"])
def test_preset_raw_parameterisation_round_trip_and_oracle_gradient(preset):
    """pack/unpack are inverse for every constraint kind, host and oracle agree on the raw vector, and the host's
    analytic objective (chain rules, prior derivatives) equals the oracle's autograd objective."""
    d, n = 4, 25
    rng = np.random.default_rng(3)
    X, y = rng.random((n, d)), rng.standard_normal(n)
    spec = gp_spec.from_preset(preset, d, np.zeros(d), np.ones(d))
    p = gp_spec.initial_params(spec)
    raw = gp_spec.pack_raw(spec, p)
    q = gp_spec.unpack_raw(spec, raw)
    assert np.allclose(q.lengthscale, p.lengthscale) and math.isclose(q.noise, p.noise, rel_tol=1e-12)
    assert math.isclose(q.outputscale, p.outputscale, rel_tol=1e-12)
    ospec = _ospec(spec)
    assert np.allclose(go.pack_raw(ospec, go.initial_params(ospec)), raw)
    Xn, ys = go.normalize_inputs(ospec, X), go.standardize_targets(y)[0]
    raw = raw + 0.05 * rng.standard_normal(raw.shape)
    f0, g0 = go.fit_objective(ospec, raw, Xn, ys)  # torch.distributions + autograd (oracle/fit_objective.py)
    # the product's hand-written chain rules / prior derivatives around a data term (here the oracle's numpy one)
    dt = go.data_term(ospec, go.unpack_raw(ospec, raw), Xn, ys)
    f1, g1 = gp_spec.objective_from_data_term(spec, raw, n, dt.value,
                                              np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls]))
    assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    for i in range(len(raw)):  # and a coarse finite-difference sanity check (these objectives are ill-conditioned)
        e = np.zeros_like(raw)
        e[i] = 1e-5
        fd = (go.fit_objective(ospec, raw + e, Xn, ys)[0] - go.fit_objective(ospec, raw - e, Xn, ys)[0]) / 2e-5
        assert math.isclose(fd, g0[i], rel_tol=1e-3, abs_tol=1e-5)


@pytest.mark.parametrize("preset", ["HVARFNER", "BOTORCH"])
def test_multitask_botorch_presets_host_objective_equals_the_autograd_oracle(preset):
    """Per-task noise / mean slots, the target-scaled task covariance and the Beta prior on the task correlations:
    the host's chain rules around a data term against the oracle's torch.distributions + autograd objective."""
    d, n, T = 5, 40, 3
    X, Xt, y = make_tl_problem(60, d - 1, n, T=T, seed=4)
    spec = gp_spec.from_preset(preset, d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=T)
    ospec = _ospec(spec)
    p = gp_spec.initial_params(spec)
    assert np.shape(p.noise) == (T,) and np.shape(p.mean) == (T,)
    raw = gp_spec.pack_raw(spec, p)
    assert len(raw) == 2 * T + spec.dn + T * T + T == len(gp_spec.raw_bounds(spec))
    assert gp_spec.raw_bounds(spec) == go.raw_bounds(ospec)
    assert np.allclose(go.pack_raw(ospec, go.initial_params(ospec)), raw)
    rng = np.random.default_rng(5)
    raw = raw + 0.1 * rng.standard_normal(raw.shape)
    raw[:T] = np.abs(raw[:T]) + 2e-4  # noises stay inside their box
    q = gp_spec.unpack_raw(spec, raw)
    assert np.isclose(q.task_B()[0, 0], 1.0) and not np.isclose(q.task_B_unscaled()[0, 0], 1.0)
    theta = gp_spec.theta_from_params(spec, q)
    h0 = 3 + spec.dn + T * T
    assert len(theta) == h0 + 2 * T and np.allclose(theta[h0:h0 + T], q.noise) and np.allclose(theta[h0 + T:], q.mean)
    Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
    oq = go.unpack_raw(ospec, raw)
    assert np.allclose(oq.task_B(), q.task_B()) and np.allclose(oq.noise, q.noise)
    dt = go.data_term(ospec, oq, Xn, ys)
    grad_theta = np.concatenate([[0.0, 0.0, dt.g_outputscale], dt.g_ls, dt.g_task_B.reshape(-1), dt.g_noise, dt.g_mean])
    f1, g1 = gp_spec.objective_from_data_term(spec, raw, len(y), dt.value, grad_theta)
    f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
    assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    # the prior term is really there for BOTORCH (and only there)
    plain = gp_spec.from_preset("HVARFNER", d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=T)
    fp, _ = gp_spec.objective_from_data_term(plain, raw, len(y), dt.value, grad_theta)
    assert (preset == "BOTORCH") == (abs(fp - f1) > 1e-6)


def test_composite_kernel_parameterisation_against_the_autograd_oracle():
    """ProductKernel / AdditiveKernel (baybe/kernels/composite.py:60-91): raw vector layout, bounds, theta layout and the
    host's chain rules (per-factor lengthscale / outputscale slots) against oracle/fit_objective.py."""
    from baybe_amd.kernels import (AdditiveKernel, GammaPrior, LogNormalPrior, MaternKernel, ProductKernel, RBFKernel, ScaleKernel,
                                   apply_kernel_spec)
    from baybe_amd.exceptions import IncompatibilityError

    d, n = 4, 30
    rng = np.random.default_rng(0)
    X, Xt, y = make_problem(100, d, n, seed=2)
    for kern in (ProductKernel([MaternKernel(2.5, GammaPrior(3, 1)), ScaleKernel(RBFKernel(), GammaPrior(2, 0.5))]),
                 ScaleKernel(AdditiveKernel([ScaleKernel(MaternKernel(1.5)), ScaleKernel(RBFKernel(LogNormalPrior(0, 1))),
                                             MaternKernel(0.5)]), GammaPrior(2, 0.15)),
                 # the nested entry of the reference's kernel matrix (tests/test_iterations.py:294-296): (M * M) + (M + M)
                 AdditiveKernel([ProductKernel([MaternKernel(2.5, GammaPrior(3, 1)), MaternKernel(1.5)]),
                                 AdditiveKernel([ScaleKernel(MaternKernel(2.5), GammaPrior(2, 0.5)), MaternKernel(0.5)])])):
        spec = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern)
        if type(kern).__name__ == "AdditiveKernel":
            assert spec.combine == "grouped" and [f.group for f in spec.factors] == [0, 0, 1, 2]
        ospec = _ospec(spec)
        F = spec.n_factors
        raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        assert np.allclose(go.pack_raw(ospec, go.initial_params(ospec)), raw) and gp_spec.raw_bounds(spec) == go.raw_bounds(ospec)
        raw = raw + 0.2 * rng.standard_normal(raw.shape)
        raw[0] = abs(raw[0]) + 1e-3  # (the noise slot is box-constrained: raw = natural value)
        q = gp_spec.unpack_raw(spec, raw)
        assert np.allclose(gp_spec.pack_raw(spec, q), raw)
        theta = gp_spec.theta_from_params(spec, q)
        assert len(theta) == 3 + F * d + F and np.allclose(theta[-F:], q.factor_os)
        Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
        dt = go.data_term(ospec, go.unpack_raw(ospec, raw), Xn, ys)
        grad_theta = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale]] + dt.g_member_ls + [dt.g_member_scale])
        f1, g1 = gp_spec.objective_from_data_term(spec, raw, n, dt.value, grad_theta)
        f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
        assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    # PiecewisePolynomialKernel(q): the oracle's two statements of it (numpy value + product-rule derivative; torch value +
    # autograd) agree, alone and as a factor
    from baybe_amd.kernels import PiecewisePolynomialKernel
    for q in range(4):
        for kern in (ScaleKernel(PiecewisePolynomialKernel(q, GammaPrior(3, 1), 2.0)),
                     ProductKernel([PiecewisePolynomialKernel(q, None, 3.0), RBFKernel()])):
            spec = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern)
            ospec = _ospec(spec)
            raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
            raw = raw + 0.1 * rng.standard_normal(raw.shape)
            raw[0] = abs(raw[0]) + 1e-3
            Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
            dt = go.data_term(ospec, go.unpack_raw(ospec, raw), Xn, ys)
            grad_theta = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale]] + (dt.g_member_ls + [dt.g_member_scale] if spec.factors else [dt.g_ls]))
            f1, g1 = gp_spec.objective_from_data_term(spec, raw, n, dt.value, grad_theta)
            f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
            assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    # RQKernel: alpha slots (softplus chain, no prior) alone, scaled and inside composites
    from baybe_amd.kernels import RQKernel
    for kern in (RQKernel(GammaPrior(3, 1)), ScaleKernel(RQKernel(None, 0.7), GammaPrior(2, 0.5)),
                 ProductKernel([MaternKernel(2.5), ScaleKernel(RQKernel())]), AdditiveKernel([RQKernel(), RQKernel(None, 2.0), RBFKernel()])):
        spec = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern)
        ospec = _ospec(spec)
        raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        assert np.allclose(go.pack_raw(ospec, go.initial_params(ospec)), raw) and gp_spec.raw_bounds(spec) == go.raw_bounds(ospec)
        raw = raw + 0.2 * rng.standard_normal(raw.shape)
        raw[0] = abs(raw[0]) + 1e-3
        q = gp_spec.unpack_raw(spec, raw)
        assert np.allclose(gp_spec.pack_raw(spec, q), raw) and len(q.alpha) == spec.n_factors
        Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
        dt = go.data_term(ospec, go.unpack_raw(ospec, raw), Xn, ys)
        head = [[dt.g_noise, dt.g_mean, dt.g_outputscale]]
        grad_theta = np.concatenate(head + (dt.g_member_ls + [dt.g_member_scale] if spec.factors else [dt.g_ls]) + [dt.g_alpha])
        assert len(grad_theta) == len(gp_spec.theta_from_params(spec, q))
        f1, g1 = gp_spec.objective_from_data_term(spec, raw, n, dt.value, grad_theta)
        f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
        assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    with pytest.raises(IncompatibilityError):  # a sum inside a product would have to be multiplied out (shared parameters): a different model
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)),
                          ProductKernel([MaternKernel(2.5), AdditiveKernel([RBFKernel(), MaternKernel(1.5)])]))
    with pytest.raises(IncompatibilityError):  # so is a scaled product inside a product (one outputscale over two factors)
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)),
                          ProductKernel([MaternKernel(2.5), ScaleKernel(ProductKernel([RBFKernel(), MaternKernel(1.5)]))]))
    # nesting of the same type flattens: same factors, same raw layout (gpytorch registers kernels.0.kernels.0, kernels.0.kernels.1,
    # kernels.1 in this order)
    nested = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)),
                               AdditiveKernel([AdditiveKernel([ScaleKernel(MaternKernel(1.5)), RBFKernel(GammaPrior(3, 1))]), MaternKernel(0.5)]))
    flat = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)),
                             AdditiveKernel([ScaleKernel(MaternKernel(1.5)), RBFKernel(GammaPrior(3, 1)), MaternKernel(0.5)]))
    assert nested.factor_kinds == flat.factor_kinds and nested.combine == flat.combine == "sum"
    assert np.array_equal(gp_spec.pack_raw(nested, gp_spec.initial_params(nested)), gp_spec.pack_raw(flat, gp_spec.initial_params(flat)))
    assert gp_spec.raw_bounds(nested) == gp_spec.raw_bounds(flat)
    with pytest.raises(IncompatibilityError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), ProductKernel([RBFKernel()] * 5))


def test_fast_objective_equals_the_general_functions_for_every_single_kernel_model():
    """``gp_spec.FastObjective`` (the vectorised raw <-> theta map and objective assembly the fit uses for single-task,
    single-kernel models) against ``unpack_raw`` / ``theta_from_params`` / ``objective_from_data_term``: every preset, every
    kernel kind incl. RQ (alpha slot) and piecewise-polynomial, scaled and unscaled, box and softplus constraints."""
    from baybe_amd.kernels import (GammaPrior, LogNormalPrior, MaternKernel, PiecewisePolynomialKernel, RBFKernel, RQKernel,
                                   ScaleKernel, apply_kernel_spec)

    d, n = 6, 37
    rng = np.random.default_rng(11)
    specs = [gp_spec.from_preset(pr, d, np.zeros(d), np.ones(d)) for pr in gp_spec.PRESETS]
    for kern in (MaternKernel(1.5, GammaPrior(3, 1)), ScaleKernel(RBFKernel(LogNormalPrior(0.2, 0.7), 0.4), GammaPrior(2, 0.2), 3.0),
                 RQKernel(GammaPrior(2, 2)), ScaleKernel(RQKernel()), ScaleKernel(PiecewisePolynomialKernel(2, GammaPrior(3, 1), 2.0))):
        specs.append(apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern))
    for spec in specs:
        assert gp_spec.FastObjective.applies(spec)
        fast = gp_spec.FastObjective(spec, n)
        raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec)) + 0.3 * rng.standard_normal(len(gp_spec.raw_bounds(spec)))
        for i, (lo, _) in enumerate(gp_spec.raw_bounds(spec)):
            if lo is not None:
                raw[i] = lo + abs(raw[i] - lo) + 1e-3
        theta, nat = fast.theta(raw)
        ref_theta = gp_spec.theta_from_params(spec, gp_spec.unpack_raw(spec, raw))
        assert theta.shape == ref_theta.shape and np.array_equal(theta, ref_theta)
        val, grad_theta = float(rng.standard_normal()), rng.standard_normal(len(theta))
        f0, g0 = gp_spec.objective_from_data_term(spec, raw, n, val, grad_theta.copy())
        f1, g1 = fast.objective(raw, nat, val, grad_theta.copy())
        assert math.isclose(f0, f1, rel_tol=1e-14, abs_tol=1e-15) and np.allclose(g0, g1, rtol=1e-14, atol=1e-16)
    # round 6: the ICM models of the BAYBE preset take the vectorised assembly too (task factors behind the kernel slots); per-task
    # noise / mean, unit scaling and correlation priors stay on the general functions
    from baybe_amd.kernels import ICMKernelFactory, IndexKernel, ProductKernel

    class Space:
        comp_rep_columns = tuple(f"x{j}" for j in range(d - 1)) + ("task",)
        task_idx, n_tasks = d - 1, 3

    tls = [gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=T) for T in (2, 4)]
    free = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=3)
    apply_kernel_spec(free, ProductKernel([IndexKernel(num_tasks=3, rank=2, parameter_names=["task"]), RQKernel(GammaPrior(2, 2))]), Space())
    for spec in tls + [free]:
        assert gp_spec.FastObjective.applies(spec)
        fast = gp_spec.FastObjective(spec, n)
        import torch

        torch.manual_seed(2)
        raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec)) + 0.3 * rng.standard_normal(len(gp_spec.raw_bounds(spec)))
        for i, (lo, _) in enumerate(gp_spec.raw_bounds(spec)):
            if lo is not None:
                raw[i] = lo + abs(raw[i] - lo) + 1e-3
        theta, nat = fast.theta(raw)
        ref_theta = gp_spec.theta_from_params(spec, gp_spec.unpack_raw(spec, raw))
        assert theta.shape == ref_theta.shape and np.allclose(theta, ref_theta, rtol=1e-15, atol=0)
        val, grad_theta = float(rng.standard_normal()), rng.standard_normal(len(theta))
        f0, g0 = gp_spec.objective_from_data_term(spec, raw, n, val, grad_theta.copy())
        f1, g1 = fast.objective(raw, nat, val, grad_theta.copy())
        assert g0.shape == g1.shape
        assert math.isclose(f0, f1, rel_tol=1e-14, abs_tol=1e-15) and np.allclose(g0, g1, rtol=1e-13, atol=1e-16)
    scaled = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=2)
    scaled.task_unit_scale = True
    assert not gp_spec.FastObjective.applies(scaled)


# ---- backtesting driver: lookup semantics (simulation/lookup.py:19-150), no device needed ---------------
def test_lookup_dataframe_callable_and_impute_modes():
    import pandas as pd

    from baybe_amd.simulation import _cumargmax, look_up_targets

    class T:  # NumericalTarget-shaped
        def __init__(self, name, minimize=False):
            self.name, self.minimize = name, minimize

    lookup = pd.DataFrame({"x": [1, 2, 3], "c": ["a", "b", "a"], "y": [10.0, 20.0, 5.0], "z": [1.0, 2.0, 3.0]})
    q = pd.DataFrame({"x": [2, 1], "c": ["b", "a"]}, index=[7, 9])
    look_up_targets(q, [T("y"), T("z")], lookup)
    assert q["y"].tolist() == [20.0, 10.0] and q["z"].tolist() == [2.0, 1.0] and q.index.tolist() == [7, 9]
    q = pd.DataFrame({"x": [3, 4], "c": ["a", "a"]})
    with pytest.raises(IndexError):
        look_up_targets(q.copy(), [T("y")], lookup[["x", "c", "y"]])
    for mode, want in (("worst", 5.0), ("best", 20.0), ("mean", 35.0 / 3)):
        qq = q.copy()
        look_up_targets(qq, [T("y")], lookup[["x", "c", "y"]], mode)
        assert qq["y"].tolist() == [5.0, pytest.approx(want)]
    qq = q.copy()
    look_up_targets(qq, [T("y", minimize=True)], lookup[["x", "c", "y"]], "worst")
    assert qq["y"].tolist() == [5.0, 20.0]  # worst for a minimised target is the largest value
    qq = q.copy()
    look_up_targets(qq, [T("y")], lambda df: pd.DataFrame({"y": df["x"] * 2.0}, index=df.index))
    assert qq["y"].tolist() == [6.0, 8.0]
    assert _cumargmax(np.array([1.0, 3.0, 2.0, 3.0, 5.0])).tolist() == [0, 1, 1, 3, 4]


def test_scenario_rollout_cases_and_content_hash():
    """``_Rollouts.cases`` of simulation/scenarios.py:26-91 and the content key of the resident candidate matrix."""
    from baybe_amd.recommenders import _content_hash
    from baybe_amd.simulation import _rollout_cases

    assert _rollout_cases(2, 2, 11) == [{"Random_Seed": 11, "Initial_Data": 0}, {"Random_Seed": 11, "Initial_Data": 1},
                                        {"Random_Seed": 12, "Initial_Data": 0}, {"Random_Seed": 12, "Initial_Data": 1}]
    assert [c["Random_Seed"] for c in _rollout_cases(None, 3, None)] == [1337, 1338, 1339]
    assert len(_rollout_cases(3, None, 5)) == 3 and all(np.isnan(c["Initial_Data"]) for c in _rollout_cases(3, None, 5))
    with pytest.raises(ValueError):
        _rollout_cases(None, None, 1)
    a = np.random.default_rng(0).random((5000, 7))
    h = _content_hash(a)
    a[4321, 5] += 1e-12  # any row, any column
    assert _content_hash(a) != h and _content_hash(a) == _content_hash(a.copy())
    # the frame key of the resident candidate matrix (every byte stays hashed, on a persistent pool in 2 MB pieces):
    # an in-place edit of ONE cell of a large frame (pool path) changes it; a copy keys the same
    import pandas as pd

    from baybe_amd.recommenders import _frame_content_hash

    df = pd.DataFrame(np.random.default_rng(1).integers(0, 11, size=(600_000, 4)) / 10.0, columns=list("abcd")).copy()
    k = _frame_content_hash(df)
    assert _frame_content_hash(df.copy()) == k
    df.iloc[599_999, 3] += 1e-12
    assert _frame_content_hash(df) != k
    df.iloc[599_999, 3] -= 1e-12
    df.iloc[300_001, 0] = 0.55
    assert _frame_content_hash(df) != k


def test_simulate_transfer_learning_partitions_by_task_and_trains_on_the_other_tasks():
    """``simulate_transfer_learning`` (simulation/transfer_learning.py:16-99) with a recommender stub (no device): one
    scenario per task, candidates of that task only, every lookup row of the other tasks as training data, the rollout
    structure of ``simulate_scenarios``."""
    import pandas as pd

    from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective, TaskParameter
    from baybe_amd.simulation import simulate_transfer_learning

    vals = np.arange(4) / 3.0
    params = [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals), TaskParameter("task", ["A", "B", "C"])]
    space = SearchSpace.from_product(params)
    exp = space.discrete.exp_rep
    lookup = exp.copy()
    lookup["yield"] = -(lookup["x0"] - 0.3) ** 2 - (lookup["x1"] - 0.6) ** 2 + lookup["task"].map({"A": 0.0, "B": 0.5, "C": -0.5})
    seen = []

    class FirstRows:  # recommends the first candidates, records what it was given
        def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
            cand = searchspace.discrete.get_candidates()[0] if hasattr(searchspace.discrete, "get_candidates") else searchspace.discrete.exp_rep
            seen.append((set(cand["task"]), set(measurements["task"]), len(measurements)))
            return cand.iloc[:batch_size]

    camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), FirstRows())
    res = simulate_transfer_learning(camp, lookup, batch_size=2, n_doe_iterations=3, n_mc_iterations=2)
    assert list(res.columns[:3]) == ["Scenario", "Random_Seed", "Initial_Data"]
    assert sorted(res["Scenario"].unique()) == ["A", "B", "C"] and len(res) == 3 * 2 * 3
    assert set(res["Random_Seed"]) == {1337, 1338}
    per_task = len(exp) // 3
    for cand_tasks, train_tasks, n_train in seen:
        assert len(cand_tasks) == 1  # only the scenario's task is recommended from
        (t,) = cand_tasks
        # the other tasks' rows are all there from the start; the own task only through the loop's measurements
        assert train_tasks - {t} == {"A", "B", "C"} - {t} and n_train >= 2 * per_task
    first = [s for s in seen if s[2] == 2 * per_task]
    assert len(first) == 3 * 2  # the first call of every case sees exactly the off-task data
    with pytest.raises(TypeError):
        simulate_transfer_learning(camp, lambda df: df)
    no_task = Campaign(SearchSpace.from_product(params[:2]), SingleTargetObjective(NumericalTarget("yield")), FirstRows())
    with pytest.raises(NotImplementedError):
        simulate_transfer_learning(no_task, lookup.drop(columns=["task"]))


def test_simulate_scenarios_groupby_partitions_the_search_space():
    """``groupby`` (simulation/scenarios.py:235-334): one loop per group of equal values of the named parameters, the search
    restricted to that group, the group's values in leading columns."""
    import pandas as pd

    from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
    from baybe_amd.simulation import simulate_scenarios

    vals = np.arange(4) / 3.0
    params = [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals), NumericalDiscreteParameter("g", [0.0, 1.0])]
    space = SearchSpace.from_product(params)
    lookup = space.discrete.exp_rep.copy()
    lookup["yield"] = lookup["x0"] + lookup["x1"] + 10 * lookup["g"]
    seen = []

    class FirstRows:
        def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
            cand = searchspace.discrete.get_candidates()[0] if hasattr(searchspace.discrete, "get_candidates") else searchspace.discrete.exp_rep
            seen.append(set(cand["g"]))
            return cand.iloc[:batch_size]

    camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), FirstRows())
    res = simulate_scenarios({"s": camp}, lookup, batch_size=2, n_doe_iterations=2, groupby=["g"])
    assert list(res.columns[:4]) == ["Scenario", "Random_Seed", "Initial_Data", "g"]
    assert sorted(res["g"].unique()) == [0.0, 1.0] and len(res) == 2 * 2
    assert all(len(gs) == 1 for gs in seen) and {next(iter(gs)) for gs in seen} == {0.0, 1.0}
    # measurements of a group come from that group only
    for gval, block in res.groupby("g"):
        flat = [v for row in block["yield_Measurements"] for v in row]
        assert all((v >= 10) == (gval == 1.0) for v in flat)
    plain = simulate_scenarios({"s": camp}, lookup, batch_size=2, n_doe_iterations=2)
    assert "g" not in plain.columns[:4] and len(plain) == 2


def test_noise_percent_perturbs_the_measured_parameters_and_rows_are_still_matched():
    """``noise_percent`` (simulation/core.py:187-195 -> ``add_parameter_noise``): every measured batch enters the campaign
    with its numerical parameter values off the grid by at most that percentage; the campaign marks the nearest grid rows as
    measured (fuzzy matching), so nothing is recommended twice; the run is reproducible under ``random_seed``."""
    import pandas as pd

    from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
    from baybe_amd.dataframe import add_parameter_noise
    from baybe_amd.simulation import simulate_experiment

    vals = 1.0 + np.arange(5) / 4.0
    params = [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals)]
    space = SearchSpace.from_product(params)
    lookup = space.discrete.exp_rep.copy()
    lookup["yield"] = lookup["x0"] * lookup["x1"]
    batches = []

    class FirstRows:
        def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
            cand = searchspace.discrete.get_candidates()[0] if hasattr(searchspace.discrete, "get_candidates") else searchspace.discrete.exp_rep
            batches.append((cand.index[:batch_size].tolist(), None if measurements is None else measurements.copy()))
            return cand.iloc[:batch_size]

    camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), FirstRows())
    res = simulate_experiment(camp, lookup, batch_size=3, n_doe_iterations=4, random_seed=7, noise_percent=3.0)
    picked = [i for idx, _ in batches for i in idx]
    assert len(picked) == len(set(picked)) == 12  # fuzzy matching kept the bookkeeping intact
    last = batches[-1][1]
    grid = set(np.round(vals, 12))
    off = [v for v in last["x0"].tolist() + last["x1"].tolist() if round(v, 12) not in grid]
    assert len(off) >= 0.9 * 2 * len(last)  # the values really left the grid ...
    exp = space.discrete.exp_rep
    for (_, row), true_idx in zip(last.iterrows(), [i for idx, _ in batches[:-1] for i in idx]):
        for c in ("x0", "x1"):  # ... by at most 3 %
            assert abs(row[c] / exp.loc[true_idx, c] - 1.0) <= 0.03 + 1e-12
        assert row["yield"] == lookup.loc[true_idx, "yield"]  # targets were looked up before the noise was applied
    again = simulate_experiment(camp, lookup, batch_size=3, n_doe_iterations=4, random_seed=7, noise_percent=3.0)
    assert again.equals(res)
    df = pd.DataFrame({"x0": [1.0, 2.0], "x1": [1.0, 1.0]})
    np.random.seed(0)
    add_parameter_noise(df, params, noise_type="absolute", noise_level=0.1)
    assert (abs(df["x0"] - [1.0, 2.0]) <= 0.1).all() and (df["x0"] != [1.0, 2.0]).all()
    with pytest.raises(ValueError):
        add_parameter_noise(df, params, noise_type="nope")


def test_engine_state_pickles_without_a_device():
    """``HipGP.__getstate__`` / ``__setstate__`` carry model description, data and hyper-parameters - never the ctypes
    handle (``deepcopy`` of a fitted surrogate raised "ctypes objects containing pointers cannot be pickled").
    Exercised here without the library: the object is rebuilt but never used."""
    import copy
    import pickle

    from baybe_amd import engine, gp_spec

    gp = engine.HipGP.__new__(engine.HipGP)
    spec = gp_spec.GPSpec.baybe_default(3, np.zeros(3), np.ones(3))
    prm = gp_spec.GPParams(np.array([0.3, 0.4, 0.5]), 0.01, 0.1)
    gp.__setstate__({"device": 0, "spec": spec, "params": prm, "n": 4, "ybar": 0.5, "ysd": 2.0, "jitter": 0.0,
                     "X_train": np.arange(12.0).reshape(4, 3), "y_train": np.arange(4.0), "model_args": (None, None)})
    for clone in (copy.deepcopy(gp), pickle.loads(pickle.dumps(gp))):
        assert clone._handle is None and clone._restorable and clone.n == 4 and clone.ysd == 2.0
        assert np.array_equal(clone._X_train, gp._X_train) and clone._X_train is not gp._X_train
        assert np.array_equal(clone.params.lengthscale, prm.lengthscale) and clone.spec.d == 3
        clone.close()  # nothing to release
        assert not clone._restorable


def test_parameter_subsets_pin_the_inactive_lengthscales():
    """``parameter_names`` on the product side is host logic only: pinned raw slots (equal bounds), priors over the active
    columns, a gradient of zero in the pinned slots; the oracle's description of the same model has len(active_dims) entries."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from _problems import oracle_params, oracle_spec
    from baybe_amd import gp_spec
    from baybe_amd.exceptions import IncompatibilityError
    from baybe_amd.kernels import GammaPrior, MaternKernel, PiecewisePolynomialKernel, ProductKernel, RBFKernel, ScaleKernel, apply_kernel_spec

    class Space:
        comp_rep_columns = ("a", "b", "c", "d")

    spec = gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
    apply_kernel_spec(spec, ProductKernel([MaternKernel(2.5, GammaPrior(3, 1), parameter_names=["a", "b"]),
                                           ScaleKernel(RBFKernel(GammaPrior(2, 1), parameter_names=["b", "c", "d"]))]), Space())
    assert spec.active_mask(0).tolist() == [True, True, False, False] and spec.active_mask(1).tolist() == [False, True, True, True]
    p = gp_spec.initial_params(spec)
    assert (p.lengthscale[2:] == gp_spec.INACTIVE_LS).all() and p.factor_ls[0][0] == gp_spec.INACTIVE_LS
    raw, bounds = gp_spec.pack_raw(spec, p), gp_spec.raw_bounds(spec)
    pinned = [i for i, b in enumerate(bounds) if b[0] is not None and b[0] == b[1]]
    assert len(pinned) == 3 and all(raw[i] == bounds[i][0] for i in pinned)
    q = gp_spec.unpack_raw(spec, raw)
    assert np.array_equal(q.lengthscale, p.lengthscale) and np.array_equal(q.factor_ls[0], p.factor_ls[0])
    theta = gp_spec.theta_from_params(spec, q)
    f, g = gp_spec.objective_from_data_term(spec, raw, 10, -3.0, np.ones_like(theta))
    assert np.isfinite(f) and all(g[i] == 0.0 for i in pinned)  # no prior on, no gradient through, a pinned slot
    ospec, op = oracle_spec(spec), oracle_params(spec, p)
    assert len(op.member_ls[0]) == 2 and len(op.member_ls[1]) == 3 and list(ospec.members[1].active_dims) == [1, 2, 3]
    assert not gp_spec.FastObjective.applies(spec)
    with pytest.raises(IncompatibilityError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4)),
                          ProductKernel([MaternKernel(2.5, parameter_names=["zz"]), RBFKernel()]), None)
    with pytest.raises(ValueError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4)), MaternKernel(2.5, parameter_names=["zz"]), Space())


def _theta_layout_mll(spec, theta, Xn, ys):
    """The MLL data term and its gradient in the DEVICE's theta layout (include/baybe_hip.h) from torch autograd - the stand-in
    for ``bbh_fit_value_grad`` in the dot-product kernel test below (one task, plain likelihood): every factor is a function of
    s = sum_j x_j x'_j / w_j^2 (dot kinds) or r^2 = sum_j (x_j - x'_j)^2 / l_j^2, both with the slot values as they stand."""
    import torch

    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)
    X = torch.as_tensor(Xn[:, spec.num_idx], dtype=torch.float64)
    y = torch.as_tensor(ys, dtype=torch.float64)
    dn, F = spec.dn, spec.n_factors
    base = 3 + dn
    fos_off = base + (F - 1) * dn
    alpha_off = base + ((F - 1) * dn + F if F > 1 else 0)
    per_off = alpha_off + (F if spec.has_rq else 0)
    K = None
    for f, kind in enumerate(spec.factor_kinds):
        w = th[3 : 3 + dn] if f == 0 else th[base + (f - 1) * dn : base + f * dn]
        Xs = X / w
        if kind == "periodic":  # exp(-2 sum_j sin^2(pi Delta_j / p_j) / l_j): lengthscale slots l_j, period block p_j
            per = th[per_off + f * dn : per_off + (f + 1) * dn]
            u = math.pi * (X[:, None, :] - X[None, :, :]) / per
            k = torch.exp(-2.0 * (torch.sin(u) ** 2 / w).sum(-1))
        elif kind in gp_spec.DOT_KINDS:
            s = Xs @ Xs.T
            k = s if kind == "linear" else (s + th[alpha_off + f]) ** int(kind[-1])
        else:
            r2 = ((Xs[:, None, :] - Xs[None, :, :]) ** 2).sum(-1)
            k = {"rbf": lambda: torch.exp(-0.5 * r2),
                 "rq": lambda: (1 + r2 / (2 * th[alpha_off + f])) ** (-th[alpha_off + f]),
                 "matern52": lambda: (1 + math.sqrt(5) * torch.sqrt(r2 + 1e-300) + 5.0 / 3.0 * r2) * torch.exp(-math.sqrt(5) * torch.sqrt(r2 + 1e-300))}[kind]()
        if F > 1:
            k = k * th[fos_off + f]
        K = k if K is None else (K * k if spec.combine == "product" else K + k)
    if spec.use_outputscale:
        K = K * th[2]
    n = len(y)
    dist = torch.distributions.MultivariateNormal(th[1] * torch.ones(n, dtype=torch.float64), covariance_matrix=K + th[0] * torch.eye(n, dtype=torch.float64))
    val = dist.log_prob(y)
    (g,) = torch.autograd.grad(val, th)
    return float(val.detach()), g.numpy()


def test_linear_polynomial_and_periodic_kernels_against_the_autograd_oracle():
    """LinearKernel / PolynomialKernel / PeriodicKernel (baybe/kernels/basic.py:20-46, 135-163, 73-112): the product keeps weights
    w_j = v_j^-1/2 (Linear ARD variances) or pinned ones (Polynomial) in the lengthscale slots, the offset in the alpha slot and the
    period lengths of a Periodic kernel in a block at the end of theta; its raw vector has the
    oracle's (= gpytorch's) free parameters in the same order plus pinned slots.  Value and gradient of the assembled objective
    equal the oracle's on the free slots; pinned slots have zero gradient."""
    from _problems import oracle_params
    from baybe_amd.exceptions import IncompatibilityError
    from baybe_amd.kernels import (AdditiveKernel, GammaPrior, LinearKernel, LogNormalPrior, MaternKernel, PeriodicKernel, PolynomialKernel,
                                   ProductKernel, RBFKernel, RQKernel, ScaleKernel, apply_kernel_spec)

    class Space:
        comp_rep_columns = ("a", "b", "c", "d")

    d, n = 4, 24
    rng = np.random.default_rng(5)
    X, Xt, y = make_problem(100, d, n, seed=3)
    kernels = (LinearKernel(), ScaleKernel(LinearKernel(GammaPrior(2, 1), 0.7), GammaPrior(2, 0.5)),
               PolynomialKernel(2), ScaleKernel(PolynomialKernel(3, LogNormalPrior(0, 1), 0.5)),
               LinearKernel(parameter_names=["a", "c"]), PolynomialKernel(1, GammaPrior(2, 2), parameter_names=["b", "c", "d"]),
               AdditiveKernel([PolynomialKernel(1), PolynomialKernel(2), PolynomialKernel(3)]),  # reference tests/test_iterations.py:294
               AdditiveKernel([RBFKernel(), ScaleKernel(LinearKernel(GammaPrior(3, 2))), PolynomialKernel(1, GammaPrior(2, 1))]),
               ProductKernel([MaternKernel(2.5, GammaPrior(3, 1)), ScaleKernel(PolynomialKernel(2, None, 1.5))]),
               PeriodicKernel(), ScaleKernel(PeriodicKernel(GammaPrior(3, 1), 0.8, LogNormalPrior(0, 0.5), 1.3), GammaPrior(2, 0.5)),
               PeriodicKernel(GammaPrior(2, 1), None, GammaPrior(3, 3), parameter_names=["b", "d"]),
               AdditiveKernel([ScaleKernel(PeriodicKernel(None, None, GammaPrior(2, 2))), RQKernel(), PolynomialKernel(1)]),
               ProductKernel([RBFKernel(GammaPrior(3, 1), parameter_names=["a", "b"]),
                              ScaleKernel(PeriodicKernel(GammaPrior(3, 1), 2.0, None, 0.6, parameter_names=["b", "c"]))]))
    for kern in kernels:
        spec = apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern, Space())
        ospec = _ospec(spec)
        assert (spec.has_dot_kind or spec.has_periodic) and not gp_spec.FastObjective.applies(spec)
        p0 = gp_spec.initial_params(spec)
        raw, bounds = gp_spec.pack_raw(spec, p0), gp_spec.raw_bounds(spec)
        free = np.array([not (b[0] is not None and b[0] == b[1]) for b in bounds])
        assert np.allclose(raw[free], go.pack_raw(ospec, go.initial_params(ospec))) and free.sum() == len(go.raw_bounds(ospec))
        raw = np.where(free, raw + 0.3 * rng.standard_normal(raw.shape), raw)
        raw[0] = abs(raw[0]) + 0.05
        q = gp_spec.unpack_raw(spec, raw)
        assert np.allclose(gp_spec.pack_raw(spec, q), raw)
        oq = oracle_params(spec, q)
        assert np.allclose(go.pack_raw(ospec, oq), raw[free], rtol=1e-12, atol=1e-12)
        theta = gp_spec.theta_from_params(spec, q)
        Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
        # the theta-layout stand-in states the same covariance matrix as the oracle's numpy kernels
        val, grad_theta = _theta_layout_mll(spec, theta, Xn, ys)
        f1, g1 = gp_spec.objective_from_data_term(spec, raw, n, val, grad_theta)
        f0, g0 = go.fit_objective(ospec, raw[free], Xn, ys)
        assert math.isclose(f0, f1, rel_tol=1e-10), (kern, f0, f1)
        assert np.allclose(g0, g1[free], rtol=1e-8, atol=1e-11 * np.abs(g0).max()), (kern, g0, g1[free])
        assert (g1[~free] == 0.0).all()
        # prior variance: k(x, x) of the oracle equals the diagonal of its own cross covariance
        Xc = go.normalize_inputs(ospec, X[:7])
        assert np.allclose(go.prior_var(ospec, oq, Xc), np.diag(go.cross_cov(ospec, oq, Xc, Xc)), rtol=1e-13)
        # restart points: drawn from the priors, pinned slots stay pinned
        ps = gp_spec.sample_params_from_priors(spec, np.random.default_rng(1))
        assert np.allclose(gp_spec.pack_raw(spec, ps)[~free], raw[~free])
    with pytest.raises(IncompatibilityError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), PolynomialKernel(5))
    with pytest.raises(IncompatibilityError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), PolynomialKernel(0))


@pytest.mark.parametrize("family", ["gamma", "halfcauchy", "halfnormal", "lognormal", "normal", "smoothedbox"])
def test_every_prior_family_of_the_reference_on_every_kind_of_slot(family):
    """baybe/priors/basic.py:17-92 - the reference iterates GammaPrior(3, 1), HalfCauchyPrior(0.5), HalfNormalPrior(0.5),
    LogNormalPrior(1, 2), NormalPrior(1, 2), SmoothedBoxPrior(0, 3, 0.1) over its kernels and wraps them in
    ``ScaleKernel(outputscale_prior=HalfCauchyPrior(1))`` (tests/test_iterations.py:262-285).  Priors are host arithmetic here
    (``gp_spec._prior_logp_and_grad``: log-density and derivative by hand); the oracle takes them from ``torch.distributions`` with
    autograd (gpytorch's HalfCauchyPrior / NormalPrior / HalfNormalPrior are those classes; SmoothedBoxPrior restated from its
    source).  Lengthscales, outputscales, Linear variances, Polynomial offsets and period lengths each carry the prior once."""
    from baybe_amd import kernels as K

    prior = {"gamma": K.GammaPrior(3, 1), "halfcauchy": K.HalfCauchyPrior(0.5), "halfnormal": K.HalfNormalPrior(0.5),
             "lognormal": K.LogNormalPrior(1, 2), "normal": K.NormalPrior(1, 2), "smoothedbox": K.SmoothedBoxPrior(0, 3, 0.1)}[family]
    d, n = 3, 20
    rng = np.random.default_rng(8)
    X, Xt, y = make_problem(80, d, n, seed=6)
    kernels = (K.ScaleKernel(K.MaternKernel(2.5, prior), K.HalfCauchyPrior(1.0)), K.ScaleKernel(K.RBFKernel(K.GammaPrior(3, 1)), prior),
               K.ScaleKernel(K.LinearKernel(prior), K.HalfCauchyPrior(1.0)), K.ScaleKernel(K.PolynomialKernel(2, prior), K.HalfCauchyPrior(1.0)),
               K.ScaleKernel(K.PeriodicKernel(prior, None, prior), K.HalfCauchyPrior(1.0)), K.RQKernel(prior))
    for kern in kernels:
        spec = K.apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), kern)
        ospec = _ospec(spec)
        bounds = gp_spec.raw_bounds(spec)
        free = np.array([not (b[0] is not None and b[0] == b[1]) for b in bounds])
        raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        for trial in range(3):  # (trial 2 pushes the prior's argument beyond 3: the Gaussian tail of the smoothed box)
            raw_t = np.where(free, raw + (0.4 if trial < 2 else 2.5) * np.abs(rng.standard_normal(raw.shape)) * (1 if trial else -1), raw)
            raw_t[0] = abs(raw_t[0]) + 0.05
            q = gp_spec.unpack_raw(spec, raw_t)
            theta = gp_spec.theta_from_params(spec, q)
            Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
            val, grad_theta = _theta_layout_mll(spec, theta, Xn, ys)
            f1, g1 = gp_spec.objective_from_data_term(spec, raw_t, n, val, grad_theta)
            f0, g0 = go.fit_objective(ospec, raw_t[free], Xn, ys)
            assert math.isclose(f0, f1, rel_tol=1e-10), (family, kern, f0, f1)
            assert np.allclose(g0, g1[free], rtol=1e-7, atol=1e-10 * np.abs(g0).max()), (family, kern, g0, g1[free])
            if gp_spec.FastObjective.applies(spec):  # single stationary kernels: the vectorised assembly takes the same priors
                fast = gp_spec.FastObjective(spec, n)
                th_f, nat = fast.theta(raw_t)
                assert np.allclose(th_f, theta, rtol=1e-14)
                f2, g2 = fast.objective(raw_t, nat, val, grad_theta.copy())
                assert math.isclose(f2, f1, rel_tol=1e-12) and np.allclose(g2, g1, rtol=1e-10, atol=1e-14)
        ps = gp_spec.sample_params_from_priors(spec, np.random.default_rng(2))  # restart points exist for every family
        assert np.all(np.isfinite(gp_spec.pack_raw(spec, ps)))


def test_lean_lbfgsb_driver_is_scipys():
    """``engine.lbfgsb_minimize`` drives scipy's compiled L-BFGS-B routine (``setulb``) without the per-evaluation wrappers of
    ``scipy.optimize.minimize``; the iterates must be scipy's, bit for bit: end point, value, iteration / evaluation counts, status
    and message - on the oracle's GP fit objective (box bounds, pinned slots, an infinite value on the way), on an iteration cap,
    and when the very first evaluation fails."""
    import scipy.optimize as sopt

    from baybe_amd import engine
    from baybe_amd.kernels import GammaPrior, MaternKernel, ScaleKernel, apply_kernel_spec

    assert engine._lean_lbfgsb_usable()

    def same(make_fun, x0, bounds, maxiter):  # (a fresh objective per run: the ones below count their calls)
        ref = sopt.minimize(make_fun(), x0, jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": maxiter})
        got = engine.lbfgsb_minimize(make_fun(), x0, bounds, maxiter)
        assert isinstance(got, engine._OptResult)  # (the lean driver ran, not the fallback)
        assert np.array_equal(ref.x, got.x) and (ref.fun == got.fun or (np.isnan(ref.fun) and np.isnan(got.fun)))
        assert (ref.nit, ref.nfev, ref.status, str(ref.message)) == (got.nit, got.nfev, got.status, got.message)
        return got

    class Space:
        comp_rep_columns = ("a", "b", "c", "d")

    X, Xt, y = make_problem(200, 4, 35, seed=12)
    for kern in (None, ScaleKernel(MaternKernel(1.5, GammaPrior(3, 1), parameter_names=["a", "c"]), GammaPrior(2, 0.5))):
        spec = gp_spec.GPSpec.baybe_default(4, np.zeros(4), np.ones(4))
        if kern is not None:
            apply_kernel_spec(spec, kern, Space())
        ospec = _ospec(spec)
        Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
        bounds = gp_spec.raw_bounds(spec)
        free = np.array([not (b[0] is not None and b[0] == b[1]) for b in bounds])
        def make_fun(bad_call):
            def factory():
                calls = {"n": 0}

                def fun(raw):  # the oracle's objective on the free slots, zero gradient in the pinned ones; one evaluation reports +inf
                    calls["n"] += 1
                    if calls["n"] == bad_call:
                        return float("inf"), np.zeros_like(raw)
                    f, g = go.fit_objective(ospec, raw[free], Xn, ys)
                    full = np.zeros_like(raw)
                    full[free] = g
                    return f, full

                return fun

            return factory

        x0 = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        for maxiter, bad_call in ((15000, 0), (15000, 4), (5, 0)):
            res = same(make_fun(bad_call), x0, bounds, maxiter)
            assert (maxiter == 5) == (res.status == 1), (maxiter, res.status, res.message)
    # the first evaluation fails (NaN value, zero gradient): the same outcome from both - the engine's retry logic looks at the value
    res = same(lambda: (lambda x: (float("nan"), np.zeros_like(x))), np.ones(3), [(None, None)] * 3, 100)
    assert np.isnan(res.fun)


def test_fast_sobol_points_are_the_engines_bitwise():
    """``engine._sobol_uniform_fast`` (torch's generator for the scrambling bits + the library's host code for the scrambling and
    the Gray-code walk) against ``torch.quasirandom.SobolEngine`` itself: every point bitwise, including the engine's
    single-precision first point; and the base samples built on it are unchanged by the switch."""
    import torch

    from baybe_amd import engine

    assert engine._fast_sobol_usable()
    for S, q, seed in ((1, 1, 0), (2, 3, 1), (64, 16, 1234), (513, 96, 999999), (100, 1111, 42)):
        assert torch.equal(engine._sobol_uniform_fast(S, q, seed), engine._sobol_uniform_engine(S, q, seed)), (S, q, seed)
    z = engine.sobol_normal_base_samples(128, 12, 77)
    u = engine._sobol_uniform_engine(128, 12, 77)
    v = 0.5 + (1 - torch.finfo(torch.float64).eps) * (u - 0.5)
    assert np.array_equal(z, (torch.erfinv(2 * v - 1) * np.sqrt(2.0)).numpy())


@pytest.mark.parametrize("variant", ["positive_rank2", "free_rank1", "icm_factory_default", "icm_factory_custom"])
def test_user_supplied_task_kernels_against_the_autograd_oracle(variant):
    """``IndexKernel`` / ``PositiveIndexKernel`` objects inside a ``ProductKernel`` and ``ICMKernelFactory`` with custom base / task
    kernels (kernels/basic.py:220-248, components/kernel.py:238-337) reach the device's task table: covariance factor [T, rank]
    with rank < T, a free factor for gpytorch's ``IndexKernel`` (started from ``torch.randn``), ``unit_scale_for_target`` off.  Raw
    layout, bounds and the host's chain rules against the oracle's autograd objective; ``BetaPrior`` is refused as the reference
    refuses it (``to_gpytorch`` raises NotImplementedError, priors/basic.py:94-108)."""
    import torch

    from baybe_amd.kernels import (GammaPrior, ICMKernelFactory, IndexKernel, MaternKernel, PositiveIndexKernel, ProductKernel, RBFKernel,
                                   ScaleKernel, _prior_tuple, apply_kernel_spec)

    d, n, T = 5, 14, 3
    X, Xt, y = make_tl_problem(60, d - 1, n, T=T, seed=9)

    class Space:
        comp_rep_columns = tuple(f"x{j}" for j in range(d - 1)) + ("task",)
        n_tasks, task_idx = T, d - 1

    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=T)
    if variant == "positive_rank2":
        kern = ProductKernel([ScaleKernel(MaternKernel(1.5, GammaPrior(3, 1)), GammaPrior(2, 0.5)), PositiveIndexKernel(num_tasks=T, rank=2)])
        rank, free = 2, False
    elif variant == "free_rank1":
        kern = ProductKernel([IndexKernel(num_tasks=T, rank=1, parameter_names=["task"]), RBFKernel(GammaPrior(3, 1))])
        rank, free = 1, True
    elif variant == "icm_factory_default":
        kern = ICMKernelFactory()(Space())
        rank, free = T, False
    else:
        kern = ICMKernelFactory(base_kernel_or_factory=lambda ss, *a: MaternKernel(2.5, GammaPrior(3, 1)),
                                task_kernel_or_factory=IndexKernel(num_tasks=T, rank=2))(Space(), None, None)
        rank, free = 2, True
    apply_kernel_spec(spec, kern, Space())
    assert spec.task_rank == rank and (spec.task_factor_constraint == "none") == free and not spec.task_unit_scale
    if variant == "icm_factory_default":
        assert spec.kernel == "matern52" and spec.ls_constraint == "box"  # the preset's numerical kernel stays
    ospec = _ospec(spec)
    torch.manual_seed(3)
    p = gp_spec.initial_params(spec)
    assert p.task_W.shape == (T, rank)
    if free:
        # covar_factor, then raw_var: gpytorch's registration order - drawn at torch's DEFAULT dtype (float32 in a fresh process,
        # float64 once a BayBE prior has been converted, baybe/priors/base.py:25) and cast to double, as gpytorch's parameters are
        for default in (torch.float32, torch.float64):
            was = torch.get_default_dtype()
            try:
                torch.set_default_dtype(default)
                torch.manual_seed(3)
                pd_ = gp_spec.initial_params(spec)
                torch.manual_seed(3)
                assert np.array_equal(pd_.task_W, torch.randn(T, rank).to(torch.float64).numpy())
            finally:
                torch.set_default_dtype(was)
    raw = gp_spec.pack_raw(spec, p)
    assert len(raw) == len(gp_spec.raw_bounds(spec)) and gp_spec.raw_bounds(spec) == go.raw_bounds(ospec)
    rng = np.random.default_rng(6)
    raw = raw + 0.2 * rng.standard_normal(raw.shape)
    for k, (lo, _) in enumerate(gp_spec.raw_bounds(spec)):  # box-constrained slots (noise; the preset's lengthscales) stay inside
        if lo is not None:
            raw[k] = max(raw[k], lo + 1e-3)
    q = gp_spec.unpack_raw(spec, raw)
    oq = go.unpack_raw(ospec, raw)
    assert q.task_W.shape == (T, rank) and np.allclose(oq.task_B(), q.task_B())
    if free:
        assert (q.task_W < 0).any() or True  # (a free factor may be negative: no softplus)
        assert np.array_equal(q.task_W.reshape(-1), raw[-(T * rank + T):-T])
    Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
    dt = go.data_term(ospec, oq, Xn, ys)
    grad_theta = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls, dt.g_task_B.reshape(-1)])
    f1, g1 = gp_spec.objective_from_data_term(spec, raw, len(y), dt.value, grad_theta)
    f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
    assert math.isclose(f0, f1, rel_tol=1e-12) and np.allclose(g0, g1, rtol=1e-9, atol=1e-12 * np.abs(g0).max())
    # refusals
    with pytest.raises(ValueError):
        apply_kernel_spec(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=d - 1, n_tasks=T),
                          ProductKernel([MaternKernel(), IndexKernel(num_tasks=T + 1, rank=1)]), Space())
    with pytest.raises(NotImplementedError):
        _prior_tuple(type("BetaPrior", (), {"alpha": 1.0, "beta": 2.0})())


# ---- RFF kernel ---------------------------------------------------------------------------------------------------------------------------
def test_rff_kernel_oracle_and_the_feature_space_identities():
    """The oracle's RFF kernel (oracle/gp_oracle.py::rff_features: gpytorch ``RFFKernel._featurize`` / ``forward``) against its own
    autograd form, and the feature-space expressions csrc/bbh_rff.hip evaluates (m x m system, Woodbury) against the oracle's n x n
    posterior and objective - in numpy, so that the maths the device uses is pinned where no device exists."""
    import torch
    from _problems import oracle_params, oracle_spec
    from baybe_amd import gp_spec
    from baybe_amd.kernels import GammaPrior, RFFKernel, ScaleKernel, apply_kernel_spec
    from oracle import fit_objective as fo
    from oracle import gp_oracle as go

    d, n, D = 4, 40, 7
    rng = np.random.default_rng(5)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    apply_kernel_spec(spec, ScaleKernel(RFFKernel(D, GammaPrior(3, 1)), GammaPrior(2, 0.5)))
    assert (spec.kernel, spec.rff_num_samples, spec.use_outputscale, spec.ls_constraint) == ("rff", D, True, "softplus")
    spec.rff_weights = rng.standard_normal((d, D))
    ospec = oracle_spec(spec)
    Xt, Xc = rng.random((n, d)), rng.random((25, d))
    y = np.sin(3 * Xt[:, 0]) + Xt[:, 1]
    p = gp_spec.initial_params(spec)
    p.lengthscale = 0.3 + rng.random(d)
    p.noise, p.outputscale, p.mean = 0.02, 1.7, 0.1
    op = oracle_params(spec, p)
    Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
    K = go.cross_cov(ospec, op, Xn, Xn)
    assert np.allclose(np.diag(K), p.outputscale, rtol=1e-13)  # k(x, x) = (cos^2 + sin^2) summed / D = 1
    assert np.linalg.matrix_rank(K) == 2 * D
    nat = {"lengthscale": torch.as_tensor(p.lengthscale), "outputscale": torch.tensor(p.outputscale, dtype=torch.float64)}
    assert np.allclose(fo.train_covariance(ospec, nat, torch.as_tensor(Xn)).numpy(), K, rtol=1e-13, atol=1e-14)
    # autograd gradient of the oracle objective against central differences
    raw = go.pack_raw(ospec, op)
    f0, g0 = go.fit_objective(ospec, raw, Xn, ys)
    for i in range(len(raw)):
        e = np.zeros_like(raw)
        e[i] = 1e-6
        fd = (go.fit_objective(ospec, raw + e, Xn, ys)[0] - go.fit_objective(ospec, raw - e, Xn, ys)[0]) / 2e-6
        assert math.isclose(fd, g0[i], rel_tol=2e-6, abs_tol=1e-7), (i, fd, g0[i])
    # feature space: B = eps I + Phi^T Phi, a = B^-1 Phi^T r
    s2, os_, c = p.noise, p.outputscale, p.mean
    Phi = go.rff_features(spec.rff_weights, Xn, p.lengthscale) / math.sqrt(D)
    m = 2 * D
    B = (s2 / os_) * np.eye(m) + Phi.T @ Phi
    r = ys - c
    a = np.linalg.solve(B, Phi.T @ r)
    alpha = (r - Phi @ a) / s2
    Ky = K + s2 * np.eye(n)
    assert np.allclose(alpha, np.linalg.solve(Ky, r), rtol=1e-9, atol=1e-12)
    logdet = (n - m) * math.log(s2) + m * math.log(os_) + np.linalg.slogdet(B)[1]
    assert math.isclose(logdet, np.linalg.slogdet(Ky)[1], rel_tol=1e-11)
    trBinv = np.trace(np.linalg.inv(B))
    assert math.isclose((n - m + (s2 / os_) * trBinv) / s2, np.trace(np.linalg.inv(Ky)), rel_tol=1e-9)
    model = go.GPModel(ospec, op, Xt, y)
    mo, vo = model.posterior(Xc)
    Pc = go.rff_features(spec.rff_weights, go.normalize_inputs(ospec, Xc), p.lengthscale) / math.sqrt(D)
    ybar, ysd = go.standardize_targets(y)[1:]
    assert np.allclose(ybar + ysd * (c + Pc @ a), mo, rtol=1e-10, atol=1e-12)
    var_f = ysd**2 * s2 * np.einsum("ik,kl,il->i", Pc, np.linalg.inv(B), Pc)
    assert np.allclose(var_f, vo, rtol=1e-7, atol=1e-12)
    _, cov = model.posterior_joint(Xc[:5])
    assert np.allclose(ysd**2 * s2 * Pc[:5] @ np.linalg.inv(B) @ Pc[:5].T, cov, rtol=1e-7, atol=1e-12)


def test_rff_kernel_specifications_the_device_does_not_take():
    from baybe_amd import gp_spec
    from baybe_amd.exceptions import IncompatibleSurrogateError  # BayBE's own type for "this surrogate cannot do that" (exceptions.py:61)
    from baybe_amd.kernels import AdditiveKernel, MaternKernel, RFFKernel, apply_kernel_spec

    d = 3
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    apply_kernel_spec(spec, RFFKernel(256))  # (round 6: up to 256 frequencies)
    assert spec.rff_num_samples == 256
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    with pytest.raises(IncompatibleSurrogateError, match="256 frequencies"):
        apply_kernel_spec(spec, RFFKernel(257))
    with pytest.raises(IncompatibleSurrogateError, match="inside a"):
        apply_kernel_spec(spec, AdditiveKernel([RFFKernel(5), MaternKernel(2.5)]))
    tl = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=2)
    with pytest.raises(IncompatibleSurrogateError, match="task parameter"):
        apply_kernel_spec(tl, RFFKernel(5))
    with pytest.raises((ValueError, TypeError)):
        RFFKernel(0)


def test_qlognehvi_refuses_rff_surrogates_before_touching_a_device():
    """The extended models of qLogNEHVI condition on noise-free latent rows, which the RFF kernel's feature-space form does not have: the
    scorer refuses such surrogates by name, before any device object exists (no HIP library is needed for the refusal)."""
    from types import SimpleNamespace

    from baybe_amd.exceptions import IncompatibilityError
    from baybe_amd.nehvi import HipNEHVI

    engines = [SimpleNamespace(spec=SimpleNamespace(kernel="matern52")), SimpleNamespace(spec=SimpleNamespace(kernel="rff"))]
    with pytest.raises(IncompatibilityError, match="RFFKernel"):
        HipNEHVI(engines, [1.0, 1.0], np.zeros((3, 2)), np.zeros(2))


def test_fits_side_by_side_fall_back_to_a_sequence_without_a_device(monkeypatch):
    """``engine.fit_side_by_side`` (what ``HipCompositeImpl.fit`` runs) needs streams: where no device is visible - the CPU double of the
    handle in these tests - the jobs run in order on the calling thread; ``BBH_FIT_SIDE_BY_SIDE=0`` forces that everywhere."""
    import threading

    import torch

    from baybe_amd import engine

    seen = []

    def job(k):
        def run():
            seen.append((k, threading.current_thread().name, getattr(engine._FIT_TLS, "stream", None)))
            return k * k
        return run

    if not torch.cuda.is_available():
        assert engine.fit_side_by_side([job(k) for k in range(3)]) == [0, 1, 4]
        assert [s[0] for s in seen] == [0, 1, 2] and {s[1] for s in seen} == {threading.current_thread().name} and all(s[2] is None for s in seen)
    seen.clear()
    monkeypatch.setenv("BBH_FIT_SIDE_BY_SIDE", "0")
    assert engine.fit_side_by_side([job(k) for k in range(2)]) == [0, 1] and [s[0] for s in seen] == [0, 1]
    assert engine.fit_side_by_side([]) == [] and engine.fit_side_by_side([job(7)]) == [49]


def test_side_by_side_fits_that_need_the_global_generator_are_redone_in_sequence(monkeypatch):
    """A retry from re-sampled hyper-parameters (or a free task factor's random start) draws from torch's global generator; side by side
    the state after the group would depend on thread timing.  Such a fit raises inside the parallel phase - before anything has been
    drawn - and the whole group is fitted again target by target, as the reference does (surrogates/composite.py:101-134)."""
    import torch

    from baybe_amd import engine

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)

    class _NoStream:
        def __enter__(self):
            return None

        def __exit__(self, *a):
            return False

    monkeypatch.setattr(engine, "private_fit_stream", lambda device, slot: _NoStream())
    calls = []

    def job(k, needs_rng):
        def run():
            inside = bool(getattr(engine._FIT_TLS, "no_global_generator", False))
            calls.append((k, inside))
            if needs_rng and inside:
                raise engine._FitNeedsTheGlobalGenerator("retry")
            return (k, float(torch.rand(1)) if needs_rng else None)
        return run

    torch.manual_seed(0)
    want = float(torch.rand(1))
    torch.manual_seed(0)
    out = engine.fit_side_by_side([job(0, False), job(1, True), job(2, False)])
    assert [o[0] for o in out] == [0, 1, 2] and out[1][1] == want  # the draw happened once, in the sequential pass
    assert sorted(calls[:3]) == [(0, True), (1, True), (2, True)] and calls[3:] == [(0, False), (1, False), (2, False)]
    calls.clear()
    assert [o[0] for o in engine.fit_side_by_side([job(0, False), job(1, False)])] == [0, 1] and len(calls) == 2  # (no rerun without need)
