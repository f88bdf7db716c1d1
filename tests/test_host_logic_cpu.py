"""Host-side fit bookkeeping (baybe_amd/gp_spec.py) against the oracle, using the oracle's data
term in place of the device call (checker role only)."""

import numpy as np
import pytest

from _problems import make_problem, make_tl_problem
from baybe_amd import gp_spec
from oracle import gp_oracle as go


def _ospec(spec):
    return go.GPSpec(
        d=spec.d, num_idx=spec.num_idx, lo=spec.lo[spec.num_idx], hi=spec.hi[spec.num_idx], kernel=spec.kernel,
        task_idx=spec.task_idx, n_tasks=spec.n_tasks, use_outputscale=spec.use_outputscale,
        ls_constraint=spec.ls_constraint, ls_lower=spec.ls_lower, ls_prior=spec.ls_prior, ls_init=spec.ls_init,
        noise_lower=spec.noise_lower, noise_prior=spec.noise_prior, noise_init=spec.noise_init,
        outputscale_prior=spec.outputscale_prior, criterion=spec.criterion)


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
