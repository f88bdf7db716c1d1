"""The other acquisition functions of baybe/acquisition/acqfs.py:161-290 on the device vs the oracle:
analytic (PM, PSTD, UCB, EI, LogEI, PI) and MC (qEI, qPI, qSR, qUCB, qPSTD), q=1 and with pending
points; through the recommender for batch sizes 1 and 2 (tests/test_iterations.py:319-336 analogue)."""

import numpy as np
import pytest

from _problems import fixed_theta, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    N, d, n = 2000, 5, 60
    X, Xt, y = make_problem(N, d, n, seed=9)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    ls, nz, _ = fixed_theta(d)
    p = gp_spec.GPParams(np.full(d, ls), nz, 0.02)
    gp = engine.HipGP(0)
    gp.set_model(spec, Xt, y)
    gp.factorize(p)
    om = go.GPModel(go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), go.GPParams(p.lengthscale, p.noise, p.mean), Xt, y)
    return gp, om, X


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_analytic_family(setup, sign):
    from oracle import gp_oracle as go

    gp, om, X = setup
    mo, vo = om.posterior(X)
    m, v = gp.posterior(X)
    bf = go.best_f_from_model(om, sign)
    for kind in go.ANALYTIC_KINDS:
        for maximize in ((True, False) if kind == "PSTD" else (True,)):
            ref = go.analytic_acq(kind, mo, vo, bf, sign, beta=0.7, maximize=maximize)
            got = gp.analytic_acq(kind, m, v, bf, sign, beta=0.7, maximize=maximize).cpu().numpy()
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-12), (kind, np.abs(got - ref).max())
            assert int(np.argmax(got)) == int(np.argmax(ref))


def test_logei_tail_is_finite_and_accurate(setup):
    import torch

    from oracle import gp_oracle as go

    gp, _, _ = setup
    mu = torch.tensor([-50.0, -5.0, 0.0, 3.0], dtype=torch.float64, device="cuda")
    var = torch.tensor([1e-4, 1.0, 1.0, 1e-14], dtype=torch.float64, device="cuda")
    got = gp.analytic_acq("LogEI", mu, var, 1.0, 1.0).cpu().numpy()
    ref = go.analytic_acq("LogEI", mu.cpu().numpy(), var.cpu().numpy(), 1.0, 1.0)
    assert np.isfinite(got).all() and np.allclose(got, ref, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_mc_family_q1_and_pending(setup, sign):
    from oracle import gp_oracle as go

    gp, om, X = setup
    mo, vo = om.posterior(X)
    m, v = gp.posterior(X)
    bf = go.best_f_from_model(om, sign)
    z1 = go.sobol_normal_base_samples(256, 1, 3)[:, 0]
    pend = X[[4, 9]]
    keep = np.ones(300, bool)
    keep[[4, 9]] = False
    Xc = X[:300][keep]
    z3 = go.sobol_normal_base_samples(256, 3, 3)
    gp.set_pending(pend)
    mc, vc = gp.posterior(Xc)
    cross = gp.cross_cov(Xc)
    for kind in go.MC_KINDS:
        ref = go.mc_acq_q1(kind, mo, vo, z1, bf, sign, beta=0.4)
        got = gp.mc_acq(kind, m, v, z1, bf, sign, beta=0.4).cpu().numpy()
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-10), (kind, np.abs(got - ref).max())
        refp = np.array([go.mc_acq_joint(kind, *om.posterior_joint(np.vstack([x[None, :], pend])), z3, bf, sign, beta=0.4)
                         for x in Xc[:80]])
        gotp = gp.mc_acq(kind, mc, vc, z3, bf, sign, beta=0.4, cross=cross).cpu().numpy()[:80]
        assert np.allclose(gotp, refp, rtol=1e-8, atol=1e-9), (kind, np.abs(gotp - refp).max())
    gp.set_pending(None)


@pytest.mark.parametrize("abbr", ["qEI", "qPI", "qSR", "qUCB", "qPSTD", "PM", "PSTD", "UCB", "EI", "LogEI", "PI"])
def test_through_the_recommender(abbr):
    from _baybe_shim import NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
    from baybe_amd.acquisition import convert_acqf
    from baybe_amd.exceptions import IncompatibleAcquisitionFunctionError
    from baybe_amd.recommenders import HipBotorchRecommender

    rng = np.random.default_rng(5)
    vals = np.arange(7) / 6.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 15, replace=False)].copy()
    Xm = meas[["x0", "x1", "x2"]].to_numpy(float)
    meas["y"] = -((Xm - 0.4) ** 2).sum(1) + 0.02 * rng.standard_normal(len(Xm))
    obj = SingleTargetObjective(NumericalTarget("y"))
    rec = HipBotorchRecommender(acquisition_function=abbr)
    acqf = convert_acqf(abbr)
    got = rec.recommend(1, space, obj, meas)
    assert len(got) == 1
    vals_ = rec.acquisition_values(exp.iloc[:40], space, obj, meas)
    assert np.isfinite(vals_.to_numpy()).all()
    if acqf.is_analytic:
        with pytest.raises(IncompatibleAcquisitionFunctionError):
            rec.recommend(2, space, obj, meas)
        with pytest.raises(IncompatibleAcquisitionFunctionError):
            rec.recommend(1, space, obj, meas, pending_experiments=exp.iloc[:1])
    else:
        got2 = rec.recommend(2, space, obj, meas, pending_experiments=exp.iloc[:1])
        assert len(set(got2.index)) == 2
