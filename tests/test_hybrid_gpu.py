"""Hybrid search spaces on the device (``HipRecommenderImpl._recommend_hybrid``: every discrete row crossed with scrambled-Sobol
points of the continuous box in ONE scoring pass, compass refinement of the best starts; reference: botorch/hybrid.py:30-163).
The recommended point must be at least as good as the best point of a dense (discrete rows x continuous grid) sweep scored by the
same device model, and the oracle must agree with the device on its value."""

import itertools

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def test_hybrid_recommendation_reaches_the_dense_grid_optimum_on_the_device():
    from types import SimpleNamespace

    import torch

    from _problems import oracle_params, oracle_spec
    from _replay import HybridSpace
    from baybe_amd import engine
    from baybe_amd.recommenders import HipBotorchRecommender
    from oracle import gp_oracle as go

    rng = np.random.default_rng(0)
    levels = np.linspace(0, 1, 7)
    disc = pd.DataFrame(list(itertools.product(levels, levels, levels)), columns=["d0", "d1", "d2"])  # 343 discrete rows
    bounds = pd.DataFrame({"c0": [0.0, 1.0], "c1": [-1.0, 2.0]}, index=["min", "max"])
    space = HybridSpace(disc, bounds)

    def truth(M):
        return -((M[:, :3] - 0.5) ** 2).sum(1) - (M[:, 3] - 0.3) ** 2 - 0.3 * (M[:, 4] - 1.2) ** 2

    M = np.hstack([disc.to_numpy()[rng.choice(343, 25)], rng.random((25, 1)), rng.uniform(-1, 2, (25, 1))])
    meas = pd.DataFrame(M, columns=list(space.comp_rep_columns)).assign(y=truth(M))
    objective = SimpleNamespace(targets=(SimpleNamespace(name="y", minimize=False, transformation=None),), is_multi_output=False)
    rec = HipBotorchRecommender(n_raw_samples=64, n_restarts=10)
    torch.manual_seed(5)
    got = rec.recommend(3, space, objective, meas)
    assert list(got.columns) == ["d0", "d1", "d2", "c0", "c1"] and len(got) == 3
    assert got["c0"].between(0, 1).all() and got["c1"].between(-1, 2).all()
    eng = rec._surrogate_model.engine
    g = np.linspace(0, 1, 21)
    cont = np.array([[a, -1 + 3 * b] for a, b in itertools.product(g, g)])
    dense = np.hstack([np.repeat(disc.to_numpy(), len(cont), axis=0), np.tile(cont, (len(disc), 1))])  # 151 263 points
    z = engine.sobol_normal_base_samples(512, 1, 99)[:, 0]
    bf = eng.best_f(1.0)
    dscore = eng.qlogei(*eng.posterior(dense), z, bf).cpu().numpy()
    first = got.iloc[[0]].to_numpy(dtype=float)
    m1, v1 = eng.posterior(first)
    s1 = eng.qlogei(m1, v1, z, bf).cpu().numpy()[0]
    assert s1 >= dscore.max() - 1e-3, (s1, dscore.max())
    om = go.GPModel(oracle_spec(eng.spec), oracle_params(eng.spec, eng.params), eng._X_train, eng._y_train)
    assert abs(go.qlogei_q1(*om.posterior(first), z, go.best_f_from_model(om))[0] - s1) < 1e-8
    comp = got.to_numpy(dtype=float)
    assert np.linalg.norm(comp[0] - comp[1]) > 1e-3 and np.linalg.norm(comp[1] - comp[2]) > 1e-3


@pytest.mark.parametrize("dc,levels,q", [(1, 5, 3), (2, 4, 3), (3, 3, 2), (4, 2, 1)])
def test_hybrid_picks_reach_the_enumeration_oracles_optimum(dc, levels, q):
    """The parity path of hybrid spaces.  ``oracle/hybrid_oracle.py`` restates what
    /root/reference/baybe/recommenders/pure/bayesian/botorch/hybrid.py:30-163 computes through ``optimize_acqf_mixed`` - per discrete
    row the maximum of the acquisition function over the continuous box, the best row wins, sequential greedy for a batch - as a dense
    grid per row plus an L-BFGS-B polish.  Step by step (the oracle conditions on the DEVICE's earlier picks, so that one step is
    compared at a time): the device's pick is a point of the winning row's basin whose acquisition value, scored by the oracle,
    reaches the oracle's optimum to 1e-6."""
    from types import SimpleNamespace

    import torch

    from _problems import oracle_params, oracle_spec
    from _replay import HybridSpace
    from baybe_amd import engine
    from baybe_amd.recommenders import HipBotorchRecommender
    from conftest import record_deviation
    from oracle import gp_oracle as go
    from oracle import hybrid_oracle as ho

    rng = np.random.default_rng(10 + dc)
    lv = np.linspace(0, 1, levels)
    disc = pd.DataFrame(list(itertools.product(lv, lv)), columns=["d0", "d1"])
    cb = np.array([[0.0, -1.0, 0.5, 0.0][:dc], [1.0, 2.0, 1.5, 2.0][:dc]])
    bounds = pd.DataFrame(cb, index=["min", "max"], columns=[f"c{a}" for a in range(dc)])
    space = HybridSpace(disc, bounds)
    opt = np.array([0.3, 1.2, 0.9, 1.4][:dc])

    def truth(M):
        return -((M[:, :2] - 0.45) ** 2).sum(1) - (0.6 * (M[:, 2:] - opt) ** 2).sum(1) + 0.05 * np.sin(5.0 * M[:, 2])

    n = 18
    M = np.hstack([disc.to_numpy()[rng.choice(len(disc), n)], cb[0] + rng.random((n, dc)) * (cb[1] - cb[0])])
    meas = pd.DataFrame(M, columns=list(space.comp_rep_columns)).assign(y=truth(M))
    objective = SimpleNamespace(targets=(SimpleNamespace(name="y", minimize=False, transformation=None),), is_multi_output=False)
    rec = HipBotorchRecommender()
    torch.manual_seed(7)
    got = rec.recommend(q, space, objective, meas)
    torch.manual_seed(7)
    engine.draw_sampler_seed()  # (the raw samples' Sobol seed)
    seed = engine.draw_sampler_seed()  # the acquisition function's sampler seed of that call
    eng = rec._surrogate_model.engine
    om = go.GPModel(oracle_spec(eng.spec), oracle_params(eng.spec, eng.params), eng._X_train, eng._y_train)
    best_f = go.best_f_from_model(om)
    picks = got.to_numpy(dtype=float)
    D = disc.to_numpy(dtype=float)
    worst = 0.0
    for step in range(q):
        pend = picks[:step]
        z = go.sobol_normal_base_samples(512, 1 + step, seed)
        row, c_opt, v_opt, _ = ho.mixed_step(om, D, cb, pend, z, best_f)
        v_dev = float(ho._scores(om, picks[step : step + 1], pend, z, best_f, 1.0)[0])
        gap = v_opt - v_dev
        worst = max(worst, gap)
        print(f"d_c = {dc} step {step}: oracle row {row} value {v_opt:.9f} at {c_opt}; device row {picks[step, :2]} value {v_dev:.9f} at {picks[step, 2:]}")
        assert gap <= 1e-6, (dc, step, gap)
        assert (picks[step, :2] == D).all(axis=1).any()  # the discrete part is a row of the subspace, bit for bit
    record_deviation(f"hybrid_enumeration_value_gap_dc{dc}", worst, 1e-6)


def test_hybrid_search_refuses_more_continuous_dimensions_than_it_was_validated_for():
    from types import SimpleNamespace

    from _replay import HybridSpace
    from baybe_amd.exceptions import IncompatibilityError
    from baybe_amd.recommenders import HipBotorchRecommender

    dc = 5
    disc = pd.DataFrame({"d0": [0.0, 0.5, 1.0]})
    bounds = pd.DataFrame(np.array([[0.0] * dc, [1.0] * dc]), index=["min", "max"], columns=[f"c{a}" for a in range(dc)])
    space = HybridSpace(disc, bounds)
    rng = np.random.default_rng(0)
    M = np.hstack([rng.choice([0.0, 0.5, 1.0], (8, 1)), rng.random((8, dc))])
    meas = pd.DataFrame(M, columns=list(space.comp_rep_columns)).assign(y=-((M - 0.4) ** 2).sum(1))
    objective = SimpleNamespace(targets=(SimpleNamespace(name="y", minimize=False, transformation=None),), is_multi_output=False)
    with pytest.raises(IncompatibilityError, match="continuous parameters"):
        HipBotorchRecommender().recommend(1, space, objective, meas)
