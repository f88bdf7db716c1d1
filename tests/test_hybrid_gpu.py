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
