"""End-to-end regression check on a benchmark domain of the reference:
``hartmann_3d_discretized`` (``/root/reference/benchmarks/domains/hartmann/convergence.py:36-88``) - three
``NumericalDiscreteParameter``s with 25 levels each (15 625 candidates), Hartmann-3 as a minimised target, the scenarios
"Random Recommender" vs the Bayesian recommender, run through ``simulate_scenarios``.  The reference records such runs as
convergence curves; the check here is the curve's content: the GP recommender closes in on the grid optimum and beats
random search by a wide margin."""

import numpy as np
import pandas as pd
import pytest

from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective

pytestmark = pytest.mark.gpu

# Hartmann-3 (the constants of botorch.test_functions.synthetic.Hartmann(dim=3)); global minimum -3.86278 at
# (0.114614, 0.555649, 0.852547)
_ALPHA = np.array([1.0, 1.2, 3.0, 3.2])
_A = np.array([[3.0, 10.0, 30.0], [0.1, 10.0, 35.0], [3.0, 10.0, 30.0], [0.1, 10.0, 35.0]])
_P = 1e-4 * np.array([[3689, 1170, 2673], [4699, 4387, 7470], [1091, 8732, 5547], [381, 5743, 8828]])


def hartmann3(X):
    inner = (_A[None, :, :] * (X[:, None, :] - _P[None, :, :]) ** 2).sum(-1)
    return -(_ALPHA[None, :] * np.exp(-inner)).sum(-1)


class RandomRecommender:
    """Scenario "Random Recommender" of the benchmark: uniform draws from the remaining candidates."""

    def __init__(self, seed):
        self._rng = np.random.default_rng(seed)

    def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
        cand, _ = searchspace.discrete.get_candidates()
        return cand.iloc[self._rng.choice(len(cand), batch_size, replace=False)]


def test_hartmann_3d_discretized_convergence():
    from baybe_amd.recommenders import HipBotorchRecommender
    from baybe_amd.simulation import simulate_scenarios

    assert abs(hartmann3(np.array([[0.114614, 0.555649, 0.852547]]))[0] + 3.86278) < 1e-4
    levels = np.linspace(0, 1, 25)
    space = SearchSpace.from_product([NumericalDiscreteParameter(n, levels) for n in ("x1", "x2", "x3")])
    grid_best = hartmann3(space.discrete.exp_rep.to_numpy(dtype=float)).min()
    assert grid_best < -3.8

    def lookup(df):
        return pd.DataFrame({"target": hartmann3(df[["x1", "x2", "x3"]].to_numpy(dtype=float))}, index=df.index)

    obj = SingleTargetObjective(NumericalTarget("target", minimize=True))
    rng = np.random.default_rng(0)
    inits = []
    for _ in range(2):  # the two-phase default starts from random points: here 5 per Monte-Carlo run
        init = space.discrete.exp_rep.iloc[rng.choice(15625, 5, replace=False)].copy()
        init["target"] = lookup(init)["target"]
        inits.append(init)
    scenarios = {"Random Recommender": Campaign(space, obj, RandomRecommender(1)),
                 "HIP Recommender": Campaign(space, obj, HipBotorchRecommender())}
    res = simulate_scenarios(scenarios, lookup, batch_size=3, n_doe_iterations=10, initial_data=inits, n_mc_iterations=None,
                             random_seed=1337)
    assert len(res) == 2 * 2 * 10 and set(res["Scenario"]) == set(scenarios)
    last = res[res["Iteration"] == 9].groupby("Scenario")["target_CumBest"]
    hip, rnd = last.get_group("HIP Recommender"), last.get_group("Random Recommender")
    # minimisation: CumBest is the lowest value seen; 30 experiments out of 15 625 candidates per run
    assert (hip < grid_best + 0.05).all(), (hip.tolist(), grid_best)  # at (or next to) the grid optimum in every run
    assert hip.mean() < rnd.mean() - 0.2, (hip.tolist(), rnd.tolist())  # random search: -3.2 / -3.5 after 30 draws
    curve = res[(res["Scenario"] == "HIP Recommender") & (res["Initial_Data"] == 0)]["target_CumBest"].to_numpy()
    assert (np.diff(curve) <= 1e-12).all()  # a convergence curve: monotone
