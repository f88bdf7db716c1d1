"""Writes ``tests/golden/reference_traces.npz``: what the REFERENCE's own objects hand to the plug-in recommender, and what came back.

Run in the build container (``python tests/golden/make_reference_traces.py``), where ``/root/reference/baybe`` imports
(``tests/_reference.py``) but no GPU exists: the real ``baybe.Campaign`` / ``SearchSpace`` / ``TwoPhaseMetaRecommender`` /
``simulate_experiment`` drive ``baybe_amd.plugin.make_baybe_classes()``'s recommender with the oracle standing in for the device
(``tests/_oracle_engine.py``).  Every ``recommend(batch_size, searchspace, objective, measurements, pending_experiments)`` call that
reaches the plug-in is recorded at that boundary:

    comp rep of the discrete subspace [N, d] + its column names, the keep-mask of the call, scaling bounds, task column,
    the measurements (comp-rep columns + target columns), target names / directions, pending rows (comp rep), batch size,
    torch's global RNG state on entry (the MC sampler seeds are drawn from it) -> the returned index labels

``tests/test_reference_replay_gpu.py`` feeds the same calls, in the same order and on one recommender object per scenario, to
``HipBotorchRecommender`` on the GPU box - where the reference tree does not exist - and expects the same labels: the reference's own
call sequence (cached fits, shrinking candidate masks, pending rows, task / Pareto objectives) checked on the hardware path.

The labels stored here are the CPU double's, i.e. the oracle's arithmetic under the product's host code; the CPU suite separately checks
them against the oracle run independently of the product (``tests/test_reference_campaign_cpu.py``).
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE.parent.parent))


class _Patch:
    def __init__(self):
        self._undo = []

    def setattr(self, obj, name, value, raising=True):
        self._undo.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def undo(self):
        for obj, name, old in reversed(self._undo):
            setattr(obj, name, old)


def _f(X):
    return -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * X[:, 0])


def _record_fit(arrays, k, recommender):
    """The fitted hyper-parameters (raw vector per target model) and objective values of the call's surrogate: two complete
    L-BFGS-B runs on a flat criterion end at different points, so the replay holds the device's FIT to the value and
    compares LABELS on the recorded hyper-parameters when the two fits' picks differ."""
    from baybe_amd import gp_spec

    model = recommender._surrogate_model
    subs = list(model.models) if hasattr(model, "models") else [model]
    for i, sub in enumerate(subs):
        eng = sub.engine
        arrays[f"{k}_raw{i}"] = gp_spec.pack_raw(eng.spec, eng.params)
        info = getattr(sub, "_fit_info", None)
        arrays[f"{k}_fun{i}"] = np.asarray([np.nan if info is None else float(info.fun)])


def main(out_path: Path):
    from _reference import reference_baybe

    reference_baybe()
    import _oracle_engine

    patch = _Patch()
    _oracle_engine.install(patch)
    from baybe import Campaign
    from baybe.objectives import ParetoObjective
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter, TaskParameter
    from baybe.recommenders import RandomRecommender, TwoPhaseMetaRecommender
    from baybe.searchspace import SearchSpace
    from baybe.settings import Settings
    from baybe.simulation.core import simulate_experiment
    from baybe.targets import NumericalTarget

    from baybe_amd.plugin import make_baybe_classes

    S, C, R = make_baybe_classes()
    arrays: dict = {}
    scenarios: dict = {}
    current: list = []
    orig = R.recommend

    def recording_recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
        state = torch.get_rng_state().numpy().copy()
        out = orig(self, batch_size, searchspace, objective, measurements, pending_experiments)
        sd = searchspace.discrete
        comp = sd.comp_rep
        assert isinstance(comp.index, pd.RangeIndex) and comp.index.start == 0 and comp.index.step == 1
        cols = list(comp.columns)
        mask = getattr(sd, "mask_keep", None)
        mask = np.ones(len(comp), bool) if mask is None else np.asarray(mask, bool)
        names = [t.name for t in objective.targets]
        k = f"a{len(arrays)}"
        arrays[k + "_comp"] = comp.to_numpy(dtype=np.float64)
        arrays[k + "_mask"] = mask
        arrays[k + "_bounds"] = searchspace.scaling_bounds[cols].to_numpy(dtype=np.float64)
        arrays[k + "_meas_x"] = searchspace.transform(measurements, allow_extra=True)[cols].to_numpy(dtype=np.float64)
        arrays[k + "_meas_y"] = measurements[names].to_numpy(dtype=np.float64)
        if pending_experiments is not None and len(pending_experiments):
            arrays[k + "_pend"] = searchspace.transform(pending_experiments, allow_extra=True)[cols].to_numpy(dtype=np.float64)
        arrays[k + "_rng"] = state
        arrays[k + "_out"] = np.asarray(out.index, dtype=np.int64)
        _record_fit(arrays, k, self)
        current.append({"key": k, "batch_size": int(batch_size), "columns": cols, "targets": names,
                        "minimize": [bool(t.minimize) for t in objective.targets],
                        "multi_output": bool(objective.is_multi_output), "task_idx": searchspace.task_idx,
                        "n_tasks": int(searchspace.n_tasks), "has_pending": k + "_pend" in arrays})
        return out

    R.recommend = recording_recommend

    def scenario(name):
        current.clear()

        def done():
            scenarios[name] = list(current)

        return done

    vals10 = np.arange(10) / 9
    space3 = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals10) for i in range(3)])

    # -- configs[0], both directions: two consecutive batches of a campaign ------------------------------------------------------
    for minimize in (False, True):
        done = scenario("cfg1_min" if minimize else "cfg1_max")
        rng = np.random.default_rng(0)
        exp = space3.discrete.exp_rep
        meas = exp.iloc[rng.choice(len(exp), 20, replace=False)].copy()
        y = _f(meas.to_numpy()) + 0.05 * rng.standard_normal(20)
        meas["yield"] = -y if minimize else y
        camp = Campaign(space3, NumericalTarget("yield", minimize=minimize).to_objective(), R())
        camp.add_measurements(meas)
        torch.manual_seed(1337)
        first = camp.recommend(3)
        camp.recommend(3)
        first["yield"] = (-1 if minimize else 1) * _f(first[["x0", "x1", "x2"]].to_numpy())
        camp.add_measurements(first)
        camp.recommend(2)  # refit with three more measurements
        done()

    # -- pending experiments (tests/test_pending_experiments.py:100-128) ------------------------------------------------------------
    done = scenario("pending")
    rng = np.random.default_rng(1)
    vals6 = np.arange(6) / 5
    space6 = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals6) for i in range(3)])
    camp = Campaign(space6, NumericalTarget("yield").to_objective(), TwoPhaseMetaRecommender(recommender=R()))
    camp.allow_recommending_already_recommended = True
    camp.allow_recommending_already_measured = True
    meas = space6.discrete.exp_rep.iloc[rng.choice(216, 8, replace=False)].copy()
    meas["yield"] = _f(meas.to_numpy()) + 0.05 * rng.standard_normal(8)
    camp.add_measurements(meas)
    with Settings(random_seed=1337):
        rec1 = camp.recommend(3)
    camp.clear_cache()
    with Settings(random_seed=1337):
        camp.recommend(batch_size=3, pending_experiments=rec1)
    done()

    # -- transfer learning: TaskParameter, candidates of the active task ----------------------------------------------------------------
    done = scenario("task")
    rng = np.random.default_rng(7)
    space_t = SearchSpace.from_product([NumericalDiscreteParameter("x0", vals6), NumericalDiscreteParameter("x1", vals6),
                                        TaskParameter("task", ["src", "tgt"], active_values=["tgt"])])
    grid = pd.DataFrame([(a, b) for a in vals6 for b in vals6], columns=["x0", "x1"])
    meas = pd.concat([grid.iloc[rng.choice(36, 14, replace=False)].assign(task="src"),
                      grid.iloc[rng.choice(36, 5, replace=False)].assign(task="tgt")], ignore_index=True)
    X = meas[["x0", "x1"]].to_numpy()
    meas["yield"] = -((X - 0.4) ** 2).sum(1) * np.where(meas["task"] == "src", 0.9, 1.0) + np.where(meas["task"] == "src", 0.2, 0.0)
    camp = Campaign(space_t, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    torch.manual_seed(3)
    camp.recommend(2)
    camp.recommend(1)
    done()

    # -- Pareto objective: two targets, one minimised -------------------------------------------------------------------------------------
    done = scenario("pareto")
    rng = np.random.default_rng(4)
    vals5 = np.arange(5) / 4
    space5 = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals5) for i in range(3)])
    rows = space5.discrete.exp_rep.iloc[rng.choice(125, 10, replace=False)].copy()
    X = rows.to_numpy()
    rows["a"] = -((X - 0.25) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    rows["b"] = ((X - 0.75) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    camp = Campaign(space5, ParetoObjective([NumericalTarget("a"), NumericalTarget("b", minimize=True)]), R())
    camp.add_measurements(rows)
    torch.manual_seed(11)
    camp.recommend(2)
    done()

    # -- the reference's backtesting loop (simulation/core.py:21-239) over a mixed-encoding space -------------------------------------------
    done = scenario("simulate_experiment")
    space_m = SearchSpace.from_product([CategoricalParameter("cat", ["A", "B", "C"], encoding="OHE"),
                                        CategoricalParameter("switch", ["on", "off"], encoding="INT"),
                                        NumericalDiscreteParameter("num", [1.0, 2.0, 4.0, 7.0, 9.0])])

    def lookup(df):
        bonus = df["cat"].map({"A": 0.0, "B": 0.3, "C": 0.1}).to_numpy() + (df["switch"] == "on").to_numpy() * 0.2
        return pd.DataFrame({"t": bonus - ((df["num"].to_numpy() - 4.0) / 8.0) ** 2}, index=df.index)

    camp = Campaign(space_m, NumericalTarget("t").to_objective(),
                    TwoPhaseMetaRecommender(initial_recommender=RandomRecommender(), recommender=R()))
    res = simulate_experiment(camp, lookup, batch_size=2, n_doe_iterations=5, random_seed=59)
    assert len(res) == 5
    done()

    R.recommend = orig
    patch.undo()
    np.savez_compressed(out_path, meta=np.frombuffer(json.dumps(scenarios).encode(), dtype=np.uint8), **arrays)
    n_calls = sum(len(v) for v in scenarios.values())
    print(f"wrote {out_path}: {len(scenarios)} scenarios, {n_calls} recommend() calls, {out_path.stat().st_size} bytes")


if __name__ == "__main__":
    main(Path(sys.argv[1]) if len(sys.argv) > 1 else HERE / "reference_traces.npz")
