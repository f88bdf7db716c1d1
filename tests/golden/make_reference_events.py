"""Writes ``tests/golden/reference_events.npz``: the scenarios of ``tests/test_reference_campaign_cpu.py`` that round 4 only had on the
CPU double, recorded at the plug-in boundary from the REFERENCE's own ``Campaign`` so that the GPU box - where
``/root/reference`` does not exist - replays them through ``libbaybe_hip.so`` and checks labels AND values:

  desirability   ``DesirabilityObjective(as_pre_transformation=True)``: recommend + ``posterior_stats`` (objectives/desirability.py:322-346)
  subsets        ``DiscreteBatchConstraint`` -> ``recommend_discrete_with_subsets`` (botorch/discrete.py:21-75), the subset masks recorded
  pending17      16 pending rows + batch 3, and a batch of 17 (joint q-batches beyond 16 points)
  readbacks      ``posterior_stats`` / ``acquisition_values`` (with and without 16 pending rows) / ``joint_acquisition_value`` (campaign.py:676-899)
  task           ``TaskParameter`` with a non-zero active task: recommend + ``posterior_stats`` on candidates of that task
  composite      a user ``ScaleKernel(Matern * RBF)`` surrogate through ``kernel_or_factory``: two batches
  pareto         ``ParetoObjective``: recommend + ``acquisition_values`` (qLogNEHVI read-back)

Run in the build container: ``python tests/golden/make_reference_events.py``.  Recorded per event: kind, the arrays the plug-in was handed
(comp rep, keep-mask, bounds, measurements in the modeled quantities, pending rows, candidate rows, subset masks), torch's RNG state on
entry, and what came back (index labels / values).  The values are the oracle double's under the product's host code
(``tests/_oracle_engine.py``); ``tests/test_reference_events_gpu.py`` holds the device to them at 1e-6.
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

from make_reference_traces import _Patch, _f, _record_fit  # noqa: E402


def main(out_path: Path):
    from _reference import reference_baybe

    reference_baybe()
    import _oracle_engine

    patch = _Patch()
    _oracle_engine.install(patch)
    from baybe import Campaign
    from baybe.constraints import DiscreteBatchConstraint
    from baybe.kernels import MaternKernel, ProductKernel, RBFKernel, ScaleKernel
    from baybe.objectives import DesirabilityObjective, ParetoObjective
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter, TaskParameter
    from baybe.priors import GammaPrior
    from baybe.searchspace import SearchSpace
    from baybe.targets import NumericalTarget

    from baybe_amd.plugin import make_baybe_classes
    from baybe_amd.surrogates import modeled_quantities, pre_transformed

    S, C, R = make_baybe_classes()
    arrays: dict = {}
    scenarios: dict = {}
    current: list = []

    def context(k, searchspace, objective, measurements, pending):
        sd = searchspace.discrete
        comp = sd.comp_rep
        assert isinstance(comp.index, pd.RangeIndex) and comp.index.start == 0 and comp.index.step == 1
        cols = list(comp.columns)
        mask = getattr(sd, "mask_keep", None)
        mask = np.ones(len(comp), bool) if mask is None else np.asarray(mask, bool)
        mq = modeled_quantities(objective)
        names = [q.name for q in mq]
        arrays[k + "_comp"] = comp.to_numpy(dtype=np.float64)
        arrays[k + "_mask"] = mask
        arrays[k + "_bounds"] = searchspace.scaling_bounds[cols].to_numpy(dtype=np.float64)
        arrays[k + "_meas_x"] = searchspace.transform(measurements, allow_extra=True)[cols].to_numpy(dtype=np.float64)
        arrays[k + "_meas_y"] = pre_transformed(objective, measurements)[names].to_numpy(dtype=np.float64)
        has_pend = pending is not None and len(pending) > 0
        if has_pend:
            arrays[k + "_pend"] = searchspace.transform(pending, allow_extra=True)[cols].to_numpy(dtype=np.float64)
        return {"key": k, "columns": cols, "targets": names, "minimize": [bool(getattr(q, "minimize", False)) for q in mq],
                "multi_output": bool(objective.is_multi_output), "task_idx": searchspace.task_idx, "n_tasks": int(searchspace.n_tasks),
                "has_pending": has_pend, "n_subsets": int(getattr(sd, "n_subsets", 0))}

    orig_rec, orig_acq, orig_joint = R.recommend, R.acquisition_values, R.joint_acquisition_value
    orig_stats, orig_cstats = S.posterior_stats, C.posterior_stats

    def rec_recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
        state = torch.get_rng_state().numpy().copy()
        k = f"e{len(arrays)}"
        sd = searchspace.discrete
        seen_masks = []
        cls, inner = type(sd), None
        if getattr(sd, "n_subsets", 0) > 0:  # (slotted attrs class: the spy goes on the class for the duration of the call)
            inner = cls.subset_masks

            def spy(self_, candidates_exp, min_candidates=1):
                out = list(inner(self_, candidates_exp, min_candidates=min_candidates))
                seen_masks.append((candidates_exp.index.to_numpy(), [np.asarray(m, bool) for m in out]))
                return out

            cls.subset_masks = spy
        try:
            out = orig_rec(self, batch_size, searchspace, objective, measurements, pending_experiments)
        finally:
            if inner is not None:
                cls.subset_masks = inner
        ev = context(k, searchspace, objective, measurements, pending_experiments)
        if seen_masks:
            idx, masks = seen_masks[0]
            arrays[k + "_sub_index"] = idx.astype(np.int64)
            arrays[k + "_sub_masks"] = np.stack(masks)
        arrays[k + "_rng"] = state
        arrays[k + "_out"] = np.asarray(out.index, dtype=np.int64)
        _record_fit(arrays, k, self)
        current.append({**ev, "kind": "recommend", "batch_size": int(batch_size)})
        return out

    def rec_acq(self, candidates, searchspace, objective, measurements, pending_experiments=None, acquisition_function=None):
        state = torch.get_rng_state().numpy().copy()
        out = orig_acq(self, candidates, searchspace, objective, measurements, pending_experiments, acquisition_function)
        k = f"e{len(arrays)}"
        ev = context(k, searchspace, objective, measurements, pending_experiments)
        arrays[k + "_cand"] = searchspace.transform(candidates, allow_extra=True)[ev["columns"]].to_numpy(dtype=np.float64)
        arrays[k + "_rng"] = state
        arrays[k + "_out"] = out.to_numpy(dtype=np.float64)
        current.append({**ev, "kind": "acquisition_values"})
        return out

    def rec_joint(self, candidates, searchspace, objective, measurements, pending_experiments=None, acquisition_function=None):
        state = torch.get_rng_state().numpy().copy()
        out = orig_joint(self, candidates, searchspace, objective, measurements, pending_experiments, acquisition_function)
        k = f"e{len(arrays)}"
        ev = context(k, searchspace, objective, measurements, pending_experiments)
        arrays[k + "_cand"] = searchspace.transform(candidates, allow_extra=True)[ev["columns"]].to_numpy(dtype=np.float64)
        arrays[k + "_rng"] = state
        arrays[k + "_out"] = np.asarray([float(out)])
        current.append({**ev, "kind": "joint_acquisition_value"})
        return out

    def stats_recorder(orig):
        def rec_stats(self, candidates, stats=("mean", "std")):
            out = orig(self, candidates, stats)
            if getattr(self, "_recording_outer", True) and not getattr(rec_stats, "busy", False):
                k = f"e{len(arrays)}"
                cols = list(self._searchspace.discrete.comp_rep.columns)
                arrays[k + "_cand"] = self._searchspace.transform(candidates, allow_extra=True)[cols].to_numpy(dtype=np.float64)
                arrays[k + "_out"] = out.to_numpy(dtype=np.float64)
                current.append({"key": k, "kind": "posterior_stats", "columns": cols, "stat_columns": list(out.columns)})
            return out
        return rec_stats

    R.recommend, R.acquisition_values, R.joint_acquisition_value = rec_recommend, rec_acq, rec_joint
    # (a composite's posterior_stats calls its members': only the outermost call is an event)
    comp_stats = stats_recorder(orig_cstats)

    def composite_stats(self, candidates, stats=("mean", "std")):
        single_stats.busy = True
        try:
            out = orig_cstats(self, candidates, stats)
        finally:
            single_stats.busy = False
        k = f"e{len(arrays)}"
        cols = list(self.models[0]._searchspace.discrete.comp_rep.columns)
        arrays[k + "_cand"] = self.models[0]._searchspace.transform(candidates, allow_extra=True)[cols].to_numpy(dtype=np.float64)
        arrays[k + "_out"] = out.to_numpy(dtype=np.float64)
        current.append({"key": k, "kind": "posterior_stats", "columns": cols, "stat_columns": list(out.columns)})
        return out

    single_stats = stats_recorder(orig_stats)
    S.posterior_stats, C.posterior_stats = single_stats, composite_stats
    del comp_stats

    def scenario(name, **extra):
        current.clear()

        def done():
            scenarios[name] = {"events": list(current), **extra}

        return done

    def space3(levels):
        vals = np.arange(levels) / (levels - 1)
        return SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])

    # -- DesirabilityObjective(as_pre_transformation=True): the single-target path on the scalarised column ------------------------------
    done = scenario("desirability")
    rng = np.random.default_rng(31)
    space = space3(6)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(216, 14, replace=False)].copy()
    X = meas.to_numpy()
    meas["gain"] = 100.0 * np.exp(-((X - 0.4) ** 2).sum(1))
    meas["cost"] = 3.0 + 5.0 * X.sum(1)
    targets = [NumericalTarget.normalized_ramp("gain", cutoffs=(20, 100)),
               NumericalTarget.normalized_ramp("cost", cutoffs=(3, 18), descending=True)]
    objective = DesirabilityObjective(targets, weights=[2.0, 1.0], scalarizer="GEOM_MEAN", as_pre_transformation=True)
    camp = Campaign(space, objective, R())
    camp.add_measurements(meas)
    torch.manual_seed(77)
    camp.recommend(3)
    camp.posterior_stats(exp.iloc[:12])
    done()

    # -- DiscreteBatchConstraint: one greedy run per subset, the best joint value wins ------------------------------------------------------
    done = scenario("subsets")
    vals = np.arange(5) / 4
    space = SearchSpace.from_product(
        [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals), CategoricalParameter("plate", ["p", "q", "r"])],
        constraints=[DiscreteBatchConstraint(parameters=["plate"])])
    rng = np.random.default_rng(9)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 9, replace=False)].copy()
    X = meas[["x0", "x1"]].to_numpy(dtype=float)
    meas["yield"] = -((X - 0.5) ** 2).sum(1) + 0.3 * (meas["plate"] == "q")
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    torch.manual_seed(5)
    camp.recommend(3)
    done()

    # -- joint q-batches beyond 16 points: 16 pending rows + a batch of 3, and a batch of 17 ---------------------------------------------
    done = scenario("pending17")
    rng = np.random.default_rng(21)
    space = space3(7)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 12, replace=False)].copy()
    meas["yield"] = _f(meas.to_numpy()) + 0.05 * rng.standard_normal(12)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    pend = exp.drop(index=meas.index).iloc[rng.choice(len(exp) - 12, 16, replace=False)]
    torch.manual_seed(8)
    camp.recommend(3, pending_experiments=pend)
    camp.recommend(17)
    done()

    # -- read-backs of a single-target campaign --------------------------------------------------------------------------------------------
    done = scenario("readbacks")
    rng = np.random.default_rng(3)
    space = space3(6)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 12, replace=False)].copy()
    meas["yield"] = _f(meas.to_numpy()) + 0.05 * rng.standard_normal(12)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    torch.manual_seed(5)
    camp.acquisition_values(exp.iloc[:50])
    camp.posterior_stats(exp.iloc[:50])  # (after a call that carries the fit's context: the replay fits there)
    torch.manual_seed(6)
    camp.joint_acquisition_value(exp.iloc[[3, 40]])
    torch.manual_seed(7)
    camp.acquisition_values(exp.iloc[60:66], pending_experiments=exp.iloc[100:116])
    done()

    # -- transfer learning, the active task is the SECOND task value ---------------------------------------------------------------------------
    done = scenario("task")
    rng = np.random.default_rng(7)
    vals6 = np.arange(6) / 5
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
    camp.posterior_stats(space_t.discrete.exp_rep.iloc[:20])
    done()

    # -- a user kernel through kernel_or_factory ------------------------------------------------------------------------------------------------
    done = scenario("composite", kernel="scale(matern52 * rbf)")
    rng = np.random.default_rng(12)
    space = space3(6)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 15, replace=False)].copy()
    meas["yield"] = _f(meas.to_numpy()) + 0.05 * rng.standard_normal(15)
    kern = ScaleKernel(ProductKernel([MaternKernel(2.5, lengthscale_prior=GammaPrior(3, 1)), RBFKernel(lengthscale_prior=GammaPrior(3, 1))]),
                       outputscale_prior=GammaPrior(2, 0.5))
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R(surrogate_model=S(kernel_or_factory=kern)))
    camp.add_measurements(meas)
    torch.manual_seed(14)
    first = camp.recommend(2)
    first["yield"] = _f(first[["x0", "x1", "x2"]].to_numpy())
    camp.add_measurements(first)
    camp.recommend(2)
    done()

    # -- Pareto objective: recommend and the qLogNEHVI read-back ----------------------------------------------------------------------------------
    done = scenario("pareto")
    rng = np.random.default_rng(4)
    space5 = space3(5)
    rows = space5.discrete.exp_rep.iloc[rng.choice(125, 10, replace=False)].copy()
    X = rows.to_numpy()
    rows["a"] = -((X - 0.25) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    rows["b"] = ((X - 0.75) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    camp = Campaign(space5, ParetoObjective([NumericalTarget("a"), NumericalTarget("b", minimize=True)]), R())
    camp.add_measurements(rows)
    torch.manual_seed(11)
    camp.recommend(2)
    torch.manual_seed(12)
    camp.acquisition_values(space5.discrete.exp_rep.iloc[:40])
    camp.posterior_stats(space5.discrete.exp_rep.iloc[:7])
    done()

    R.recommend, R.acquisition_values, R.joint_acquisition_value = orig_rec, orig_acq, orig_joint
    S.posterior_stats, C.posterior_stats = orig_stats, orig_cstats
    patch.undo()
    np.savez_compressed(out_path, meta=np.frombuffer(json.dumps(scenarios).encode(), dtype=np.uint8), **arrays)
    n_ev = sum(len(v["events"]) for v in scenarios.values())
    print(f"wrote {out_path}: {len(scenarios)} scenarios, {n_ev} events, {out_path.stat().st_size} bytes")
    for name, sc in scenarios.items():
        print("  ", name, [e["kind"] for e in sc["events"]])


if __name__ == "__main__":
    main(Path(sys.argv[1]) if len(sys.argv) > 1 else HERE / "reference_events.npz")
