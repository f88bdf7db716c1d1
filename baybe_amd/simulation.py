"""Backtesting driver for the HIP recommend() path.

``simulate_experiment`` mirrors the closed loop of ``baybe/simulation/core.py:27-240`` —
recommend → look up → add measurements, with the ``Iteration / Num_Experiments / <t>_Measurements /
<t>_IterBest / <t>_CumBest`` result frame — for the part of its argument space a discrete GP campaign
uses: dataframe or callable lookups (``simulation/lookup.py:19-150``), ``impute_mode`` "error" /
"ignore" / "worst" / "best" / "mean", an optional initial data set.  It is plain control flow around the
campaign object it is given (BayBE's ``Campaign`` or anything with ``recommend`` / ``add_measurements`` /
``objective``); what is specific to this package is how the device state survives the deep copies the drivers
make (``simulation/core.py:124``, ``scenarios.py:296``, ``transfer_learning.py:78``): a copied recommender SHARES
the device-resident candidate matrix of the original (one upload per search space, however many cases run), a
copied surrogate carries its data and fitted hyper-parameters and rebuilds its device model on first use, and the
handles of finished cases go back to a per-device pool (``baybe_amd.engine.pool_stats``), so the cases of a
scenario run on one handle per target.  Fits restart from the prior mode in every iteration, as in the reference;
warm-starting L-BFGS-B from the previous optimum is opt-in (``HipGaussianProcessSurrogate(warm_start=True)``): on
the multi-modal marginal likelihood it settled in optima up to 3 % worse and did not save evaluations (31 vs 33 on
the 3-parameter test space).
"""

from __future__ import annotations

import warnings
from copy import deepcopy

import numpy as np
import pandas as pd

from baybe_amd.exceptions import NotEnoughPointsLeftError


class NothingToSimulateError(Exception):
    """``baybe.exceptions.NothingToSimulateError``: the loop produced no iteration."""


def _sign(target) -> float:
    return -1.0 if bool(getattr(target, "minimize", False)) else 1.0


def look_up_targets(queries: pd.DataFrame, targets, lookup, impute_mode: str = "error") -> None:
    """Fill the target columns of ``queries`` in place (``simulation/lookup.py:19-150``)."""
    names = [t.name for t in targets]
    if callable(lookup):
        out = lookup(queries)
        queries[out.columns] = out
        return
    if not isinstance(lookup, pd.DataFrame):
        raise ValueError("Unsupported lookup mechanism.")
    pcols = [c for c in queries.columns if c not in names]
    merged = queries[pcols].reset_index().merge(lookup[pcols + names].drop_duplicates(subset=pcols), on=pcols, how="left")
    vals = merged.set_index("index")[names]
    missing = vals.isna().any(axis=1)
    if missing.any():
        if impute_mode == "ignore":
            raise AssertionError("impute_mode 'ignore': the search space was not reduced to the lookup rows.")
        if impute_mode == "error":
            raise IndexError(f"Cannot look up target values for {queries.loc[missing[missing].index[0], pcols].to_dict()}.")
        for t in targets:
            col = lookup[t.name]
            fill = {"worst": col.max() if _sign(t) < 0 else col.min(), "best": col.min() if _sign(t) < 0 else col.max(),
                    "mean": col.mean()}.get(impute_mode)
            if fill is None:
                raise ValueError(f"unsupported impute_mode {impute_mode!r}")
            vals.loc[missing, t.name] = fill
    for n in names:
        queries[n] = vals[n].to_numpy()


def _cumargmax(arr: np.ndarray) -> np.ndarray:
    cummax = np.maximum.accumulate(arr)
    jumps = np.nonzero(arr == cummax)[0]
    out = np.zeros_like(arr, dtype=int)
    out[jumps] = jumps
    return np.maximum.accumulate(out)


def simulate_experiment(campaign, lookup, /, *, batch_size: int = 1, n_doe_iterations: int | None = None,
                        initial_data: pd.DataFrame | None = None, random_seed: int | None = None,
                        impute_mode: str = "error", noise_percent: float | None = None) -> pd.DataFrame:
    """One closed optimisation loop; returns the reference's result frame (``simulation/core.py:64-82``)."""
    if getattr(campaign, "objective", None) is None:
        raise ValueError("The given campaign has no objective defined, hence there are no targets to be tracked.")
    if not (isinstance(lookup, pd.DataFrame) or callable(lookup)):
        raise TypeError("The lookup can either be a pandas dataframe or a callable.")
    if impute_mode == "ignore" and not isinstance(lookup, pd.DataFrame):
        raise ValueError("Impute mode 'ignore' is only available for dataframe lookups.")
    if random_seed is not None:
        import torch

        torch.manual_seed(random_seed)  # governs the MC sampler seed and fit restarts (Settings(random_seed))
        np.random.seed(random_seed)
    campaign = deepcopy(campaign)
    targets = list(campaign.objective.targets)
    if initial_data is not None and not initial_data.empty:
        campaign.add_measurements(initial_data)
    if impute_mode == "ignore":
        pcols = [c for c in lookup.columns if c not in [t.name for t in targets]]
        campaign.toggle_discrete_candidates(lookup[pcols], exclude=True, complement=True)
    limit = n_doe_iterations if n_doe_iterations is not None else np.inf
    k, n_exp, rows = 0, 0, []
    while k < limit:
        try:
            measured = campaign.recommend(batch_size=batch_size)
        except NotEnoughPointsLeftError:
            warnings.warn("The simulation of the campaign ended because not sufficiently many points were left "
                          "for recommendation", UserWarning)
            break
        n_exp += len(measured)
        look_up_targets(measured, targets, lookup, impute_mode)
        rows.append({"Iteration": k, "Num_Experiments": n_exp,
                     **{f"{t.name}_Measurements": measured[t.name].to_list() for t in targets}})
        if noise_percent:  # imperfect execution of the recommendation: relative noise on the numerical parameter values
            from baybe_amd.dataframe import add_parameter_noise

            parameters = getattr(campaign, "parameters", None) or campaign.searchspace.parameters
            add_parameter_noise(measured, parameters, noise_type="relative_percent", noise_level=noise_percent)
        campaign.add_measurements(measured)
        k += 1
    if not rows:
        raise NothingToSimulateError()
    results = pd.DataFrame(rows)
    for t in targets:
        raw = np.array(results[f"{t.name}_Measurements"].tolist(), dtype=float)
        tr = _sign(t) * raw
        it_idx = np.argmax(tr, axis=1)
        iterbest = np.take_along_axis(raw, it_idx[:, None], axis=1)[:, 0]
        results[f"{t.name}_IterBest"] = iterbest
        results[f"{t.name}_CumBest"] = iterbest[_cumargmax(np.max(tr, axis=1))]
    return results


_DEFAULT_SEED = 1337  # baybe/simulation/scenarios.py:23


def _rollout_cases(n_mc_iterations, n_initial_data, random_seed) -> list[dict]:
    """``_Rollouts.cases`` (simulation/scenarios.py:26-91): seeds count up from the first one; with
    ``n_mc_iterations=None`` every initial data set is paired with its own seed, otherwise seeds x data sets."""
    first = _DEFAULT_SEED if random_seed is None else int(random_seed)
    if n_mc_iterations is None:
        if n_initial_data is None:
            raise ValueError(
                "Setting the number of Monte Carlo iterations to `None` requires that initial data is specified. "
                "Perhaps you forgot to do so? If not, consider setting the number of iterations to 1."
            )
        return [{"Random_Seed": first + i, "Initial_Data": i} for i in range(n_initial_data)]
    if int(n_mc_iterations) < 1:
        raise ValueError("n_mc_iterations must be >= 1")
    data = range(n_initial_data) if n_initial_data else [float("nan")]
    return [{"Random_Seed": first + s, "Initial_Data": i} for s in range(int(n_mc_iterations)) for i in data]


def _simulate_partitions(campaign, lookup, groupby, **kwargs) -> pd.DataFrame:
    """``_simulate_groupby`` (simulation/scenarios.py:235-334): with ``groupby`` parameter names the discrete search space
    is split into the groups of equal values of those parameters and the loop is run once per group, recommending from that
    group only; the group's values lead the result rows.  Groups in which nothing can be simulated are skipped."""
    if not groupby:
        return simulate_experiment(campaign, lookup, **kwargs)
    groupby = list(groupby)
    exp = campaign.searchspace.discrete.exp_rep
    param_cols = [c for c in exp.columns]
    frames = []
    for key, part in exp.groupby(groupby, sort=True):
        focused = deepcopy(campaign)
        focused.toggle_discrete_candidates(part[param_cols], exclude=True, complement=True)
        try:
            res = simulate_experiment(focused, lookup, **kwargs)
        except NothingToSimulateError:
            continue
        key = key if isinstance(key, tuple) else (key,)
        head = pd.DataFrame([key] * len(res), columns=groupby, index=res.index)
        frames.append(pd.concat([head, res], axis=1))
    if not frames:
        raise NothingToSimulateError()
    return pd.concat(frames, ignore_index=True)


def simulate_scenarios(scenarios: dict, lookup, /, *, batch_size: int = 1, n_doe_iterations: int | None = None,
                       initial_data: list | None = None, groupby: list | None = None, n_mc_iterations: int | None = 1,
                       random_seed: int | None = None, impute_mode: str = "error",
                       noise_percent: float | None = None) -> pd.DataFrame:
    """``baybe.simulation.scenarios.simulate_scenarios`` (simulation/scenarios.py:94-232) for discrete GP campaigns:
    every scenario (a campaign) is run once per rollout case (random seed x initial data set) through
    ``simulate_experiment``; the result frames are concatenated with the leading columns ``Scenario``, ``Random_Seed``
    and ``Initial_Data`` (NaN without initial data), as the reference's ``unpack_simulation_results`` does.

    The reference fans the cases out to worker processes (xyzpy) when ``parallelize_simulation_runs`` is set; here they
    run one after the other on the device, where a case is a few milliseconds per iteration, and the recommender
    objects of the scenarios keep their device handles (each case works on a deep copy of the campaign's host state
    only).  ``groupby`` partitions the search space as in the reference (one loop per group, the group's values in leading
    columns after ``Initial_Data``), ``noise_percent`` perturbs the numerical parameter values of every measured batch before it
    is added (``add_parameter_noise``)."""
    if not scenarios:
        raise ValueError("no scenarios given")
    if initial_data is not None and len(initial_data) < 1:  # ``_Rollouts.n_initial_data`` carries a ge(1) validator
        raise ValueError("'initial_data' must contain at least one data set when it is given.")
    cases = _rollout_cases(n_mc_iterations, len(initial_data) if initial_data is not None else None, random_seed)
    frames = []
    for name, campaign in scenarios.items():
        for case in cases:
            idx = case["Initial_Data"]
            data = None if initial_data is None else initial_data[int(idx)]
            res = _simulate_partitions(campaign, lookup, groupby, batch_size=batch_size, n_doe_iterations=n_doe_iterations,
                                       initial_data=data, random_seed=case["Random_Seed"], impute_mode=impute_mode,
                                       noise_percent=noise_percent)
            head = pd.DataFrame({"Scenario": name, "Random_Seed": case["Random_Seed"], "Initial_Data": idx}, index=res.index)
            frames.append(pd.concat([head, res], axis=1))
    return pd.concat(frames, ignore_index=True)


def simulate_transfer_learning(campaign, lookup: pd.DataFrame, /, *, batch_size: int = 1, n_doe_iterations: int | None = None,
                               groupby: list | None = None, n_mc_iterations: int = 1,
                               random_seed: int | None = None) -> pd.DataFrame:
    """``baybe.simulation.transfer_learning.simulate_transfer_learning`` (simulation/transfer_learning.py:16-99): the
    search space is partitioned into its tasks, and every task is simulated as its own scenario with the lookup rows of
    all OTHER tasks as training data (``lookup`` is both the loop-closing element and the source of off-task data, hence
    dataframe lookups and discrete spaces only).  Result: the frame of ``simulate_scenarios`` with the tasks in the
    ``Scenario`` column."""
    if not isinstance(lookup, pd.DataFrame):
        raise TypeError("simulate_transfer_learning needs a dataframe lookup (it also supplies the off-task training data).")
    space_type = getattr(getattr(campaign.searchspace, "type", None), "name", "DISCRETE")
    if str(space_type).upper() != "DISCRETE":
        raise NotImplementedError("Currently, only purely discrete search spaces are supported.")
    parameters = getattr(campaign, "parameters", None) or campaign.searchspace.parameters
    task_params = [p for p in parameters if type(p).__name__ == "TaskParameter"]
    if len(task_params) != 1:
        raise NotImplementedError("Currently, transfer learning supports exactly one task parameter.")
    task_param = task_params[0]
    scenarios = {}
    for task in task_param.values:
        # a campaign that recommends for this task only ...
        campaign_task = deepcopy(campaign)
        campaign_task.toggle_discrete_candidates(pd.DataFrame({task_param.name: [task]}), exclude=True, complement=True)
        # ... and knows every measurement of the other tasks
        campaign_task.add_measurements(lookup[lookup[task_param.name] != task])
        scenarios[task] = campaign_task
    return simulate_scenarios(scenarios, lookup, batch_size=batch_size, n_doe_iterations=n_doe_iterations, groupby=groupby,
                              n_mc_iterations=n_mc_iterations, random_seed=random_seed, impute_mode="ignore")
