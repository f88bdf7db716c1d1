"""The drop-in boundary in BayBE's own types: ``baybe_amd.plugin.make_baybe_classes`` builds subclasses of
``Surrogate`` / ``BayesianRecommender`` without the "multiple bases have instance lay-out conflict" of two slotted
attrs bases, constructible as ``cls()`` like BayBE's test loops do (tests/test_iterations.py:75-105).
Runs against layout replicas of the bases (tests/_baybe_layout.py); on a GPU box tests/test_plugin_gpu.py drives the
same classes end to end."""

import numpy as np
import pandas as pd
import pytest

import _baybe_layout as bl
from _baybe_shim import NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd import plugin
from baybe_amd.exceptions import IncompatibilityError, IncompatibleAcquisitionFunctionError


@pytest.fixture(scope="module")
def classes():
    return plugin.make_baybe_classes(bl.Surrogate, bl.BayesianRecommender, "DISCRETE")


def test_two_slotted_attrs_bases_do_conflict_which_is_why_the_mixins_have_no_fields():
    from attrs import define, field

    @define
    class Impl:
        x = field(default=0)

    with pytest.raises(TypeError, match="lay-out conflict"):
        define(type("Broken", (Impl, bl.Surrogate), {}))


def test_subclasses_construct_like_baybes_test_loops_do(classes):
    Sur, Comp, Rec = classes
    s = Sur()
    assert isinstance(s, bl.Surrogate) and issubclass(Sur, bl.Surrogate) and not hasattr(s, "__dict__")  # still slotted
    assert s.preset == "BAYBE" and s.kernel == "matern52" and s._searchspace is None and s._measurements_hash is None
    assert Sur.supports_transfer_learning is True and Sur.supports_multi_output is False
    assert not Sur.is_available and Sur.is_available() is False  # no HIP device here: skipped, not crashing
    assert isinstance(Sur.from_preset("EDBO"), Sur) and Sur.from_preset("EDBO").preset == "EDBO"
    assert s.to_dict() == {"type": "HipGaussianProcessSurrogate"}  # the serialisation mixin of the base is inherited
    c = s.replicate()
    assert isinstance(c, Comp) and c.template is s and Comp.supports_multi_output
    r = Rec()
    assert isinstance(r, bl.BayesianRecommender) and isinstance(r, bl.PureRecommender) and isinstance(r, bl.RecommenderProtocol)
    assert isinstance(r._surrogate_model, Sur) and r.acquisition_function is None and r._objective is None
    assert Rec.compatibility == "DISCRETE" and Rec.supports_discrete_subset_generating_constraints is True
    assert Rec(surrogate_model=Sur(preset="CHEN"), acquisition_function="qUCB", max_n_subsets=3).max_n_subsets == 3
    # the reference's options for continuous / hybrid optimisation are accepted with its defaults and validators (botorch/core.py:69-135):
    # constructor calls carry over, purely discrete optimisation ignores them as the reference does
    assert (r.sequential_continuous, r.hybrid_sampler, r.sampling_percentage, r.n_restarts, r.n_raw_samples) == (True, None, 1.0, 10, 64)
    assert Rec(hybrid_sampler="FPS", sampling_percentage=0.3, n_restarts=3, n_raw_samples=16, sequential_continuous=False).hybrid_sampler == "FPS"
    with pytest.raises(ValueError, match="between 0 and 1"):
        Rec(sampling_percentage=1.5)
    with pytest.raises(ValueError):
        Rec(hybrid_sampler="Farthest")
    with pytest.raises(ValueError):
        Rec(n_restarts=0)
    with pytest.raises(RuntimeError, match="deprecated"):
        Rec(allow_repeated_recommendations=True)  # the base's __attrs_post_init__ still runs
    with pytest.raises(TypeError):
        Rec(Sur())  # keyword-only, as BotorchRecommender (botorch/core.py:46)


def _context():
    vals = np.arange(6) / 5.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(2)])
    meas = space.discrete.exp_rep.iloc[[1, 7, 20, 33]].copy()
    meas["y"] = [0.1, 0.4, 0.3, 0.2]
    return space, SingleTargetObjective(NumericalTarget("y")), meas


def test_baybes_recommend_drives_the_native_overrides(classes):
    """``BayesianRecommender.recommend`` is the base's own method; it must reach the HIP surrogate's ``fit`` through
    the overridden ``_setup_botorch_acqf`` - which, without a HIP device, fails loudly (no CPU fallback)."""
    from baybe_amd import HipUnavailableError

    Sur, _, Rec = classes
    assert Rec.recommend is bl.BayesianRecommender.recommend
    space, obj, meas = _context()
    r = Rec()
    with pytest.raises(NotImplementedError, match="objective"):
        r.recommend(2, space, None, meas)
    with pytest.raises(NotImplementedError, match="empty training data"):
        r.recommend(2, space, obj, pd.DataFrame())
    with pytest.raises(HipUnavailableError):
        r.recommend(2, space, obj, meas)
    assert r.calls == ["BayesianRecommender.recommend"] and r._objective is obj


def test_native_flow_after_the_fit_uses_the_reference_signatures(classes, monkeypatch):
    """With the device work stubbed out: base.recommend -> _setup_botorch_acqf (ours) -> PureRecommender.recommend ->
    _recommend_with_discrete_parts(searchspace, batch_size, pending_experiments=...) (ours) -> exp_rep rows."""
    Sur, _, Rec = classes
    space, obj, meas = _context()
    r = Rec()
    seen = {}

    def fake_setup(self, searchspace, objective, measurements, pending_experiments=None, acquisition_function=None):
        self._objective, self._acqf_in_use, self._pending_comp = objective, self._get_acquisition_function(objective), None
        seen["setup"] = True

    def fake_without_subsets(self, sd, cand, batch_size, return_values=False, keep_mask=None):
        seen["n_candidates"] = len(cand)
        return cand.index[:batch_size]

    monkeypatch.setattr(Rec, "_setup_acqf", fake_setup)
    monkeypatch.setattr(Rec, "_recommend_discrete_without_subsets", fake_without_subsets)
    rec = r.recommend(3, space, obj, meas)
    assert seen == {"setup": True, "n_candidates": 36} and list(rec.index) == [0, 1, 2]
    assert r.calls == ["BayesianRecommender.recommend", "PureRecommender.recommend"]
    # the joint q-batch of qLogEI (the default) holds up to 64 points (the reference has no cap), the other MC functions 16
    with pytest.raises(IncompatibilityError, match="exceeds 64"):
        r.recommend(65, space, obj, meas)


def test_objective_and_acquisition_checks_happen_before_any_fit(classes):
    from _baybe_shim import ParetoObjective

    _, _, Rec = classes
    space, obj, meas = _context()

    class DesirabilityObjective:  # several targets, single output (objectives/desirability.py)
        is_multi_output = False
        targets = (NumericalTarget("y"), NumericalTarget("z"))

    with pytest.raises(IncompatibilityError, match="scalarise"):
        Rec()._setup_botorch_acqf(space, DesirabilityObjective(), meas)
    with pytest.raises(IncompatibleAcquisitionFunctionError, match="multi-output objective"):
        Rec(acquisition_function="qLogNEHVI")._setup_botorch_acqf(space, obj, meas)
    with pytest.raises(IncompatibleAcquisitionFunctionError, match="single-output acquisition"):
        Rec(acquisition_function="qLogEI")._setup_botorch_acqf(space, ParetoObjective([NumericalTarget("y"), NumericalTarget("z")]), meas)

    class Transformed:
        name, minimize = "y", False
        transformation = type("BellTransformation", (), {})()

    with pytest.raises(IncompatibilityError, match="BellTransformation"):
        Rec()._setup_botorch_acqf(space, SingleTargetObjective(Transformed()), meas)


def test_standalone_classes_share_the_implementation(classes):
    from baybe_amd.recommenders import HipBotorchRecommender, HipRecommenderImpl
    from baybe_amd.surrogates import HipGaussianProcessSurrogate, HipGPSurrogateImpl

    Sur, _, Rec = classes
    assert HipGPSurrogateImpl in Sur.__mro__ and HipGPSurrogateImpl in HipGaussianProcessSurrogate.__mro__
    assert HipRecommenderImpl in Rec.__mro__ and HipRecommenderImpl in HipBotorchRecommender.__mro__
    assert Sur.fit is HipGaussianProcessSurrogate.fit and Rec._recommend_discrete is HipBotorchRecommender._recommend_discrete
    assert HipBotorchRecommender().max_n_subsets == 10 and HipBotorchRecommender.compatibility == "HYBRID"


def test_reference_constructor_arguments_of_the_gp_surrogate(classes):
    """``GaussianProcessSurrogate(kernel_or_factory=..., fit_criterion_or_factory=..., mean_or_factory=..., likelihood_or_factory=...)``
    (gaussian_process/core.py:149-207): kernel objects and kernel factories, fit criteria by member / name / factory; custom
    mean / likelihood components (gpytorch objects) are refused at construction."""
    from enum import Enum

    from baybe_amd.kernels import GammaPrior, MaternKernel, ScaleKernel
    from baybe_amd.surrogates import HipGaussianProcessSurrogate, resolve_fit_criterion, resolve_kernel_argument

    Sur, _, _ = classes
    space, obj, meas = _context()
    X, y = np.zeros((4, 2)), np.zeros(4)
    kern = ScaleKernel(MaternKernel(1.5, GammaPrior(3, 1)))
    for cls in (Sur, HipGaussianProcessSurrogate):
        s = cls(kernel_or_factory=kern)
        assert s.kernel_or_factory is kern and s.kernel == "matern52" and s.fit_criterion_or_factory is None
        with pytest.raises(IncompatibilityError, match="gpytorch objects"):
            cls(mean_or_factory=object())
        with pytest.raises(IncompatibilityError, match="gpytorch objects"):
            cls(likelihood_or_factory=object())
    assert resolve_kernel_argument("matern52", None, space, X, y) == "matern52"
    assert resolve_kernel_argument("matern52", kern, space, X, y) is kern  # a kernel object is callable-free: taken as is
    seen = {}

    def factory(searchspace, train_x, train_y):  # KernelFactoryProtocol.__call__
        seen["args"] = (searchspace, tuple(train_x.shape), tuple(train_y.shape))
        return kern

    assert resolve_kernel_argument("matern52", factory, space, X, y) is kern and seen["args"] == (space, (4, 2), (4, 1))

    class FakeGpytorchKernel:
        pass

    FakeGpytorchKernel.__module__ = "gpytorch.kernels.matern_kernel"
    with pytest.raises(IncompatibilityError, match="gpytorch kernel"):
        resolve_kernel_argument("matern52", FakeGpytorchKernel(), space, X, y)

    class FitCriterion(Enum):  # components/fit_criterion.py:18-24
        MARGINAL_LOG_LIKELIHOOD = "MARGINAL_LOG_LIKELIHOOD"
        LEAVE_ONE_OUT_PSEUDOLIKELIHOOD = "LEAVE_ONE_OUT_PSEUDOLIKELIHOOD"

    assert resolve_fit_criterion(None, space, X, y) is None
    assert resolve_fit_criterion(FitCriterion.MARGINAL_LOG_LIKELIHOOD, space, X, y) == "mll"
    assert resolve_fit_criterion("LEAVE_ONE_OUT_PSEUDOLIKELIHOOD", space, X, y) == "loo"
    assert resolve_fit_criterion(lambda sp, tx, ty: FitCriterion.LEAVE_ONE_OUT_PSEUDOLIKELIHOOD, space, X, y) == "loo"
    with pytest.raises(ValueError, match="unknown fit criterion"):
        resolve_fit_criterion("ELBO", space, X, y)


class _StubEngine:
    """Stands in for ``engine.HipGP`` in the host-logic tests below (no device): records the calls ``Surrogate.fit`` makes."""

    created = 0

    def __init__(self, device=0):
        type(self).created += 1
        self.device, self.calls, self.spec = device, [], None

    def set_model(self, spec, X, y):
        self.spec, self.n = spec, len(y)
        self.calls.append(("set_model", X.shape, float(np.sum(y))))

    def fit(self, p0=None, **kw):
        from baybe_amd import gp_spec
        from baybe_amd.engine import FitInfo

        self.calls.append(("fit", p0 is not None))
        return FitInfo(gp_spec.initial_params(self.spec), 0.0, 1, 1, 0, "ok")

    def factorize(self, params):
        self.calls.append(("factorize",))

    def _dev(self):
        import torch

        return torch.device("cpu")


def test_surrogate_fit_host_logic_with_a_stub_engine(monkeypatch):
    """``Surrogate.fit`` (surrogates/base.py:387-465) around the device calls: unchanged context -> no refit (the measurements are
    keyed by content, not by identity, index or memory layout); any changed value, another objective or search-space encoding
    refits; missing target values raise; ``kernel_or_factory`` / ``fit_criterion_or_factory`` reach the model description."""
    import baybe_amd.engine as engine_mod
    from baybe_amd.kernels import GammaPrior, MaternKernel, ScaleKernel
    from baybe_amd.surrogates import HipGaussianProcessSurrogate

    monkeypatch.setattr(engine_mod, "HipGP", _StubEngine)
    space, obj, meas = _context()
    s = HipGaussianProcessSurrogate()
    s.fit(space, obj, meas)
    eng = s._engine
    assert [c[0] for c in eng.calls] == ["set_model", "fit"] and eng.spec.kernel == "matern52" and eng.spec.criterion == "mll"
    s.fit(space, obj, meas.copy())  # equal content, another object
    s.fit(space, obj, meas.set_index(pd.Index([10, 11, 12, 13])))  # another index
    assert len(eng.calls) == 2 and s._engine is eng
    edited = meas.copy()
    edited.loc[edited.index[0], "y"] += 0.5
    s.fit(space, obj, edited)
    assert [c[0] for c in eng.calls] == ["set_model", "fit", "set_model", "fit"] and s._engine is eng  # the same handle refits
    s.fit(space, SingleTargetObjective(NumericalTarget("y", minimize=True)), edited)  # another objective: refit
    assert len(eng.calls) == 6
    with pytest.raises(ValueError, match="Missing target values"):
        s.fit(space, obj, edited.assign(y=[0.1, np.nan, 0.3, 0.2]))
    # warm starts hand the previous optimum to the next fit
    w = HipGaussianProcessSurrogate(warm_start=True)
    w.fit(space, obj, meas)
    w.fit(space, obj, edited)
    assert [c for c in w._engine.calls if c[0] == "fit"] == [("fit", False), ("fit", True)]
    # the reference's constructor arguments
    kern = ScaleKernel(MaternKernel(1.5, GammaPrior(3, 1)), GammaPrior(2, 0.5))
    k = HipGaussianProcessSurrogate(kernel_or_factory=lambda sp, tx, ty: kern, fit_criterion_or_factory="LEAVE_ONE_OUT_PSEUDOLIKELIHOOD")
    k.fit(space, obj, meas)
    assert k._engine.spec.kernel == "matern32" and k._engine.spec.use_outputscale and k._engine.spec.criterion == "loo"
    with pytest.raises(ValueError, match="preset fixes the kernel"):
        HipGaussianProcessSurrogate(preset="EDBO", kernel_or_factory=kern).fit(space, obj, meas)
    fixed = HipGaussianProcessSurrogate(fixed_hyperparameters=object())
    fixed.fit(space, obj, meas)
    assert [c[0] for c in fixed._engine.calls] == ["set_model", "factorize"] and fixed._fit_info is None


def test_recommender_host_logic_with_a_stub_engine(monkeypatch):
    """``Campaign.recommend`` -> ``recommend`` -> ``_recommend_with_discrete_parts`` around the device calls (pure/base.py:210-283,
    botorch/core.py:143-186): the whole comp rep stays resident and is re-uploaded only when its CONTENT changes; the rows that are
    candidates in a call travel as a mask; positions come back as index labels of the search space; measured / recommended /
    pending rows are never among the candidates the engine sees."""
    import torch

    import baybe_amd.engine as engine_mod
    from _baybe_shim import Campaign
    from baybe_amd.engine import GreedyResult
    from baybe_amd.recommenders import HipBotorchRecommender

    seen = {"uploads": 0, "greedy": []}

    class Engine(_StubEngine):
        def best_f(self, *a, **k):
            return 0.0

        def greedy_qlogei(self, Xd, q, alive=None, X_pending=None, **kw):
            live = np.arange(Xd.shape[0]) if alive is None else np.nonzero(alive.numpy())[0]
            seen["greedy"].append((Xd.shape, None if alive is None else int(alive.sum()), None if X_pending is None else len(X_pending)))
            return GreedyResult([int(i) for i in live[-q:]], [0.0] * q)  # (the LAST live rows: positions != labels below)

    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):  # "upload": stay on the CPU, count the calls that move the candidate matrix
        if self.ndim == 2:
            seen["uploads"] += 1
        return self

    monkeypatch.setattr(engine_mod, "HipGP", Engine)
    monkeypatch.setattr(engine_mod, "draw_sampler_seed", lambda: 7, raising=False)
    monkeypatch.setattr(torch.Tensor, "to", to_cpu)
    vals = np.arange(5) / 4.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(2)])
    exp = space.discrete.exp_rep
    rec = HipBotorchRecommender()
    camp = Campaign(space, SingleTargetObjective(NumericalTarget("y")), rec, allow_recommending_already_measured=False)
    meas = exp.iloc[[0, 6, 12]].copy()
    meas["y"] = [0.1, 0.5, 0.3]
    camp.add_measurements(meas)
    got = camp.recommend(2)
    N = len(exp)
    assert seen["uploads"] == 1 and seen["greedy"][-1] == ((N, 2), N - 3, None)  # 3 measured rows masked out, nothing pending
    assert list(got.index) == [N - 2, N - 1] and got.equals(exp.loc[got.index])
    got2 = camp.recommend(2)  # the two recommended rows are no candidates any more; same resident matrix
    assert seen["uploads"] == 1 and seen["greedy"][-1][1] == N - 5 and list(got2.index) == [N - 4, N - 3]
    pend = exp.iloc[[N - 5]]
    got3 = camp.recommend(1, pending_experiments=pend)
    assert seen["greedy"][-1][1:] == (N - 8, 1) and list(got3.index) == [N - 6] and seen["uploads"] == 1
    # stand-alone call with a candidate frame that is a subset with its own labels
    rec2 = HipBotorchRecommender()
    sub = space.filtered(np.isin(np.arange(N), [3, 4, 9, 20]))
    out = rec2.recommend(2, sub, camp.objective, camp.measurements)
    assert list(out.index) == [9, 20]
    with pytest.raises(Exception):  # more rows than candidates (NotEnoughPointsLeftError)
        rec2.recommend(5, sub, camp.objective, camp.measurements)
    monkeypatch.setattr(torch.Tensor, "to", real_to)


def test_replicated_surrogate_carries_every_constructor_argument(monkeypatch):
    """``HipCompositeImpl.fit`` (surrogates/composite.py:54-60, 101-123: deep copies of the template): a multi-target fit of a template
    built with ``kernel_or_factory`` / ``fit_criterion_or_factory`` / a preset reaches the model description of EVERY target with
    those choices (they were dropped and every target silently fitted the default Matern-5/2 + MLL)."""
    import baybe_amd.engine as engine_mod
    from baybe_amd.kernels import GammaPrior, MaternKernel, ScaleKernel
    from baybe_amd.surrogates import HipGaussianProcessSurrogate

    monkeypatch.setattr(engine_mod, "HipGP", _StubEngine)
    space, obj, meas = _context()
    meas = meas.assign(y2=[0.4, 0.1, 0.2, 0.3])

    class _Obj:
        is_multi_output = True
        targets = (NumericalTarget("y"), NumericalTarget("y2", minimize=True))

    kern = ScaleKernel(MaternKernel(1.5, GammaPrior(3, 1)), GammaPrior(2, 0.5))
    comp = HipGaussianProcessSurrogate(kernel_or_factory=kern, fit_criterion_or_factory="LEAVE_ONE_OUT_PSEUDOLIKELIHOOD",
                                       warm_start=True).replicate()
    comp.fit(space, _Obj(), meas)
    assert len(comp.models) == 2
    for i, m in enumerate(comp.models):
        assert m._target_index == i and m.warm_start
        assert m._engine.spec.kernel == "matern32" and m._engine.spec.use_outputscale and m._engine.spec.criterion == "loo"
    edbo = HipGaussianProcessSurrogate(preset="EDBO").replicate()
    edbo.fit(space, _Obj(), meas)
    assert all(m.preset == "EDBO" and m._engine.spec.use_outputscale for m in edbo.models)


def test_measurement_key_is_process_independent():
    """The fit-cache key of the measurements (``_frame_hash``) holds no salted ``hash()`` of strings: the same frame keys identically
    in another interpreter (a pickled surrogate must not retrain on its first fit after loading); unhashable cell values are fine."""
    import subprocess
    import sys
    from pathlib import Path

    ROOT = Path(__file__).resolve().parents[1]
    code = ("import pandas as pd, sys; sys.path.insert(0, %r); from baybe_amd.surrogates import _frame_hash; "
            "print(_frame_hash(pd.DataFrame({'a': [1.0, 2.0], 'lab': ['u', 'v'], 'y': [0.5, 0.25]})))") % str(ROOT)
    outs = {subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True,
                           env={**__import__("os").environ, "PYTHONHASHSEED": seed}).stdout for seed in ("1", "2", "3")}
    assert len(outs) == 1
    from baybe_amd.surrogates import _frame_hash

    a = pd.DataFrame({"x": [1.0, 2.0], "lists": [[1, 2], [3]]})
    b = pd.DataFrame({"x": [1.0, 2.0], "lists": [[1, 2], [4]]})
    assert _frame_hash(a) == _frame_hash(a.copy()) and _frame_hash(a) != _frame_hash(b)
