"""Layout replicas of the two BayBE base classes the HIP plug-in subclasses (TEST INFRASTRUCTURE).

BayBE cannot be imported in the build container (no ``cattrs``), so ``tests/test_plugin_layout_cpu.py`` checks
``baybe_amd.plugin.make_baybe_classes`` against stand-ins that reproduce what matters for subclassing:

* ``Surrogate``: a *slotted* ``@define`` class over ``ABC``, a slot-less protocol and a slot-less serialisation
  mixin, with the six ``init=False`` runtime fields and the abstract ``_fit`` / ``_posterior`` hooks
  (``/root/reference/baybe/surrogates/base.py:81-130, 274-306, 467-469``); ``is_available`` is a class-level property.
* ``PureRecommender``: ``@define(slots=False)`` with the three deprecated keyword-only flags and the
  ``recommend -> _recommend_with_discrete_parts(searchspace, batch_size, pending_experiments=...)`` ->
  ``_recommend_discrete`` call flow (``recommenders/pure/base.py:37-140, 248-310``).
* ``BayesianRecommender``: ``@define`` with ``_surrogate_model`` (alias ``surrogate_model``), ``acquisition_function``,
  ``_objective``, ``_botorch_acqf`` and a ``recommend`` that validates, calls ``_setup_botorch_acqf`` and defers to the
  parent (``recommenders/pure/bayesian/base.py:42-69, 130-170``).
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import ClassVar, Protocol, runtime_checkable

from attrs import define, field


class classproperty:
    def __init__(self, fn):
        self.fn = fn

    def __get__(self, _, cls):
        return self.fn(cls)


class SurrogateProtocol(Protocol):
    __slots__ = ()

    def fit(self, searchspace, objective, measurements) -> None: ...

    def to_botorch(self): ...


class SerialMixin:
    __slots__ = ()

    def to_dict(self) -> dict:
        return {"type": type(self).__name__}


@define
class Surrogate(ABC, SurrogateProtocol, SerialMixin):
    supports_transfer_learning: ClassVar[bool]
    supports_multi_output: ClassVar[bool] = False

    _searchspace = field(init=False, default=None, eq=False)
    _objective = field(init=False, default=None, eq=False)
    _measurements = field(init=False, default=None, eq=False)
    _measurements_hash: str = field(init=False, default=None, eq=False)
    _input_scaler = field(init=False, default=None, eq=False)
    _output_scaler = field(init=False, default=None, eq=False)

    @classproperty
    def is_available(cls) -> bool:
        return True

    def to_botorch(self):
        raise RuntimeError("replica: the real base wraps the surrogate into an AdapterModel here")

    def replicate(self):
        raise RuntimeError("replica: the real base builds a CompositeSurrogate here")

    def fit(self, searchspace, objective, measurements) -> None:
        raise RuntimeError("replica: the real base's fit() plumbing must be overridden by the HIP surrogate")

    @abstractmethod
    def _posterior(self, candidates_comp_scaled, /): ...

    @abstractmethod
    def _fit(self, train_x, train_y) -> None: ...


@runtime_checkable
class RecommenderProtocol(Protocol):
    __slots__ = ()

    def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None): ...


@define(slots=False)
class PureRecommender(ABC, RecommenderProtocol):
    compatibility: ClassVar[object]
    supports_discrete_subset_generating_constraints: ClassVar[bool] = False

    _deprecated_allow_repeated_recommendations: bool = field(alias="allow_repeated_recommendations", default=None, kw_only=True)
    _deprecated_allow_recommending_already_measured: bool = field(alias="allow_recommending_already_measured", default=None, kw_only=True)
    _deprecated_allow_recommending_pending_experiments: bool = field(alias="allow_recommending_pending_experiments", default=None, kw_only=True)

    def __attrs_post_init__(self):
        if any(v is not None for v in (self._deprecated_allow_repeated_recommendations,
                                       self._deprecated_allow_recommending_already_measured,
                                       self._deprecated_allow_recommending_pending_experiments)):
            raise RuntimeError("replica: allow_* flags are deprecated")

    def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
        self.calls.append("PureRecommender.recommend")
        return self._recommend_with_discrete_parts(searchspace, batch_size, pending_experiments=pending_experiments)

    def _recommend_discrete(self, subspace_discrete, candidates_exp, batch_size):
        raise NotImplementedError

    def _recommend_with_discrete_parts(self, searchspace, batch_size, pending_experiments):
        raise RuntimeError("replica: the HIP recommender overrides the candidate extraction")


@define
class BayesianRecommender(PureRecommender, ABC):
    _surrogate_model = field(alias="surrogate_model", factory=lambda: (_ for _ in ()).throw(
        RuntimeError("replica: the base default is baybe's GaussianProcessSurrogate")))
    acquisition_function = field(default=None)
    _objective = field(default=None, init=False, eq=False)
    _botorch_acqf = field(default=None, init=False, eq=False)
    calls: list = field(factory=list, init=False, eq=False, repr=False)  # (replica only) trace of the call flow

    def _setup_botorch_acqf(self, searchspace, objective, measurements, pending_experiments=None) -> None:
        raise RuntimeError("replica: the real base builds BoTorch objects here - must be overridden")

    def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None):
        if objective is None:
            raise NotImplementedError("Recommenders of type 'BayesianRecommender' require that an objective is specified.")
        if measurements is None or measurements.empty:
            raise NotImplementedError("Recommenders of type 'BayesianRecommender' do not support empty training data.")
        self.calls.append("BayesianRecommender.recommend")
        self._setup_botorch_acqf(searchspace, objective, measurements, pending_experiments)
        return super().recommend(batch_size=batch_size, searchspace=searchspace, objective=objective,
                                 measurements=measurements, pending_experiments=pending_experiments)
