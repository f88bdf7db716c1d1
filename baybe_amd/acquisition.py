"""Declarative acquisition functions of the HIP path (mirror of ``baybe/acquisition/acqfs.py``).

Only what the hot path scores on the device is defined: qLogEI (``acqfs.py:219-223``) and the
closed-form/posterior read-backs built from (mean, variance).  BoTorch's defaults that BayBE
does not expose are recorded as explicit fields (sample count 512; fat=True, tau_relu=1e-6,
tau_max=1e-2 are compiled into the kernels)."""

from __future__ import annotations

from typing import ClassVar

from attrs import define, field
from attrs.validators import ge, instance_of


@define(frozen=True)
class qLogExpectedImprovement:
    """Logarithmic Monte-Carlo expected improvement (BoTorch qLogExpectedImprovement)."""

    abbreviation: ClassVar[str] = "qLogEI"
    supports_batching: ClassVar[bool] = True
    supports_pending_experiments: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = False
    is_mc: ClassVar[bool] = True

    n_mc_samples: int = field(default=512, validator=[instance_of(int), ge(1)])
    """Sobol base samples (BoTorch default for qLogEI; BayBE has no knob for it)."""


    kind: ClassVar[str] = "qLogEI"
    is_analytic: ClassVar[bool] = False


qLogEI = qLogExpectedImprovement


def _mc_class(name: str, abbr: str, doc: str, with_beta: bool = False):
    """Declarative MC acquisition function scored by ``bbh_mc_acq_*`` (acqfs.py:180-290)."""
    ns = {
        "__doc__": doc, "abbreviation": abbr, "kind": abbr, "supports_batching": True, "supports_pending_experiments": True,
        "supports_multi_output": False, "is_mc": True, "is_analytic": False,
        "__annotations__": {"abbreviation": ClassVar[str], "kind": ClassVar[str], "supports_batching": ClassVar[bool],
                            "supports_pending_experiments": ClassVar[bool], "supports_multi_output": ClassVar[bool],
                            "is_mc": ClassVar[bool], "is_analytic": ClassVar[bool], "n_mc_samples": int},
        "n_mc_samples": field(default=512, validator=[instance_of(int), ge(1)]),
    }
    if with_beta:
        ns["__annotations__"]["beta"] = float
        ns["beta"] = field(default=0.2, converter=float)
    return define(frozen=True)(type(name, (), ns))


def _analytic_class(name: str, abbr: str, doc: str, with_beta: bool = False, with_maximize: bool = False):
    """Declarative analytic acquisition function scored by ``bbh_analytic_acq`` (q = 1 only)."""
    ns = {
        "__doc__": doc, "abbreviation": abbr, "kind": abbr, "supports_batching": False, "supports_pending_experiments": False,
        "supports_multi_output": False, "is_mc": False, "is_analytic": True,
        "__annotations__": {"abbreviation": ClassVar[str], "kind": ClassVar[str], "supports_batching": ClassVar[bool],
                            "supports_pending_experiments": ClassVar[bool], "supports_multi_output": ClassVar[bool],
                            "is_mc": ClassVar[bool], "is_analytic": ClassVar[bool]},
    }
    if with_beta:
        ns["__annotations__"]["beta"] = float
        ns["beta"] = field(default=0.2, converter=float)
    if with_maximize:
        ns["__annotations__"]["maximize"] = bool
        ns["maximize"] = field(default=True, validator=instance_of(bool))
    return define(frozen=True)(type(name, (), ns))


qExpectedImprovement = _mc_class("qExpectedImprovement", "qEI", "Monte Carlo based expected improvement.")
qProbabilityOfImprovement = _mc_class("qProbabilityOfImprovement", "qPI", "Monte Carlo based probability of improvement.")
qSimpleRegret = _mc_class("qSimpleRegret", "qSR", "Monte Carlo based simple regret.")
qUpperConfidenceBound = _mc_class("qUpperConfidenceBound", "qUCB", "Monte Carlo based upper confidence bound.", with_beta=True)
qPosteriorStandardDeviation = _mc_class("qPosteriorStandardDeviation", "qPSTD", "Monte Carlo based posterior standard deviation.")
PosteriorMean = _analytic_class("PosteriorMean", "PM", "Posterior mean.")
PosteriorStandardDeviation = _analytic_class("PosteriorStandardDeviation", "PSTD", "Posterior standard deviation.", with_maximize=True)
UpperConfidenceBound = _analytic_class("UpperConfidenceBound", "UCB", "Analytical upper confidence bound.", with_beta=True)
ExpectedImprovement = _analytic_class("ExpectedImprovement", "EI", "Analytical expected improvement.")
LogExpectedImprovement = _analytic_class("LogExpectedImprovement", "LogEI", "Logarithmic analytical expected improvement.")
ProbabilityOfImprovement = _analytic_class("ProbabilityOfImprovement", "PI", "Analytical probability of improvement.")
qEI, qPI, qSR, qUCB, qPSTD = (qExpectedImprovement, qProbabilityOfImprovement, qSimpleRegret, qUpperConfidenceBound,
                              qPosteriorStandardDeviation)
PM, PSTD, UCB, EI, LogEI, PI = (PosteriorMean, PosteriorStandardDeviation, UpperConfidenceBound, ExpectedImprovement,
                                LogExpectedImprovement, ProbabilityOfImprovement)
_SINGLE_OUTPUT = {c.abbreviation: c for c in (qExpectedImprovement, qProbabilityOfImprovement, qSimpleRegret,
                                               qUpperConfidenceBound, qPosteriorStandardDeviation, PosteriorMean,
                                               PosteriorStandardDeviation, UpperConfidenceBound, ExpectedImprovement,
                                               LogExpectedImprovement, ProbabilityOfImprovement)}


def _convert_ref(value):
    if value is None or isinstance(value, float):
        return value
    if isinstance(value, int):
        return float(value)
    return tuple(float(v) for v in value)


@define(frozen=True)
class qLogNoisyExpectedHypervolumeImprovement:
    """Logarithmic Monte-Carlo noisy expected hypervolume improvement (``acqfs.py:477-484``)."""

    abbreviation: ClassVar[str] = "qLogNEHVI"
    supports_batching: ClassVar[bool] = True
    supports_pending_experiments: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = True
    is_mc: ClassVar[bool] = True

    reference_point = field(default=None, converter=_convert_ref)
    """None: computed from the measured targets; float: the factor of ``compute_ref_point``;
    iterable: the coordinates themselves (``acqfs.py:337-347``)."""

    prune_baseline: bool = field(default=True, validator=instance_of(bool))
    """Auto-prune baseline points that are unlikely to be Pareto-optimal."""

    n_mc_samples: int = field(default=128, validator=[instance_of(int), ge(1)])
    """Sobol base samples (BoTorch default for multi-objective MC acquisition functions)."""


qLogNEHVI = qLogNoisyExpectedHypervolumeImprovement


def convert_acqf(acqf):
    """``baybe.acquisition.utils.convert_acqf``: accept abbreviations / BayBE objects."""
    if acqf is None or isinstance(acqf, (qLogExpectedImprovement, qLogNoisyExpectedHypervolumeImprovement)):
        return acqf
    if type(acqf) in _SINGLE_OUTPUT.values():
        return acqf
    name = acqf if isinstance(acqf, str) else type(acqf).__name__
    for abbr, cls in _SINGLE_OUTPUT.items():
        if name in (abbr, cls.__name__):
            kw = {}
            for attr in ("beta", "maximize"):
                if not isinstance(acqf, str) and hasattr(acqf, attr) and attr in {a.name for a in cls.__attrs_attrs__}:
                    kw[attr] = getattr(acqf, attr)
            return cls(**kw)
    if name in ("qLogEI", "qLogExpectedImprovement"):
        return qLogExpectedImprovement()
    if name in ("qLogNEHVI", "qLogNoisyExpectedHypervolumeImprovement"):
        kw = {}
        if not isinstance(acqf, str):
            kw = {"reference_point": getattr(acqf, "reference_point", None),
                  "prune_baseline": getattr(acqf, "prune_baseline", True)}
        return qLogNoisyExpectedHypervolumeImprovement(**kw)
    from baybe_amd.exceptions import IncompatibleAcquisitionFunctionError

    raise IncompatibleAcquisitionFunctionError(
        f"The HIP recommender scores the MC / analytic EI-PI-UCB-SR families and qLogNEHVI on the device; '{name}' is not available on this path."
    )
