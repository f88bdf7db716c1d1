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


qLogEI = qLogExpectedImprovement


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
    name = acqf if isinstance(acqf, str) else type(acqf).__name__
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
        f"The HIP recommender scores qLogEI / qLogNEHVI on the device; '{name}' is not available on this path."
    )
