"""Host-side mirror of BayBE's GP surrogate for the HIP path.

``HipGaussianProcessSurrogate`` satisfies ``baybe.surrogates.base.SurrogateProtocol``
(``surrogates/base.py:49-78``): ``fit(searchspace, objective, measurements)`` with the same
caching / validation behaviour as ``Surrogate.fit`` (``base.py:387-465``), and the read-back
``posterior_stats`` (``base.py:308-384``).  The model it assembles is the BAYBE preset of
``GaussianProcessSurrogate._fit`` (``gaussian_process/core.py:272-341``) — see
``baybe_amd.gp_spec.GPSpec.baybe_default`` — fitted and evaluated by libbaybe_hip.

Search space / objective / targets are duck-typed on the attributes BayBE's own classes expose
(``transform``, ``scaling_bounds``, ``task_idx``, ``n_tasks``, ``targets``, ``minimize``), so real
BayBE objects work unchanged; INTEGRATION.md shows the subclassing stub for serialisation.
"""

from __future__ import annotations

from typing import ClassVar

import numpy as np
import pandas as pd
from attrs import define, field

from baybe_amd import _lib
from baybe_amd.exceptions import IncompatibilityError, IncompatibleSurrogateError, ModelNotTrainedError
from baybe_amd.gp_spec import GPSpec


def _frame_hash(df: pd.DataFrame) -> str:
    import joblib

    return joblib.hash(df)


def _has_edbo_encoding(searchspace) -> bool:
    """Any substance parameter with a MORDRED / RDKIT / RDKIT2DDESCRIPTORS encoding (presets/edbo.py:39-54)."""
    params = getattr(getattr(searchspace, "discrete", None), "parameters", ()) or ()
    for prm in params:
        enc = getattr(prm, "encoding", None)
        name = str(getattr(enc, "name", enc)).upper() if enc is not None else ""
        if name in ("MORDRED", "RDKIT", "RDKIT2DDESCRIPTORS") and type(prm).__name__ == "SubstanceParameter":
            return True
    return False


def _target_sign(target) -> float:
    """+1 maximise / -1 minimise.  Only identity transformations are on the HIP path (row a9)."""
    tr = getattr(target, "transformation", None)
    if tr is not None and type(tr).__name__ not in ("IdentityTransformation",):
        raise IncompatibilityError(
            f"Target '{target.name}' carries a '{type(tr).__name__}'; the HIP path supports identity "
            f"transformations with optional minimisation only."
        )
    return -1.0 if getattr(target, "minimize", False) else 1.0


@define
class HipGaussianProcessSurrogate:
    """A Gaussian process surrogate evaluated on an MI355X (BAYBE preset)."""

    supports_transfer_learning: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = False

    kernel = field(default="matern52")
    """``"matern12" | "matern32" | "matern52" | "rbf"`` within the BAYBE preset (box constraints,
    dimension-scaled Gamma priors), or a kernel specification object — ``baybe_amd.kernels`` or
    BayBE's own ``MaternKernel`` / ``RBFKernel`` / ``ScaleKernel`` — handled like ``Kernel.to_gpytorch``."""

    use_outputscale: bool = field(default=False)
    """Wrap the base kernel in a ScaleKernel (user kernels); the BAYBE preset has none."""

    preset: str = field(default="BAYBE", converter=lambda v: str(getattr(v, "value", v)).upper())
    """``GaussianProcessPreset`` (presets/core.py:8-27): BAYBE | BOTORCH | CHEN | EDBO | EDBO_SMOOTHED |
    HVARFNER — prior tables, constraints and initial values as data (``gp_spec.from_preset``)."""

    device: int = field(default=0)
    """HIP device ordinal."""

    fixed_hyperparameters = field(default=None, eq=False)
    """Optional ``GPParams`` to skip the fit (kernel-parity / benchmarking mode)."""

    # runtime state (not part of the specification; mirrors gaussian_process/core.py:211-212)
    _engine = field(init=False, default=None, eq=False, repr=False)
    _searchspace = field(init=False, default=None, eq=False, repr=False)
    _objective = field(init=False, default=None, eq=False, repr=False)
    _measurements_hash = field(init=False, default=None, eq=False, repr=False)
    _fit_info = field(init=False, default=None, eq=False, repr=False)
    _target_index = field(default=None, eq=False, repr=False)
    """Which target of a multi-target objective this (replicated) model represents (None = the
    single target of a single-target objective)."""

    @classmethod
    def from_preset(cls, preset, **kwargs):
        """``GaussianProcessSurrogate.from_preset`` (gaussian_process/core.py:215-246)."""
        return cls(preset=preset, **kwargs)

    @classmethod
    def is_available(cls) -> bool:
        """False without the shared library or a HIP device (``surrogates/base.py:121-128``)."""
        return _lib.is_available()

    # ---- SurrogateProtocol ---------------------------------------------------------------------
    def fit(self, searchspace, objective, measurements: pd.DataFrame) -> None:
        n_targets = len(objective.targets)
        if n_targets > 1 and not self.supports_multi_output and self._target_index is None:
            raise IncompatibleSurrogateError(
                f"You attempted to train a single-output surrogate in a {n_targets}-target multi-output "
                f"context. Use '.replicate()'."
            )
        # Unchanged context -> no refit (surrogates/base.py:418-424).  The reference compares the
        # search space / objective objects by value; what the model depends on is their encoding
        # (columns, scaling bounds, task column) and the target definition, so that is the key.
        mhash = (
            tuple(searchspace.comp_rep_columns),
            np.asarray(searchspace.scaling_bounds.to_numpy(), dtype=np.float64).tobytes(),
            getattr(searchspace, "task_idx", None),
            int(getattr(searchspace, "n_tasks", 1)),
            tuple((t.name, bool(getattr(t, "minimize", False))) for t in objective.targets),
            self._target_index,
            self.preset,
            _frame_hash(measurements),
        )
        if self._engine is not None and mhash == self._measurements_hash:
            self._searchspace, self._objective = searchspace, objective
            return
        target = objective.targets[self._tix]
        names = [t.name for t in objective.targets]
        if measurements[[target.name]].isna().any().any():
            if n_targets == 1:
                raise ValueError(f"Missing target values are not supported: {names}")  # handle_missing_values
            measurements = measurements.dropna(subset=[target.name])  # composite.py:101-123 per-target filter
        from baybe_amd.engine import HipGP

        comp = searchspace.transform(measurements, allow_extra=True)
        train_x = np.ascontiguousarray(comp.to_numpy(dtype=np.float64))
        train_y = measurements[target.name].to_numpy(dtype=np.float64)
        bounds = np.asarray(searchspace.scaling_bounds.to_numpy(), dtype=np.float64)
        task_idx = getattr(searchspace, "task_idx", None)
        n_tasks = int(getattr(searchspace, "n_tasks", 1))
        if self.preset != "BAYBE":
            from baybe_amd.gp_spec import from_preset

            if not isinstance(self.kernel, str) or self.kernel != "matern52" or self.use_outputscale:
                raise ValueError("a preset fixes the kernel; pass either preset=... or kernel=...")
            spec = from_preset(self.preset, train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks,
                               edbo_encodings=_has_edbo_encoding(searchspace))
        elif isinstance(self.kernel, str):
            spec = GPSpec.baybe_default(train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks,
                                        kernel=self.kernel)
            spec.use_outputscale = bool(self.use_outputscale)
        else:
            from baybe_amd.kernels import apply_kernel_spec

            spec = GPSpec.baybe_default(train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks)
            apply_kernel_spec(spec, self.kernel)
        if self._engine is None:
            self._engine = HipGP(self.device)
        self._engine.set_model(spec, train_x, train_y)
        if self.fixed_hyperparameters is not None:
            self._engine.factorize(self.fixed_hyperparameters)
            self._fit_info = None
        else:
            self._fit_info = self._engine.fit()
        self._searchspace, self._objective, self._measurements_hash = searchspace, objective, mhash

    def to_botorch(self):
        raise IncompatibilityError(
            "HipGaussianProcessSurrogate does not wrap a BoTorch model: the posterior and the acquisition "
            "are evaluated by libbaybe_hip. Use posterior_mean_var() / posterior_stats(), or a "
            "BotorchRecommender with baybe's GaussianProcessSurrogate for BoTorch-only features."
        )

    @property
    def _tix(self) -> int:
        return 0 if self._target_index is None else int(self._target_index)

    # ---- native read-backs ---------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None or self._objective is None:
            raise ModelNotTrainedError("The surrogate must be trained before a posterior can be computed.")
        return self._engine

    @property
    def sign(self) -> float:
        return _target_sign(self._objective.targets[self._tix])

    def posterior_mean_var(self, candidates_comp):
        """(mean, var) device tensors [N] for comp-rep candidates (numpy or torch)."""
        return self.engine.posterior(candidates_comp)

    def posterior_stats(self, candidates: pd.DataFrame, stats=("mean", "std")) -> pd.DataFrame:
        """``Surrogate.posterior_stats`` (surrogates/base.py:308-384): ``<target>_<stat>`` columns."""
        eng = self.engine
        for s in (x for x in stats if isinstance(x, float)):
            if not 0.0 < s < 1.0:
                raise ValueError(
                    f"Posterior quantile statistics can only be computed for quantiles between 0 and 1 "
                    f"(non-inclusive). Provided value: '{s}' as part of '{stats=}'."
                )
        comp = self._searchspace.transform(candidates, allow_extra=True)
        mean, var = eng.posterior(np.ascontiguousarray(comp.to_numpy(dtype=np.float64)))
        mean, var = mean.cpu().numpy(), np.maximum(var.cpu().numpy(), 0.0)
        name = self._objective.targets[self._tix].name
        out = pd.DataFrame(index=candidates.index)
        for s in stats:
            if isinstance(s, float):
                from scipy.stats import norm

                out[f"{name}_Q_{s}"] = mean + np.sqrt(var) * norm.ppf(s)
            elif s == "mean":
                out[f"{name}_mean"] = mean
            elif s == "std":
                out[f"{name}_std"] = np.sqrt(var)
            elif s == "var":
                out[f"{name}_var"] = var
            else:
                raise TypeError(f"The HIP posterior does not support the statistic '{s}'.")
        return out

    def replicate(self):
        """One independent copy per target (``surrogates/base.py:136-150``, ``composite.py:101-134``)."""
        return HipCompositeSurrogate(template=self)


@define
class HipCompositeSurrogate:
    """Per-target replication of a single-output HIP surrogate (``surrogates/composite.py``)."""

    supports_transfer_learning: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = True

    template: HipGaussianProcessSurrogate = field(factory=HipGaussianProcessSurrogate)
    _models: list = field(init=False, factory=list, eq=False, repr=False)
    _objective = field(init=False, default=None, eq=False, repr=False)

    def fit(self, searchspace, objective, measurements: pd.DataFrame) -> None:
        m = len(objective.targets)
        if len(self._models) != m:
            self._models = [
                HipGaussianProcessSurrogate(kernel=self.template.kernel, use_outputscale=self.template.use_outputscale,
                                            preset=self.template.preset, device=self.template.device, target_index=i)
                for i in range(m)
            ]
        for model in self._models:
            model.fit(searchspace, objective, measurements)
        self._objective = objective

    @property
    def models(self):
        if not self._models:
            raise ModelNotTrainedError("The surrogate must be trained first.")
        return self._models

    def to_botorch(self):
        return self.template.to_botorch()

    def posterior_stats(self, candidates: pd.DataFrame, stats=("mean", "std")) -> pd.DataFrame:
        return pd.concat([m.posterior_stats(candidates, stats) for m in self.models], axis=1)
