"""Host-side mirror of BayBE's GP surrogate for the HIP path.

``HipGaussianProcessSurrogate`` satisfies ``baybe.surrogates.base.SurrogateProtocol``
(``surrogates/base.py:49-78``): ``fit(searchspace, objective, measurements)`` with the same
caching / validation behaviour as ``Surrogate.fit`` (``base.py:387-465``), and the read-back
``posterior_stats`` (``base.py:308-384``).  The model it assembles is the BAYBE preset of
``GaussianProcessSurrogate._fit`` (``gaussian_process/core.py:272-341``) — see
``baybe_amd.gp_spec.GPSpec.baybe_default`` — fitted and evaluated by libbaybe_hip.

Search space / objective / targets are duck-typed on the attributes BayBE's own classes expose
(``transform``, ``scaling_bounds``, ``task_idx``, ``n_tasks``, ``targets``, ``minimize``), so real
BayBE objects work unchanged.

Layout: the behaviour lives in field-less mixins with empty ``__slots__`` (``HipGPSurrogateImpl``,
``HipCompositeImpl``); the attrs fields are attached by ``attrs.make_class`` — here onto a bare base
(the stand-alone classes below), and in ``baybe_amd.plugin.make_baybe_classes`` onto BayBE's own slotted
``Surrogate`` (``surrogates/base.py:81-82``), whose runtime fields (``_searchspace``, ``_objective``,
``_measurements_hash``) are then inherited instead of redeclared.  Two slotted attrs bases cannot be combined
("multiple bases have instance lay-out conflict"); one slotted base plus a slot-less mixin can.
"""

from __future__ import annotations

from typing import ClassVar

import numpy as np
import pandas as pd
import attrs
from attrs import field

from baybe_amd import _lib
from baybe_amd.exceptions import IncompatibilityError, IncompatibleSurrogateError, ModelNotTrainedError
from baybe_amd.gp_spec import GPSpec


def _frame_hash(df: pd.DataFrame):
    """Content key of the measurements frame for the "unchanged context -> no refit" test (the reference compares
    ``hash(measurements)``, surrogates/base.py:418-424): column names, dtypes and every column's values (xxh3 over the column
    buffers) - independent of the frame's memory layout and of its index, 40 us for a 70-row frame where pickling the frame for
    ``joblib.hash`` took 0.8 ms of every ``recommend()`` call."""
    try:
        import xxhash
    except ImportError:  # pragma: no cover
        import joblib

        return joblib.hash(df)
    # every part is an xxh3 digest or a plain integer - no ``hash()`` of strings, which is salted per process (PYTHONHASHSEED): the
    # key of a pickled surrogate has to match in the process that loads it, or its first fit there would needlessly retrain
    dtypes = tuple(str(t) for t in df.dtypes)
    head = xxhash.xxh3_64_intdigest(repr((tuple(map(str, df.columns)), dtypes)).encode())
    parts = [df.shape[0], df.shape[1], head]
    if df.size and len(set(dtypes)) == 1 and dtypes[0] != "object":  # one numeric block: a single buffer
        parts.append(xxhash.xxh3_64_intdigest(memoryview(np.ascontiguousarray(df.to_numpy())).cast("B")))
        return tuple(parts)
    for j in range(df.shape[1]):
        a = df.iloc[:, j].to_numpy()
        if a.dtype == object:  # labels (strings), or anything else with a stable repr; unhashable values are fine here
            parts.append(xxhash.xxh3_64_intdigest("\x1f".join(map(repr, a.tolist())).encode()))
        elif a.size:
            parts.append(xxhash.xxh3_64_intdigest(memoryview(np.ascontiguousarray(a)).cast("B")))
    return tuple(parts)


def _has_substance_parameter(searchspace) -> bool:
    """``_dispatch`` of the BAYBE preset (presets/baybe.py:150-171): any ``SubstanceParameter`` in the search space."""
    params = getattr(searchspace, "parameters", None)
    if params is None:
        params = getattr(getattr(searchspace, "discrete", None), "parameters", ()) or ()
    return any(type(prm).__name__ == "SubstanceParameter" for prm in params)


def _has_edbo_encoding(searchspace) -> bool:
    """Any substance parameter with a MORDRED / RDKIT / RDKIT2DDESCRIPTORS encoding (presets/edbo.py:39-54)."""
    params = getattr(getattr(searchspace, "discrete", None), "parameters", ()) or ()
    for prm in params:
        enc = getattr(prm, "encoding", None)
        name = str(getattr(enc, "name", enc)).upper() if enc is not None else ""
        if name in ("MORDRED", "RDKIT", "RDKIT2DDESCRIPTORS") and type(prm).__name__ == "SubstanceParameter":
            return True
    return False


class _Availability:
    """Value of the ``is_available`` class attribute: truthy / falsy like BayBE's ``classproperty``
    (``if not cls.is_available: skip``, ``surrogates/base.py:121-128``, ``tests/test_iterations.py:75-88``) and
    callable for the stand-alone spelling ``cls.is_available()``."""

    def __bool__(self) -> bool:
        return _lib.is_available()

    __call__ = __bool__

    def __repr__(self) -> str:
        return f"<is_available: {bool(self)}>"


class _availability_property:
    def __get__(self, _obj, _cls):
        return _Availability()


def _target_sign(target) -> float:
    """+1 maximise / -1 minimise.  Only identity transformations are on the HIP path (row a9)."""
    tr = getattr(target, "transformation", None)
    if tr is not None and type(tr).__name__ not in ("IdentityTransformation",):
        raise IncompatibilityError(
            f"Target '{target.name}' carries a '{type(tr).__name__}'; the HIP path supports identity "
            f"transformations with optional minimisation only."
        )
    return -1.0 if getattr(target, "minimize", False) else 1.0


def modeled_quantities(objective) -> tuple:
    """The quantities the surrogate models (``Objective._modeled_quantities``, objectives/base.py:55-63): the targets themselves,
    except for ``DesirabilityObjective(as_pre_transformation=True)``, whose ONE modeled quantity is the desirability score computed
    from all targets before fitting (objectives/desirability.py:155-172)."""
    mq = getattr(objective, "_modeled_quantities", None)
    return tuple(mq) if mq is not None else tuple(objective.targets)


def pre_transformed(objective, measurements: pd.DataFrame) -> pd.DataFrame:
    """``objective._pre_transform(measurements, allow_extra=True)`` (surrogates/base.py:454): the columns the model is trained on, named
    after the modeled quantities.  The default pipes the target columns through (objectives/base.py:161-176); the desirability
    objective scalarises them on the host (objectives/desirability.py:322-346) - after which the path IS the single-target one."""
    fn = getattr(objective, "_pre_transform", None)
    return measurements if fn is None else fn(measurements, allow_extra=True)


def _objective_key(objective) -> tuple:
    """What of the objective the trained model depends on (part of the fit-cache key)."""
    targets = tuple((t.name, bool(getattr(t, "minimize", False)), repr(getattr(t, "transformation", None)))
                    for t in objective.targets)
    return (type(objective).__name__, targets, getattr(objective, "as_pre_transformation", None),
            repr(getattr(objective, "weights", None)), str(getattr(objective, "scalarizer", None)))


class HipGPSurrogateImpl:
    """Behaviour of the GP surrogate evaluated on an MI355X (no fields: see the module docstring)."""

    __slots__ = ()

    supports_transfer_learning: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = False
    is_available = _availability_property()
    """False without the shared library or a HIP device (``surrogates/base.py:121-128``)."""

    @classmethod
    def from_preset(cls, preset, **kwargs):
        """``GaussianProcessSurrogate.from_preset`` (gaussian_process/core.py:215-246)."""
        return cls(preset=preset, **kwargs)

    # ---- SurrogateProtocol ---------------------------------------------------------------------
    def fit(self, searchspace, objective, measurements: pd.DataFrame) -> None:
        quantities = modeled_quantities(objective)
        n_targets = len(quantities)
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
            _objective_key(objective),
            self._target_index,
            self.preset,
            _frame_hash(measurements),
        )
        if self._engine is not None and mhash == self._measurements_hash:
            self._searchspace, self._objective = searchspace, objective
            return
        target = quantities[self._tix]
        # the raw target columns this model's quantity is computed from (all of them for a pre-transformed desirability score)
        needs = getattr(objective, "_model_quantities_to_target_names", None)
        names = list(needs[target.name]) if needs is not None else [target.name]
        missing = np.zeros(len(measurements), dtype=bool)
        for nm in names:  # (on the columns' arrays: the frame-level reduction costs 0.6 ms)
            missing |= pd.isna(measurements[nm].to_numpy())
        if missing.any():
            if n_targets == 1:
                raise ValueError(f"Missing target values are not supported: {[t.name for t in objective.targets]}")  # handle_missing_values
            measurements = measurements.loc[~missing]  # composite.py:101-123 per-target filter
        from baybe_amd.engine import HipGP

        comp = searchspace.transform(measurements, allow_extra=True)
        train_x = np.ascontiguousarray(comp.to_numpy(dtype=np.float64))
        train_y = pre_transformed(objective, measurements)[target.name].to_numpy(dtype=np.float64)  # surrogates/base.py:454
        bounds = np.asarray(searchspace.scaling_bounds.to_numpy(), dtype=np.float64)
        task_idx = getattr(searchspace, "task_idx", None)
        n_tasks = int(getattr(searchspace, "n_tasks", 1))
        kernel = resolve_kernel_argument(self.kernel, self.kernel_or_factory, searchspace, train_x, train_y)
        if self.preset != "BAYBE":
            from baybe_amd.gp_spec import from_preset

            if not isinstance(kernel, str) or kernel != "matern52" or self.use_outputscale:
                raise ValueError("a preset fixes the kernel; pass either preset=... or kernel=...")
            spec = from_preset(self.preset, train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks,
                               edbo_encodings=_has_edbo_encoding(searchspace))
        elif isinstance(kernel, str) and kernel == "matern52" and not self.use_outputscale and _has_substance_parameter(searchspace):
            # BayBE{Kernel,Mean,Likelihood}Factory delegate to the Chen components for chemical search spaces
            from baybe_amd.gp_spec import from_preset

            spec = from_preset("CHEN", train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks)
        elif isinstance(kernel, str):
            spec = GPSpec.baybe_default(train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks,
                                        kernel=kernel)
            spec.use_outputscale = bool(self.use_outputscale)
        else:
            from baybe_amd.kernels import apply_kernel_spec

            spec = GPSpec.baybe_default(train_x.shape[1], bounds[0], bounds[1], task_idx=task_idx, n_tasks=n_tasks)
            apply_kernel_spec(spec, kernel, searchspace)
        criterion = resolve_fit_criterion(self.fit_criterion_or_factory, searchspace, train_x, train_y)
        if criterion is not None:
            spec.criterion = criterion
        if self._engine is None:
            self._engine = HipGP(self.device)
        self._fit_on_engine(spec, train_x, train_y)
        self._searchspace, self._objective, self._measurements_hash = searchspace, objective, mhash

    def _fit_on_engine(self, spec, train_x, train_y):
        self._engine.set_model(spec, train_x, train_y)
        if self.fixed_hyperparameters is not None:
            self._engine.factorize(self.fixed_hyperparameters)
            self._fit_info = None
        else:
            # warm_start (opt-in; the reference always restarts from the prior modes): begin L-BFGS-B at the previous optimum
            prev = self._fit_info.params if (self.warm_start and self._fit_info is not None) else None
            if prev is not None:
                from baybe_amd.gp_spec import initial_params, pack_raw

                if pack_raw(spec, prev).shape != pack_raw(spec, initial_params(spec)).shape:
                    prev = None  # the model changed shape (another search space / task count)
            try:
                self._fit_info = self._engine.fit(p0=prev)
            except Exception:
                if prev is None:
                    raise
                self._fit_info = self._engine.fit()

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
        return _target_sign(modeled_quantities(self._objective)[self._tix])

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
        name = modeled_quantities(self._objective)[self._tix].name  # (``_modeled_quantity_names``, surrogates/base.py:345-352)
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

    _composite_class: ClassVar[type] = None  # set below / by make_baybe_classes

    def replicate(self):
        """One independent copy per target (``surrogates/base.py:136-150``, ``composite.py:101-134``)."""
        return type(self)._composite_class(template=self)

    # ---- abstract hooks of baybe.surrogates.base.Surrogate (base.py:274-306, 467-469) ------------------------
    def _fit(self, train_x, train_y) -> None:
        raise NotImplementedError("HipGaussianProcessSurrogate.fit() builds and fits the device model itself.")

    def _posterior_comp(self, candidates_comp, /):
        """BoTorch ``Posterior`` for comp-rep candidates [..., q, d] - compatibility read-back only (needs
        botorch; the hot path never constructs it): t-batches of q = 1 use the fused kernel, q-batches the joint
        posterior."""
        import torch
        from botorch.posteriors import GPyTorchPosterior
        from gpytorch.distributions import MultivariateNormal

        eng = self.engine
        X = torch.as_tensor(candidates_comp, dtype=torch.float64)
        batch, q, d = X.shape[:-2], X.shape[-2], X.shape[-1]
        flat = X.reshape(-1, q, d)
        if q == 1:
            mean, var = eng.posterior(flat[:, 0, :].contiguous())
            mean, cov = mean.cpu().reshape(*batch, 1), var.cpu().clamp_min(0.0).reshape(*batch, 1, 1)
        else:
            ms, cs = zip(*(eng.posterior_joint(x.numpy()) for x in flat))
            mean = torch.as_tensor(np.stack(ms)).reshape(*batch, q)
            cov = torch.as_tensor(np.stack(cs)).reshape(*batch, q, q)
        return GPyTorchPosterior(MultivariateNormal(mean, cov))

    _posterior = _posterior_comp  # no scaler sits in front of the device model (gp/core.py:253-265 returns None)

    def posterior(self, candidates: pd.DataFrame, *, joint: bool = True):
        """``Surrogate.posterior`` (surrogates/base.py:213-247)."""
        import torch

        _ = self.engine  # ModelNotTrainedError before training
        comp = self._searchspace.transform(candidates, allow_extra=True)
        t = torch.as_tensor(np.ascontiguousarray(comp.to_numpy(dtype=np.float64)))
        return self._posterior_comp(t if joint else t.unsqueeze(-2))


class HipCompositeImpl:
    """Per-target replication of a single-output HIP surrogate (``surrogates/composite.py``)."""

    __slots__ = ()

    supports_transfer_learning: ClassVar[bool] = True
    supports_multi_output: ClassVar[bool] = True
    is_available = _availability_property()

    def fit(self, searchspace, objective, measurements: pd.DataFrame) -> None:
        m = len(modeled_quantities(objective))
        if len(self._models) != m:
            # one copy of the template per target with EVERY constructor argument carried over (the reference deep-copies the
            # template, surrogates/composite.py:54-60): kernel / kernel_or_factory / fit_criterion_or_factory / preset / ...
            t = self.template
            self._models = [attrs.evolve(t, target_index=i, fixed_hyperparameters=_per_target(t.fixed_hyperparameters, i))
                            for i in range(m)]
        # the targets' fits side by side: one host thread and one stream each (the reference fits them in sequence,
        # surrogates/composite.py:101-134; the results do not depend on the order - every model has its own handle and data)
        from baybe_amd.engine import fit_side_by_side

        fit_side_by_side([(lambda mod=model: mod.fit(searchspace, objective, measurements)) for model in self._models],
                         device=getattr(self.template, "device", 0))
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


def _per_target(fixed, i):
    """``fixed_hyperparameters`` of a replicated surrogate: one ``GPParams`` for every target, or a sequence
    with one entry per target."""
    return fixed[i] if isinstance(fixed, (list, tuple)) else fixed


def _only_default(name):
    def check(_, attribute, value):
        if value is not None:
            raise IncompatibilityError(f"'{name}': custom mean / likelihood components are gpytorch objects; the HIP path evaluates the "
                                       f"components of the GP presets (preset=...) only.")
    return check


_CRITERIA = {"MARGINAL_LOG_LIKELIHOOD": "mll", "MLL": "mll", "LEAVE_ONE_OUT_PSEUDOLIKELIHOOD": "loo", "LOO": "loo"}


def resolve_kernel_argument(kernel, kernel_or_factory, searchspace, train_x, train_y):
    """The kernel specification a fit uses: ``kernel_or_factory`` if given - a kernel object as is, a factory (any callable that is
    not itself a kernel object: ``KernelFactoryProtocol.__call__(searchspace, train_x, train_y)``, components/kernel.py) called
    with torch tensors - else ``kernel``.  gpytorch kernel objects (which the reference also accepts) are refused."""
    if kernel_or_factory is None:
        return kernel
    k = kernel_or_factory
    if type(k).__name__ == "ICMKernelFactory" and hasattr(k, "base_kernel_factory"):
        # BayBE's own ICMKernelFactory (components/kernel.py:238-337) returns a gpytorch product; its two member factories return
        # BayBE kernel OBJECTS, which is what this path evaluates: call them and keep the product declarative
        import torch

        from baybe_amd.kernels import ProductKernel

        args = (searchspace, torch.as_tensor(train_x), torch.as_tensor(train_y).reshape(-1, 1))

        def member(f):
            try:
                return f(*args)
            except TypeError:  # (searchspace, objective, measurements) signature of newer factories
                return f(searchspace, None, None)

        base_f, task_f = k.base_kernel_factory, k.task_kernel_factory
        base = "BAYBE" if type(base_f).__name__ == "_BayBENumericalKernelFactory" else member(base_f)
        return ProductKernel([base, member(task_f)])
    if callable(k) and not type(k).__name__.endswith("Kernel"):
        import torch

        k = k(searchspace, torch.as_tensor(train_x), torch.as_tensor(train_y).reshape(-1, 1))
    if type(k).__module__.startswith("gpytorch"):
        raise IncompatibilityError("gpytorch kernel objects are not evaluated on the HIP path; pass a BayBE kernel specification.")
    return k


def resolve_fit_criterion(value, searchspace, train_x, train_y):
    """'mll' | 'loo' | None (keep the preset's choice) from a ``FitCriterion`` member / its name / a factory of one
    (components/fit_criterion.py:18-40)."""
    if value is None:
        return None
    if callable(value) and not hasattr(value, "value"):
        import torch

        value = value(searchspace, torch.as_tensor(train_x), torch.as_tensor(train_y).reshape(-1, 1))
    name = str(getattr(value, "value", value)).upper()
    if name not in _CRITERIA:
        raise ValueError(f"unknown fit criterion {value!r}; available: MARGINAL_LOG_LIKELIHOOD, LEAVE_ONE_OUT_PSEUDOLIKELIHOOD")
    return _CRITERIA[name]


def gp_surrogate_fields(with_runtime_state: bool = True) -> dict:
    """attrs fields of the GP surrogate.  ``with_runtime_state=False`` leaves out the fields BayBE's ``Surrogate``
    base already declares (``surrogates/base.py:93-103``)."""
    f = {
        # "matern12" | "matern32" | "matern52" | "rbf" within the BAYBE preset (box constraints, dimension-scaled
        # Gamma priors), or a kernel specification object - baybe_amd.kernels or BayBE's own MaternKernel /
        # RBFKernel / ScaleKernel / ProductKernel - handled like Kernel.to_gpytorch
        "kernel": field(default="matern52"),
        # The reference's constructor arguments (gaussian_process/core.py:149-207), accepted under their own names: a BayBE ``Kernel``
        # object or a kernel factory ``(searchspace, train_x, train_y) -> Kernel`` (takes precedence over ``kernel``); a
        # ``FitCriterion`` member / name or a factory of one.  Custom mean / likelihood components are gpytorch objects, which this
        # path does not evaluate: anything but None is refused at construction.
        "kernel_or_factory": field(default=None, kw_only=True),
        "fit_criterion_or_factory": field(default=None, kw_only=True),
        "mean_or_factory": field(default=None, kw_only=True, validator=_only_default("mean_or_factory")),
        "likelihood_or_factory": field(default=None, kw_only=True, validator=_only_default("likelihood_or_factory")),
        # wrap the base kernel in a ScaleKernel (user kernels); the BAYBE preset has none
        "use_outputscale": field(default=False),
        # GaussianProcessPreset (presets/core.py:8-27): BAYBE | BOTORCH | CHEN | EDBO | EDBO_SMOOTHED | HVARFNER
        "preset": field(default="BAYBE", converter=lambda v: str(getattr(v, "value", v)).upper()),
        "device": field(default=0),  # HIP device ordinal
        "fixed_hyperparameters": field(default=None, eq=False),  # GPParams: skip the fit (parity / benchmarking)
        # start every refit at the previous optimum instead of the prior modes (backtesting loops; not what the reference
        # does, so off by default - see baybe_amd/simulation.py)
        "warm_start": field(default=False, eq=False),
        # runtime state, not part of the specification (pattern of gaussian_process/core.py:211-212)
        "_engine": field(init=False, default=None, eq=False, repr=False),
        "_fit_info": field(init=False, default=None, eq=False, repr=False),
        # which target of a multi-target objective this (replicated) model represents
        "_target_index": field(default=None, eq=False, repr=False),
    }
    if with_runtime_state:
        for name in ("_searchspace", "_objective", "_measurements_hash"):
            f[name] = field(init=False, default=None, eq=False, repr=False)
    return f


def composite_fields(template_factory) -> dict:
    return {
        "template": field(factory=template_factory),
        "_models": field(init=False, factory=list, eq=False, repr=False),
        "_objective": field(init=False, default=None, eq=False, repr=False),
    }


HipGaussianProcessSurrogate = attrs.make_class("HipGaussianProcessSurrogate", gp_surrogate_fields(),
                                               bases=(HipGPSurrogateImpl,), slots=True)
HipGaussianProcessSurrogate.__doc__ = "A Gaussian process surrogate evaluated on an MI355X (stand-alone class)."
HipCompositeSurrogate = attrs.make_class("HipCompositeSurrogate", composite_fields(HipGaussianProcessSurrogate),
                                         bases=(HipCompositeImpl,), slots=True)
HipCompositeSurrogate.__doc__ = "Per-target replication of ``HipGaussianProcessSurrogate`` (stand-alone class)."
for _c in (HipGaussianProcessSurrogate, HipCompositeSurrogate):
    _c.__module__ = __name__
HipGPSurrogateImpl._composite_class = HipCompositeSurrogate
