"""Host-side mirror of ``BotorchRecommender`` for purely discrete search spaces, on the HIP path.

Same call structure as the reference:
``recommend`` (``pure/bayesian/base.py:130-197``) -> fit the surrogate and set up the acquisition
(``_setup_botorch_acqf``, ``base.py:87-111``; best_f / X_pending semantics of
``acquisition/_builder.py:195-334``) -> ``_recommend_with_discrete_parts``
(``pure/base.py:248-310``) -> ``_recommend_discrete`` (``botorch/core.py:155-184``) ->
``recommend_discrete_without_subsets`` (``botorch/discrete.py:78-142``), whose
``optimize_acqf_discrete`` call is replaced by ``HipGP.greedy_qlogei``.

Differences that are deliberate: the comp-rep rows of the
candidates are taken positionally from ``subspace_discrete.comp_rep`` (no re-encoding of N rows,
no float-key merge), so exactly ``batch_size`` index labels are returned even if the search
space contains duplicate rows.
"""

from __future__ import annotations

from typing import ClassVar

import attrs
import numpy as np
import pandas as pd
from attrs import field
from attrs.validators import ge, gt, instance_of

from baybe_amd import _lib
from baybe_amd.acquisition import convert_acqf, qLogExpectedImprovement, qLogNoisyExpectedHypervolumeImprovement
from baybe_amd.engine import draw_sampler_seed, sobol_normal_base_samples
from baybe_amd.exceptions import (
    IncompatibilityError,
    IncompatibleAcquisitionFunctionError,
    NotEnoughPointsLeftError,
)
from baybe_amd.surrogates import HipCompositeImpl, HipGaussianProcessSurrogate, _availability_property


MAX_HYBRID_CONTINUOUS = 4  # continuous parameters of a hybrid / continuous space the derivative-free search is validated for


def _hash_buffers(bufs) -> int:
    """64-bit content key of a list of byte buffers: the library's host-side multiply-fold hash over 4 MB pieces on native threads
    (``bbh_content_key``).  python-xxhash keeps the GIL, so the "8 threads" of round 4 hashed the 160 MB comp rep of a 1e6 x 20 grid
    at one core's rate - 5 ms of every ``recommend()``; natively it is memory-bound."""
    import ctypes as C
    import os

    from baybe_amd import _lib

    lib = _lib.load_library()
    n = len(bufs)
    arrs = [np.frombuffer(b, dtype=np.uint8) for b in bufs]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    lens = (C.c_int64 * n)(*[a.size for a in arrs])
    total = sum(a.size for a in arrs)
    threads = 1 if total < (1 << 22) else max(1, min(32, os.cpu_count() or 1))
    return int(lib.bbh_content_key(ptrs, lens, n, threads))


def _content_hash(arr: np.ndarray):
    """Content hash of a numeric array: xxh3 over its buffer (the transposed view if that is the contiguous one - a
    single-dtype DataFrame hands out its block that way - so nothing is copied), large buffers in pieces on a thread
    pool (xxhash releases the GIL)."""
    arr = np.asarray(arr)
    if arr.size == 0:
        return hash((arr.shape, str(arr.dtype)))
    if arr.dtype == object:
        return hash((arr.shape, tuple(arr.ravel().tolist())))
    if not arr.flags.c_contiguous:
        arr = arr.T if arr.T.flags.c_contiguous else np.ascontiguousarray(arr)
    return _hash_buffers([memoryview(arr).cast("B")])


def _frame_content_hash(df: pd.DataFrame):
    """Content hash of a DataFrame's values without materialising them as one array: every column's buffer in pieces on the
    thread pool; columns that are not plain numeric arrays go through ``_content_hash``.  Equal keys
    mean equal content up to the collision chance of a 64-bit hash (2^-64 per pair of contents); the same content in another memory layout (a row-major block against a column-major copy) may key
    differently, which costs one re-upload and nothing else."""
    if df.shape[1] == 0 or not df.iloc[:, 0].to_numpy().flags.c_contiguous:
        return _content_hash(df.to_numpy())  # one 2-D block: ``to_numpy`` is a view, its columns are strided
    cols = [df.iloc[:, j].to_numpy() for j in range(df.shape[1])]
    plain = [a for a in cols if a.dtype != object and a.flags.c_contiguous]
    other = [a for a in cols if not (a.dtype != object and a.flags.c_contiguous)]
    return hash((_hash_buffers([memoryview(a).cast("B") for a in plain]), tuple(_content_hash(a) for a in other),
                 tuple(a.dtype != object and a.flags.c_contiguous for a in cols)))


def _is_multi_output(objective) -> bool:
    """``Objective.is_multi_output`` (objectives/base.py): False for single-target AND desirability objectives."""
    flag = getattr(objective, "is_multi_output", None)
    return bool(flag) if flag is not None else len(objective.targets) > 1


def _check_objective(objective, acqf) -> None:
    """What of ``baybe.objectives`` is on the device path, checked before anything is fitted: single targets and
    Pareto stacking of identity / negated targets.  A ``DesirabilityObjective`` is
    single-output in the reference (scalarised, optimised with qLogEI, ``objectives/desirability.py:222-265``) -
    treating it as Pareto because it has several targets would answer a different question."""
    kind = type(objective).__name__
    from baybe_amd.surrogates import _target_sign, modeled_quantities

    quantities = modeled_quantities(objective)
    if len(quantities) > 1 and not _is_multi_output(objective):
        # one model per target + a per-sample scalarisation objective (objectives/desirability.py:229-265): not a device path
        raise IncompatibilityError(
            f"Objectives of type '{kind}' scalarise several modeled targets per posterior sample; the HIP path supports "
            f"single-target and Pareto objectives and 'DesirabilityObjective(as_pre_transformation=True)', which scalarises the "
            f"measurements before fitting. Use BotorchRecommender otherwise."
        )
    if _is_multi_output(objective) and not acqf.supports_multi_output:
        raise IncompatibleAcquisitionFunctionError(
            f"You attempted to use a single-output acquisition function in a "
            f"{len(quantities)}-target multi-output context."
        )
    if not _is_multi_output(objective) and acqf.supports_multi_output:
        raise IncompatibleAcquisitionFunctionError(
            f"The acquisition function '{type(acqf).__name__}' needs a multi-output objective, but a single-target "
            f"objective was given."
        )
    for target in quantities:  # identity transformations (+ minimisation) only: raises IncompatibilityError
        _target_sign(target)


def _check_continuous_part(cont) -> None:
    """What of ``SubspaceContinuous`` the HIP path's search handles: box-bounded parameters.  Linear / nonlinear / cardinality /
    interpoint constraints need the reference's constrained optimiser."""
    for name in ("constraints_lin_eq", "constraints_lin_ineq", "constraints_nonlin", "constraints_cardinality"):
        if len(getattr(cont, name, ()) or ()):
            raise IncompatibilityError(
                f"The continuous subspace carries '{name}'; the HIP path searches box-bounded continuous parameters only. "
                f"Use BotorchRecommender."
            )
    dc = len(getattr(getattr(cont, "comp_rep_bounds", None), "columns", ()))
    if dc > MAX_HYBRID_CONTINUOUS:  # (before anything is fitted)
        raise IncompatibilityError(
            f"{dc} continuous parameters: the HIP path searches the continuous box derivative-free (Sobol raw samples + pattern "
            f"search) and is held to the reference's enumeration (optimize_acqf_mixed) for up to {MAX_HYBRID_CONTINUOUS} of them "
            f"(tests/test_hybrid_gpu.py); use BotorchRecommender, whose gradient optimiser is the tool beyond that.")


class HipRecommenderImpl:
    """Behaviour of the Bayesian recommender scoring the full discrete candidate set on an MI355X.  No fields: they
    are attached by ``attrs.make_class`` - below for the stand-alone class, in ``baybe_amd.plugin`` on top of BayBE's
    ``BayesianRecommender``, whose ``recommend`` (``pure/bayesian/base.py:130-197``) then drives the overrides
    ``_setup_botorch_acqf`` / ``_recommend_with_discrete_parts`` / ``_recommend_discrete`` defined here."""

    __slots__ = ()

    supports_discrete_subset_generating_constraints: ClassVar[bool] = True
    is_available = _availability_property()

    # ---- copies (simulation/core.py:124, scenarios.py:296, transfer_learning.py:78 deep-copy whole campaigns) --------
    _SHARED_ON_COPY: ClassVar[tuple] = ("_cand_cache", "shard")  # resident candidate matrix (up to 160 MB of HBM): shared, read-only
    _DROPPED_ON_COPY: ClassVar[tuple] = ("_nehvi",)  # per-call acquisition state holding device handles: rebuilt by every recommend()

    def __deepcopy__(self, memo):
        from copy import deepcopy

        cls = type(self)
        new = cls.__new__(cls)
        memo[id(self)] = new
        for a in attrs.fields(cls):
            val = getattr(self, a.name)
            if a.name in self._DROPPED_ON_COPY:
                val = None
            elif a.name not in self._SHARED_ON_COPY:
                val = deepcopy(val, memo)
            object.__setattr__(new, a.name, val)
        return new

    def __getstate__(self):
        """Pickling: as ``__deepcopy__``, but the device-resident candidate matrix stays behind too (it is uploaded again
        from the search space on first use)."""
        return {a.name: (None if a.name in self._DROPPED_ON_COPY or a.name == "_cand_cache" else getattr(self, a.name))
                for a in attrs.fields(type(self))}

    def __setstate__(self, state):
        for k, v in state.items():
            object.__setattr__(self, k, v)

    # ---- BayesianRecommender surface -----------------------------------------------------------
    def _get_acquisition_function(self, objective, override=None):
        """Native acquisition spec for the context (default choice: pure/bayesian/base.py:70-74); BayBE's own
        acquisition objects are mapped by class name (``baybe_amd.acquisition.convert_acqf``)."""
        chosen = override if override is not None else self.acquisition_function
        if chosen is None:
            return qLogNoisyExpectedHypervolumeImprovement() if _is_multi_output(objective) else qLogExpectedImprovement()
        return convert_acqf(chosen)

    def get_surrogate(self, searchspace, objective, measurements):
        if _is_multi_output(objective) and not self._surrogate_model.supports_multi_output:
            self._surrogate_model = self._surrogate_model.replicate()  # pure/bayesian/base.py:35-39
        self._surrogate_model.fit(searchspace, objective, measurements)
        if self.shard is not None and self.shard.world > 1:
            # every rank fitted the same data, but a fit retry re-samples its start point from the priors (private
            # RNG): rank 0's hyper-parameters are the model on every rank
            model = self._surrogate_model
            for sub in (model.models if isinstance(model, HipCompositeImpl) else [model]):
                params = self.shard.agree(sub.engine.params)
                if self.shard.rank != 0:
                    sub.engine.factorize(params)
        return self._surrogate_model

    def _setup_botorch_acqf(self, searchspace, objective, measurements, pending_experiments=None) -> None:
        """Override point of ``BayesianRecommender.recommend`` (base.py:87-111, called at base.py:166-168): the native
        set-up instead of ``acqf.to_botorch(...)``."""
        self._setup_acqf(searchspace, objective, measurements, pending_experiments)

    def _setup_acqf(self, searchspace, objective, measurements, pending_experiments=None, acquisition_function=None):
        """Native counterpart of ``_setup_botorch_acqf``: fit (cached), best_f, pending rows."""
        cont = getattr(searchspace, "continuous", None)
        if cont is not None and not getattr(cont, "is_empty", True):
            _check_continuous_part(cont)  # hybrid / continuous spaces: box-bounded continuous parameters only
        self._objective = objective
        acqf = self._get_acquisition_function(objective, acquisition_function)
        _check_objective(objective, acqf)
        self._acqf_in_use = acqf
        if pending_experiments is not None and not acqf.supports_pending_experiments:
            raise IncompatibleAcquisitionFunctionError(
                f"The chosen acquisition function of type '{type(acqf).__name__}' does not support pending experiments."
            )
        surrogate = self.get_surrogate(searchspace, objective, measurements)
        self._pending_comp = None
        if pending_experiments is not None and len(pending_experiments):
            pend = searchspace.transform(pending_experiments, allow_extra=True)  # _builder.py:326-334
            self._pending_comp = np.ascontiguousarray(pend.to_numpy(dtype=np.float64))
        self._nehvi = None
        if _is_multi_output(objective):
            from baybe_amd.nehvi import HipNEHVI, compute_ref_point

            models = surrogate.models
            signs = np.array([m.sign for m in models])
            names = [t.name for t in objective.targets]
            ref = acqf.reference_point
            if not isinstance(ref, tuple):  # _builder.py:301-317: from completely measured rows
                complete = measurements[names].dropna()
                if complete.empty:
                    raise ValueError(
                        "For calculating a default reference point, at least one configuration must have a "
                        "measured value for all targets. Set 'reference_point' explicitly."
                    )
                kw = {} if ref is None else {"factor": ref}
                ref = compute_ref_point(complete.to_numpy(dtype=np.float64) * signs[None, :], **kw)
            X_base = np.ascontiguousarray(searchspace.transform(measurements, allow_extra=True).to_numpy(dtype=np.float64))
            self._nehvi = HipNEHVI([m.engine for m in models], signs, X_base, np.asarray(ref, dtype=np.float64),
                                   n_mc_samples=acqf.n_mc_samples, prune_baseline=acqf.prune_baseline,
                                   device=models[0].engine.device)
            self._best_f = None
        else:
            self._best_f = surrogate.engine.best_f(surrogate.sign)  # _builder.py:141-161, 256-265
        return surrogate, acqf

    def _check_batch_size(self, batch_size, pending_experiments=None) -> None:
        """The joint q'-batch kernels hold 1 + (pending + earlier picks) <= 16 points; the reference has no such
        limit, so say so before any candidate is scored (single-target MC acquisition functions only: qLogNEHVI
        caches picks into its baseline and analytic functions have q = 1)."""
        n_pend = len(self._pending_comp) if self._pending_comp is not None else (
            0 if pending_experiments is None else len(pending_experiments))
        acqf = self._acqf_in_use
        if acqf is not None and (acqf.supports_multi_output or getattr(acqf, "is_analytic", False)):
            return
        # qLogEI: up to 64 points (beyond 16 through bbh_qlogei_pending_big, the factor in a global workspace); the other MC
        # acquisition functions: 16
        cap = (_lib.MAX_PENDING_BIG if getattr(acqf, "kind", None) == "qLogEI" else _lib.MAX_PENDING) + 1
        if batch_size + n_pend > cap:
            raise IncompatibilityError(
                f"batch_size ({batch_size}) + pending experiments ({n_pend}) exceeds {cap}, the largest "
                f"joint q-batch of the HIP kernels; recommend in smaller batches (marking earlier ones as pending)."
            )

    def recommend(self, batch_size, searchspace, objective=None, measurements=None, pending_experiments=None) -> pd.DataFrame:
        if objective is None:
            raise NotImplementedError(
                "Recommenders of type 'BayesianRecommender' require that an objective is specified."
            )
        if measurements is None or measurements.empty:
            raise NotImplementedError("Recommenders of type 'BayesianRecommender' do not support empty training data.")
        self._acqf_in_use = self._get_acquisition_function(objective)
        self._pending_comp = None
        self._check_batch_size(batch_size, pending_experiments)  # before the fit
        self._setup_botorch_acqf(searchspace, objective, measurements, pending_experiments)
        return self._recommend_with_discrete_parts(searchspace, batch_size, pending_experiments=pending_experiments)

    def _recommend_with_discrete_parts(self, searchspace, batch_size, pending_experiments=None) -> pd.DataFrame:
        sd = searchspace.discrete
        self._check_batch_size(batch_size)
        cont = getattr(searchspace, "continuous", None)
        if cont is not None and not getattr(cont, "is_empty", True):  # hybrid space (pure/base.py:268-303)
            if getattr(sd, "is_empty", False):
                return self._recommend_hybrid(searchspace, pd.DataFrame(), batch_size)
            candidates_exp, _ = sd.get_candidates()
            return self._recommend_hybrid(searchspace, candidates_exp, batch_size)
        mask = getattr(sd, "mask_keep", None)  # FilteredSubspaceDiscrete (searchspace/_filtered.py:14-44)
        if (isinstance(mask, np.ndarray) and mask.dtype == bool and len(mask) == len(sd.exp_rep)
                and getattr(sd, "n_subsets", 0) == 0):
            # The candidate set is "the rows of the resident matrix where the keep-mask is set": the mask goes to
            # the device as it is and the two N-row dataframe copies of get_candidates() (50 ms at 1e6 rows) are
            # not made.  Same candidates, same order, same first-index ties.
            if int(mask.sum()) < batch_size:
                raise NotEnoughPointsLeftError(
                    f"Using the current settings, there are fewer than {batch_size} possible data points left to recommend."
                )
            idxs = self._recommend_discrete_without_subsets(sd, None, batch_size, keep_mask=mask)
            return sd.exp_rep.loc[idxs, :]
        candidates_exp, _ = sd.get_candidates()
        if len(candidates_exp) < batch_size:
            raise NotEnoughPointsLeftError(
                f"Using the current settings, there are fewer than {batch_size} possible data points left to recommend."
            )
        idxs = self._recommend_discrete(searchspace.discrete, candidates_exp, batch_size)
        return searchspace.discrete.exp_rep.loc[idxs, :]

    # ---- discrete optimisation -----------------------------------------------------------------
    def _sampler_seed(self) -> int:
        """MC sampler seed from torch's global RNG (``engine.draw_sampler_seed``); with row shards rank 0's draw."""
        seed = draw_sampler_seed()
        return int(self.shard.agree(seed)) if self.shard is not None else seed

    def _candidates_on_device(self, subspace_discrete, candidates_exp, keep_mask=None):
        """(X_dev, alive, labels): the *whole* comp rep of the discrete subspace as a device-resident
        fp64 matrix (uploaded once per search space, cached), plus a uint8 mask of the rows that are
        candidates in this call (``None`` = all) and the index labels of the resident rows.

        Campaign.recommend shrinks the candidate set by a few rows per iteration
        (``campaign.py:549-572``); masking instead of re-encoding / re-uploading N x d doubles removes
        the O(N) pandas work and the PCIe copy from every call after the first.
        Row order = search-space order, so first-index tie-breaking is unchanged."""
        import torch

        comp_rep = subspace_discrete.comp_rep
        # The resident copy is keyed on the CONTENT of the comp rep (xxh3 over all N x d doubles: ~10 ms at 1e6 x 20,
        # against 50 ms for re-encoding and re-uploading them), so an edit of any row is noticed
        # (hashed column by column: a column of a numeric frame is a view of its block, whereas ``to_numpy()`` of a frame
        # with one block per column interleaves all N x d values into a new array - 11 ms of every call at 1e6 x 20)
        idx = comp_rep.index
        idx_key = (idx.start, idx.stop, idx.step) if isinstance(idx, pd.RangeIndex) else _content_hash(np.asarray(idx))
        key = (comp_rep.shape, tuple(comp_rep.columns), _frame_content_hash(comp_rep), idx_key)
        if self._cand_cache is None or self._cand_cache[0] != key:
            values = comp_rep.to_numpy(dtype=np.float64)
            X = torch.from_numpy(np.ascontiguousarray(values))
            if self.shard is not None:
                if self.shard.N_total != len(comp_rep):
                    raise ValueError(
                        f"RowShard was built for {self.shard.N_total} rows but the discrete subspace has {len(comp_rep)}"
                    )
                X = X[self.shard.start : self.shard.stop]
            self._cand_cache = (key, X.to(self._engine._dev()), comp_rep.index)
        _, Xd, labels = self._cand_cache
        alive = None
        if keep_mask is not None:
            if not keep_mask.all():
                mask = keep_mask.astype(np.uint8)
                if self.shard is not None:
                    mask = mask[self.shard.start : self.shard.stop]
                alive = torch.from_numpy(mask).to(Xd.device)
        elif len(candidates_exp) != len(labels) or not candidates_exp.index.equals(labels):
            pos = labels.get_indexer(candidates_exp.index)
            if (pos < 0).any():
                raise KeyError("candidates contain rows that are not part of the discrete subspace")
            mask = np.zeros(len(labels), dtype=np.uint8)
            mask[pos] = 1
            if self.shard is not None:
                mask = mask[self.shard.start : self.shard.stop]
            alive = torch.from_numpy(mask).to(Xd.device)
        return Xd, alive, labels

    @property
    def _engine(self):
        model = self._surrogate_model
        return model.models[0].engine if isinstance(model, HipCompositeImpl) else model.engine

    def _recommend_discrete(self, subspace_discrete, candidates_exp: pd.DataFrame, batch_size: int) -> pd.Index:
        assert self._objective is not None
        acqf = self._acqf_in_use
        if batch_size > 1 and not acqf.supports_batching:
            raise IncompatibleAcquisitionFunctionError(
                f"The '{self.__class__.__name__}' only works with Monte Carlo acquisition functions for batch sizes > 1."
            )
        n_sub = getattr(subspace_discrete, "n_subsets", 0)
        if n_sub > 0:
            return self._recommend_discrete_with_subsets(subspace_discrete, candidates_exp, batch_size)
        return self._recommend_discrete_without_subsets(subspace_discrete, candidates_exp, batch_size)

    def _recommend_discrete_without_subsets(self, subspace_discrete, candidates_exp, batch_size, return_values=False,
                                            keep_mask=None):
        surrogate = self._surrogate_model
        acqf = self._acqf_in_use
        Xd, alive, labels = self._candidates_on_device(subspace_discrete, candidates_exp, keep_mask)
        if self._nehvi is not None:
            res = self._nehvi.greedy(Xd, batch_size, seed=self._sampler_seed(), prune_seed=self._sampler_seed(),
                                     X_pending=self._pending_comp, alive=alive, shard=self.shard)
            idxs = labels[np.asarray(res.indices, dtype=np.int64)]
            return (idxs, res) if return_values else idxs
        if acqf.is_analytic:  # q = 1 by construction (supports_batching is False)
            eng = surrogate.engine
            mean, var = eng.posterior(Xd)
            scores = self._analytic_scores(eng, acqf, mean, var, surrogate.sign, alive)
            val, idx = eng.argmax(scores)
            if self.shard is not None:
                val, idx, _ = self.shard.global_argmax(val, idx, Xd)
            from baybe_amd.engine import GreedyResult

            res = GreedyResult([int(idx)], [float(val)])
        else:
            res = surrogate.engine.greedy_qlogei(
                Xd, batch_size, S=acqf.n_mc_samples, seed=self._sampler_seed(), sign=surrogate.sign,
                X_pending=self._pending_comp, best_f=self._best_f, shard=self.shard, kind=acqf.kind,
                beta=getattr(acqf, "beta", 0.2), alive=alive,
            )
        idxs = labels[np.asarray(res.indices, dtype=np.int64)]
        return (idxs, res) if return_values else idxs

    def _recommend_discrete_with_subsets(self, subspace_discrete, candidates_exp, batch_size) -> pd.Index:
        """``recommend_discrete_with_subsets`` (botorch/discrete.py:21-75): one greedy run per
        batch-constraint subset, the batch with the highest joint acquisition value wins."""
        if subspace_discrete.n_subsets <= self.max_n_subsets:
            masks = subspace_discrete.subset_masks(candidates_exp, min_candidates=batch_size)
        else:
            masks = subspace_discrete.sample_subset_masks(candidates_exp, self.max_n_subsets, min_candidates=batch_size)
        best = None
        for mask in masks:
            subset = candidates_exp.loc[mask]
            idxs = self._recommend_discrete_without_subsets(subspace_discrete, subset, batch_size)
            comp = subspace_discrete.comp_rep.loc[idxs].to_numpy(dtype=np.float64)
            val = self._joint_value(comp)
            if best is None or val > best[1]:
                best = (idxs, val)
        if best is None:
            from baybe_amd.exceptions import IncompatibilityError as _E

            raise _E("No feasible subset with enough candidates was found.")
        return best[0]

    # ---- hybrid and continuous spaces --------------------------------------------------------------
    def _recommend_hybrid(self, searchspace, candidates_exp: pd.DataFrame, batch_size: int) -> pd.DataFrame:
        """``recommend_hybrid_without_subsets`` (botorch/hybrid.py:30-163) as a data-parallel search.  The reference hands BoTorch's
        ``optimize_acqf_mixed`` one fixed-feature dictionary per discrete row and runs a multi-start gradient optimiser over the
        continuous parameters for each - "a brute-force calculation ... computationally expensive" (its own docstring).  Here the
        brute force is one scoring pass: every discrete candidate row is crossed with ``n_raw_samples`` scrambled-Sobol points of
        the continuous box (the raw samples of BoTorch's initialiser), all rows are scored on the device, and the ``n_restarts`` best
        rows are refined in their continuous coordinates by a compass search (2 d_c probes per start and iteration, all starts in
        one pass, step halved when no probe improves) - no acquisition gradients exist on this path.  Batches are built greedily with
        the chosen points pending, as ``optimize_acqf_mixed`` does.  Same options (``hybrid_sampler`` "Random",
        ``sampling_percentage``, ``n_restarts``, ``n_raw_samples``), same return frame (discrete part in experimental representation,
        indexed by the discrete candidate, plus the continuous columns); purely continuous spaces arrive here through
        ``PureRecommender._recommend_continuous`` with an empty discrete part.  Not a parity path: both sides are stochastic
        multi-start searches, compared by the acquisition value they reach (tests/test_hybrid_*.py)."""
        import torch

        from baybe_amd.engine import GreedyResult  # noqa: F401

        cont, disc = searchspace.continuous, searchspace.discrete
        _check_continuous_part(cont)
        acqf, surrogate = self._acqf_in_use, self._surrogate_model
        if self._nehvi is not None:
            raise IncompatibilityError("Multi-output objectives in hybrid / continuous spaces are not on the HIP path; use BotorchRecommender.")
        if batch_size > 1 and not acqf.supports_batching:
            raise IncompatibleAcquisitionFunctionError(
                f"The '{self.__class__.__name__}' only works with Monte Carlo acquisition functions for batch sizes > 1.")
        if getattr(disc, "n_subsets", 0) > 0:
            raise IncompatibilityError("Discrete subset-generating constraints in hybrid spaces are not on the HIP path.")
        eng = surrogate.engine
        cnames = list(cont.comp_rep_bounds.columns)
        cb = cont.comp_rep_bounds.to_numpy(dtype=np.float64)  # [2, d_c]
        dc = len(cnames)
        has_disc = candidates_exp is not None and len(candidates_exp.columns) > 0 and len(candidates_exp) > 0
        if has_disc:
            Dcomp = disc.comp_rep.loc[candidates_exp.index]
            n_keep = int(np.ceil(self.sampling_percentage * len(Dcomp)))
            if self.hybrid_sampler is not None and n_keep < len(Dcomp):
                Dcomp = Dcomp.iloc[self._sample_discrete_rows(Dcomp, n_keep)]
            D = np.ascontiguousarray(Dcomp.to_numpy(dtype=np.float64))
            labels = Dcomp.index
        else:
            D = np.zeros((1, 0))
            labels = pd.RangeIndex(1)
        Nd, dd = D.shape
        # raw samples of the continuous box: scrambled Sobol, as many per discrete row as a few million rows allow.  The reference runs
        # a gradient optimiser per start; this search has only the raw samples and a compass refinement, so the sample count grows
        # with the continuous dimension (512 per dimension) where the row budget allows, and beyond the dimension the
        # search was validated for (tests/test_hybrid_gpu.py: d_c <= 6) it says so
        R = int(max(1, min(max(self.n_raw_samples, 512 * dc), 4_000_000 // max(Nd, 1))))
        sob = torch.quasirandom.SobolEngine(dimension=max(dc, 1), scramble=True, seed=self._sampler_seed())
        U = sob.draw(R, dtype=torch.float64).numpy()[:, :dc]
        Craw = cb[0] + U * (cb[1] - cb[0])
        dev = eng._dev()
        Dd, Cd = torch.from_numpy(D).to(dev), torch.from_numpy(np.ascontiguousarray(Craw)).to(dev)
        X = torch.cat([Dd.repeat_interleave(R, dim=0), Cd.repeat(Nd, 1)], dim=1).contiguous()  # row i R + r = [disc i | cont r]
        base = self._pending_comp if self._pending_comp is not None else np.zeros((0, dd + dc))
        seed = self._sampler_seed()
        mean, var = eng.posterior(X)
        picks, pick_labels = [], []
        for _step in range(batch_size):
            pend = np.vstack([base] + picks) if picks else base
            scores = self._mc_or_analytic(eng, acqf, X, mean, var, pend, seed, surrogate.sign)
            k = int(min(self.n_restarts, X.shape[0], 64))  # (bbh_topk returns at most 64 rows; the reference's n_restarts has no cap)
            _, top = eng.topk(scores, k)
            top = [int(t) for t in top if t >= 0]
            starts = X[torch.as_tensor(top, device=X.device)].cpu().numpy()
            best_row, _ = self._compass_refine(eng, acqf, starts, dd, cb, pend, seed, surrogate.sign, iters=min(96, 24 + 8 * dc))
            picks.append(best_row.reshape(1, -1))
            if has_disc:  # the label of the refined winner's discrete part (copied bit for bit from its start row)
                pick_labels.append(labels[int(np.nonzero((D == best_row[:dd]).all(axis=1))[0][0])])
        out_cont = pd.DataFrame(np.vstack(picks)[:, dd:], columns=[str(c) for c in cnames])
        if has_disc:
            rec_disc = disc.exp_rep.loc[pd.Index(pick_labels)]
            out_cont.index = rec_disc.index
            return pd.concat([rec_disc, out_cont], axis=1)
        return out_cont

    def _sample_discrete_rows(self, Dcomp: pd.DataFrame, n_keep: int) -> list:
        """Positions of the discrete rows a hybrid recommendation enumerates: ``sample_numerical_df(candidates_comp, n, method=
        hybrid_sampler)`` of the reference (botorch/hybrid.py:92-96, utils/sampling_algorithms.py:185-229) - "Random" is pandas' own
        ``DataFrame.sample`` (numpy's global generator, the same draw as the reference's), "FPS" the reference's
        ``farthest_point_sampling`` itself where BayBE is importable (the plug-in lives inside it)."""
        name = str(getattr(self.hybrid_sampler, "value", self.hybrid_sampler))
        if self.shard is not None:  # row shards: every rank must draw the same subsample - the agreed sampler seed, not a global state
            if name != "Random":
                raise IncompatibilityError("hybrid_sampler='FPS' under row shards is not on the HIP path; use 'Random'.")
            return np.random.default_rng(self._sampler_seed()).choice(len(Dcomp), n_keep, replace=False).tolist()
        if name == "Random":
            return Dcomp.reset_index(drop=True).sample(n_keep).index.tolist()
        if name == "FPS":
            try:
                from baybe.utils.sampling_algorithms import farthest_point_sampling  # type: ignore
            except Exception as ex:  # noqa: BLE001
                raise IncompatibilityError("hybrid_sampler='FPS' needs BayBE's own farthest_point_sampling (baybe is not importable here); "
                                           "use 'Random' or sampling_percentage=1.") from ex
            return list(farthest_point_sampling(Dcomp.to_numpy(), n_keep))
        raise ValueError(f"Unrecognized sampling method: '{name}'.")

    def _mc_or_analytic(self, eng, acqf, X, mean, var, pend, seed, sign):
        if acqf.is_analytic:
            if len(pend):
                raise IncompatibleAcquisitionFunctionError("Analytic acquisition functions score single points only.")
            return self._analytic_scores(eng, acqf, mean, var, sign)
        return self._mc_scores_with_pending(eng, acqf, X, mean, var, pend, seed, sign)

    def _compass_refine(self, eng, acqf, starts: np.ndarray, dd: int, cb: np.ndarray, pend, seed, sign, iters: int = 24):
        """Compass (pattern) search on the continuous coordinates of ``starts`` [k, d], all starts advancing together: per
        iteration 2 d_c probes per start (plus and minus the start's step along every continuous axis, clipped to the box) are
        scored in ONE device pass; a start moves to its best improving probe, otherwise its step halves.  Returns the best row and
        its value.  (Same base samples for every evaluation: the MC acquisition is a deterministic function of the point.)"""
        k, d = starts.shape
        dc = d - dd
        cur = starts.copy()
        span = cb[1] - cb[0]
        if dc == 0:
            m, v = eng.posterior(cur)
            val = self._mc_or_analytic(eng, acqf, cur, m, v, pend, seed, sign).cpu().numpy()
            i = int(np.argmax(val))
            return cur[i], float(val[i])
        step = np.full(k, 0.125)  # fraction of the box, per start
        m, v = eng.posterior(cur)
        val = self._mc_or_analytic(eng, acqf, cur, m, v, pend, seed, sign).cpu().numpy()
        for _ in range(iters):
            probes = np.repeat(cur, 2 * dc, axis=0)  # [k * 2 dc, d]: start s, axis a, sign
            for a in range(dc):
                for sg, off in ((+1.0, 0), (-1.0, 1)):
                    rows = np.arange(k) * 2 * dc + 2 * a + off
                    probes[rows, dd + a] = np.clip(cur[:, dd + a] + sg * step * span[a], cb[0, a], cb[1, a])
            pm, pv = eng.posterior(probes)
            pval = self._mc_or_analytic(eng, acqf, probes, pm, pv, pend, seed, sign).cpu().numpy().reshape(k, 2 * dc)
            j = pval.argmax(axis=1)
            better = pval[np.arange(k), j] > val
            rows = np.arange(k) * 2 * dc + j
            cur[better] = probes[rows[better]]
            val[better] = pval[np.arange(k), j][better]
            step[~better] *= 0.5
            if (step < 1e-5).all():  # every start has converged to 1e-5 of the box
                break
        i = int(np.argmax(val))
        return cur[i], float(val[i])

    def _analytic_scores(self, eng, acqf, mean, var, sign, alive=None):
        return eng.analytic_acq(acqf.kind, mean, var, self._best_f, sign, getattr(acqf, "beta", 0.2),
                                getattr(acqf, "maximize", True), alive)

    # ---- read-backs (Campaign.acquisition_values / joint_acquisition_value) ---------------------
    def _joint_value(self, comp: np.ndarray) -> float:
        """qLogEI of one q-batch (candidate = first row, the others enter as pending rows)."""
        if self._nehvi is not None:
            # incremental NEHVI: the batch value is the value of its last point given the others
            self._nehvi.prepare(self._sampler_seed(), np.vstack([comp[:-1]] + ([self._pending_comp] if self._pending_comp is not None else [])) if len(comp) > 1 or self._pending_comp is not None else None)
            sc = self._nehvi.score(self._nehvi.outputs[0].engine._as_dev(comp[-1:]))
            return float(sc.cpu().numpy()[0])
        surrogate = self._surrogate_model
        eng = surrogate.engine
        acqf = self._acqf_in_use
        base = self._pending_comp if self._pending_comp is not None else np.zeros((0, comp.shape[1]))
        pend = np.vstack([comp[1:], base])
        seed = self._sampler_seed()
        mean, var = eng.posterior(comp[:1])
        if acqf.is_analytic:
            if len(pend):
                raise IncompatibleAcquisitionFunctionError("Analytic acquisition functions score single points only.")
            return float(self._analytic_scores(eng, acqf, mean, var, surrogate.sign).cpu().numpy()[0])
        s = self._mc_scores_with_pending(eng, acqf, comp[:1], mean, var, pend, seed, surrogate.sign)
        return float(s.cpu().numpy()[0])

    def _mc_scores_with_pending(self, eng, acqf, comp, mean, var, pend, seed, sign):
        """MC acquisition values of the t-batches [x_i ; pend] for the read-back APIs.  Up to 15 pending rows ride on the handle's
        pending state; beyond that (qLogEI only, up to 63 - the same limit ``_check_batch_size`` admits for recommend()) the
        columns come from ``cross_cov_many`` and the joint kernel takes the pending statistics explicitly, as in the greedy loop."""
        beta = getattr(acqf, "beta", 0.2)
        if len(pend) == 0:
            z = sobol_normal_base_samples(acqf.n_mc_samples, 1, seed)[:, 0]
            return eng.mc_acq(acqf.kind, mean, var, z, self._best_f, sign, beta)
        z = sobol_normal_base_samples(acqf.n_mc_samples, 1 + len(pend), seed)
        if len(pend) > _lib.MAX_PENDING:
            if acqf.kind != "qLogEI" or len(pend) > _lib.MAX_PENDING_BIG:
                raise IncompatibilityError(
                    f"{len(pend)} pending / batch rows exceed the largest joint q-batch of the HIP kernels for '{type(acqf).__name__}' "
                    f"({_lib.MAX_PENDING_BIG + 1} points for qLogEI, {_lib.MAX_PENDING + 1} otherwise).")
            cross = eng.cross_cov_many(comp, pend)
            s = eng.qlogei_pending_big(mean, var, cross, pend, z, self._best_f, sign)
        else:
            eng.set_pending(pend)
            cross = eng.cross_cov(comp)
            s = eng.mc_acq(acqf.kind, mean, var, z, self._best_f, sign, beta, cross=cross)
        eng.set_pending(None)
        return s

    def acquisition_values(self, candidates: pd.DataFrame, searchspace, objective, measurements,
                           pending_experiments=None, acquisition_function=None) -> pd.Series:
        """``BayesianRecommender.acquisition_values`` (base.py:199-238): one value per candidate."""
        surrogate, acqf = self._setup_acqf(searchspace, objective, measurements, pending_experiments, acquisition_function)
        comp = np.ascontiguousarray(searchspace.transform(candidates, allow_extra=True).to_numpy(dtype=np.float64))
        if self._nehvi is not None:
            self._nehvi.prepare(self._sampler_seed(), self._pending_comp)
            sc = self._nehvi.score(self._nehvi.outputs[0].engine._as_dev(comp))
            return pd.Series(sc.cpu().numpy(), index=candidates.index)
        eng = surrogate.engine
        mean, var = eng.posterior(comp)
        if acqf.is_analytic:
            s = self._analytic_scores(eng, acqf, mean, var, surrogate.sign)
            return pd.Series(s.cpu().numpy(), index=candidates.index)
        pend = self._pending_comp if self._pending_comp is not None else np.zeros((0, comp.shape[1]))
        s = self._mc_scores_with_pending(eng, acqf, comp, mean, var, pend, self._sampler_seed(), surrogate.sign)
        return pd.Series(s.cpu().numpy(), index=candidates.index)

    def joint_acquisition_value(self, candidates: pd.DataFrame, searchspace, objective, measurements,
                                pending_experiments=None, acquisition_function=None) -> float:
        """``BayesianRecommender.joint_acquisition_value`` (base.py:240-269)."""
        self._setup_acqf(searchspace, objective, measurements, pending_experiments, acquisition_function)
        comp = np.ascontiguousarray(searchspace.transform(candidates, allow_extra=True).to_numpy(dtype=np.float64))
        return self._joint_value(comp)


def _convert_hybrid_sampler(value):
    """``DiscreteSamplingMethod`` (utils/sampling_algorithms.py:175-182) by value or member; None = no sampling."""
    if value is None:
        return None
    name = str(getattr(value, "value", value))
    if name not in ("Random", "FPS"):
        raise ValueError(f"'{value}' is not a valid DiscreteSamplingMethod (Random, FPS)")
    return value


def _validate_percentage(_, attribute, value):
    if not 0 <= value <= 1:  # botorch/core.py:123-135
        raise ValueError(f"Hybrid sampling percentage needs to be between 0 and 1 but is {value}")


def recommender_fields(with_base_fields: bool = True, surrogate_factory=HipGaussianProcessSurrogate) -> dict:
    """attrs fields of the recommender.  ``with_base_fields=False`` leaves out what BayBE's ``BayesianRecommender``
    declares itself (``acquisition_function``, ``_objective``; pure/bayesian/base.py:46-66); ``_surrogate_model`` is
    redeclared in both cases because its default is the HIP surrogate."""
    f = {
        "_surrogate_model": field(alias="surrogate_model", factory=surrogate_factory),
        "max_n_subsets": field(default=10, validator=[instance_of(int), ge(1)], kw_only=True),  # botorch/core.py:93-98
        # The reference's options for continuous / hybrid optimisation (botorch/core.py:69-91), accepted with its defaults and
        # validators so that constructor calls carry over; none of them affects purely discrete optimisation (the reference says so
        # for n_restarts / n_raw_samples), which is the only kind of search space this recommender takes.
        "sequential_continuous": field(default=True, validator=instance_of(bool), kw_only=True),
        "hybrid_sampler": field(default=None, converter=_convert_hybrid_sampler, kw_only=True),
        "sampling_percentage": field(default=1.0, validator=_validate_percentage, kw_only=True),
        "n_restarts": field(default=10, validator=[instance_of(int), gt(0)], kw_only=True),
        "n_raw_samples": field(default=64, validator=[instance_of(int), gt(0)], kw_only=True),
        # optional baybe_amd.distributed.RowShard: this process scores only its row range and joins one all-gather
        # per selection step (multi-GPU)
        "shard": field(default=None, eq=False, repr=False, kw_only=True),
        "_best_f": field(default=None, init=False, eq=False, repr=False),
        "_pending_comp": field(default=None, init=False, eq=False, repr=False),
        "_cand_cache": field(default=None, init=False, eq=False, repr=False),
        "_nehvi": field(default=None, init=False, eq=False, repr=False),
        "_acqf_in_use": field(default=None, init=False, eq=False, repr=False),
    }
    if with_base_fields:
        f["acquisition_function"] = field(default=None, converter=convert_acqf, kw_only=True)
        f["_objective"] = field(default=None, init=False, eq=False, repr=False)
    return f


HipBotorchRecommender = attrs.make_class("HipBotorchRecommender", recommender_fields(), bases=(HipRecommenderImpl,),
                                         kw_only=True, slots=False)
HipBotorchRecommender.__doc__ = "Bayesian recommender scoring the full discrete candidate set on an MI355X (stand-alone)."
HipBotorchRecommender.__module__ = __name__
HipBotorchRecommender.compatibility = "HYBRID"
