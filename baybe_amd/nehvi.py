"""qLogNoisyExpectedHypervolumeImprovement on the HIP path (BayBE's default for ParetoObjective,
``baybe/acquisition/acqfs.py:477-484``; arguments per ``baybe/acquisition/_builder.py:301-324``).

Structure.  BoTorch draws f(x) *jointly* with the baseline values f(X_b) through a
cached Cholesky root, per MC sample s and per (independent) output o.  That joint draw is the
posterior of a GP whose training set is extended by the baseline points as noise-free observations
of the sampled values F_b,s:

    f_o(x)_s = E[f_o(x) | D_o, f_o(X_b) = F_b,s] + sd[f_o(x) | D_o, f_o(X_b)] * z_x,s,o

so the per-candidate work is exactly the fused posterior kernel on an *extended model*
(``bbh_set_model_ex`` with a noise mask): one variance pass, and the S conditional means as a
contraction of the same cross-covariance with S target columns (``bbh_posterior_columns``).
Set-up per selection step, on the device (round 5; it was 5 ms of host work per step plus 50 ms of
pruning per call): the extended model's own Cholesky factor IS the sampler -

    y_ext,s = c + L_ext [t; z_s],   t = L^-1 (y - c) on the training rows,

reproduces the measurements in its first n rows and draws the baseline's joint posterior sample
mu_b + chol(Sigma_b) z_s in its last n_b rows (L_ext's last block row is [K_bn L^-T, chol(Sigma_b)]),
and the weight columns of the model conditioned on that sample are L_ext^-T [t; z_s].  So per target:
``bbh_set_model_ex`` + ``bbh_factorize`` of the extended model, then ``bbh_nehvi_samples`` (sampled
baseline values into a device array, weight columns installed) - no baseline posterior call, no host
Cholesky, no (n + n_b) x S host array; the three targets run from three host threads on three
streams.  ``bbh_cells_build_dev`` decomposes every sample's non-dominated region on the device (one
wavefront per sample), ``bbh_qlognehvi_cells`` scores against the device-resident cell lists, and
the pruning counts come from ``bbh_pareto_frequency_dev`` on samples drawn the same way.  The base
samples themselves (torch's scrambled Sobol engine, as BoTorch) stay on the host; the next step's
draw is prepared while the device scores the current one.
"""

from __future__ import annotations

import ctypes as C
import math
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from baybe_amd import _lib
from baybe_amd.box_decomposition import pack_cells_native, pareto_mask
from baybe_amd.engine import GreedyResult, HipGP, _dp, _native_sobol_usable, draw_sampler_seed, sobol_normal_base_samples

PRUNE_SAMPLES = 2048  # prune_inferior_points_multi_objective(num_samples=2048)


def compute_ref_point(array, maximize=None, factor: float = 0.1) -> np.ndarray:
    """``_ExpectedHypervolumeImprovement.compute_ref_point`` (acqfs.py:369-426)."""
    array = np.asarray(array, dtype=np.float64)
    if array.ndim != 2:
        raise ValueError("The specified data array must have exactly two dimensions.")
    mx = np.where(np.ones(array.shape[1], bool) if maximize is None else np.asarray(maximize), 1.0, -1.0)
    a = array * mx[None, :]
    lo, hi = a.min(axis=0), a.max(axis=0)
    return (lo - factor * (hi - lo)) * mx


def _unique_rows(X: np.ndarray):
    """(rows of X without repeats in first-occurrence order, their indices, index of every row's representative).
    Repeated baseline points (replicate measurements) carry one latent value: BoTorch's joint draw gives the copies the same
    sample up to sqrt(jitter), and an extended model with two noise-free rows at one location is singular - the copies keep
    their base-sample columns (the Sobol dimension counts them) but only the first enters the model and the decompositions."""
    if len(X) == 0:
        return X, np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    _, first, inverse = np.unique(X, axis=0, return_index=True, return_inverse=True)
    order = np.sort(first)
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first)] = np.arange(len(first))
    return X[order], order, rank[np.asarray(inverse).reshape(-1)]


def _chol_with_jitter(A: np.ndarray) -> np.ndarray:
    """psd_safe_cholesky: plain attempt, then jitter 1e-8 * 10^i."""
    jit = 0.0
    for attempt in range(4):
        try:
            return np.linalg.cholesky(A + jit * np.eye(A.shape[0]))
        except np.linalg.LinAlgError:
            jit = 1e-8 * 10**attempt
    raise np.linalg.LinAlgError("baseline posterior covariance is not positive definite")


@dataclass
class _Output:
    engine: HipGP  # fitted model of this target
    ext: HipGP  # the same model conditioned on sampled baseline values
    sign: float


class HipNEHVI:
    """qLogNEHVI scorer over m independent HIP GPs (q = 1 t-batches, pending points cached into the
    baseline exactly as BoTorch's ``cache_pending=True``)."""

    def __init__(self, engines, signs, X_baseline, ref_point, n_mc_samples: int = 128, prune_baseline: bool = True,
                 device: int = 0):
        if not 1 <= len(engines) <= _lib.MAX_OBJECTIVES:
            raise ValueError(f"1..{_lib.MAX_OBJECTIVES} objectives are supported")
        self.m = len(engines)
        if any(getattr(e.spec, "kernel", None) == "rff" for e in engines):
            from baybe_amd.exceptions import IncompatibilityError

            # (the extended models condition on noise-free latent rows, which the RFF kernel's uniform-noise feature-space form does not have)
            raise IncompatibilityError("qLogNEHVI (ParetoObjective) is not available with an RFFKernel surrogate on the HIP path.")
        self.outputs = [_Output(e, HipGP(device), float(s)) for e, s in zip(engines, signs)]
        self.signs = np.asarray(signs, dtype=np.float64)
        self.X_baseline = np.ascontiguousarray(np.atleast_2d(X_baseline), dtype=np.float64)
        self.ref = np.asarray(ref_point, dtype=np.float64)
        self.S = int(n_mc_samples)
        self.prune = bool(prune_baseline)
        self._lib = _lib.load_library()
        self._prepared = False
        self._streams = None
        self.concurrent = True  # False: the targets' passes one after the other (per-kernel timing, A/B)
        self._pruned = None
        # BBH_NEHVI_HOST=1: the round-4 host set-up (baseline posterior -> host Cholesky -> host samples -> host box
        # decompositions -> (n + nb) x S host array per target), kept as the cross-check of the device set-up
        self.device_setup = os.environ.get("BBH_NEHVI_HOST", "0") != "1"
        # BBH_NEHVI_DRAW=host: base samples from the host path (torch's generator and erfinv) - the A/B form of the device draw
        self.device_draw = os.environ.get("BBH_NEHVI_DRAW", "device") != "host"
        self._factorize_lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=self.m) if self.m > 1 else None
        self._z_next = None  # (key, future): base samples of the next selection step, drawn while the device scores
        self._cells_on_device = False
        self.last_setup_ms = {}

    # ---- set-up ----------------------------------------------------------------------------------
    def _baseline_posteriors(self, Xb):
        mus, Ls = [], []
        for out in self.outputs:
            mu, cov = out.engine.posterior_joint(Xb)
            mus.append(mu)
            Ls.append(_chol_with_jitter(cov))
        return mus, Ls

    def _draw(self, S: int, nb: int, seed: int) -> np.ndarray:
        return sobol_normal_base_samples(S, (nb + 1) * self.m, seed).reshape(S, nb + 1, self.m)

    def _base_samples(self, S: int, nb: int, seed: int) -> np.ndarray:
        """[S, nb + 1, m] base samples of a step with nb baseline points (the candidate's are the last row); taken from the
        draw prepared during the previous scoring pass when it matches."""
        key = (S, nb, seed)
        nxt, self._z_next = self._z_next, None
        if nxt is not None and nxt[0] == key:
            return nxt[1].result()
        return self._draw(S, nb, seed)

    def prefetch_base_samples(self, nb: int, seed: int):
        """Draw the base samples of the NEXT selection step on a host thread (the number of baseline points after the pick
        is known before the pick is): the engine's scrambling and the normal transform overlap the device's scoring pass."""
        if self._pool is None:
            return
        key = (self.S, nb, seed)
        self._z_next = (key, self._pool.submit(self._draw, self.S, nb, seed))

    def _extend(self, o: int, Xb: np.ndarray, z_o, S: int, Fb_dev, want_columns: bool):
        """Target o: extended model (baseline rows as noise-free observations), its factorisation, and the samples /
        weight columns drawn through the factor itself (``bbh_nehvi_samples``).  Runs on a host thread per target.
        ``z_o``: the target's base samples as a host array [S, nb], or (device draw [S, ld], ld, device int32 offsets [nb]) when the
        draw was made on the device (``bbh_nehvi_samples_dev``)."""
        out = self.outputs[o]
        eng = out.engine
        Xt, yt = eng._X_train, eng._y_train
        nb = len(Xb)
        X_ext = np.vstack([Xt, Xb])
        y_ext = np.concatenate([yt, np.full(nb, eng.ybar)])  # (the baseline rows' values are not read)
        mask = np.concatenate([np.ones(len(yt), np.uint8), np.zeros(nb, np.uint8)])
        out.ext.set_model(eng.spec, X_ext, y_ext, noise_mask=mask, standardization=(eng.ybar, eng.ysd))
        # The tile-dataflow factorisation needs every tile of a launch co-resident (one 135 KB-LDS workgroup per CU); two or three
        # targets' launches overlapping can exceed the 256 CUs and starve each other until the spin limit - the
        # factorisations take turns, the sample and weight-column kernels stay concurrent.
        with self._factorize_lock:  # (bbh_factorize returns after its stream has drained: it reads the Cholesky flag back)
            out.ext.factorize(eng.params)
        if isinstance(z_o, tuple):
            z_dev, ld, cols_dev = z_o
            out.ext._check(
                self._lib.bbh_nehvi_samples_dev(out.ext._h, z_dev.data_ptr(), ld, cols_dev.data_ptr(), S, nb, float(self.signs[o]), o,
                                                self.m, Fb_dev.data_ptr(), 1 if want_columns else 0),
                "bbh_nehvi_samples_dev",
            )
        else:
            z_o = np.ascontiguousarray(z_o, dtype=np.float64)
            out.ext._check(
                self._lib.bbh_nehvi_samples(out.ext._h, _dp(z_o), S, nb, float(self.signs[o]), o, self.m, Fb_dev.data_ptr(),
                                            1 if want_columns else 0),
                "bbh_nehvi_samples",
            )
        if want_columns:
            out.ext._ncols = S

    def _for_each_target(self, fn):
        if self._pool is None:
            return [fn(0)]
        return [f.result() for f in [self._pool.submit(fn, o) for o in range(self.m)]]

    def _baseline_samples_dev(self, Xb: np.ndarray, z, want_columns: bool, S: int | None = None, first=None):
        """Oriented baseline objective samples [S, nb, m] as a device tensor (all targets).  ``z``: host base samples [S, >= nb, m],
        or a device draw [S, n_all * m] over all baseline rows (``first``: which of them the unique rows ``Xb`` are)."""
        import torch

        nb = len(Xb)
        eng0 = self.outputs[0].engine
        on_device = isinstance(z, torch.Tensor)
        S = int(z.shape[0]) if S is None else S
        Fb_dev = torch.empty((S, nb, self.m), dtype=torch.float64, device=eng0._dev())
        if on_device:  # offsets of (baseline row b, target o) within a row of the draw: row-of-draw index * m + o
            base = np.arange(nb, dtype=np.int64) if first is None else np.asarray(first, dtype=np.int64)
            cols = torch.from_numpy((base[None, :] * self.m + np.arange(self.m)[:, None]).astype(np.int32)).to(eng0._dev())
            ld = int(z.shape[1])
            arg = lambda o: (z, ld, cols[o])  # noqa: E731
        else:
            arg = lambda o: z[:, :nb, o]  # noqa: E731
        torch.cuda.synchronize(eng0.device)  # (Fb_dev's allocation, the draw and the offsets vs the targets' own streams)
        self._for_each_target(lambda o: self._extend(o, Xb, arg(o), S, Fb_dev, want_columns))
        torch.cuda.synchronize(eng0.device)  # the targets write Fb_dev on their own streams
        return Fb_dev

    def prune_points(self, Xb: np.ndarray, seed: int) -> np.ndarray:
        """Keep baseline points that are Pareto-optimal and above the reference point in at least one
        of 2048 joint posterior samples (``prune_inferior_points_multi_objective``)."""
        import time

        Xb_all = Xb
        Xb, first, rep = _unique_rows(Xb_all)
        nb = len(Xb)
        t0 = time.perf_counter()
        draw_on_device = self.device_setup and self.device_draw and _native_sobol_usable()
        if draw_on_device:  # 2048 x (n_b m) values: the transform alone was 19.9 ms on the host
            z = self.outputs[0].ext.sobol_normal_dev(PRUNE_SAMPLES, len(Xb_all) * self.m, seed)
        else:
            z = sobol_normal_base_samples(PRUNE_SAMPLES, len(Xb_all) * self.m, seed).reshape(PRUNE_SAMPLES, len(Xb_all), self.m)
            if nb < len(Xb_all):
                z = np.ascontiguousarray(z[:, first, :])
        self.last_prune_ms = {"base_samples": 1e3 * (time.perf_counter() - t0)}
        t0 = time.perf_counter()
        counts = np.zeros(nb, dtype=np.int64)
        ref = np.ascontiguousarray(self.ref, dtype=np.float64)
        eng = self.outputs[0].engine
        if self.device_setup:
            obj_dev = self._baseline_samples_dev(Xb, z, want_columns=False, S=PRUNE_SAMPLES, first=first if draw_on_device else None)
            self.last_prune_ms["extend"] = 1e3 * (time.perf_counter() - t0)
            h = self.outputs[0].ext
            h._check(
                self._lib.bbh_pareto_frequency_dev(h._h, obj_dev.data_ptr(), PRUNE_SAMPLES, nb, self.m, _dp(ref),
                                                   counts.ctypes.data_as(_lib.c_int64_p)),
                "bbh_pareto_frequency_dev",
            )
        else:
            zc = np.ascontiguousarray(z.transpose(2, 0, 1))  # [m, S, nb]: contiguous operands for the BLAS products
            mus, Ls = self._baseline_posteriors(Xb)
            obj = np.empty((PRUNE_SAMPLES, nb, self.m))
            for o in range(self.m):
                obj[:, :, o] = (mus[o][None, :] + zc[o] @ Ls[o].T) * self.signs[o]
            obj = np.ascontiguousarray(obj)
            eng._check(
                self._lib.bbh_pareto_frequency(eng._h, _dp(obj), PRUNE_SAMPLES, nb, self.m, _dp(ref),
                                               counts.ctypes.data_as(_lib.c_int64_p)),
                "bbh_pareto_frequency",
            )
        self.last_prune_ms["total_after_base_samples"] = 1e3 * (time.perf_counter() - t0)
        idx = np.nonzero(counts[rep] > 0)[0]  # (a repeated point shares the verdict of its representative)
        return Xb_all[idx] if len(idx) else Xb_all[:0]

    def _prepare_host(self, Xb: np.ndarray, z: np.ndarray):
        """The round-4 set-up: samples and box decompositions on the host, full (n + nb) x S target columns per target."""
        nb = len(Xb)
        Fb = np.empty((self.S, nb, self.m))
        mus = None
        if nb:
            mus, Ls = self._baseline_posteriors(Xb)
            for o in range(self.m):
                Fb[:, :, o] = mus[o][None, :] + z[:, :nb, o] @ Ls[o].T
        self.cell_off, self.cell_lo, self.cell_ll = pack_cells_native(Fb * self.signs[None, None, :], self.ref)
        self._cells_on_device = False
        for o, out in enumerate(self.outputs):
            eng = out.engine
            Xt, yt = eng._X_train, eng._y_train
            X_ext = np.vstack([Xt, Xb]) if nb else Xt
            y_ext = np.concatenate([yt, mus[o]]) if nb else yt
            mask = np.concatenate([np.ones(len(yt), np.uint8), np.zeros(nb, np.uint8)])
            out.ext.set_model(eng.spec, X_ext, y_ext, noise_mask=mask, standardization=(eng.ybar, eng.ysd))
            out.ext.factorize(eng.params)
            Y = np.empty((len(y_ext), self.S))
            Y[: len(yt), :] = yt[:, None]
            if nb:
                Y[len(yt):, :] = Fb[:, :, o].T
            out.ext.set_mean_columns(Y)

    def prepare(self, seed: int, extra_baseline: np.ndarray | None = None, prune_seed: int | None = None):
        """Sample the baseline, decompose, and condition the per-output models (one selection step)."""
        import time

        t0 = time.perf_counter()
        tm = {}
        if self._pruned is None:  # pruning happens once, when the acquisition function is built
            Xb0 = self.X_baseline
            if self.prune and len(Xb0):
                Xb0 = self.prune_points(Xb0, draw_sampler_seed() if prune_seed is None else prune_seed)
            self._pruned = Xb0
            tm["prune"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        Xb = self._pruned
        if extra_baseline is not None and len(extra_baseline):
            Xb = np.vstack([Xb, np.atleast_2d(extra_baseline)])  # cache_pending: picks join the baseline
        Xb_all = Xb
        z = self._base_samples(self.S, len(Xb_all), seed)
        self.zx = np.ascontiguousarray(z[:, len(Xb_all), :])  # [S, m] base samples of the candidate
        Xb, first, _ = _unique_rows(Xb_all)
        nb = len(Xb)
        if nb < len(Xb_all):  # repeated baseline points: one latent value each (their base-sample columns stay counted)
            z = np.ascontiguousarray(z[:, np.concatenate([first, [len(Xb_all)]]), :])
        tm["base_samples"] = time.perf_counter() - t1
        t1 = time.perf_counter()
        if self.device_setup and nb:
            Fb_dev = self._baseline_samples_dev(Xb, z, want_columns=True)
            tm["extend"] = time.perf_counter() - t1
            t1 = time.perf_counter()
            h = self.outputs[0].ext
            total, over = C.c_int64(), C.c_int64()
            ref = np.ascontiguousarray(self.ref, dtype=np.float64)
            h._check(self._lib.bbh_cells_build_dev(h._h, Fb_dev.data_ptr(), self.S, nb, self.m, _dp(ref), C.byref(total), C.byref(over)),
                     "bbh_cells_build_dev")
            self._cells_on_device = over.value == 0
            self.n_cells = int(total.value)
            if not self._cells_on_device:  # a sample's bound list exceeded the kernel's capacity (or nb > 512): host form
                self.cell_off, self.cell_lo, self.cell_ll = pack_cells_native(Fb_dev.cpu().numpy(), self.ref)
                self.n_cells = int(self.cell_off[-1])
            tm["cells"] = time.perf_counter() - t1
        else:
            self._prepare_host(Xb, z)
            self.n_cells = int(self.cell_off[-1])
            tm["host_setup"] = time.perf_counter() - t1
        self.X_b_current = Xb_all
        self._prepared = True
        tm["total"] = time.perf_counter() - t0
        self.last_setup_ms = {k: 1e3 * v for k, v in tm.items()}

    def cells(self):
        """(cell_off [S + 1], cell_lo [K, m], cell_loglen [K, m]) of the current step as host arrays."""
        if not self._cells_on_device:
            return self.cell_off, self.cell_lo, self.cell_ll
        h = self.outputs[0].ext
        off = np.zeros(self.S + 1, dtype=np.int64)
        lo, ll = np.zeros((self.n_cells, self.m)), np.zeros((self.n_cells, self.m))
        h._check(self._lib.bbh_cells_read_dev(h._h, off.ctypes.data_as(_lib.c_int64_p), _dp(lo), _dp(ll)), "bbh_cells_read_dev")
        return off, lo, ll

    # ---- scoring ---------------------------------------------------------------------------------
    def _target_streams(self, device):
        """One HIP stream per target (created once; the extended models' handles are bound to them), or [] for a single target."""
        if self._streams is None:
            import os

            import torch

            self._streams = []
            if self.m > 1 and os.environ.get("BBH_NEHVI_STREAMS", "1") != "0":
                self._streams = [torch.cuda.Stream(device=device) for _ in self.outputs]
                for out, st in zip(self.outputs, self._streams):
                    with torch.cuda.stream(st):
                        out.ext.use_current_torch_stream()
        return self._streams

    def score(self, X_dev, alive=None, sync: bool = True):
        import torch

        assert self._prepared, "call prepare() first"
        tmats, vars_ = [], []
        # The targets' passes are independent: each target's extended model enqueues its variance pass and its conditional-mean
        # columns on a stream of its own, so that the launch tails of one target (13 % of a columns launch at 1e5 candidates) fill
        # with another target's workgroups; the cell kernel runs on the first target's stream after the others have been joined.
        # BBH_NEHVI_STREAMS=0: everything on the handles' default stream, one launch after the other (A/B).
        streams = self._target_streams(X_dev.device)
        if streams:
            cur = torch.cuda.current_stream(X_dev.device)
            for out, st in zip(self.outputs, streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    _, var = out.ext.posterior(X_dev)
                    tmats.append(out.ext.posterior_columns(X_dev, sample_major=True))
                vars_.append(var)
                if not self.concurrent:  # (the handles stay bound to their streams: one target after the other = wait for each)
                    st.synchronize()
            for st in streams[1:]:
                streams[0].wait_stream(st)
        else:
            for out in self.outputs:
                _, var = out.ext.posterior(X_dev)
                tmats.append(out.ext.posterior_columns(X_dev, sample_major=True))
                vars_.append(var)
        N = X_dev.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=X_dev.device)
        tp = (C.c_void_p * self.m)(*[t.data_ptr() for t in tmats])
        vp = (C.c_void_p * self.m)(*[v.data_ptr() for v in vars_])
        h = self.outputs[0].ext
        sg = np.ascontiguousarray(self.signs)
        zx = np.ascontiguousarray(self.zx)
        if self._cells_on_device:
            rc = self._lib.bbh_qlognehvi_cells(h._h, self.m, N, tp, vp, _dp(sg), _dp(zx), self.S,
                                               alive.data_ptr() if alive is not None else None, scores.data_ptr())
        else:
            off = np.ascontiguousarray(self.cell_off, dtype=np.int64)
            rc = self._lib.bbh_qlognehvi_sm(
                h._h, self.m, N, tp, vp, _dp(sg), _dp(zx), self.S, off.ctypes.data_as(_lib.c_int64_p),
                _dp(self.cell_lo) if len(self.cell_lo) else None, _dp(self.cell_ll) if len(self.cell_ll) else None,
                alive.data_ptr() if alive is not None else None, scores.data_ptr(),
            )
        h._check(rc, "bbh_qlognehvi")
        if sync:
            torch.cuda.synchronize(X_dev.device)  # tmats / vars_ must outlive the kernel
        else:
            self._keep = (tmats, vars_)  # (the caller synchronises; the operands live until the next pass)
        return scores

    def greedy(self, X_dev, q: int, seed: int | None = None, prune_seed: int | None = None,
               X_pending: np.ndarray | None = None, alive=None, shard=None) -> GreedyResult:
        """Sequential greedy of optimize_acqf_discrete: pending points and each pick join the
        baseline (``cache_pending=True``).  With ``shard`` (``RowShard``) every rank scores its rows
        and one all-gather per step makes the pick global; the set-up is replicated (all ranks must
        use the same seeds, i.e. the same torch RNG state, as for the single-target path)."""
        import torch

        X_dev = self.outputs[0].engine._as_dev(X_dev)
        d = self.outputs[0].engine.spec.d
        if seed is None:
            seed = draw_sampler_seed()
        alive = torch.ones(X_dev.shape[0], dtype=torch.uint8, device=X_dev.device) if alive is None else alive.clone()
        picks: list[np.ndarray] = []
        if X_pending is not None and len(X_pending):
            picks.append(np.atleast_2d(np.asarray(X_pending, dtype=np.float64)))
        indices, values = [], []
        for step in range(q):
            self.prepare(seed, np.vstack(picks) if picks else None, prune_seed)
            scores = self.score(X_dev, alive, sync=False)
            if step + 1 < q:  # the next step's base samples: drawn on a host thread while the device scores this one
                self.prefetch_base_samples(len(self.X_b_current) + 1, seed)
            val, idx = self.outputs[0].ext.argmax(scores) if X_dev.shape[0] else (-math.inf, -1)
            torch.cuda.synchronize(X_dev.device)
            if shard is not None:
                val, gidx, row = shard.global_argmax(val, idx, X_dev)
                if shard.owns(gidx):
                    alive[shard.to_local(gidx)] = 0
                idx = gidx
            else:
                row = X_dev[idx, :d].cpu().numpy()
                alive[idx] = 0
            indices.append(int(idx))
            values.append(float(val))
            picks.append(np.asarray(row, dtype=np.float64).reshape(1, d))
        return GreedyResult(indices, values)
