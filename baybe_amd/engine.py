"""Host driver of the HIP GP engine: one handle = one single-output GP on one MI355X.

Mirrors, step by step, what BayBE triggers inside BoTorch on the recommend() path
(SURVEY.md §3.2-3.4): fit_gpytorch_mll -> posterior -> qLogEI -> optimize_acqf_discrete.
PyTorch-ROCm tensors are used purely as device containers (``data_ptr()``); all arithmetic
happens in libbaybe_hip through the C-ABI of ``include/baybe_hip.h``.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import threading
from dataclasses import dataclass

import numpy as np
from scipy import optimize as sopt

from baybe_amd import _lib
from baybe_amd._lib import HipError, HipUnavailableError
from baybe_amd.gp_spec import (
    FastObjective,
    GPParams,
    GPSpec,
    initial_params,
    sample_params_from_priors,
    objective_from_data_term,
    pack_raw,
    raw_bounds,
    theta_from_params,
    unpack_raw,
)

MAX_PENDING = _lib.MAX_PENDING
MAX_PENDING_BIG = _lib.MAX_PENDING_BIG


def _dp(a: np.ndarray):
    return a.ctypes.data_as(_lib.c_double_p)


_SOBOL_MAXBIT = 30
_SOBOL_STATE: dict = {}  # dimension -> unscrambled direction numbers [q, 30] (int64 numpy), as the engine initialises them
_FAST_SOBOL = None  # None: unchecked, True / False after the first use


def _sobol_uniform_engine(S: int, q: int, seed: int):
    """[S, q] scrambled Sobol points from torch's engine itself (float64 tensor)."""
    import torch

    return torch.quasirandom.SobolEngine(dimension=q, scramble=True, seed=int(seed)).draw(S, dtype=torch.float64)


def _sobol_uniform_fast(S: int, q: int, seed: int):
    """The same points without the engine's per-dimension loops: torch's generator draws the scrambling bits in the engine's
    order (``SobolEngine._scramble``: the shift bits [q, 30], then the lower-triangular matrices [q, 30, 30]), the library's
    host code scrambles the direction numbers and walks the Gray-code sequence (``bbh_sobol_scramble`` / ``bbh_sobol_draw``:
    integer arithmetic, bitwise the engine's points).  768 dimensions x 2048 points: 25 ms -> 4 ms on the GPU box's host."""
    import torch

    lib = _lib.load_library()
    state0 = _SOBOL_STATE.get(q)
    if state0 is None:
        st = torch.zeros(q, _SOBOL_MAXBIT, dtype=torch.long)
        torch._sobol_engine_initialize_state_(st, q)
        state0 = _SOBOL_STATE[q] = st.numpy().copy()
    g = torch.Generator()
    g.manual_seed(int(seed))
    shift_bits = torch.randint(2, (q, _SOBOL_MAXBIT), generator=g).numpy()
    ltm = np.ascontiguousarray(torch.randint(2, (q, _SOBOL_MAXBIT, _SOBOL_MAXBIT), generator=g).numpy())
    shift = np.ascontiguousarray(shift_bits @ (np.int64(1) << np.arange(_SOBOL_MAXBIT, dtype=np.int64)))
    state = state0.copy()
    p64 = _lib.c_int64_p
    if lib.bbh_sobol_scramble(state.ctypes.data_as(p64), ltm.ctypes.data_as(p64), q) != 0:
        raise RuntimeError("bbh_sobol_scramble failed")
    u = np.empty((S, q), dtype=np.float64)
    if lib.bbh_sobol_draw(state.ctypes.data_as(p64), shift.ctypes.data_as(p64), S, q, _dp(u)) != 0:
        raise RuntimeError("bbh_sobol_draw failed")
    return torch.from_numpy(u)


def _fast_sobol_usable() -> bool:
    """The fast path uses two private torch entry points and mirrors ``SobolEngine._scramble``: the first use compares it with
    the engine on three small cases and falls back to the engine for good on any difference (``BBH_FAST_SOBOL=0`` forces that)."""
    global _FAST_SOBOL
    if _FAST_SOBOL is None:
        import os

        import torch

        ok = os.environ.get("BBH_FAST_SOBOL", "1") != "0"
        try:
            for S, q, seed in ((9, 1, 3), (33, 7, 123456), (130, 41, 999)) if ok else ():
                ok = ok and torch.equal(_sobol_uniform_fast(S, q, seed), _sobol_uniform_engine(S, q, seed))
        except Exception:  # noqa: BLE001
            ok = False
        _FAST_SOBOL = bool(ok)
    return _FAST_SOBOL


def sobol_normal_base_samples(S: int, q: int, seed: int) -> np.ndarray:
    """Base samples of botorch's SobolQMCNormalSampler: scrambled Sobol (torch's engine, the same
    one BoTorch uses) -> v = 0.5 + (1 - eps)(u - 0.5) -> sqrt(2) erfinv(2 v - 1).  [S, q] fp64."""
    import torch

    # The engine's scrambling runs a few tiny tensor ops (tril of a [q, 30, 30] matrix ...); with torch's
    # intra-op pool at its default size each of them wakes every host thread - 17 ms per engine on the
    # 128-thread GPU box, 1.3 ms on one thread.  The values do not depend on the thread count.
    nthreads = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        u = _sobol_uniform_fast(S, q, seed) if _fast_sobol_usable() else _sobol_uniform_engine(S, q, seed)
        if S * q >= 1 << 18:  # the normal transform of a pruning draw (2048 x 768 values) on a few threads: elementwise, same values
            torch.set_num_threads(min(8, nthreads))
        v = 0.5 + (1 - torch.finfo(torch.float64).eps) * (u - 0.5)
        return (torch.erfinv(2 * v - 1) * math.sqrt(2)).numpy().copy()
    finally:
        torch.set_num_threads(nthreads)


_FIT_TLS = threading.local()  # .stream: the torch stream fits started from this thread enqueue on (private_fit_stream)
_FIT_STREAMS: dict = {}
_FIT_POOL = None


@contextlib.contextmanager
def private_fit_stream(device: int, slot: int):
    """``HipGP.fit`` calls made by this thread inside the block run on a stream of their own (one per (device, slot), created once)."""
    import torch

    key = (int(device), int(slot))
    st = _FIT_STREAMS.get(key)
    if st is None:
        st = _FIT_STREAMS[key] = torch.cuda.Stream(device=int(device))
    prev = getattr(_FIT_TLS, "stream", None)
    _FIT_TLS.stream = st
    try:
        yield st
    finally:
        _FIT_TLS.stream = prev


def fit_side_by_side(jobs, device: int = 0):
    """Run the callables ``jobs`` (each one fits one model) on host threads, each with a private stream: the results in order.
    One shared pool of four threads (``baybe/surrogates/composite.py:101-134`` fits its targets one after the other)."""
    global _FIT_POOL
    jobs = list(jobs)
    if len(jobs) <= 1 or os.environ.get("BBH_FIT_SIDE_BY_SIDE", "1") == "0":
        return [job() for job in jobs]
    import torch

    if not torch.cuda.is_available():  # (no device, no streams: the CPU test double of the handle fits in sequence)
        return [job() for job in jobs]
    if _FIT_POOL is None:
        from concurrent.futures import ThreadPoolExecutor

        _FIT_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="bbh-fit")

    def run(slot, job):
        _FIT_TLS.no_global_generator = True
        try:
            with private_fit_stream(device, slot):
                return job()
        finally:
            _FIT_TLS.no_global_generator = False

    futures = [_FIT_POOL.submit(run, k % 4, job) for k, job in enumerate(jobs)]
    results, rerun = [], False
    for f in futures:  # (every job is waited for before anything is decided: none may still be running when the sequence starts)
        try:
            results.append(f.result())
        except _FitNeedsTheGlobalGenerator:
            results.append(None)
            rerun = True
    if rerun:
        # a fit wants torch's global generator: side by side its state would depend on which thread draws first.  Nothing has drawn
        # from it yet (the parallel attempts started from the deterministic prior modes), so the whole group is fitted again in order
        return [job() for job in jobs]
    return results


_NATIVE_SOBOL = None  # None: unchecked, True / False after the first use


def _sobol_state0(q: int) -> np.ndarray:
    """Unscrambled direction numbers [q, 30] as torch's engine initialises them (cached per dimension count)."""
    import torch

    state0 = _SOBOL_STATE.get(q)
    if state0 is None:
        st = torch.zeros(q, _SOBOL_MAXBIT, dtype=torch.long)
        torch._sobol_engine_initialize_state_(st, q)
        state0 = _SOBOL_STATE[q] = st.numpy().copy()
    return state0


def sobol_normal_native(S: int, q: int, seed: int) -> np.ndarray:
    """``sobol_normal_base_samples`` by the library's host code alone (``bbh_sobol_normal``: MT19937 scrambling bits, Gray-code walk,
    inverse error function) - the CPU twin of the device draw, equal to the torch path up to the last bits of erfinv."""
    lib = _lib.load_library()
    state0 = _sobol_state0(q)
    out = np.empty((S, q), dtype=np.float64)
    if lib.bbh_sobol_normal(state0.ctypes.data_as(_lib.c_int64_p), int(seed) & 0xFFFFFFFFFFFFFFFF, S, q, _dp(out)) != 0:
        raise RuntimeError("bbh_sobol_normal failed")
    return out


def _native_sobol_usable() -> bool:
    """The native draw restates torch's CPU generator (MT19937, one 32-bit output per ``randint`` element) and mirrors
    ``SobolEngine._scramble``: the first use compares it with the engine path on three small cases (a wrong bit anywhere gives
    unrelated points, so 1e-13 decides) and falls back to the host path for good on any difference (``BBH_NATIVE_SOBOL=0`` forces that)."""
    global _NATIVE_SOBOL
    if _NATIVE_SOBOL is None:
        import os

        ok = os.environ.get("BBH_NATIVE_SOBOL", "1") != "0"
        try:
            for S, q, seed in ((9, 1, 3), (33, 7, 123456), (130, 41, 999)) if ok else ():
                ok = ok and bool(np.allclose(sobol_normal_native(S, q, seed), sobol_normal_base_samples(S, q, seed), rtol=1e-13, atol=0))
        except Exception:  # noqa: BLE001
            ok = False
        _NATIVE_SOBOL = bool(ok)
    return _NATIVE_SOBOL


def draw_sampler_seed() -> int:
    """MCSampler seed when none is given: ``torch.randint(0, 1000000, (1,))`` from torch's global
    RNG at the first acquisition evaluation (so ``active_settings.random_seed`` governs it)."""
    import torch

    return int(torch.randint(0, 1000000, (1,)).item())


# ---- scipy's L-BFGS-B without scipy's per-evaluation wrappers -----------------------------------------------------------------
# ``scipy.optimize.minimize(method="L-BFGS-B")`` is what ``botorch.fit.fit_gpytorch_mll`` runs, and the iterates have to stay
# scipy's.  Its driver (``_minimize_lbfgsb``) is a short loop around the compiled reverse-communication routine ``setulb``;
# around every objective evaluation it puts ``ScalarFunction`` / ``MemoizeJac`` bookkeeping (array comparisons, copies, lowest-value
# tracking): ~25 us per evaluation - a quarter of an evaluation of a small model (0.09 ms on the device).  The loop
# below is that driver with the evaluation called directly: same routine, same work arrays, same stopping logic, same messages -
# bitwise the same iterates (``tests/test_host_logic_cpu.py::test_lean_lbfgsb_driver_is_scipys``).  ``setulb`` is a private
# interface: the first use checks the lean driver against ``minimize`` on a small problem and falls back to ``minimize`` for good
# if they differ in any bit or the call fails (another scipy version).
_LEAN_LBFGSB = None  # None: unchecked, True / False after the first use


class _OptResult:
    __slots__ = ("x", "fun", "nit", "nfev", "status", "message")

    def __init__(self, x, fun, nit, nfev, status, message):
        self.x, self.fun, self.nit, self.nfev, self.status, self.message = x, fun, nit, nfev, status, message


def _lbfgsb_lean(fun, x0, bounds, maxiter, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxls=20):
    from scipy.optimize import _lbfgsb_py as lb

    m = maxcor
    factr = ftol / np.finfo(float).eps
    x0 = np.asarray(x0, dtype=np.float64).ravel()
    n = x0.shape[0]
    nbd = np.zeros(n, np.int32)
    low, up = np.zeros(n, np.float64), np.zeros(n, np.float64)
    if bounds is not None:
        lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
        hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
        if (lo > hi).any():
            raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
        x0 = np.clip(x0, lo, hi)
        for i in range(n):
            has_lo, has_hi = np.isfinite(lo[i]), np.isfinite(hi[i])
            if has_lo:
                low[i] = lo[i]
            if has_hi:
                up[i] = hi[i]
            nbd[i] = (2 if has_hi else 1) if has_lo else (3 if has_hi else 0)
    x = np.array(x0, dtype=np.float64)
    f = np.array(0.0, dtype=np.int32)
    g = np.zeros((n,), dtype=np.int32)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, dtype=np.int32)
    task, ln_task = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
    lsave, isave, dsave = np.zeros(4, dtype=np.int32), np.zeros(44, dtype=np.int32), np.zeros(29, dtype=np.float64)
    nit = nfev = 0
    while True:
        g = np.asarray(g, dtype=np.float64)
        lb._lbfgsb.setulb(m, x, low, up, nbd, f, g, factr, gtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        if task[0] == 3:  # f and g at the current x
            f, g = fun(np.copy(x))
            nfev += 1
        elif task[0] == 1:  # new iteration
            nit += 1
            if nit >= maxiter:
                task[0], task[1] = 5, 504
            elif nfev > maxfun:
                task[0], task[1] = 5, 502
        else:
            break
    status = 0 if task[0] == 4 else (1 if (nfev > maxfun or nit >= maxiter) else 2)
    return _OptResult(x, f, nit, nfev, status, lb.status_messages[task[0]] + ": " + lb.task_messages[task[1]])


def _lean_lbfgsb_usable() -> bool:
    global _LEAN_LBFGSB
    if _LEAN_LBFGSB is None:
        try:
            A = np.array([[3.0, 0.5, 0.1], [0.5, 2.0, 0.3], [0.1, 0.3, 1.5]])
            b = np.array([1.0, -2.0, 0.5])

            def probe(x):
                return float(0.5 * x @ A @ x - b @ x + 0.1 * np.log1p(x @ x)), A @ x - b + 0.2 * x / (1.0 + x @ x)

            bnd = [(None, None), (-0.1, None), (0.25, 0.25)]
            x0 = np.array([0.7, -0.4, 0.2])
            ref = sopt.minimize(probe, x0, jac=True, method="L-BFGS-B", bounds=bnd, options={"maxiter": 50})
            got = _lbfgsb_lean(probe, x0, bnd, 50)
            _LEAN_LBFGSB = bool(np.array_equal(ref.x, got.x) and ref.fun == got.fun and ref.nit == got.nit and ref.nfev == got.nfev
                                and ref.status == got.status and str(ref.message) == got.message)
        except Exception:  # noqa: BLE001  (another scipy: private interface changed)
            _LEAN_LBFGSB = False
    return _LEAN_LBFGSB


def lbfgsb_minimize(fun, x0, bounds, maxiter):
    """``scipy.optimize.minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": maxiter})`` - through the
    lean driver above when it reproduces scipy bit for bit here (``BBH_LEAN_LBFGSB=0`` forces ``minimize``)."""
    if os.environ.get("BBH_LEAN_LBFGSB", "1") != "0" and _lean_lbfgsb_usable():
        return _lbfgsb_lean(fun, x0, bounds, maxiter)
    return sopt.minimize(fun, x0, jac=True, method="L-BFGS-B", bounds=bounds, options={"maxiter": maxiter})


@dataclass
class FitInfo:
    params: GPParams
    fun: float
    nit: int
    nfev: int
    status: int
    message: str


@dataclass
class GreedyResult:
    indices: list  # candidate positions, in selection order
    values: list  # acquisition value of each greedy step


class ModelFittingError(RuntimeError):
    """Hyper-parameter fit / factorisation failed (mirrors baybe.exceptions.ModelFittingError)."""


class _FitNeedsTheGlobalGenerator(ModelFittingError):
    """Raised inside ``fit_side_by_side`` by a fit that would have to draw from torch's global generator (a retry from re-sampled
    hyper-parameters, random start values of a free task factor): the group is then fitted in sequence, so that the generator is
    consumed in the reference's order (surrogates/composite.py:101-134 fits target by target)."""


# Device handles of closed / collected HipGP objects, per device ordinal: a backtesting run creates (and drops) one
# model per scenario case (``simulate_scenarios`` deep-copies the campaign, simulation/scenarios.py:296); re-using the
# handle keeps its streams, events, workspaces and grow-only device buffers instead of paying hipMalloc / hipFree per case.
_HANDLE_POOL: dict = {}
_HANDLE_POOL_MAX = 8  # idle handles kept per device
_POOL_KEEP_BYTES = 64 << 20  # per buffer of an idle handle (bbh_trim)
pool_stats = {"created": 0, "reused": 0}


def _pool_key(device: int):
    """A handle reads its A/B switches (``BBH_*`` environment variables) when it is created: handles are
    only handed on between objects created under the same switches."""
    import os

    return (int(device), tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("BBH_"))))


def _acquire_handle(lib, device: int):
    free = _HANDLE_POOL.get(_pool_key(device))
    if free:
        pool_stats["reused"] += 1
        return free.pop()
    h = C.c_void_p()
    rc = lib.bbh_create(int(device), C.byref(h))
    if rc != 0:
        raise HipUnavailableError(
            f"bbh_create(device={device}) failed with {rc}: no usable HIP device. "
            "This path has no CPU fallback."
        )
    pool_stats["created"] += 1
    return h


def drain_handle_pool():
    """Destroy the idle device handles (tests; before a process gives its device away)."""
    lib = _lib.load_library()
    for free in _HANDLE_POOL.values():
        while free:
            lib.bbh_destroy(free.pop())


class HipGP:
    """A GP surrogate living on one HIP device.

    The object can be copied and pickled (``copy.deepcopy`` of campaigns / surrogates is what the reference's
    backtesting drivers do: ``simulation/core.py:124``, ``scenarios.py:296``, ``transfer_learning.py:78``,
    ``surrogates/composite.py:54``): the copy carries the model description, the training data and the fitted
    hyper-parameters - not the device handle - and re-creates its device state (handle from the pool, upload,
    factorisation with the same hyper-parameters) the first time it is used."""

    def __init__(self, device: int = 0):
        self._libobj = _lib.load_library()
        self._handle = _acquire_handle(self._libobj, device)
        self._pool_key = _pool_key(device)
        self.device = int(device)
        self.spec: GPSpec | None = None
        self.params: GPParams | None = None
        self.n = 0
        self.ybar = 0.0
        self.ysd = 1.0
        self.jitter = 0.0
        self._comm = None  # (rank, world) once bbh_comm_init has run on this handle
        self._model_args = None  # (noise_mask, standardization) of the last set_model: what a copy needs to rebuild
        self._X_train = self._y_train = None

    # ---- plumbing -------------------------------------------------------------------------
    @property
    def _lib(self):
        """The bound shared library (loaded on first use in a copied / unpickled object)."""
        if self._libobj is None:
            self._libobj = _lib.load_library()
        return self._libobj

    @property
    def _h(self):
        """The device handle; a copied / unpickled object builds its device state here, on first use."""
        if self._handle is None and getattr(self, "_restorable", False):
            self._restore()
        return self._handle

    def close(self):
        if getattr(self, "_handle", None):
            key = self._pool_key
            if self._comm is not None or len(_HANDLE_POOL.setdefault(key, [])) >= _HANDLE_POOL_MAX:
                self._lib.bbh_destroy(self._handle)  # a handle that owns a communicator is not handed on
            else:  # back to its defaults: legacy stream, no timing; buffers above 64 MB are released (an idle handle must not pin HBM)
                self._lib.bbh_set_stream(self._handle, None)
                self._lib.bbh_set_slice_rows(self._handle, 0)
                self._lib.bbh_timing_enable(self._handle, 0)
                self._lib.bbh_trim(self._handle, _POOL_KEEP_BYTES)
                _HANDLE_POOL[key].append(self._handle)
            self._handle = None
        self._restorable = False

    # ---- copies ---------------------------------------------------------------------------
    def __getstate__(self):
        return {
            "device": self.device, "spec": self.spec, "params": self.params, "n": self.n, "ybar": self.ybar,
            "ysd": self.ysd, "jitter": self.jitter, "X_train": self._X_train, "y_train": self._y_train,
            "model_args": self._model_args,
        }

    def __setstate__(self, st):
        self._libobj = None
        self._handle = None
        self.device, self.spec, self.params = st["device"], st["spec"], st["params"]
        self.n, self.ybar, self.ysd, self.jitter = st["n"], st["ybar"], st["ysd"], st["jitter"]
        self._X_train, self._y_train, self._model_args = st["X_train"], st["y_train"], st["model_args"]
        self._comm = None
        self._restorable = True

    def _restore(self):
        self._restorable = False
        self._handle = _acquire_handle(self._lib, self.device)
        self._pool_key = _pool_key(self.device)
        if self.spec is not None and self._X_train is not None:
            params = self.params
            mask, std = self._model_args or (None, None)
            self.set_model(self.spec, self._X_train, self._y_train, noise_mask=mask, standardization=std)
            if params is not None:
                self.factorize(params)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc < 0:
            msg = self._lib.bbh_last_error(self._h)
            raise HipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    def _torch(self):
        import torch

        return torch

    def _dev(self):
        return self._torch().device("cuda", self.device)

    def use_current_torch_stream(self):
        """Enqueue on torch's current stream of this device (containers and kernels then share
        one queue; the default is the legacy null stream, which torch's default stream is)."""
        torch = self._torch()
        self._stream_ptr = int(torch.cuda.current_stream(self.device).cuda_stream)
        self._check(self._lib.bbh_set_stream(self._h, C.c_void_p(self._stream_ptr)), "bbh_set_stream")

    def selftest(self):
        self._check(self._lib.bbh_selftest(self._h), "bbh_selftest")

    # ---- model ----------------------------------------------------------------------------
    def set_model(self, spec: GPSpec, X_train: np.ndarray, y_train: np.ndarray, noise_mask: np.ndarray | None = None,
                  standardization: tuple | None = None):
        """``noise_mask`` [n] (0 = noise-free latent value) and ``standardization`` = (ybar, ysd) are
        used by the qLogNEHVI machinery (model conditioned on sampled baseline values)."""
        X = np.ascontiguousarray(X_train, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y_train, dtype=np.float64).reshape(-1))
        if X.ndim != 2 or X.shape[1] != spec.d or X.shape[0] != y.shape[0]:
            raise ValueError("X_train must be [n, d] and y_train [n]")
        desc = _lib.ModelDesc(
            _lib.KERNEL_KINDS[spec.kernel],
            spec.d,
            -1 if spec.task_idx is None else int(spec.task_idx),
            int(spec.n_tasks),
            1 if spec.use_outputscale else 0,
            _lib.CRITERIA[spec.criterion],
            1 if (getattr(spec, "hadamard", False) and spec.n_tasks > 1) else 0,
        )
        factors = getattr(spec, "factors", None)
        if factors:  # composite kernel: ProductKernel / AdditiveKernel of stationary factors
            desc.n_factors = len(factors)
            desc.combine = {"product": 0, "sum": 1, "grouped": 2}[spec.combine]
            for k, f in enumerate(factors):
                desc.factor_kind[k] = _lib.KERNEL_KINDS[f.kernel]
                desc.factor_scaled[k] = 1 if f.scaled else 0
                desc.factor_group[k] = int(getattr(f, "group", 0))
        lo = np.ascontiguousarray(spec.lo, dtype=np.float64)
        hi = np.ascontiguousarray(spec.hi, dtype=np.float64)
        if lo.shape[0] != spec.d or hi.shape[0] != spec.d:
            raise ValueError("spec.lo / spec.hi must have one entry per comp-rep column")
        if spec.kernel == "rff":
            spec = self._with_rff_weights(spec)
        mask_p = None
        if noise_mask is not None:
            mask = np.ascontiguousarray(noise_mask, dtype=np.uint8)
            mask_p = mask.ctypes.data_as(C.POINTER(C.c_uint8))
        given, yb, ys = (1, float(standardization[0]), float(standardization[1])) if standardization else (0, 0.0, 1.0)
        with self._on_thread_stream():  # (the uploads complete inside the call: bbh_set_model_ex synchronises its stream)
            self._check(
                self._lib.bbh_set_model_ex(self._h, C.byref(desc), X.shape[0], _dp(X), _dp(y), _dp(lo), _dp(hi), mask_p,
                                           given, yb, ys),
                "bbh_set_model_ex",
            )
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.bbh_get_standardization(self._h, C.byref(a), C.byref(b)), "bbh_get_standardization")
        self.spec, self.n, self.ybar, self.ysd = spec, X.shape[0], a.value, b.value
        self.params = None
        self._X_train = X
        self._y_train = y
        self._model_args = (None if noise_mask is None else np.array(noise_mask, dtype=np.uint8),
                            None if not standardization else (float(standardization[0]), float(standardization[1])))

    def _with_rff_weights(self, spec: GPSpec) -> GPSpec:
        """The frequencies of an RFF kernel go to the handle before the model: drawn here when the spec carries none - gpytorch's
        ``RFFKernel._init_weights`` draws ``torch.randn(d, D)`` (global generator, the lengthscale's dtype) at the first forward of a
        fit, over the kernel's active columns -, and kept on a copy of the spec so that copies / re-created handles use the same ones."""
        import copy

        from baybe_amd.exceptions import IncompatibleSurrogateError  # (BayBE's own class where it is importable)

        if spec.factors or spec.n_tasks > 1 or spec.task_idx is not None or spec.criterion != "mll":
            raise IncompatibleSurrogateError("On the HIP path the RFF kernel is the single kernel of a single-task model fitted by its "
                                             "marginal likelihood (alone or in a ScaleKernel).")
        D = int(spec.rff_num_samples or 0)
        if D < 1:
            raise ValueError("rff_num_samples must be at least 1")  # (kernels/basic.py:183-200 validates the same)
        if D > _lib.MAX_RFF_SAMPLES:
            raise IncompatibleSurrogateError(f"RFFKernel(num_samples={D}): the HIP path holds up to {_lib.MAX_RFF_SAMPLES} frequencies "
                                             f"(a {2 * _lib.MAX_RFF_SAMPLES} x {2 * _lib.MAX_RFF_SAMPLES} feature-space system).")
        mask = spec.active_mask(0)
        d_act = spec.dn if mask is None else int(mask.sum())
        if spec.rff_weights is None:
            spec = copy.copy(spec)
            spec.rff_weights = self._torch().randn(d_act, D, dtype=self._torch().float64).numpy().copy()
        W = np.asarray(spec.rff_weights, dtype=np.float64)
        if W.shape != (d_act, D):
            raise ValueError(f"rff_weights must be [{d_act}, {D}] (active numerical columns x num_samples)")
        full = np.zeros((spec.dn, D))  # (columns the kernel does not act on: zero frequencies - their pinned lengthscale is immaterial)
        full[np.arange(spec.dn) if mask is None else np.flatnonzero(mask)] = W
        self._check(self._lib.bbh_set_rff_weights(self._h, _dp(np.ascontiguousarray(full)), spec.dn, D), "bbh_set_rff_weights")
        return spec

    def data_term(self, params: GPParams):
        """Device data term (MLL or LOO) and its gradient in theta layout; (None, None) if the
        train covariance is not positive definite."""
        return self._data_term_theta(theta_from_params(self.spec, params))

    def _data_term_theta(self, theta: np.ndarray):
        val = C.c_double()
        grad = np.zeros_like(theta)
        rc = self._check(self._lib.bbh_fit_value_grad(self._h, _dp(theta), C.byref(val), _dp(grad)), "bbh_fit_value_grad")
        if rc == 1:
            return None, None
        return val.value, grad

    def fit(self, p0: GPParams | None = None, maxiter: int = 15000, max_attempts: int = 5) -> FitInfo:
        """scipy L-BFGS-B over the raw parameters with scipy's defaults, as
        ``botorch.fit.fit_gpytorch_mll`` does (call site gaussian_process/core.py:340-341);
        every objective evaluation is one ``bbh_fit_value_grad`` on the device.

        Like BoTorch's ``_fit_fallback`` the fit is retried (up to ``max_attempts`` times) from
        hyper-parameters re-sampled from their priors when an attempt ends abnormally or at a
        non-finite point; ``ModelFittingError`` is raised when every attempt fails."""
        with self._on_thread_stream():
            return self._fit(p0, maxiter, max_attempts)

    @contextlib.contextmanager
    def _on_thread_stream(self):
        """Fits side by side (the targets of a CompositeSurrogate, one host thread each): on the handles' default stream - the legacy
        null stream, ONE queue per device - their evaluations would run one after the other however many threads submit them (round 5:
        50.6 ms for three fits against 56.6 ms in sequence).  Inside ``private_fit_stream`` the handle enqueues on the thread's private
        stream for the duration of the block; ``set_model``, every evaluation and the final factorisation end with a synchronisation
        of that stream, so nothing is in flight on it when the handle goes back to its own."""
        st = getattr(_FIT_TLS, "stream", None)
        if st is None or getattr(self, "_thread_stream_depth", 0) > 0:
            yield
            return
        prev = getattr(self, "_stream_ptr", 0)
        self._check(self._lib.bbh_set_stream(self._h, C.c_void_p(int(st.cuda_stream))), "bbh_set_stream")
        self._thread_stream_depth = 1
        try:
            yield
        finally:
            self._thread_stream_depth = 0
            self._check(self._lib.bbh_set_stream(self._h, C.c_void_p(prev)), "bbh_set_stream")

    def _fit(self, p0, maxiter, max_attempts) -> FitInfo:
        spec = self.spec
        n = self.n

        fast = FastObjective(spec, n) if FastObjective.applies(spec) else None

        def fun(raw):
            if fast is not None:  # single-task, single-kernel models: the same maps as below in a few vectorised operations
                theta, nat = fast.theta(raw)
                val, g = self._data_term_theta(theta)
                if val is None:
                    return float("inf"), np.zeros_like(raw)
                return fast.objective(raw, nat, val, g)
            p = unpack_raw(spec, raw)
            val, g = self.data_term(p)
            if val is None:
                return float("inf"), np.zeros_like(raw)
            return objective_from_data_term(spec, raw, n, val, g, params=p)

        last_msg = ""
        no_rng = bool(getattr(_FIT_TLS, "no_global_generator", False))  # (inside fit_side_by_side: see _FitNeedsTheGlobalGenerator)
        if no_rng and p0 is None and spec.n_tasks > 1 and getattr(spec, "task_factor_constraint", "softplus") == "none":
            raise _FitNeedsTheGlobalGenerator("random start values of the task factor")
        for attempt in range(max_attempts):
            if attempt > 0 and no_rng:
                raise _FitNeedsTheGlobalGenerator(f"retry after: {last_msg}")
            start = p0 if (attempt == 0 and p0 is not None) else (
                initial_params(spec) if attempt == 0 else sample_params_from_priors(spec))
            res = lbfgsb_minimize(fun, pack_raw(spec, start), raw_bounds(spec), maxiter)
            last_msg = str(res.message)
            ok = np.all(np.isfinite(res.x)) and np.isfinite(res.fun) and "ABNORMAL" not in last_msg.upper()
            if not ok:
                continue
            params = unpack_raw(spec, res.x)
            try:
                self.factorize(params)
            except ModelFittingError as ex:
                last_msg = str(ex)
                continue
            return FitInfo(params, float(res.fun), int(res.nit), int(res.nfev), int(res.status), last_msg)
        raise ModelFittingError(f"All attempts to fit the model have failed (last: {last_msg}).")

    def factorize(self, params: GPParams):
        theta = theta_from_params(self.spec, params)
        jit = C.c_double()
        try:
            self._check(self._lib.bbh_factorize(self._h, _dp(theta), C.byref(jit)), "bbh_factorize")
        except HipError as ex:
            raise ModelFittingError(str(ex)) from ex
        self.params = params
        self.jitter = jit.value

    # ---- posterior ------------------------------------------------------------------------
    def _as_dev(self, X):
        torch = self._torch()
        if isinstance(X, np.ndarray):
            X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64))
        if X.dtype != torch.float64:
            X = X.to(torch.float64)
        X = X.to(self._dev())
        if X.dim() != 2 or X.shape[1] < self.spec.d:
            raise ValueError("candidates must be [N, d]")
        if X.stride(1) != 1:
            X = X.contiguous()
        if X.shape[0] == 1 and X.stride(0) < X.shape[1]:
            # a single row that came from a column-major block (``frame.iloc[[i]].to_numpy()``) counts as contiguous with a row
            # stride of 1: give it the row stride the C-ABI asks for (ldx >= d)
            X = X.as_strided(X.shape, (X.shape[1], 1))
        return X

    def posterior(self, X, unfused: bool = False, out=None):
        """Marginal posterior (mean, var) of N candidates as device tensors [N] (fp64); ``out`` = (mean, var) tensors to write into."""
        torch = self._torch()
        X = self._as_dev(X)
        N = X.shape[0]
        if out is not None:
            mean, var = out
        else:
            mean = torch.empty(N, dtype=torch.float64, device=X.device)
            var = torch.empty(N, dtype=torch.float64, device=X.device)
        fn = self._lib.bbh_posterior_unfused if unfused else self._lib.bbh_posterior
        self._check(fn(self._h, X.data_ptr(), N, X.stride(0), mean.data_ptr(), var.data_ptr()), "bbh_posterior")
        return mean, var

    def posterior_joint(self, Xq: np.ndarray):
        """Joint posterior (mean [q], cov [q,q]) of a small point set (host arrays)."""
        Xq = np.ascontiguousarray(np.atleast_2d(Xq), dtype=np.float64)
        q = Xq.shape[0]
        mean, cov = np.empty(q), np.empty((q, q))
        self._check(self._lib.bbh_posterior_joint(self._h, _dp(Xq), q, _dp(mean), _dp(cov)), "bbh_posterior_joint")
        return mean, cov

    def set_mean_columns(self, Y: np.ndarray):
        """Alternative target columns Y [n, S] (original scale) for ``posterior_columns``."""
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        if Y.ndim != 2 or Y.shape[0] != self.n:
            raise ValueError("Y must be [n, S]")
        self._ncols = Y.shape[1]
        self._check(self._lib.bbh_set_mean_columns(self._h, _dp(Y), Y.shape[1]), "bbh_set_mean_columns")

    def posterior_columns(self, X, sample_major: bool = False):
        """Posterior means of the candidates under each target column: [N, S], or [S, N] with ``sample_major`` (the layout
        the qLogNEHVI scoring kernel reads coalesced)."""
        torch = self._torch()
        X = self._as_dev(X)
        N = X.shape[0]
        tmat = torch.empty((self._ncols, N) if sample_major else (N, self._ncols), dtype=torch.float64, device=X.device)
        fn = self._lib.bbh_posterior_columns_sm if sample_major else self._lib.bbh_posterior_columns
        self._check(fn(self._h, X.data_ptr(), N, X.stride(0), tmat.data_ptr()), "bbh_posterior_columns")
        return tmat

    def train_posterior_mean(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        self._check(self._lib.bbh_train_posterior_mean(self._h, _dp(out)), "bbh_train_posterior_mean")
        return out

    def best_f(self, sign: float = 1.0) -> float:
        """max_i objective(posterior mean at x_i)  (baybe/acquisition/_builder.py:141-161,256-265)."""
        return float((sign * self.train_posterior_mean()).max())

    # ---- qLogEI ---------------------------------------------------------------------------
    def qlogei(self, mean, var, z: np.ndarray, best_f: float, sign: float = 1.0, alive=None):
        torch = self._torch()
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
        N = mean.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        self._check(
            self._lib.bbh_qlogei_q1(
                self._h, mean.data_ptr(), var.data_ptr(), N, _dp(z), z.shape[0], float(best_f), float(sign),
                alive.data_ptr() if alive is not None else None, scores.data_ptr(),
            ),
            "bbh_qlogei_q1",
        )
        return scores

    def qlogei_topk(self, mean, var, z: np.ndarray, best_f: float, sign: float = 1.0, k: int = 1, alive=None, scores=None):
        """q' = 1 qLogEI scores AND their k best in one call (``bbh_qlogei_q1_topk``: scores kernel, one selection kernel, results
        in host-mapped memory, one synchronisation): (scores [N] device tensor, values [k], indices [k]); entries beyond the
        number of scored candidates are (-inf, -1).  ``scores``: an fp64 device tensor [N] to write into (a step loop's buffer)."""
        torch = self._torch()
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
        N = mean.shape[0]
        if scores is None:
            scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        k = int(min(k, N))
        vals, idx = np.empty(k), np.empty(k, dtype=np.int64)
        self._check(
            self._lib.bbh_qlogei_q1_topk(
                self._h, mean.data_ptr(), var.data_ptr(), N, _dp(z), z.shape[0], float(best_f), float(sign),
                alive.data_ptr() if alive is not None else None, scores.data_ptr(), k, _dp(vals), idx.ctypes.data_as(_lib.c_int64_p),
            ),
            "bbh_qlogei_q1_topk",
        )
        return scores, vals, idx

    def mc_acq(self, kind: str, mean, var, z: np.ndarray, best_f: float = 0.0, sign: float = 1.0, beta: float = 0.2,
               alive=None, cross=None):
        """MC acquisition values (qLogEI, qEI, qPI, qSR, qUCB, qPSTD) of N t-batches [x_i ; pending]."""
        torch = self._torch()
        z = np.ascontiguousarray(z, dtype=np.float64)
        N = mean.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        k = _lib.ACQ_KINDS[kind]
        al = alive.data_ptr() if alive is not None else None
        if cross is None:
            z = z.reshape(-1)
            rc = self._lib.bbh_mc_acq_q1(self._h, k, mean.data_ptr(), var.data_ptr(), N, _dp(z), z.shape[0], float(best_f),
                                         float(sign), float(beta), al, scores.data_ptr())
        else:
            rc = self._lib.bbh_mc_acq_pending(self._h, k, mean.data_ptr(), var.data_ptr(), cross.data_ptr(), N, _dp(z),
                                              z.shape[0], float(best_f), float(sign), float(beta), al, scores.data_ptr())
        self._check(rc, "bbh_mc_acq")
        return scores

    def analytic_acq(self, kind: str, mean, var, best_f: float = 0.0, sign: float = 1.0, beta: float = 0.2,
                     maximize: bool = True, alive=None):
        """Analytic acquisition values (PM, PSTD, UCB, EI, LogEI, PI) of N single candidates."""
        torch = self._torch()
        N = mean.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        self._check(
            self._lib.bbh_analytic_acq(self._h, _lib.ACQ_KINDS[kind], mean.data_ptr(), var.data_ptr(), N, float(best_f),
                                       float(sign), float(beta), 1 if maximize else 0,
                                       alive.data_ptr() if alive is not None else None, scores.data_ptr()),
            "bbh_analytic_acq",
        )
        return scores

    def score_qlogei(self, X, z: np.ndarray, best_f: float, sign: float = 1.0, alive=None, want_posterior: bool = True):
        """Fused scoring pass: posterior + q'=1 qLogEI in one kernel.  Returns (scores, mean, var);
        mean/var are None unless ``want_posterior``."""
        torch = self._torch()
        X = self._as_dev(X)
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1)
        N = X.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=X.device)
        mean = torch.empty(N, dtype=torch.float64, device=X.device) if want_posterior else None
        var = torch.empty(N, dtype=torch.float64, device=X.device) if want_posterior else None
        self._check(
            self._lib.bbh_score_qlogei(
                self._h, X.data_ptr(), N, X.stride(0), _dp(z), z.shape[0], float(best_f), float(sign),
                alive.data_ptr() if alive is not None else None,
                mean.data_ptr() if mean is not None else None, var.data_ptr() if var is not None else None,
                scores.data_ptr(),
            ),
            "bbh_score_qlogei",
        )
        return scores, mean, var

    def set_pending(self, X_pending: np.ndarray | None):
        """Pending points = base pending + greedy picks (candidate first, then pending)."""
        if X_pending is None or len(X_pending) == 0:
            self._check(self._lib.bbh_pending_set(self._h, None, 0, None, None), "bbh_pending_set")
            return None, None
        P = np.ascontiguousarray(X_pending, dtype=np.float64)
        p = P.shape[0]
        if p > MAX_PENDING:
            raise ValueError(f"at most {MAX_PENDING} pending points are supported by the HIP path")
        mp = np.empty(p)
        cpp = np.empty((p, p))
        self._check(self._lib.bbh_pending_set(self._h, _dp(P), p, _dp(mp), _dp(cpp)), "bbh_pending_set")
        self._p = p
        return mp, cpp

    def cross_cov(self, X):
        torch = self._torch()
        X = self._as_dev(X)
        N = X.shape[0]
        cross = torch.empty((N, self._p), dtype=torch.float64, device=X.device)
        self._check(self._lib.bbh_cross_cov(self._h, X.data_ptr(), N, X.stride(0), cross.data_ptr()), "bbh_cross_cov")
        return cross

    def cross_cov_many(self, X, X_pending: np.ndarray):
        """[N, p] posterior cross-covariances with ANY number of pending points: one mean-only pass per chunk of <= 15 (a column
        does not depend on the other pending points).  Leaves the handle's pending state at the last chunk."""
        torch = self._torch()
        P = np.ascontiguousarray(np.atleast_2d(X_pending), dtype=np.float64)
        cols = []
        for c0 in range(0, len(P), MAX_PENDING):
            self.set_pending(P[c0 : c0 + MAX_PENDING])
            cols.append(self.cross_cov(X))
        return cols[0] if len(cols) == 1 else torch.cat(cols, dim=1).contiguous()

    def qlogei_pending_big(self, mean, var, cross, X_pending: np.ndarray, z: np.ndarray, best_f: float, sign: float = 1.0,
                           alive=None, stats=None):
        """qLogEI of N t-batches [x_i ; pending] with explicit pending statistics (``bbh_qlogei_pending_big``), 1 ... 63
        pending points: ``stats`` = (mean [p], cov [p, p]) of the pending points, by default from ``posterior_joint``; their
        cross-covariances from ``cross_cov`` / ``cross_cov_many``."""
        torch = self._torch()
        P = np.ascontiguousarray(np.atleast_2d(X_pending), dtype=np.float64)
        p = P.shape[0]
        if not 1 <= p <= MAX_PENDING_BIG or cross.shape[1] != p:
            raise ValueError(f"qlogei_pending_big handles 1 ... {MAX_PENDING_BIG} pending points")
        mp, cpp = stats if stats is not None else self.posterior_joint(P)
        z = np.ascontiguousarray(z, dtype=np.float64)
        N = mean.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        cpp = np.ascontiguousarray(cpp, dtype=np.float64)
        self._check(
            self._lib.bbh_qlogei_pending_big(
                self._h, mean.data_ptr(), var.data_ptr(), cross.data_ptr(), N, p, _dp(np.ascontiguousarray(mp)), _dp(cpp), _dp(z),
                z.shape[0], float(best_f), float(sign), alive.data_ptr() if alive is not None else None, scores.data_ptr(),
            ),
            "bbh_qlogei_pending_big",
        )
        return scores

    def qlogei_pending(self, mean, var, cross, z: np.ndarray, best_f: float, sign: float = 1.0, alive=None):
        torch = self._torch()
        z = np.ascontiguousarray(z, dtype=np.float64)
        S = z.shape[0]
        N = mean.shape[0]
        scores = torch.empty(N, dtype=torch.float64, device=mean.device)
        self._check(
            self._lib.bbh_qlogei_pending(
                self._h, mean.data_ptr(), var.data_ptr(), cross.data_ptr(), N, _dp(z), S, float(best_f), float(sign),
                alive.data_ptr() if alive is not None else None, scores.data_ptr(),
            ),
            "bbh_qlogei_pending",
        )
        return scores

    def sobol_normal_dev(self, S: int, q: int, seed: int):
        """[S, q] base samples of ``SobolQMCNormalSampler`` as a device tensor, produced on the device (``bbh_sobol_normal_dev``;
        asynchronous on the handle's stream): what ``sobol_normal_base_samples`` returns, up to the last bits of erfinv."""
        torch = self._torch()
        out = torch.empty((S, q), dtype=torch.float64, device=self._dev())
        state0 = _sobol_state0(q)
        self._check(self._lib.bbh_sobol_normal_dev(self._h, state0.ctypes.data_as(_lib.c_int64_p), int(seed) & 0xFFFFFFFFFFFFFFFF, S, q,
                                                   out.data_ptr()), "bbh_sobol_normal_dev")
        return out

    # ---- selection ------------------------------------------------------------------------
    def argmax(self, scores):
        v, i = C.c_double(), C.c_int64()
        self._check(self._lib.bbh_argmax(self._h, scores.data_ptr(), scores.shape[0], C.byref(v), C.byref(i)), "bbh_argmax")
        return v.value, i.value

    def topk(self, scores, k: int):
        vals = np.empty(k)
        idx = np.empty(k, dtype=np.int64)
        self._check(
            self._lib.bbh_topk(self._h, scores.data_ptr(), scores.shape[0], k, _dp(vals), idx.ctypes.data_as(_lib.c_int64_p)),
            "bbh_topk",
        )
        return vals, idx

    # ---- row-sharded selection through the library's own RCCL communicator ---------------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(256)
        n = self._lib.bbh_comm_unique_id(buf, 256)
        if n <= 0:
            raise HipError("bbh_comm_unique_id failed: RCCL (librccl.so.1) could not be loaded")
        return buf.raw[:n]

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        self._check(self._lib.bbh_comm_init(self._h, int(rank), int(world), unique_id, len(unique_id)), "bbh_comm_init")
        self._comm = (int(rank), int(world))

    def allgather_topk(self, scores, row_offset: int, k: int):
        """Global top-k over all shards (one ncclAllGather of k (score, global index) pairs per rank)."""
        vals, idx = np.empty(k), np.empty(k, dtype=np.int64)
        self._check(
            self._lib.bbh_allgather_topk(self._h, scores.data_ptr() if scores.shape[0] else None, scores.shape[0],
                                         int(row_offset), int(k), _dp(vals), idx.ctypes.data_as(_lib.c_int64_p)),
            "bbh_allgather_topk",
        )
        return vals, idx

    def allgather_argmax(self, scores, row_offset: int, X):
        """One greedy step over all shards: (score, global row index, comp-rep row [d]) of the global winner."""
        val, gidx, row = C.c_double(), C.c_int64(), np.empty(self.spec.d)
        N = scores.shape[0]
        self._check(
            self._lib.bbh_allgather_argmax(self._h, scores.data_ptr() if N else None, N, int(row_offset),
                                           X.data_ptr() if N else None, X.stride(0) if N else self.spec.d, C.byref(val),
                                           C.byref(gidx), _dp(row)),
            "bbh_allgather_argmax",
        )
        return val.value, gidx.value, row

    def set_slice_rows(self, rows: int):
        """Row count the sample-slice heuristics of the joint q'-batch / qLogNEHVI kernels use instead of each call's own N
        (``bbh_set_slice_rows``; 0 = per call).  A row shard passes the GLOBAL count to add every candidate's partial sums in the
        order the unsharded pass uses: scores are then bit-identical across shard layouts, at the price of fewer slices (less
        parallelism) per rank."""
        self._check(self._lib.bbh_set_slice_rows(self._h, int(rows)), "bbh_set_slice_rows")

    # ---- instrumentation ------------------------------------------------------------------
    def posterior_kernel_form(self) -> str:
        """Which form of the fused posterior kernel the last variance pass ran as."""
        return {0: "windowed", 1: "cooperative", 2: "materialised", 3: "cooperative-2sweep", 4: "cooperative-generic",
                5: "register-resident", 6: "feature-space"}.get(self._lib.bbh_last_posterior_form(self._h), "none")

    def timing(self, enable, families=None):
        """HIP-event timing of the kernel families on / off; ``families`` (names as for ``timing_read``) restricts the events to those
        families - an event between two back-to-back kernels costs the stream ~5 us."""
        code = 0
        if enable:
            code = 1 if not families else 2 * sum(1 << _lib.TIMED_FAMILIES[f] for f in families)
        self._check(self._lib.bbh_timing_enable(self._h, code), "bbh_timing_enable")

    def timing_read(self, reset: bool = True, family: str = "posterior"):
        """(total ms, launches) of one kernel family since the last reset: ``"posterior"`` (variance passes of the fused
        kernel), ``"cross"`` (its mean-only passes), ``"pending"`` (joint q'-batch acquisition kernels)."""
        ms, cnt = C.c_double(), C.c_int64()
        self._check(self._lib.bbh_timing_read_family(self._h, _lib.TIMED_FAMILIES[family], C.byref(ms), C.byref(cnt),
                                                     1 if reset else 0), "bbh_timing_read_family")
        return ms.value, cnt.value

    # ---- optimize_acqf_discrete with qLogEI -----------------------------------------------
    def greedy_qlogei(self, *args, **kwargs) -> "GreedyResult":
        """``_greedy_qlogei`` with the shard's slice geometry set for its duration only (an exception inside the loop used
        to leave ``slice_rows`` on a handle that goes back to the pool)."""
        import inspect

        # (``shard`` may arrive positionally: read it from the bound arguments of the real signature - ADVICE r5)
        shard = inspect.signature(self._greedy_qlogei).bind(*args, **kwargs).arguments.get("shard")
        repro = shard is not None and getattr(shard, "reproducible", False)
        if repro:
            self.set_slice_rows(shard.N_total)
        try:
            return self._greedy_qlogei(*args, **kwargs)
        finally:
            if repro:
                self.set_slice_rows(0)

    def _greedy_qlogei(
        self,
        X,
        q: int,
        S: int = 512,
        seed: int | None = None,
        sign: float = 1.0,
        X_pending: np.ndarray | None = None,
        best_f: float | None = None,
        z_by_q: dict | None = None,
        shard=None,
        kind: str = "qLogEI",
        beta: float = 0.2,
        alive=None,
        speculate: bool = True,
    ) -> GreedyResult:
        """Sequential greedy of ``optimize_acqf_discrete(acqf, q, choices, unique=True)``.

        Step 0 runs the fused posterior over all candidates and caches (mean, var); every later
        step only needs the cross-covariances with the points picked so far (mean-only pass).
        ``alive`` (uint8 device tensor) restricts the candidates to a subset of the resident rows.
        ``shard`` (a ``baybe_amd.distributed.RowShard``) makes every selection a global one: the
        local winner is all-gathered (score, global index, row) and the global first-index argmax
        wins on every rank.
        """
        torch = self._torch()
        X = self._as_dev(X)
        N = X.shape[0]
        d = self.spec.d
        if seed is None and z_by_q is None:
            seed = draw_sampler_seed()
        if best_f is None:
            best_f = self.best_f(sign)
        base = np.zeros((0, d)) if X_pending is None or len(X_pending) == 0 else np.atleast_2d(np.asarray(X_pending, dtype=np.float64))
        alive = torch.ones(N, dtype=torch.uint8, device=X.device) if alive is None else alive.clone()
        mean = var = None
        chosen_rows: list[np.ndarray] = []
        indices, values = [], []

        def get_z(qp: int) -> np.ndarray:
            return z_by_q[qp] if z_by_q is not None else sobol_normal_base_samples(S, qp, seed)

        if q > 0 and N > 0:  # the first pass is enqueued before the host scrambles the first step's base samples (0.6 ms)
            self.set_pending(None)  # (a variance pass without pending columns: the cooperative kernel forms)
            mean, var = self.posterior(X)
        z_next = get_z(1 + base.shape[0])
        # Cross-covariance columns ahead of time.  Every later step needs cov(candidate, pending point) for the points picked
        # so far - one mean-only pass of the fused kernel over all candidates per step, whose cost does not depend on the number
        # of columns (<= 15 ride on one MFMA column block).  The picks of the later steps come almost always from the head of
        # the first step's ranking, so after the first step ONE such pass computes the columns of its top candidates; a later
        # step whose pending points are all among them gathers its columns (bit-identical values: a column is an independent
        # dot product) instead of launching its own pass, any other step falls back to its own pass.
        spec_rows, spec_pos, cross_spec, big_cross, spec_stats = None, {}, None, None, None
        first_vals = first_top = None
        b0 = base.shape[0]
        for _step in range(q):
            pend = np.vstack([base] + chosen_rows) if chosen_rows else base
            p = pend.shape[0]
            z = z_next
            if p == 0:
                if mean is None:  # first step: posterior of every candidate, cached for the later steps
                    self.set_pending(None)
                    mean, var = self.posterior(X)
                if kind == "qLogEI" and shard is None and N > 0:
                    # scores, the winner and (for later steps) the head of the ranking in ONE call
                    m_spec = min(MAX_PENDING - b0, N, 4 * q) if (q > 1 and speculate and N > 1 and b0 < MAX_PENDING) else 1
                    scores, first_vals, first_top = self.qlogei_topk(mean, var, z[:, 0], best_f, sign, max(1, m_spec), alive)
                else:
                    scores = self.mc_acq(kind, mean, var, z[:, 0], best_f, sign, beta, alive)
            else:
                if mean is None:
                    mean, var = self.posterior(X)
                if p > MAX_PENDING:
                    # more than 15 pending points (the reference has no cap): columns of earlier steps are kept, the new pick's
                    # column comes from a one-point pass; the joint kernel takes the pending statistics explicitly
                    if kind != "qLogEI":
                        raise ValueError(f"joint q-batches beyond {MAX_PENDING + 1} points are available for qLogEI only")
                    if p > MAX_PENDING_BIG:
                        raise ValueError(f"at most {MAX_PENDING_BIG} pending points (batch size + pending experiments <= "
                                         f"{MAX_PENDING_BIG + 1}) are supported by the HIP path")
                    if big_cross is None or big_cross.shape[1] != p - 1:
                        big_cross = self.cross_cov_many(X, pend)
                    else:
                        big_cross = torch.cat([big_cross, self.cross_cov_many(X, pend[-1:])], dim=1).contiguous()
                    scores = self.qlogei_pending_big(mean, var, big_cross, pend, z, best_f, sign, alive)
                else:
                    cols = None
                    if cross_spec is not None:
                        cols = [spec_pos.get(ix) for ix in indices]
                        cols = None if any(c is None for c in cols) else list(range(b0)) + cols
                    if cols is not None and kind == "qLogEI" and spec_stats is not None:
                        # columns AND pending statistics are sub-blocks of what the speculative pass left: no bbh_pending_set
                        # round trip (0.15 - 0.3 ms of host work per step)
                        cross = cross_spec[:, cols].contiguous() if len(cols) != cross_spec.shape[1] else cross_spec
                        ix = np.asarray(cols)
                        scores = self.qlogei_pending_big(mean, var, cross, pend, z, best_f, sign, alive,
                                                         stats=(spec_stats[0][ix], spec_stats[1][np.ix_(ix, ix)]))
                    else:
                        self.set_pending(pend)
                        if cols is not None:
                            cross = cross_spec[:, cols].contiguous() if len(cols) != cross_spec.shape[1] else cross_spec
                        else:
                            cross = self.cross_cov(X)
                        scores = self.mc_acq(kind, mean, var, z, best_f, sign, beta, alive, cross=cross)
                    if p == MAX_PENDING:
                        big_cross = cross  # the next step continues from these columns
            step0_global = None
            if _step == 0 and q > 1 and speculate and b0 < MAX_PENDING and (N > 1 or shard is not None):
                if shard is None:
                    m_spec = min(MAX_PENDING - b0, N, 4 * q)
                    tv, top = (first_vals, first_top) if first_top is not None and len(first_top) == m_spec else self.topk(scores, m_spec)
                    keep = [j for j, (t, v) in enumerate(zip(top, tv)) if t >= 0 and v > -math.inf]  # live candidates only
                    top = [int(top[j]) for j in keep]
                    spec_rows = X[torch.as_tensor(top, device=X.device), :d].cpu().numpy() if top else None
                else:
                    # row shards: the head of the GLOBAL ranking with its rows - one all-gather of each rank's local head; its
                    # first entry is this step's winner, so the step needs no collective of its own
                    m_spec = min(MAX_PENDING - b0, shard.N_total, 4 * q)
                    lv, li = self.topk(scores, min(m_spec, N)) if N > 0 else (np.empty(0), np.empty(0, dtype=np.int64))
                    tv, top, rows = shard.global_topk_rows(lv, li, m_spec, X, d)
                    if len(top):
                        step0_global = (float(tv[0]), int(top[0]), rows[0].copy())
                    keep = [j for j, v in enumerate(tv) if v > -math.inf]
                    top = [int(top[j]) for j in keep]
                    spec_rows = rows[keep] if keep else None
                if top:
                    spec_stats = self.set_pending(np.vstack([base, spec_rows]))  # (mean, cov) of base + speculative points
                    cross_spec = self.cross_cov(X) if N > 0 else torch.empty((0, b0 + len(top)), dtype=torch.float64, device=X.device)
                    spec_pos = {ix: b0 + j for j, ix in enumerate(top)}
            if _step + 1 < q:  # host-side Sobol scrambling of the next step overlaps the device work of this one
                z_next = get_z(2 + p)
            if step0_global is not None:  # (sharded first step: the winner came with the speculative head)
                val, gidx, row = step0_global
                if shard.owns(gidx):
                    alive[shard.to_local(gidx)] = 0
                idx = gidx
            elif shard is not None and shard.rccl_bound(self):  # payload built on the device, one ncclAllGather
                val, gidx, row = self.allgather_argmax(scores, shard.start, X)
                if shard.owns(gidx):
                    alive[shard.to_local(gidx)] = 0
                idx = gidx
            elif shard is not None:
                val, idx = self.argmax(scores) if N > 0 else (-math.inf, -1)
                val, gidx, row = shard.global_argmax(val, idx, X)
                if shard.owns(gidx):
                    alive[shard.to_local(gidx)] = 0
                idx = gidx
            else:
                if _step == 0 and first_top is not None:
                    val, idx = float(first_vals[0]), int(first_top[0])
                else:
                    val, idx = self.argmax(scores) if N > 0 else (-math.inf, -1)
                if idx in spec_pos and spec_rows is not None:  # the winner is one of the speculative heads: its row is on the host already
                    row = spec_rows[spec_pos[idx] - b0]
                else:
                    row = X[idx, :d].cpu().numpy()
                alive[idx] = 0
            indices.append(int(idx))
            values.append(float(val))
            chosen_rows.append(np.asarray(row, dtype=np.float64).reshape(1, d))
        self.set_pending(None if (base.shape[0] == 0 or base.shape[0] > MAX_PENDING) else base)
        return GreedyResult(indices, values)
