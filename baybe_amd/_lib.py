"""ctypes binding of ``libbaybe_hip.so`` (C-ABI declared in ``include/baybe_hip.h``).

There is deliberately NO fallback: if the shared library is missing, or no HIP
device is present, every entry point raises ``HipUnavailableError``.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_NAME = "libbaybe_hip.so"
_lib = None

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class HipUnavailableError(RuntimeError):
    """The HIP extension (or a HIP device) is not available; nothing else can run the path."""


class HipError(RuntimeError):
    """A libbaybe_hip call failed."""


class ModelDesc(C.Structure):
    """``bbh_model_desc`` (include/baybe_hip.h)."""

    _fields_ = [
        ("kernel_kind", C.c_int32),
        ("d", C.c_int32),
        ("task_col", C.c_int32),
        ("n_tasks", C.c_int32),
        ("use_outputscale", C.c_int32),
        ("criterion", C.c_int32),
        ("hadamard", C.c_int32),
        ("n_factors", C.c_int32),
        ("combine", C.c_int32),
        ("factor_kind", C.c_int32 * 4),
        ("factor_scaled", C.c_int32 * 4),
        ("factor_group", C.c_int32 * 4),
    ]


KERNEL_KINDS = {"matern12": 0, "matern32": 1, "matern52": 2, "rbf": 3,
                "piecewise0": 4, "piecewise1": 5, "piecewise2": 6, "piecewise3": 7, "rq": 8,
                "linear": 9, "poly1": 10, "poly2": 11, "poly3": 12, "poly4": 13, "periodic": 14, "rff": 15}  # enum bbh_kernel_kind
CRITERIA = {"mll": 0, "loo": 1}
MAX_PENDING = 15  # pending points per cross-covariance pass / in the handle's pending state (BBH_MAX_PENDING)
MAX_PENDING_BIG = 63  # joint q'-batches through bbh_qlogei_pending_big: q' = 1 + pending <= 64 (qLogEI)
MAX_RFF_SAMPLES = 256  # RFFKernel(num_samples): frequencies of the feature-space model (csrc/bbh_rff.hip: m = 2 D <= 512)
MAX_OBJECTIVES = 4
TIMED_FAMILIES = {"posterior": 0, "cross": 1, "pending": 2, "columns": 3, "nehvi": 4, "q1": 5, "select": 6}  # enum bbh_timed_family
ACQ_KINDS = {"qLogEI": 0, "qEI": 1, "qPI": 2, "qSR": 3, "qUCB": 4, "qPSTD": 5,
             "PM": 10, "PSTD": 11, "UCB": 12, "EI": 13, "LogEI": 14, "PI": 15}

# name -> (restype, argtypes); every symbol include/baybe_hip.h declares
SIGNATURES = {
    "bbh_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bbh_destroy": (C.c_int, [C.c_void_p]),
    "bbh_last_error": (C.c_char_p, [C.c_void_p]),
    "bbh_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bbh_version": (C.c_int, []),
    "bbh_selftest": (C.c_int, [C.c_void_p]),
    "bbh_set_rff_weights": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32]),
    "bbh_set_model": (
        C.c_int,
        [C.c_void_p, C.POINTER(ModelDesc), C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p],
    ),
    "bbh_set_model_ex": (
        C.c_int,
        [C.c_void_p, C.POINTER(ModelDesc), C.c_int64, c_double_p, c_double_p, c_double_p, c_double_p,
         C.POINTER(C.c_uint8), C.c_int, C.c_double, C.c_double],
    ),
    "bbh_set_slice_rows": (C.c_int, [C.c_void_p, C.c_int64]),
    "bbh_trim": (C.c_int, [C.c_void_p, C.c_int64]),
    "bbh_theta_len": (C.c_int64, [C.c_void_p]),
    "bbh_get_standardization": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "bbh_fit_value_grad": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    "bbh_factorize": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "bbh_posterior": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "bbh_posterior_unfused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "bbh_posterior_joint": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    "bbh_set_mean_columns": (C.c_int, [C.c_void_p, c_double_p, C.c_int64]),
    "bbh_posterior_columns": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "bbh_posterior_columns_sm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "bbh_train_posterior_mean": (C.c_int, [C.c_void_p, c_double_p]),
    "bbh_qlogei_q1": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_double_p, C.c_int64, C.c_double, C.c_double,
         C.c_void_p, C.c_void_p],
    ),
    "bbh_qlogei_q1_topk": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_double_p, C.c_int64, C.c_double, C.c_double,
         C.c_void_p, C.c_void_p, C.c_int64, c_double_p, c_int64_p],
    ),
    "bbh_score_qlogei": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, c_double_p, C.c_int64, C.c_double, C.c_double, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "bbh_pending_set": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, c_double_p, c_double_p]),
    "bbh_cross_cov": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "bbh_qlogei_pending": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_double_p, C.c_int64, C.c_double,
         C.c_double, C.c_void_p, C.c_void_p],
    ),
    "bbh_qlognehvi": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_double_p, c_double_p,
         C.c_int64, c_int64_p, c_double_p, c_double_p, C.c_void_p, C.c_void_p],
    ),
    "bbh_qlognehvi_sm": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_double_p, c_double_p,
         C.c_int64, c_int64_p, c_double_p, c_double_p, C.c_void_p, C.c_void_p],
    ),
    "bbh_qlogei_pending_big": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, c_double_p, c_double_p, c_double_p, C.c_int64,
         C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    ),
    "bbh_mc_acq_q1": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, c_double_p, C.c_int64, C.c_double, C.c_double,
         C.c_double, C.c_void_p, C.c_void_p],
    ),
    "bbh_mc_acq_pending": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_double_p, C.c_int64, C.c_double,
         C.c_double, C.c_double, C.c_void_p, C.c_void_p],
    ),
    "bbh_analytic_acq": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int32,
         C.c_void_p, C.c_void_p],
    ),
    "bbh_pareto_frequency": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_int64_p]),
    "bbh_pareto_frequency_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_int64_p]),
    "bbh_nehvi_samples": (C.c_int, [C.c_void_p, c_double_p, C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "bbh_nehvi_samples_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int32]),
    "bbh_cells_build_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_int64_p, c_int64_p]),
    "bbh_cells_read_dev": (C.c_int, [C.c_void_p, c_int64_p, c_double_p, c_double_p]),
    "bbh_qlognehvi_cells": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_double_p, c_double_p,
         C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "bbh_cells_create": (C.c_int, [c_double_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, C.POINTER(C.c_void_p), c_int64_p]),
    "bbh_cells_get": (C.c_int, [C.c_void_p, c_int64_p, c_double_p, c_double_p]),
    "bbh_cells_destroy": (C.c_int, [C.c_void_p]),
    "bbh_sobol_scramble": (C.c_int, [c_int64_p, c_int64_p, C.c_int64]),
    "bbh_sobol_draw": (C.c_int, [c_int64_p, c_int64_p, C.c_int64, C.c_int64, c_double_p]),
    "bbh_sobol_normal": (C.c_int, [c_int64_p, C.c_uint64, C.c_int64, C.c_int64, c_double_p]),
    "bbh_sobol_normal_dev": (C.c_int, [C.c_void_p, c_int64_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p]),
    "bbh_content_key": (C.c_uint64, [C.POINTER(C.c_void_p), c_int64_p, C.c_int32, C.c_int32]),
    "bbh_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, c_double_p, c_int64_p]),
    "bbh_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, c_double_p, c_int64_p]),
    "bbh_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int64]),
    "bbh_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "bbh_comm_destroy": (C.c_int, [C.c_void_p]),
    "bbh_allgather_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, c_double_p, c_int64_p]),
    "bbh_allgather_argmax": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, c_double_p, c_int64_p, c_double_p],
    ),
    "bbh_last_posterior_form": (C.c_int, [C.c_void_p]),
    "bbh_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bbh_timing_read": (C.c_int, [C.c_void_p, c_double_p, c_int64_p, C.c_int]),
    "bbh_timing_read_family": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, c_int64_p, C.c_int]),
    "bbh_flow_trace_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "bbh_tiles_trace_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
}


def library_path() -> Path:
    env = os.environ.get("BAYBE_AMD_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / _LIB_NAME


def load_library():
    """Load the shared library and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP/HSA runtime: load it FIRST so that libbaybe_hip.so binds to
    # the same libamdhip64 (same SONAME) instead of pulling a second runtime from /opt/rocm into
    # the process (two HSA runtimes in one process leave torch with "No HIP GPUs are available").
    import torch  # noqa: F401

    path = library_path()
    if not path.exists():
        raise HipUnavailableError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C baybe_amd/csrc`. There is no CPU fallback for this path."
        )
    try:
        lib = C.CDLL(str(path))
    except OSError as ex:  # missing ROCm runtime etc.
        raise HipUnavailableError(f"cannot load {path}: {ex}") from ex
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def is_available() -> bool:
    """True iff the library loads AND a HIP device can be opened."""
    try:
        lib = load_library()
    except (HipUnavailableError, AttributeError):
        return False
    h = C.c_void_p()
    rc = lib.bbh_create(0, C.byref(h))
    if rc != 0:
        return False
    lib.bbh_destroy(h)
    return True
