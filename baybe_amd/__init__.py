"""baybe_amd — MI355X-native GP recommend() hot path for BayBE (HIP kernels behind a C-ABI).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C-ABI of
``include/baybe_hip.h``), the ctypes binding, the host driver (``engine``), and the host-side
mirror of BayBE's surrogate / recommender plug-in surface (``surrogates``, ``recommenders``).
"""

from baybe_amd._lib import HipError, HipUnavailableError, is_available, library_path

__all__ = ["HipError", "HipUnavailableError", "is_available", "library_path"]
__version__ = "0.1.0"
