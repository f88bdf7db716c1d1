"""The C-ABI library loads here (no GPU) and exports every symbol include/baybe_hip.h declares;
without a device the product path fails loudly (no CPU fallback)."""

import re
from pathlib import Path

import pytest

from baybe_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "baybe_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_bound_and_exported():
    lib = _lib.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert name in _lib.SIGNATURES, f"{name} declared in the header but not bound in _lib.py"
        assert hasattr(lib, name), f"{name} not exported by libbaybe_hip.so"
    assert set(_lib.SIGNATURES) == set(names)


def test_version():
    assert _lib.load_library().bbh_version() >= 100


def test_no_device_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from baybe_amd import engine

    assert _lib.is_available() is False
    with pytest.raises(_lib.HipUnavailableError):
        engine.HipGP(0)


def test_product_code_never_imports_the_oracle():
    for path in (ROOT / "baybe_amd").rglob("*.py"):
        src = path.read_text()
        assert "import oracle" not in src and "from oracle" not in src, path
        # nor the test-only stand-ins (tests/_stubs: cattrs, botorch objective wrappers) or the CPU double of the device
        for name in ("_stubs", "_oracle_engine", "_reference", "_replay"):
            assert name not in src, (path, name)
