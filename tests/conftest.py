import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def record_deviation(name: str, value: float, tolerance: float) -> None:
    """Observed maximum deviation of a parity comparison next to the tolerance it was held to - written to
    ``gpurun_out/observed_deviations.json`` (merged back by gpurun; the round's copy is committed under ``profiles/``), so that a
    tolerance is never the only number on record ("the observed maximum is printed but recorded nowhere")."""
    import json
    from pathlib import Path

    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        fn = out / "observed_deviations.json"
        data = json.loads(fn.read_text()) if fn.exists() else {}
        data[name] = {"observed_max": float(value), "tolerance": float(tolerance)}
        fn.write_text(json.dumps(data, indent=1, sort_keys=True))
    except OSError:
        pass
