"""The oracle must be an *independent* restatement: it shares no logic with the product (half of the
fit objective used to be compared with a copy of itself) and only tests / smoke() / bench's CPU-baseline leg use it."""

import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
MIN_CHARS = 30  # shorter lines (``except np.linalg.LinAlgError:``, ``return out`` ...) carry no logic of their own


def _logic_lines(paths):
    found = {}
    for path in paths:
        in_doc = False
        for no, line in enumerate(path.read_text().splitlines(), 1):
            text = line.strip()
            if text.count('"""') % 2 == 1:
                in_doc = not in_doc
                continue
            if in_doc or not text or text.startswith(("#", "import ", "from ", '"""')):
                continue
            text = text.split("  #")[0].strip()
            if len(re.sub(r"\s", "", text)) >= MIN_CHARS:
                found.setdefault(text, f"{path.relative_to(ROOT)}:{no}")
    return found


def test_oracle_and_product_share_no_lines():
    oracle = _logic_lines(sorted((ROOT / "oracle").glob("*.py")))
    product = _logic_lines(sorted((ROOT / "baybe_amd").glob("*.py")))
    shared = sorted(set(oracle) & set(product))
    assert not shared, "\n".join(f"{oracle[t]} == {product[t]}: {t}" for t in shared)


def test_fit_objective_is_built_from_library_parts():
    """oracle/fit_objective.py: torch.distributions log-densities, softplus and autograd - no hand-written
    derivative, no import of the product."""
    src = (ROOT / "oracle" / "fit_objective.py").read_text()
    for needle in ("torch.autograd.grad", "MultivariateNormal(", "Gamma(", "LogNormal(", "F.softplus", "cholesky_inverse"):
        assert needle in src, needle
    for path in (ROOT / "oracle").glob("*.py"):
        assert "baybe_amd" not in re.sub(r'""".*?"""', "", path.read_text(), flags=re.S), path
