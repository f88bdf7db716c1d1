"""Latency of a hybrid-space recommend() on the HIP path: every discrete row x n_raw_samples Sobol points scored in one
pass, compass refinement of the n_restarts best, greedy batch.  Three shapes; the discrete part all-numerical, two or four continuous
parameters; 25 measurements; batch 3.  Also the purely continuous case (no discrete rows)."""
import itertools
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, pandas as pd, torch
from _replay import HybridSpace
from baybe_amd.recommenders import HipBotorchRecommender

rng = np.random.default_rng(0)
for levels, dd, dc in ((7, 3, 2), (10, 4, 2), (10, 5, 4), (0, 0, 3)):
    disc = pd.DataFrame(list(itertools.product(*[np.linspace(0, 1, levels)] * dd)) if dd else np.zeros((0, 0)), columns=[f"d{i}" for i in range(dd)])
    bounds = pd.DataFrame({f"c{i}": [0.0, 1.0 + i] for i in range(dc)}, index=["min", "max"])
    space = HybridSpace(disc, bounds)
    cols = list(space.comp_rep_columns)
    n = 25
    D = disc.to_numpy()[rng.choice(len(disc), n)] if dd else np.zeros((n, 0))
    C = rng.random((n, dc)) * (1.0 + np.arange(dc))
    M = np.hstack([D, C])
    y = -((M - 0.4) ** 2).sum(1)
    meas = pd.DataFrame(M, columns=cols).assign(y=y)
    objective = SimpleNamespace(targets=(SimpleNamespace(name="y", minimize=False, transformation=None),), is_multi_output=False)
    rec = HipBotorchRecommender()
    torch.manual_seed(1)
    t0 = time.perf_counter(); got = rec.recommend(3, space, objective, meas); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); got = rec.recommend(3, space, objective, meas); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    rows = max(len(disc), 1) * min(rec.n_raw_samples, 4_000_000 // max(len(disc), 1))
    print(f"discrete rows {len(disc):6d} (d = {dd}) x continuous d = {dc}: {rows:8d} scored rows per step, n_raw_samples {rec.n_raw_samples}, n_restarts {rec.n_restarts}; "
          f"recommend(3): first {first:7.1f} ms, then {np.median(ts):6.1f} ms (min {min(ts):.1f})", flush=True)
