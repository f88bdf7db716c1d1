"""End-to-end recommend() through the plug-in surface at BASELINE size (1e6 x 20 grid, n_train = 512,
batch 5): fit + best_f + upload + greedy batch + index mapping, first call vs cached second call."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, pandas as pd, torch
from _baybe_shim import NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd.recommenders import HipBotorchRecommender

N, d, n, q = 1_000_000, 20, 512, 5
rng = np.random.default_rng(0)
t0 = time.time()
df = pd.DataFrame(rng.integers(0, 11, size=(N, d)) / 10.0, columns=[f"x{i}" for i in range(d)])
space = SearchSpace.from_dataframe(df)
print(f"search space built in {time.time() - t0:.1f} s (host, not part of the path)")
exp = space.discrete.exp_rep
meas = exp.iloc[np.random.default_rng(1).choice(N, n, replace=False)].copy()
Xm = meas.to_numpy(float)
meas["y"] = -((Xm - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * Xm[:, 0]) + 0.05 * rng.standard_normal(n)
obj = SingleTargetObjective(NumericalTarget("y"))
rec = HipBotorchRecommender()
torch.manual_seed(0)
for call in range(3):
    mask = np.ones(N, bool); mask[meas.index] = False
    if call == 2:  # third call: new measurements -> refit, smaller candidate set, same resident matrix
        extra = exp.loc[got.index].copy(); Xe = extra.to_numpy(float)
        extra["y"] = -((Xe - 0.5) ** 2).sum(1)
        meas = pd.concat([meas, extra]); mask[got.index] = False
    sp = space.filtered(mask)
    torch.cuda.synchronize(); t0 = time.time()
    got = rec.recommend(q, sp, obj, meas)
    torch.cuda.synchronize(); t1 = time.time()
    fi = rec._surrogate_model._fit_info
    print(f"recommend(batch={q}) call {call}: {1e3 * (t1 - t0):.1f} ms  (fit nfev {fi.nfev if fi else None})  -> {got.index.tolist()}")
