"""cProfile of a cached recommend() call at BASELINE size (what besides the greedy batch costs time on the host?)."""
import cProfile, pstats, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, pandas as pd, torch
from _baybe_shim import NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd.recommenders import HipBotorchRecommender

N, d, n, q = 1_000_000, 20, 512, 5
rng = np.random.default_rng(0)
df = pd.DataFrame(rng.integers(0, 11, size=(N, d)) / 10.0, columns=[f"x{i}" for i in range(d)])
space = SearchSpace.from_dataframe(df)
exp = space.discrete.exp_rep
meas = exp.iloc[np.random.default_rng(1).choice(N, n, replace=False)].copy()
Xm = meas.to_numpy(float)
meas["y"] = -((Xm - 0.5) ** 2).sum(1) + 0.05 * rng.standard_normal(n)
obj = SingleTargetObjective(NumericalTarget("y"))
rec = HipBotorchRecommender()
mask = np.ones(N, bool); mask[meas.index] = False
sp = space.filtered(mask)
rec.recommend(q, sp, obj, meas); rec.recommend(q, sp, obj, meas)
torch.cuda.synchronize(); t0 = time.time(); rec.recommend(q, sp, obj, meas); torch.cuda.synchronize()
print(f"cached recommend: {1e3 * (time.time() - t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); rec.recommend(q, sp, obj, meas); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
