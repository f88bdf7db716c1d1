"""Latency of recommend() on a small space (BASELINE configs[0] shape: 1000 candidates, n_train = 20,
batch 3) - the regime of backtesting loops: where does a call spend its time?"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd.recommenders import HipBotorchRecommender

rng = np.random.default_rng(0)
vals = np.arange(10) / 9.0
space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
exp = space.discrete.exp_rep
def f(X): return -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(6.28 * X[:, 0])
meas = exp.iloc[rng.choice(len(exp), 20, replace=False)].copy()
meas["yield"] = f(meas.to_numpy(dtype=float)) + 0.05 * rng.standard_normal(20)
rec = HipBotorchRecommender()
camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), rec)
camp.add_measurements(meas)
torch.manual_seed(0)
t0 = time.perf_counter(); camp.recommend(3); torch.cuda.synchronize(); print(f"first recommend: {(time.perf_counter()-t0)*1e3:.1f} ms")
ts = []
for it in range(10):
    got = camp.recommend(3)
    new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)) + 0.05 * rng.standard_normal(len(new))
    camp.add_measurements(new)
    t0 = time.perf_counter(); camp.recommend(3); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("recommend after new measurements (refit) ms:", " ".join(f"{t:.1f}" for t in ts))
sur = rec._surrogate_model
print("fit info:", getattr(sur, "_fit_info", None))
import cProfile, pstats
got = camp.recommend(3); new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)); camp.add_measurements(new)
pr = cProfile.Profile(); pr.enable(); camp.recommend(3); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
