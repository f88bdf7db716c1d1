"""Where does a configs[4] ``recommend(5)`` go?  The bench's own e2e scenario (1e5 x 15 grid, 3 targets, n = 256,
qLogNEHVI S = 512) under cProfile: a call on unchanged measurements (pruning + greedy batch) and a call after new measurements
(three fits in addition)."""
import cProfile
import pstats
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, pandas as pd, torch
import bench
from baybe_amd.acquisition import qLogNoisyExpectedHypervolumeImprovement
from baybe_amd.recommenders import HipBotorchRecommender

N, d, n, S, q = 100_000, 15, 256, 512, 5
X, Xt, y = bench.synth_problem(N, d, n, 0)
ys = bench.synth_pareto_targets(Xt)
space = bench._BenchSpace(X)
cols = list(space.comp_rep_columns)
meas = pd.DataFrame(Xt, columns=cols)
names = [f"y{o}" for o in range(3)]
for nm, yo in zip(names, ys):
    meas[nm] = yo
objective = SimpleNamespace(targets=tuple(SimpleNamespace(name=nm, minimize=False, transformation=None) for nm in names), is_multi_output=True)
rec = HipBotorchRecommender(acquisition_function=qLogNoisyExpectedHypervolumeImprovement(n_mc_samples=S))


def call(m, label, prof=False):
    torch.manual_seed(0)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if prof else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    got = rec.recommend(q, space, objective, m)
    torch.cuda.synchronize()
    if pr:
        pr.disable()
    print(f"{label}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
    if pr:
        st = pstats.Stats(pr).stats
        rows = sorted(((tt, ct, nc, f"{Path(k[0]).name}:{k[1]}({k[2]})") for k, (cc, nc, tt, ct, _) in st.items()), reverse=True)[:28]
        print("own_us   cum_us  calls  function")
        for tt, ct, nc, name in rows:
            print(f"{tt * 1e6:7.0f} {ct * 1e6:8.0f} {nc:6d}  {name}")
        nv = rec._nehvi
        if nv is not None:
            print("last prune parts", getattr(nv, "last_prune_ms", None), "last set-up", nv.last_setup_ms)
    return got


call(meas, "first call")
call(meas, "unchanged")
got = call(meas, "unchanged (profiled)", prof=True)
call(meas, "unchanged again")
more = pd.concat([meas, got.assign(**{nm: float(np.mean(yo)) for nm, yo in zip(names, ys)})], ignore_index=True)
call(more, "after new measurements (profiled)", prof=True)
call(more, "unchanged after refit")
more2 = pd.concat([more, got.assign(**{nm: float(np.mean(yo)) + 0.01 for nm, yo in zip(names, ys)})], ignore_index=True)
call(more2, "after new measurements again")
