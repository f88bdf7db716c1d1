"""Where does the 8 - 30 ms device stall of a campaign-like loop at 1e5 candidates come from (round 6)?  A tiny torch kernel + synchronise
is timed (a) right after recommend() returned, (b) after add_measurements, (c) after an extra host-only pause - the first slow probe
brackets the cause."""
import sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd.recommenders import HipBotorchRecommender

x = torch.zeros(1024, device="cuda", dtype=torch.float64)


def probe():
    t0 = time.perf_counter(); x.add_(1.0); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


def f(X):
    return -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(6.28 * X[:, 0])


levels, d, n0, batch = 18, 4, 100, 5
rng = np.random.default_rng(1)
vals = np.arange(levels) / (levels - 1.0)
space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(d)])
exp = space.discrete.exp_rep
meas = exp.iloc[rng.choice(len(exp), n0, replace=False)].copy(); meas["yield"] = f(meas.to_numpy(dtype=float))
rec = HipBotorchRecommender(); camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), rec)
camp.add_measurements(meas); camp.recommend(batch)
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
for it in range(8):
    t0 = time.perf_counter(); got = camp.recommend(batch); t_rec = (time.perf_counter() - t0) * 1e3
    pa = probe() if mode in ("all", "a") else float("nan")
    new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)); camp.add_measurements(new)
    pb = probe() if mode in ("all", "b") else float("nan")
    time.sleep(0.003)
    pc = probe() if mode in ("all", "c") else float("nan")
    print(f"iteration {it}: recommend {t_rec:6.2f} ms; probe after recommend {pa:7.3f}, after add_measurements {pb:7.3f}, after 3 ms more {pc:7.3f}", flush=True)
