"""Round 6: which ingredient of set_model stalls in the campaign loop?  Inside the same loop as gpu_stall_bisect.py, after
add_measurements (where recommend() would start) time (a) a torch kernel, (b) a torch copy from a PINNED host tensor just written by the
CPU (copy engine), (c) a fresh HipGP.set_model on a spare handle, (d) the real recommend()."""
import sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd import engine, gp_spec
from baybe_amd.recommenders import HipBotorchRecommender

x = torch.zeros(4096, device="cuda", dtype=torch.float64)
pin = torch.zeros(4096, dtype=torch.float64).pin_memory()


def t(fn):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


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
spare = engine.HipGP(0)
spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
Xs, ys = rng.random((110, d)), rng.random(110)
spare.set_model(spec, Xs, ys)
order = sys.argv[1] if len(sys.argv) > 1 else "kcsr"
for it in range(8):
    got = camp.recommend(batch)
    new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)); camp.add_measurements(new)
    out = []
    for ch in order:
        if ch == "k":
            out.append(("kernel", t(lambda: x.add_(1.0))))
        elif ch == "c":
            pin.add_(1.0)
            out.append(("pinned H2D 32 KB", t(lambda: x.copy_(pin, non_blocking=True))))
        elif ch == "s":
            out.append(("spare set_model", t(lambda: spare.set_model(spec, Xs, ys))))
        elif ch == "r":
            out.append(("recommend", t(lambda: camp.recommend(batch))))
        elif ch == "z":
            time.sleep(0.02)
    print(f"iteration {it}: " + ", ".join(f"{k} {v:7.3f}" for k, v in out), flush=True)
