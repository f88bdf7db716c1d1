"""Where do the 8 ms of ``set_model`` at 104 976 candidates go?  Replays the small-space latency scenario
(18^4 grid, n ~ 110, batch 5) with ``BBH_SETMODEL_TRACE=1`` (stage stamps of ``bbh_set_model_ex`` on stderr) and times the Python
call with / without a device synchronisation in front of it: a slow call only WITHOUT the synchronisation means it waits for work the
previous ``recommend()`` left in flight, not for anything of its own."""
import os
import sys
import time
from pathlib import Path

os.environ["BBH_SETMODEL_TRACE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd import engine as _engine
from baybe_amd.recommenders import HipBotorchRecommender

_set_model = _engine.HipGP.set_model
SYNC_FIRST = [False]
TIMES = []


def _timed_set_model(self, *a, **k):
    if SYNC_FIRST[0]:
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts = (time.perf_counter() - t0) * 1e3
    else:
        ts = 0.0
    t0 = time.perf_counter()
    out = _set_model(self, *a, **k)
    TIMES.append((SYNC_FIRST[0], ts, (time.perf_counter() - t0) * 1e3))
    return out


_engine.HipGP.set_model = _timed_set_model


def f(X):
    return -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(6.28 * X[:, 0])


def main(levels=18, d=4, n0=100, batch=5):
    rng = np.random.default_rng(1)
    vals = np.arange(levels) / (levels - 1.0)
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(d)])
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), n0, replace=False)].copy(); meas["yield"] = f(meas.to_numpy(dtype=float))
    rec = HipBotorchRecommender(); camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), rec)
    camp.add_measurements(meas); camp.recommend(batch)
    for it in range(8):
        got = camp.recommend(batch); new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)); camp.add_measurements(new)
        SYNC_FIRST[0] = it % 2 == 1
        print(f"---- iteration {it}: synchronise before set_model = {SYNC_FIRST[0]}", file=sys.stderr, flush=True)
        t0 = time.perf_counter(); camp.recommend(batch); torch.cuda.synchronize()
        print(f"recommend {1e3 * (time.perf_counter() - t0):.2f} ms; set_model calls: {TIMES[-1:]}", flush=True)


if __name__ == "__main__":
    main()
