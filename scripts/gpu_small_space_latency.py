"""Latency of recommend() on small spaces - the regime of backtesting loops (BASELINE configs[0] is its smallest
member: 1000 candidates, n_train = 20, batch 3): per-call wall time after new measurements (refit + greedy batch), split into fit and
selection, for a few (candidates, measurements) shapes.  Writes gpurun_out/small_space_latency.json (copied to profiles/)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
from baybe_amd import engine as _engine
from baybe_amd.recommenders import HipBotorchRecommender

FIT_MS, FIT_NFEV = [], []
_fit = _engine.HipGP.fit


def _timed_fit(self, *a, **k):
    t0 = time.perf_counter()
    info = _fit(self, *a, **k)
    FIT_MS.append((time.perf_counter() - t0) * 1e3); FIT_NFEV.append(int(info.nfev))
    return info


_engine.HipGP.fit = _timed_fit


def f(X):
    return -((X - 0.5) ** 2).sum(1) + 0.1 * np.sin(6.28 * X[:, 0])


def run(levels, d, n0, batch, iters, seed=0):
    rng = np.random.default_rng(seed)
    vals = np.arange(levels) / (levels - 1.0)
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(d)])
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), n0, replace=False)].copy()
    meas["yield"] = f(meas.to_numpy(dtype=float)) + 0.05 * rng.standard_normal(n0)
    rec = HipBotorchRecommender()
    camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), rec)
    camp.add_measurements(meas)
    torch.manual_seed(seed)
    t0 = time.perf_counter(); camp.recommend(batch); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
    total, fit_ms, nfev = [], [], []
    for _ in range(iters):
        got = camp.recommend(batch)
        new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)) + 0.05 * rng.standard_normal(len(new))
        camp.add_measurements(new)
        FIT_MS.clear(); FIT_NFEV.clear()
        t0 = time.perf_counter(); camp.recommend(batch); torch.cuda.synchronize(); total.append((time.perf_counter() - t0) * 1e3)
        fit_ms.append(sum(FIT_MS)); nfev.append(sum(FIT_NFEV))
    out = {"candidates": len(exp), "d": d, "n_train_start": n0, "n_train_end": n0 + 2 * batch * iters, "batch": batch,
           "first_call_ms": round(first, 2), "recommend_ms_median": round(float(np.median(total)), 2),
           "recommend_ms_min": round(float(np.min(total)), 2), "recommend_ms_max": round(float(np.max(total)), 2),
           "fit_ms_median": round(float(np.median(fit_ms)), 2) if fit_ms else None,
           "fit_objective_evaluations_median": int(np.median(nfev)) if nfev else None}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    rows = [run(10, 3, 20, 3, 8), run(10, 4, 50, 3, 6), run(10, 4, 100, 5, 5), run(18, 4, 100, 5, 4)]
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "small_space_latency.json").write_text(json.dumps(rows, indent=1))
    import cProfile, pstats

    def profile_one(levels, d, n0, batch, warm):
        rng = np.random.default_rng(1)
        vals = np.arange(levels) / (levels - 1.0)
        space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(d)])
        exp = space.discrete.exp_rep
        meas = exp.iloc[rng.choice(len(exp), n0, replace=False)].copy(); meas["yield"] = f(meas.to_numpy(dtype=float))
        rec = HipBotorchRecommender(); camp = Campaign(space, SingleTargetObjective(NumericalTarget("yield")), rec)
        camp.add_measurements(meas); camp.recommend(batch)
        for _ in range(warm):
            got = camp.recommend(batch); new = got.copy(); new["yield"] = f(new.to_numpy(dtype=float)); camp.add_measurements(new)
        pr = cProfile.Profile(); pr.enable(); camp.recommend(batch); torch.cuda.synchronize(); pr.disable()
        print(f"---- profile of one recommend({batch}) after new measurements: {len(exp)} candidates, n = {len(camp.measurements)}")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
        # the same in microseconds (print_stats rounds to milliseconds): own time and cumulative time of the most expensive functions
        st = pstats.Stats(pr).stats
        rows_ = sorted(((tt, ct, nc, f"{Path(k[0]).name}:{k[1]}({k[2]})") for k, (cc, nc, tt, ct, _) in st.items()), reverse=True)[:40]
        print("own_us   cum_us  calls  function")
        for tt, ct, nc, name in rows_:
            print(f"{tt * 1e6:7.0f} {ct * 1e6:8.0f} {nc:6d}  {name}")

    profile_one(10, 3, 20, 3, 1)
    profile_one(18, 4, 100, 5, 2)
