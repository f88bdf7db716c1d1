"""A/B of the conditional-mean-columns pass (bbh_posterior_columns, qLogNEHVI support): plain kernel (128 columns per launch, every
launch recomputes the kernel values) against the cooperative kernel (512 columns per launch, kernel values exchanged through LDS).
BASELINE configs[4] shape: 1e5 x 15 candidates, n_ext = 287 training + baseline rows, S columns; also an ICM model.  Bitwise equality of
the two forms is asserted (same accumulation order)."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from _problems import make_grid, make_tl_problem
from baybe_amd import engine, gp_spec


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(name, g, X, n, S_list):
    Xd = torch.from_numpy(X).cuda()
    rng = np.random.default_rng(3)
    for S in S_list:
        g.set_mean_columns(rng.standard_normal((n, S)))
        out = {}
        for mode in ("0", "1", "2"):
            os.environ["BBH_COLUMNS_COOP"] = "0" if mode == "0" else "1"
            os.environ["BBH_COLUMNS_NT"] = "2" if mode == "2" else "1"
            for sm in (False, True):
                ms = timeit(lambda: g.posterior_columns(Xd, sample_major=sm))
                out[(mode, sm)] = (ms, g.posterior_columns(Xd, sample_major=sm).cpu().numpy())
        os.environ.pop("BBH_COLUMNS_COOP"); os.environ.pop("BBH_COLUMNS_NT")
        for sm in (False, True):
            assert np.array_equal(out[("0", sm)][1], out[("1", sm)][1]) and np.array_equal(out[("0", sm)][1], out[("2", sm)][1]), (name, S, sm)
        flops = 2.0 * X.shape[0] * S * (16 * g_nb(g))
        print(f"{name} N={X.shape[0]} n={n} S={S}: plain {out[('0', True)][0]:.3f} ms ({flops / out[('0', True)][0] / 1e9:.1f} TF/s)  "
              f"cooperative {out[('1', True)][0]:.3f} ms ({flops / out[('1', True)][0] / 1e9:.1f} TF/s)  two tiles "
              f"{out[('2', True)][0]:.3f} ms ({flops / out[('2', True)][0] / 1e9:.1f} TF/s)   [candidate-major: "
              f"{out[('0', False)][0]:.3f} / {out[('1', False)][0]:.3f} / {out[('2', False)][0]:.3f} ms]  outputs bitwise equal", flush=True)


def g_nb(g):
    return (g.n + 15) // 16


N, d = 100_000, 15
X = make_grid(N, d, 0)
for n in (287, 128, 512):
    Xt = X[np.random.default_rng(1).choice(N, n, replace=False)]
    y = -((Xt - 0.3) ** 2).sum(1)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    g = engine.HipGP(0); g.set_model(spec, Xt, y); g.fit(maxiter=20)
    case("plain", g, X, n, (512, 384, 128, 1000) if n == 287 else (512,))
    g.close()
Xtl, Xt, y = make_tl_problem(N, 8, 64, T=4, seed=5)
spec = gp_spec.GPSpec.baybe_default(9, np.zeros(9), np.ones(9), task_idx=8, n_tasks=4)
g = engine.HipGP(0); g.set_model(spec, Xt, y); g.fit(maxiter=10)
case("icm", g, Xtl, len(y), (512,))
