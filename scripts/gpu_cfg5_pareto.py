"""BASELINE configs[4]: ParetoObjective qLogNEHVI, 3 targets, 1e5 candidates, n_train=256, S=512 (and 128)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _problems import make_grid
from baybe_amd import engine, gp_spec
from baybe_amd.nehvi import HipNEHVI, compute_ref_point

N, d, n, m = 100_000, 15, 256, 3
rng = np.random.default_rng(0)
X = make_grid(N, d, 0)
Xt = X[np.random.default_rng(1).choice(N, n, replace=False)]
f1 = -((Xt - 0.25) ** 2).sum(1) + 0.05 * rng.standard_normal(n)
f2 = -((Xt - 0.75) ** 2).sum(1) + 0.05 * rng.standard_normal(n)
f3 = -np.abs(Xt - 0.5).sum(1) + 0.05 * rng.standard_normal(n)
Y = np.stack([f1, f2, f3], 1)
engines = []
t0 = time.time()
for o in range(m):
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, Y[:, o])
    fi = g.fit()
    engines.append(g)
print(f"[cfg5] 3 fits: {time.time() - t0:.2f} s (last nfev {fi.nfev})")
ref = compute_ref_point(Y)
Xd = torch.from_numpy(X).cuda()
for S in (128, 512):
    hv = HipNEHVI(engines, np.ones(m), Xt, ref, n_mc_samples=S, prune_baseline=True)
    t0 = time.time(); hv.prepare(1234, prune_seed=99); t1 = time.time()
    sc = hv.score(Xd); torch.cuda.synchronize(); t2 = time.time()
    sc = hv.score(Xd); torch.cuda.synchronize(); t3 = time.time()
    ncell = hv.n_cells / S
    print(f"[cfg5] S={S}: pruned baseline {len(hv._pruned)}/{n}, avg cells/sample {ncell:.1f}; prepare {t1 - t0:.2f} s; score {1e3 * (t3 - t2):.1f} ms ({N / (t3 - t2):.3e} cand/s); best {float(sc.max()):.4f} at {int(sc.argmax())}")
t0 = time.time(); r = hv.greedy(Xd, 2, seed=1234, prune_seed=99); print(f"[cfg5] greedy q=2 (S=512): {time.time() - t0:.2f} s -> {r.indices}")
