"""BASELINE configs[3]: transfer-learning multi-task GP (ICM), 4 tasks, 1e5 candidates, n_train=1024.
Parity against the oracle on a sample + timing (fit on the device with the LOO criterion)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from _problems import make_tl_problem
from baybe_amd import engine, gp_spec
from oracle import gp_oracle as go

N, dnum, T, npt = 100_000, 15, 4, 256
X, Xt, y = make_tl_problem(N, dnum, npt, T=T, seed=0)
d = dnum + 1
spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=dnum, n_tasks=T)
gp = engine.HipGP(0)
gp.set_model(spec, Xt, y)
p0 = gp_spec.initial_params(spec)
t0 = time.time(); val, g = gp.data_term(p0); torch.cuda.synchronize(); t1 = time.time()
for _ in range(3): val, g = gp.data_term(p0)
torch.cuda.synchronize(); t2 = time.time()
print(f"[cfg4] n={len(y)} LOO data term {val:.6f}; one evaluation {(t2 - t1) / 3 * 1e3:.2f} ms (first {1e3*(t1-t0):.1f} ms)")
ospec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=dnum, n_tasks=T)
op = go.initial_params(ospec)
t0 = time.time(); dt = go.data_term(ospec, op, go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]); t1 = time.time()
gref = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls, dt.g_task_B.reshape(-1)])
print(f"       oracle value {dt.value:.6f} ({t1 - t0:.2f} s); value rel {abs(val - dt.value) / abs(dt.value):.2e}; grad maxabs {np.abs(g - gref).max():.2e} / {np.abs(gref).max():.2e}")
t0 = time.time(); fi = gp.fit(maxiter=60); t1 = time.time()
print(f"       device fit (<=60 it): {t1 - t0:.2f} s nit {fi.nit} nfev {fi.nfev} fun {fi.fun:.6f} status {fi.status}")
Xd = torch.from_numpy(X).cuda()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); m, v = gp.posterior(Xd); torch.cuda.synchronize(); t1 = time.time()
om = go.GPModel(ospec, go.GPParams(fi.params.lengthscale, fi.params.noise, fi.params.mean, 1.0, fi.params.task_W, fi.params.task_v), Xt, y)
pick = np.random.default_rng(0).choice(N, 2000, replace=False)
mo, vo = om.posterior(X[pick])
mm, vv = m.cpu().numpy()[pick], v.cpu().numpy()[pick]
print(f"       posterior 1e5 x {d}: {(t1 - t0) * 1e3:.2f} ms ({N / (t1 - t0):.3e} cand/s); mean rel {np.max(np.abs(mm - mo) / np.abs(mo)):.2e} var rel {np.max(np.abs(vv - vo) / vo):.2e}")
r = gp.greedy_qlogei(Xd, 3, seed=5)
ro = go.optimize_acqf_discrete_qlogei(om, X[:20000], 3, seed=5)
r2 = gp.greedy_qlogei(Xd[:20000], 3, seed=5)
print(f"       greedy(1e5) {r.indices}; greedy(20k) hip {r2.indices} oracle {ro.indices}")
