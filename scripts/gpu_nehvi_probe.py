"""Where does a qLogNEHVI score deviate from the oracle?  BASELINE configs[4] (3 targets, 1e5 x 15, n = 256), S = 512 and 128:
per-sample candidate values (conditional mean from bbh_posterior_columns + sd z) against the oracle's joint draw, and the
scores of the linear-domain and the log-domain cell kernels against the oracle, on the head of the ranking + random rows."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from _problems import make_grid, oracle_spec
from baybe_amd import engine, gp_spec
from baybe_amd.nehvi import HipNEHVI, compute_ref_point
from oracle import gp_oracle as go, nehvi_oracle as no

N, d, n, m = 100_000, 15, 256, 3
rng = np.random.default_rng(0)
X = make_grid(N, d, 0)
Xt = X[np.random.default_rng(1).choice(N, n, replace=False)]
Y = np.stack([-((Xt - 0.25) ** 2).sum(1), -((Xt - 0.75) ** 2).sum(1), -np.abs(Xt - 0.5).sum(1)], 1) + 0.05 * rng.standard_normal((n, 3))
engines, models = [], []
for o in range(m):
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    g = engine.HipGP(0); g.set_model(spec, Xt, Y[:, o]); fi = g.fit(); engines.append(g)
    models.append(go.fit_gp(oracle_spec(spec), Xt, Y[:, o], params=go.GPParams(fi.params.lengthscale, fi.params.noise, fi.params.mean)))
signs = np.ones(m); ref = compute_ref_point(Y)
Xd = torch.from_numpy(X).cuda()
for S in (512, 128):
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=True)
    hv.prepare(1234, prune_seed=99)
    os.environ.pop("BBH_NEHVI_LOG", None)
    s_lin = hv.score(Xd).cpu().numpy()
    os.environ["BBH_NEHVI_LOG"] = "1"
    s_log = hv.score(Xd).cpu().numpy()
    os.environ.pop("BBH_NEHVI_LOG", None)
    keep = no.prune_baseline(models, signs, Xt, ref, 99)
    orc = no.NEHVIOracle(models, signs, Xt[keep], ref, no.sobol_normal_base_samples_nd(S, len(keep) + 1, m, 1234))
    top = np.argsort(-s_lin, kind="stable")[:12]
    pick = np.concatenate([np.random.default_rng(S).choice(N, 500, replace=False), top])
    dup = np.array([(np.abs(Xt[keep] - x).sum(1) < 1e-12).any() for x in X[pick]])
    so = orc.values(X[pick])
    print(f"S={S}: max |lin - oracle| {np.abs(s_lin[pick] - so)[~dup].max():.3e}  max |log - oracle| {np.abs(s_log[pick] - so)[~dup].max():.3e}"
          f"  max |lin - log| over all rows {np.nanmax(np.abs(s_lin - s_log)[np.isfinite(s_log) & (s_log > -30)]):.3e}")
    worst = pick[~dup][np.argsort(-np.abs(s_lin[pick] - so)[~dup])[:3]]
    print("   worst rows: lin", s_lin[worst], "log", s_log[worst], "oracle", orc.values(X[worst]), "deep-tail max |lin - log|",
          np.nanmax(np.abs(s_lin - s_log)[np.isfinite(s_log) & (s_log <= -30)]))
    Xw = torch.from_numpy(X[worst]).cuda()
    for o in range(m):
        tm = hv.outputs[o].ext.posterior_columns(Xw).cpu().numpy()
        _, var = hv.outputs[o].ext.posterior(Xw)
        fd = tm + np.sqrt(np.maximum(var.cpu().numpy(), 0))[:, None] * hv.zx[None, :, o]
        fo = np.stack([orc.candidate_samples(x)[:, o] for x in X[worst]])
        print(f"   target {o}: worst rows {worst.tolist()} max |f_dev - f_orc| per row {np.abs(fd - fo).max(1)}  (scores {s_lin[worst]}, oracle {orc.values(X[worst])})")
