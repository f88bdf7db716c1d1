"""Stage-by-stage GPU bring-up: prints the error of every device stage against the oracle.
Run on the GPU box:  python tests/gpu_scripts/gpu_first_light.py  (writes gpurun_out/first_light.log too)."""

import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np
import torch

from _problems import fixed_theta, make_problem, make_tl_problem
from baybe_amd import engine, gp_spec
from oracle import gp_oracle as go


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300))) if a.size else 0.0


def absd(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.asarray(a).size else 0.0


def o_spec_from(spec):
    from _problems import oracle_spec

    return oracle_spec(spec)


def o_params(p):
    return go.GPParams(p.lengthscale.copy(), p.noise, p.mean, p.outputscale,
                       None if p.task_W is None else p.task_W.copy(), None if p.task_v is None else p.task_v.copy())


def main():
    print("device:", torch.cuda.get_device_name(0))
    gp = engine.HipGP(0)
    gp.selftest()
    print("[ok] mfma f64 layout selftest")

    # ---------- data term (value + grad) ----------
    for (N, d, n, kern, crit, tl) in [
        (500, 5, 40, "matern52", "mll", False),
        (500, 7, 100, "rbf", "mll", False),
        (500, 4, 90, "matern32", "loo", True),
        (500, 6, 130, "matern12", "mll", False),
    ]:
        if tl:
            X, Xt, y = make_tl_problem(N, d, n // 3, T=3, seed=3)
            spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=3, kernel=kern)
            spec.use_outputscale = True
        else:
            X, Xt, y = make_problem(N, d, n, seed=1)
            spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), kernel=kern)
        spec.criterion = crit
        p = gp_spec.initial_params(spec)
        rng = np.random.default_rng(5)
        p.lengthscale = p.lengthscale * (0.7 + 0.6 * rng.random(spec.dn))
        p.mean = 0.1
        if tl:
            p.task_W = 0.3 + rng.random((3, 3))
            p.outputscale = 1.3
        gp.set_model(spec, Xt, y)
        val, g = gp.data_term(p)
        ospec = o_spec_from(spec)
        Xn = go.normalize_inputs(ospec, Xt)
        ystd, _, _ = go.standardize_targets(y)
        dt = go.data_term(ospec, o_params(p), Xn, ystd)
        gref = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls] + ([dt.g_task_B.reshape(-1)] if tl else []))
        print(f"[data_term {kern}/{crit}/tl={tl} n={Xt.shape[0]}] value rel {rel(val, dt.value):.2e}  grad maxabs {absd(g, gref):.2e} (|g|max {np.abs(gref).max():.2e})")

    # ---------- posterior: fused vs unfused vs oracle ----------
    for (N, d, n) in [(3000, 5, 40), (3000, 15, 200), (3000, 20, 300), (5000, 20, 512), (2000, 3, 20), (1000, 9, 700)]:
        X, Xt, y = make_problem(N, d, n, seed=2)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        ls, nz, c = fixed_theta(d)
        p = gp_spec.GPParams(np.full(d, ls) * (0.8 + 0.4 * np.random.default_rng(7).random(d)), nz, 0.05)
        gp.set_model(spec, Xt, y)
        gp.factorize(p)
        om = go.GPModel(o_spec_from(spec), o_params(p), Xt, y)
        mo, vo = om.posterior(X)
        m1, v1 = gp.posterior(X)
        m2, v2 = gp.posterior(X, unfused=True)
        m1, v1, m2, v2 = (t.cpu().numpy() for t in (m1, v1, m2, v2))
        print(f"[posterior n={n} d={d}] fused: mean rel {rel(m1, mo):.2e} var rel {rel(v1, vo):.2e} | unfused: mean rel {rel(m2, mo):.2e} var rel {rel(v2, vo):.2e} | min var {vo.min():.3e}")
        tm = gp.train_posterior_mean()
        print(f"    train mean rel {rel(tm, om.posterior(Xt)[0]):.2e}  best_f {gp.best_f():.6f} vs {go.best_f_from_model(om):.6f}")
        # qLogEI q=1
        z = go.sobol_normal_base_samples(512, 1, 1234)[:, 0]
        bf = go.best_f_from_model(om)
        so = go.qlogei_q1(mo, vo, z, bf)
        sg = gp.qlogei(torch.from_numpy(mo).cuda(), torch.from_numpy(vo).cuda(), z, bf).cpu().numpy()
        sg2 = gp.qlogei(torch.from_numpy(m1).cuda(), torch.from_numpy(v1).cuda(), z, bf)
        v, i = gp.argmax(sg2)
        print(f"    qlogei(q1) abs {absd(sg, so):.2e} (range {so.min():.2f}..{so.max():.2f})  argmax {i} vs {int(np.argmax(so))}  topk {gp.topk(sg2, 5)[1].tolist()} vs {go.topk_first_index(so, 5).tolist()}")

    # ---------- greedy with pending ----------
    for (N, d, n, q, minimize) in [(2000, 5, 40, 4, False), (3000, 8, 100, 3, True)]:
        X, Xt, y = make_problem(N, d, n, seed=4, minimize=minimize)
        sign = -1.0 if minimize else 1.0
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        ls, nz, c = fixed_theta(d)
        p = gp_spec.GPParams(np.full(d, ls), nz, 0.0)
        gp.set_model(spec, Xt, y)
        gp.factorize(p)
        om = go.GPModel(o_spec_from(spec), o_params(p), Xt, y)
        ro = go.optimize_acqf_discrete_qlogei(om, X, q, seed=77, sign=sign, X_pending=X[:1])
        rg = gp.greedy_qlogei(X, q, seed=77, sign=sign, X_pending=X[:1])
        print(f"[greedy N={N} q={q} min={minimize}] idx {rg.indices} vs {ro.indices}; val abs {absd(rg.values, ro.values):.2e}")
        # pending scores in detail
        pend = X[[3, 10]]
        z = go.sobol_normal_base_samples(512, 3, 5)
        bf = go.best_f_from_model(om, sign)
        so = go.qlogei_with_pending(om, X[:500], pend, z, bf, sign)
        mp, cpp = gp.set_pending(pend)
        mo_p, co_p = om.posterior_joint(pend)
        m1, v1 = gp.posterior(X[:500])
        cr = gp.cross_cov(X[:500])
        sg = gp.qlogei_pending(m1, v1, cr, z, bf, sign).cpu().numpy()
        print(f"    pending mean rel {rel(mp, mo_p):.2e} cov abs {absd(cpp, co_p):.2e}; scores abs {absd(sg, so):.2e}")

    # ---------- device fit vs oracle fit ----------
    for (N, d, n) in [(1000, 5, 60), (1000, 10, 128)]:
        X, Xt, y = make_problem(N, d, n, seed=6)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        gp.set_model(spec, Xt, y)
        t0 = time.time()
        fi = gp.fit()
        t1 = time.time()
        ospec = o_spec_from(spec)
        Xn = go.normalize_inputs(ospec, Xt)
        ystd, _, _ = go.standardize_targets(y)
        fo = go.fit_hyperparameters(ospec, Xn, ystd)
        print(f"[fit n={n} d={d}] hip {t1 - t0:.2f}s nit {fi.nit} nfev {fi.nfev} fun {fi.fun:.8f} | oracle {time.time() - t1:.2f}s nit {fo.nit} fun {fo.fun:.8f} | ls rel {rel(fi.params.lengthscale, fo.params.lengthscale):.2e} noise rel {rel(fi.params.noise, fo.params.noise):.2e}")

    # ---------- timing ----------
    for (N, d, n) in [(100000, 15, 256), (200000, 20, 512)]:
        X, Xt, y = make_problem(N, d, n, seed=0)
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        ls, nz, c = fixed_theta(d)
        gp.set_model(spec, Xt, y)
        t0 = time.time(); gp.factorize(gp_spec.GPParams(np.full(d, ls), nz, 0.0)); torch.cuda.synchronize(); tf = time.time() - t0
        Xd = torch.from_numpy(X).cuda()
        z = go.sobol_normal_base_samples(512, 1, 1234)[:, 0]
        bf = gp.best_f()
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            m, v = gp.posterior(Xd); torch.cuda.synchronize(); t1 = time.time()
            s = gp.qlogei(m, v, z, bf); torch.cuda.synchronize(); t2 = time.time()
            val, i = gp.argmax(s); t3 = time.time()
        W = n * n + 2 * n * d + 16 * n + 16 * 512
        print(f"[time N={N} d={d} n={n}] factorize {tf*1e3:.1f} ms | posterior {(t1-t0)*1e3:.2f} ms ({N/(t1-t0):.3e} cand/s, {N*W/(t1-t0)/1e12:.2f} TF alg) | qlogei {(t2-t1)*1e3:.2f} ms | argmax {(t3-t2)*1e3:.2f} ms | total {N/(t3-t0):.3e} cand/s")
        t0 = time.time(); m2, v2 = gp.posterior(Xd[:50000], unfused=True); torch.cuda.synchronize()
        print(f"    unfused 50k: {(time.time()-t0)*1e3:.1f} ms; fused-vs-unfused var rel {rel(v.cpu().numpy()[:50000], v2.cpu().numpy()):.2e}")


if __name__ == "__main__":
    main()
