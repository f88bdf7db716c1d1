"""BASELINE.json configs[1], [3] and [4] asserted at their full sizes (cfg4 / cfg5 only existed as
scripts that printed).  The oracle is the checker: live on samples / slices it finishes in seconds, and through
tests/golden/cfg2_full_ranking.npz, for which it ranked ALL 1e5 rows offline (make_golden_full_ranking.py).

Tolerances: posterior 1e-9 / 1e-8 relative, qLogEI scores 1e-8 absolute, qLogNEHVI scores 1e-6 absolute with the 99 % quantile at
1e-8 (round 2 held qLogNEHVI to 2e-5 without a reason: see test_cfg5_qlognehvi_scores_at_full_size for where 1e-6 comes from;
observed deviations are recorded in profiles/r03_observed_deviations.json), indices identical."""

import math
import os
from pathlib import Path

import numpy as np
import pytest

from _problems import fixed_theta, make_grid, make_problem, make_tl_problem, oracle_spec

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _np(t):
    return t.cpu().numpy()


# ---- configs[1]: 1e5 x 15, n = 256, Matérn-5/2, qLogEI — the oracle ranked the FULL set -----------------------
def test_cfg2_full_set_ranking_matches_the_oracle():
    import torch

    from baybe_amd import engine, gp_spec

    g = np.load(GOLD / "cfg2_full_ranking.npz")
    N, d, n, q, seed = (int(g[k]) for k in ("N", "d", "n", "q", "seed"))
    X, Xt, y = make_problem(N, d, n, seed=0)
    ls, nz, c = fixed_theta(d)
    gp = engine.HipGP(0)
    gp.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    gp.factorize(gp_spec.GPParams(np.full(d, ls), nz, c))
    Xd = torch.from_numpy(X).cuda()
    assert math.isclose(gp.best_f(), float(g["best_f"]), rel_tol=1e-10)
    # q = 1 scores of every row: ranking head, a strided sample and the checksums of all 1e5 oracle scores
    m, v = gp.posterior(Xd)
    s = _np(gp.qlogei(m, v, engine.sobol_normal_base_samples(512, 1, seed)[:, 0], gp.best_f()))
    assert float(g["min_gap_top16"]) > 1e-6  # the head of the ranking is not a numerical coin toss
    vals, idx = gp.topk(torch.from_numpy(s).cuda(), 16)
    assert np.array_equal(idx, g["top_idx"][:16]) and np.allclose(vals, g["top_val"][:16], rtol=0, atol=1e-8)
    assert np.allclose(s[::64], g["sample_scores"], rtol=0, atol=1e-8)
    assert math.isclose(float(s.sum()), float(g["score_sum"]), rel_tol=1e-10)
    assert math.isclose(float(np.abs(s).sum()), float(g["score_abs_sum"]), rel_tol=1e-10)
    # optimize_acqf_discrete(q = 5): every greedy step of the oracle ran over all remaining rows
    res = gp.greedy_qlogei(Xd, q, seed=seed)
    assert res.indices == g["greedy_idx"].tolist()
    assert np.allclose(res.values, g["greedy_val"], rtol=0, atol=1e-8)
    gp.close()


# ---- configs[2], one GPU's shard: 1e6 x 20, n = 512 - the oracle ranked the FULL set (20 minutes of CPU, offline) ----
def test_cfg3_full_set_ranking_matches_the_oracle():
    """tests/golden/cfg3_full_ranking.npz (make_golden_cfg3_full_ranking.py): the oracle scored all 1e6 rows of bench.py's
    workload in every step of a greedy batch of 5.  Asserted: best_f, top-16 of the q = 1 ranking (indices and values),
    every 256th score, checksums of all scores, the greedy batch (= what ``recommend(5)`` returns; ``bench.py`` prints the same
    indices as ``extra.greedy_q5_indices``), and posterior mean / variance on every 997th row."""
    import sys

    import torch

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    from baybe_amd import engine, gp_spec

    g = np.load(GOLD / "cfg3_full_ranking.npz")
    N, d, n, q, seed = (int(g[k]) for k in ("N", "d", "n", "q", "seed"))
    X, Xt, y = bench.synth_problem(N, d, n, 0)
    ls, nz, c = fixed_theta(d)
    gp = engine.HipGP(0)
    gp.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    gp.factorize(gp_spec.GPParams(np.full(d, ls), nz, c))
    Xd = torch.from_numpy(X).cuda()
    assert math.isclose(gp.best_f(), float(g["best_f"]), rel_tol=1e-10)
    m, v = gp.posterior(Xd)
    rows = g["post_rows"]
    assert np.allclose(_np(m)[rows], g["post_mean"], rtol=1e-9, atol=1e-12) and np.allclose(_np(v)[rows], g["post_var"], rtol=1e-8)
    s = _np(gp.qlogei(m, v, engine.sobol_normal_base_samples(512, 1, seed)[:, 0], gp.best_f()))
    assert float(g["min_gap_top16"]) > 1e-6  # the head of the ranking is not a numerical coin toss
    vals, idx = gp.topk(torch.from_numpy(s).cuda(), 16)
    assert np.array_equal(idx, g["top_idx"][:16]) and np.allclose(vals, g["top_val"][:16], rtol=0, atol=1e-8)
    from conftest import record_deviation

    record_deviation("cfg3_qlogei_scores_every_256th_of_1e6", np.abs(s[::256] - g["sample_scores"]).max(), 1e-8)
    record_deviation("cfg3_posterior_mean_rel", (np.abs(_np(m)[rows] - g["post_mean"]) / np.abs(g["post_mean"])).max(), 1e-9)
    record_deviation("cfg3_posterior_var_rel", (np.abs(_np(v)[rows] - g["post_var"]) / g["post_var"]).max(), 1e-8)
    assert np.allclose(s[::256], g["sample_scores"], rtol=0, atol=1e-8)
    assert math.isclose(float(s.sum()), float(g["score_sum"]), rel_tol=1e-10)
    assert math.isclose(float(np.abs(s).sum()), float(g["score_abs_sum"]), rel_tol=1e-10)
    res = gp.greedy_qlogei(Xd, q, seed=seed)
    assert res.indices == g["greedy_idx"].tolist()
    record_deviation("cfg3_greedy_q5_values", np.abs(np.array(res.values) - g["greedy_val"]).max(), 1e-8)
    assert np.allclose(res.values, g["greedy_val"], rtol=0, atol=1e-8)
    gp.close()


def test_bench_greedy_indices_equal_the_full_set_golden():
    """``bench.py`` (the driver's command, shortened) prints ``extra.greedy_q5_indices``: the oracle's picks over all 1e6 rows."""
    import json
    import subprocess
    import sys

    root = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-budget", "0"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    g = np.load(GOLD / "cfg3_full_ranking.npz")
    assert line["extra"]["greedy_q5_indices"] == g["greedy_idx"].tolist()
    assert line["config"]["workload"].startswith("1000000 x 20") and line["roofline"]["frac"] > 0.5


# ---- fit parity at size (it stopped at n = 128): the device's whole L-BFGS-B run against the oracle's ----------
def test_device_fit_reaches_the_oracle_optimum_at_n512():
    """configs[2]'s model (n = 512, d = 20, BAYBE preset, MLL): ``HipGP.fit`` - scipy L-BFGS-B over device evaluations, the
    factorisation as one tile-dataflow launch - against ``go.fit_hyperparameters`` (numpy / LAPACK).  Both runs stop on scipy's
    relative-reduction test (``ftol`` = 2.2e-9), so two runs whose evaluations differ in the last bits end ~1e-8 apart in the
    objective (round 5: the look-ahead factorisation changed the rounding of the factor and the device run went from 99 to 94
    evaluations, ending 2.3e-8 above the oracle's run instead of 6e-9).  What is asked: (i) the ORACLE's objective at the device's end
    point equals the device's value to 1e-9 - the same function; (ii) the end values agree to 1e-7; (iii) the hyper-parameters to
    1e-2 relative."""
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    d, n = 20, 512
    _, Xt, y = bench.synth_problem(4096, d, n, 0)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    gp = engine.HipGP(0)
    gp.set_model(spec, Xt, y)
    fi = gp.fit()
    ospec = oracle_spec(spec)
    ystd, _, _ = go.standardize_targets(y)
    fo = go.fit_hyperparameters(ospec, go.normalize_inputs(ospec, Xt), ystd)
    print(f"n=512 fit: device fun {fi.fun:.12f} nfev {fi.nfev}; oracle fun {fo.fun:.12f} nfev {fo.nfev}")
    from _problems import oracle_params

    f_at, _ = go.fit_objective(ospec, go.pack_raw(ospec, oracle_params(spec, fi.params)), go.normalize_inputs(ospec, Xt), ystd)
    assert math.isclose(f_at, fi.fun, rel_tol=1e-9, abs_tol=1e-11), (f_at, fi.fun)
    assert abs(fi.fun - fo.fun) <= 1e-7 * max(1.0, abs(fo.fun))
    assert np.allclose(fi.params.lengthscale, fo.params.lengthscale, rtol=1e-2)
    assert math.isclose(fi.params.noise, fo.params.noise, rel_tol=1e-2) and abs(fi.params.mean - fo.params.mean) <= 1e-3
    gp.close()


LS_RTOL_FLAT = 0.10  # see the docstring below: end points of complete runs on the flat LOO valley


def test_device_fit_reaches_the_oracle_optimum_at_n1024_icm(cfg4):
    """configs[3]'s model (ICM over 4 tasks, n = 1024, LOO criterion), no iteration cap: the device's complete L-BFGS-B run
    against the oracle's complete run, stored in tests/golden/cfg4_oracle_fit.npz (make_golden_cfg4_fit.py; about 1000
    evaluations of 0.3 - 0.5 s - ten minutes, so it is not repeated here).

    What can be asked of two complete runs: the LOO surface is flat along the task-covariance directions, both runs stop on
    scipy's relative-reduction test (``ftol``) with a gradient of ~5e-3 still standing, and the end point is chaotic at that
    level - the ORACLE ITSELF ends at -0.62268223 on the build container (8 BLAS threads) and at -0.62270299 on the GPU box
    (128 threads): 2.1e-5 apart in the objective, 1 % in the lengthscales.  So: (i) the oracle's objective evaluated AT the
    device's end point equals the device's value to 1e-9 (same function); (ii) the two end values agree to 5e-5 and the
    device's is not worse than the oracle's by more than that; (iii) the oracle's gradient at the device's end point is as
    small as at its own (<= 2e-2; observed 3e-4: the device run stops closer to stationarity than the golden run); (iv) the
    end points lie in the same stretch of the valley: lengthscales and noise within 10 %, task covariance within 20 % (two
    device builds that differ in the rounding of the Gram entries - fma contraction - end 3 - 6 % apart; the observed values
    are recorded in profiles/r03_observed_deviations.json)."""
    from baybe_amd import gp_spec  # noqa: F401
    from oracle import gp_oracle as go

    X, Xt, y, spec, gp = cfg4
    gold = np.load(GOLD / "cfg4_oracle_fit.npz")
    fi = gp.fit()
    ospec = oracle_spec(spec)
    ystd, _, _ = go.standardize_targets(y)
    Xn = go.normalize_inputs(ospec, Xt)
    at_dev, grad_dev = go.fit_objective(ospec, go.pack_raw(ospec, _oparams_icm(fi.params)), Xn, ystd)
    print(f"n=1024 ICM fit: device fun {fi.fun:.12f} nfev {fi.nfev}; oracle (golden) fun {float(gold['fun']):.12f} nfev {int(gold['nfev'])}; "
          f"oracle objective at the device optimum {at_dev:.12f}, |grad|_max {np.abs(grad_dev).max():.2e}")
    dev_ls = float(np.abs(fi.params.lengthscale / gold["lengthscale"] - 1.0).max())
    dev_B = float(np.abs(fi.params.task_B() / gold["task_B"] - 1.0).max())
    dev_nz = abs(fi.params.noise / float(gold["noise"]) - 1.0)
    print(f"   relative deviations from the golden end point: lengthscales {dev_ls:.3e}, task covariance {dev_B:.3e}, noise {dev_nz:.3e}")
    from conftest import record_deviation

    record_deviation("cfg4_fit_n1024_icm_objective_vs_golden_run", abs(fi.fun - float(gold["fun"])), 5e-5)
    record_deviation("cfg4_fit_n1024_icm_oracle_gradient_at_device_end", float(np.abs(grad_dev).max()), 2e-2)
    record_deviation("cfg4_fit_n1024_icm_lengthscales_rel", dev_ls, LS_RTOL_FLAT)
    record_deviation("cfg4_fit_n1024_icm_task_covariance_rel", dev_B, 2 * LS_RTOL_FLAT)
    assert abs(at_dev - fi.fun) <= 1e-9 * max(1.0, abs(fi.fun))
    assert abs(fi.fun - float(gold["fun"])) <= 5e-5 and fi.fun <= float(gold["fun"]) + 5e-5
    assert dev_ls <= LS_RTOL_FLAT and dev_B <= 2 * LS_RTOL_FLAT and dev_nz <= LS_RTOL_FLAT
    assert np.abs(grad_dev).max() <= 2e-2


def _oparams_icm(p):
    from oracle import gp_oracle as go

    return go.GPParams(np.array(p.lengthscale, dtype=float), p.noise, p.mean, p.outputscale,
                       None if p.task_W is None else p.task_W.copy(), None if p.task_v is None else p.task_v.copy(),
                       bool(getattr(p, "task_unit_scale", False)))


# ---- configs[3]: transfer learning, ICM over 4 tasks, 1e5 x (15 + task), n = 1024, LOO criterion -----------------
@pytest.fixture(scope="module")
def cfg4():
    from baybe_amd import engine, gp_spec

    N, dnum, T, per_task = 100_000, 15, 4, 256
    X, Xt, y = make_tl_problem(N, dnum, per_task, T=T, seed=0)
    d = dnum + 1
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=dnum, n_tasks=T)
    assert spec.criterion == "loo" and len(y) == 1024  # presets/baybe.py:277-281
    gp = engine.HipGP(0)
    gp.set_model(spec, Xt, y)
    yield X, Xt, y, spec, gp
    gp.close()


def test_cfg4_loo_data_term_and_gradient_at_n1024(cfg4):
    from baybe_amd import gp_spec
    from oracle import gp_oracle as go

    X, Xt, y, spec, gp = cfg4
    rng = np.random.default_rng(4)
    p = gp_spec.initial_params(spec)
    p.lengthscale = p.lengthscale * (0.8 + 0.4 * rng.random(spec.dn))
    p.task_W = 0.3 + rng.random((4, 4))
    p.mean = 0.05
    val, g = gp.data_term(p)
    ospec = oracle_spec(spec)
    op = go.GPParams(p.lengthscale.copy(), p.noise, p.mean, 1.0, p.task_W.copy(), p.task_v.copy())
    Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
    dt = go.data_term(ospec, op, Xn, ys)
    gref = np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls, dt.g_task_B.reshape(-1)])
    assert math.isclose(val, dt.value, rel_tol=1e-10)
    assert np.allclose(g, gref, rtol=1e-8, atol=1e-9 * np.abs(gref).max())
    # and through the host's chain rules against the oracle's independent torch / autograd objective
    raw = gp_spec.pack_raw(spec, p)
    f, gr = gp_spec.objective_from_data_term(spec, raw, len(y), val, g)
    fo, gro = go.fit_objective(ospec, raw, Xn, ys)
    assert math.isclose(f, fo, rel_tol=1e-10) and np.allclose(gr, gro, rtol=1e-7, atol=1e-9 * np.abs(gro).max())


def test_cfg4_fit_posterior_and_greedy_at_full_size(cfg4):
    import torch

    from baybe_amd import engine
    from oracle import gp_oracle as go

    X, Xt, y, spec, gp = cfg4
    N = len(X)
    fi = gp.fit(maxiter=40)  # a few dozen L-BFGS-B iterations of the LOO objective on the device
    assert np.isfinite(fi.fun) and (fi.params.task_B().diagonal() > 0).all()
    ospec = oracle_spec(spec)
    om = go.GPModel(ospec, go.GPParams(fi.params.lengthscale, fi.params.noise, fi.params.mean, 1.0, fi.params.task_W,
                                       fi.params.task_v), Xt, y)
    Xd = torch.from_numpy(X).cuda()
    m, v = gp.posterior(Xd)  # nb = 64: four 16-block windows, kernel values cached between the passes (HAS_TBL form)
    pick = np.random.default_rng(0).choice(N, 2000, replace=False)
    mo, vo = om.posterior(X[pick])
    assert np.allclose(_np(m)[pick], mo, rtol=1e-9, atol=1e-12) and np.allclose(_np(v)[pick], vo, rtol=1e-8)
    # the same launch with the overflow of the kernel-value cache in global slabs / without any cache: identical
    for env in ({"BBH_KV_GLOBAL": "1"}, {"BBH_KVCACHE": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            g2 = engine.HipGP(0)
        finally:
            for k, val in old.items():
                os.environ.pop(k, None) if val is None else os.environ.__setitem__(k, val)
        g2.set_model(spec, Xt, y)
        g2.factorize(fi.params)
        m2, v2 = g2.posterior(Xd)
        assert torch.allclose(m2, m, rtol=1e-12, atol=1e-13) and torch.allclose(v2, v, rtol=1e-10, atol=1e-14), env
        g2.close()
    # optimize_acqf_discrete(q = 3): oracle over a 20k-row slice, device over the same slice and over the full set
    ro = go.optimize_acqf_discrete_qlogei(om, X[:20000], 3, seed=5)
    r2 = gp.greedy_qlogei(Xd[:20000], 3, seed=5)
    assert r2.indices == ro.indices and np.allclose(r2.values, ro.values, rtol=0, atol=1e-8)
    r = gp.greedy_qlogei(Xd, 3, seed=5)
    assert len(set(r.indices)) == 3 and all(v1 >= v2 - 1e-12 for v1, v2 in zip(r.values, r2.values))
    # the full-set picks are at least as good as the slice's, and the oracle reproduces their values
    best_f = gp.best_f()
    z1 = engine.sobol_normal_base_samples(512, 1, 5)
    mo1, vo1 = om.posterior(X[r.indices[:1]])
    assert math.isclose(go.qlogei_q1(mo1, vo1, z1[:, 0], go.best_f_from_model(om))[0], r.values[0], abs_tol=1e-8)
    assert math.isclose(best_f, go.best_f_from_model(om), rel_tol=1e-9)
    z3 = engine.sobol_normal_base_samples(512, 3, 5)
    s3 = go.qlogei_with_pending(om, X[r.indices[2:3]], X[r.indices[:2]], z3, go.best_f_from_model(om))
    assert math.isclose(s3[0], r.values[2], abs_tol=1e-8)


# ---- configs[4]: ParetoObjective, qLogNEHVI, 3 targets, 1e5 x 15, n = 256, S = 128 (BoTorch default) and 512 ------
@pytest.fixture(scope="module")
def cfg5():
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    N, d, n, m = 100_000, 15, 256, 3
    rng = np.random.default_rng(0)
    X = make_grid(N, d, 0)
    Xt = X[np.random.default_rng(1).choice(N, n, replace=False)]
    Y = np.stack([-((Xt - 0.25) ** 2).sum(1), -((Xt - 0.75) ** 2).sum(1), -np.abs(Xt - 0.5).sum(1)], 1)
    Y = Y + 0.05 * rng.standard_normal(Y.shape)
    engines, models = [], []
    for o in range(m):
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        g = engine.HipGP(0)
        g.set_model(spec, Xt, Y[:, o])
        fi = g.fit()
        engines.append(g)
        models.append(go.fit_gp(oracle_spec(spec), Xt, Y[:, o],
                                params=go.GPParams(fi.params.lengthscale, fi.params.noise, fi.params.mean)))
    yield X, Xt, Y, engines, models
    for g in engines:
        g.close()


@pytest.mark.parametrize("S", [128, 512])
def test_cfg5_qlognehvi_scores_at_full_size(cfg5, S):
    import torch

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point
    from oracle import nehvi_oracle as no

    X, Xt, Y, engines, models = cfg5
    N, m = len(X), 3
    signs = np.ones(m)
    ref = compute_ref_point(Y)
    assert np.allclose(ref, no.compute_ref_point(Y))
    seed, pseed = 1234, 99
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=True)  # acqfs.py:477-484
    hv.prepare(seed, prune_seed=pseed)
    Xd = torch.from_numpy(X).cuda()
    sg = _np(hv.score(Xd))
    assert np.isfinite(sg).all()
    keep = no.prune_baseline(models, signs, Xt, ref, pseed)
    assert np.array_equal(hv._pruned, Xt[keep]) and 0 < len(keep) < len(Xt)
    orc = no.NEHVIOracle(models, signs, Xt[keep], ref, no.sobol_normal_base_samples_nd(S, len(keep) + 1, m, seed))
    assert hv.n_cells == sum(len(c[0]) for c in orc.cells)  # same box decompositions, sample by sample
    top = np.argsort(-sg, kind="stable")[:12]
    # 500 random rows, the head of the device's ranking, and grid rows that ARE baseline points (singular joint covariance: the
    # candidate's conditional variance is rounding noise around zero, and the sign of that noise decides whether the 1 x 1 jitter
    # rule applies - see tests/test_nehvi_gpu.py; those rows are held to "no improvement", all others to 1e-8)
    base_rows = np.array([int(np.nonzero((np.abs(X - xb).sum(1) < 1e-12))[0][0]) for xb in Xt[keep][:8]])
    pick = np.concatenate([np.random.default_rng(S).choice(N, 500, replace=False), top, base_rows])
    so = orc.values(X[pick])
    from conftest import record_deviation

    dup = np.array([(np.abs(Xt[keep] - x).sum(1) < 1e-12).any() for x in X[pick]])
    dev = np.abs(sg[pick] - so)
    # Tolerance: the smoothing constant tau_relu = 1e-6 makes a score as sensitive as 1 / tau_relu to the sampled value f of a
    # candidate that lands within ~tau_relu of a cell boundary in a sample that dominates its sum.  Device and oracle agree on
    # those values to 3.5e-13 (different but equivalent factorisations: extended model vs cached baseline factor,
    # scripts/gpu_nehvi_probe.py), which bounds the score deviation by 3.5e-7; observed: one row of 1024 at 8.5e-8 (value
    # -22.8, both cell kernels - linear- and log-domain - agree with each other to 5e-10 there), every other row <= 4e-10.
    record_deviation(f"qlognehvi_scores_cfg5[S={S}]", dev[~dup].max(), 1e-6)
    record_deviation(f"qlognehvi_scores_cfg5_median[S={S}]", float(np.median(dev[~dup])), 1e-6)
    assert dup.sum() >= 8
    assert np.allclose(sg[pick][~dup], so[~dup], rtol=0, atol=1e-6), dev[~dup].max()
    assert np.quantile(dev[~dup], 0.99) <= 1e-8
    assert (sg[pick][dup] < so[~dup].max() - 5).all() and (so[dup] < so[~dup].max() - 5).all()
    # among the sample and the head of the device's ranking, oracle and device agree on the best row
    assert int(np.argmax(so)) == int(np.argmax(sg[pick]))


def test_cfg5_greedy_pair_matches_the_oracle(cfg5):
    import torch

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point
    from oracle import nehvi_oracle as no

    X, Xt, Y, engines, models = cfg5
    m, S, seed, pseed, NS = 3, 128, 1234, 99, 3000
    signs = np.ones(m)
    ref = compute_ref_point(Y)
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=True)
    Xd = torch.from_numpy(X).cuda()
    res = hv.greedy(Xd[:NS], 2, seed=seed, prune_seed=pseed)
    keep = no.prune_baseline(models, signs, Xt, ref, pseed)
    alive = np.ones(NS, bool)
    picks, vals = [], []
    for _ in range(2):  # the oracle's optimize_acqf_discrete: picks join the baseline (cache_pending), same seed
        Xb = np.vstack([Xt[keep]] + [X[i][None, :] for i in picks])
        orc = no.NEHVIOracle(models, signs, Xb, ref, no.sobol_normal_base_samples_nd(S, len(Xb) + 1, m, seed))
        v = np.full(NS, -np.inf)
        v[alive] = orc.values(X[:NS][alive])
        picks.append(int(np.argmax(v)))
        vals.append(v[picks[-1]])
        alive[picks[-1]] = False
    assert res.indices == picks and np.allclose(res.values, vals, rtol=0, atol=1e-6)
    # full set: two distinct rows, each at least as good as the slice's pick of the same step
    full = hv.greedy(Xd, 2, seed=seed, prune_seed=pseed)
    assert len(set(full.indices)) == 2 and full.values[0] >= res.values[0] - 1e-12
