"""qLogNEHVI on the device against the oracle (BASELINE configs[4] in miniature): scores of q=1
t-batches, baseline pruning, greedy batches; minimised targets; partial measurements (per-target
training sets) through the plug-in surface."""

import numpy as np
import pandas as pd
import pytest

from _problems import make_grid

pytestmark = pytest.mark.gpu
NEHVI_ATOL = 1e-8  # qLogNEHVI scores, absolute (the tolerance of the qLogEI scores)


def _targets(X, rng, noise=0.05):
    f1 = -((X - 0.25) ** 2).sum(1) + noise * rng.standard_normal(len(X))
    f2 = -((X - 0.75) ** 2).sum(1) + noise * rng.standard_normal(len(X))
    f3 = -np.abs(X - 0.5).sum(1) + noise * rng.standard_normal(len(X))
    return np.stack([f1, f2, f3], 1)


def _setup(m, n=24, N=150, d=3, seed=0, signs=None):
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go

    rng = np.random.default_rng(seed)
    X = make_grid(N, d, seed)
    Xt = make_grid(4 * n, d, seed + 1)[:n]
    Y = _targets(Xt, rng)[:, :m]
    signs = np.ones(m) if signs is None else np.asarray(signs, float)
    engines, models = [], []
    for o in range(m):
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        g = engine.HipGP(0)
        g.set_model(spec, Xt, Y[:, o])
        fi = g.fit()
        engines.append(g)
        ospec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        models.append(go.fit_gp(ospec, Xt, Y[:, o], params=go.GPParams(fi.params.lengthscale, fi.params.noise, fi.params.mean)))
    return X, Xt, Y, signs, engines, models


@pytest.mark.parametrize("m,signs", [(2, None), (3, None), (2, [1.0, -1.0])])
def test_scores_match_oracle(m, signs):
    import torch

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point
    from oracle import nehvi_oracle as no

    X, Xt, Y, signs, engines, models = _setup(m, signs=signs)
    ref = compute_ref_point(Y * signs[None, :])
    S, seed = 32, 11
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=False)
    hv.prepare(seed)
    sg = hv.score(torch.from_numpy(X).cuda()).cpu().numpy()
    z = no.sobol_normal_base_samples_nd(S, len(Xt) + 1, m, seed)
    orc = no.NEHVIOracle(models, signs, Xt, ref, z)
    so = orc.values(X[:60])
    # Candidates coinciding with a baseline point have a singular joint covariance.  BoTorch draws a candidate through the
    # cached baseline factor (sample_cached_cholesky): only its 1 x 1 conditional variance goes through psd_safe_cholesky -
    # the oracle restates that and the device applies the same rule (bbh_safe_sd).  For such a row that variance is rounding
    # noise around zero (|v| ~ 1e-16), and whether the 1e-8 jitter applies depends on its SIGN - in BoTorch as here - so the
    # row's value is one of two deep-tail numbers (sd = 1e-4 or ~1e-8) on either side; they are held to "no improvement" (far
    # below the best value), every other row to 1e-8.  Round 2 held the regular rows to 2e-5 without a reason: the samples
    # agree to 1e-14, the scores to ~3e-10 (single-precision logarithms under the power tau_max = 0.01 in the cell kernel).
    from conftest import record_deviation

    dup = np.array([(np.abs(Xt - x).sum(1) < 1e-12).any() for x in X[:60]])
    dev = np.abs(sg[:60] - so)
    tag = f"m={m},signs={'mixed' if (signs < 0).any() else 'max'}"
    record_deviation(f"qlognehvi_scores_small[{tag}]", dev[~dup].max(), NEHVI_ATOL)
    assert dup.sum() >= 1
    assert np.allclose(sg[:60][~dup], so[~dup], rtol=0, atol=NEHVI_ATOL), dev[~dup].max()
    assert (sg[:60][dup] < so[~dup].max() - 5).all() and (so[dup] < so[~dup].max() - 5).all()
    assert int(np.argmax(sg[:60])) == int(np.argmax(so))
    # cells on the device side equal the oracle's per-sample decompositions
    assert hv.n_cells == sum(len(c[0]) for c in orc.cells)
    # the targets' passes overlap on one stream per target (round 4); in sequence they give bitwise the same scores
    assert hv.concurrent and len(hv._streams) == m
    hv.concurrent = False
    assert np.array_equal(hv.score(torch.from_numpy(X).cuda()).cpu().numpy(), sg)


def test_pruning_and_greedy_match_oracle():
    import torch

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point
    from oracle import nehvi_oracle as no

    m = 2
    X, Xt, Y, signs, engines, models = _setup(m, n=20, N=120, seed=3)
    ref = compute_ref_point(Y)
    S, seed, pseed = 32, 5, 9
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=S, prune_baseline=True)
    res = hv.greedy(torch.from_numpy(X).cuda(), 2, seed=seed, prune_seed=pseed)
    keep = no.prune_baseline(models, signs, Xt, ref, pseed)
    assert np.array_equal(hv._pruned, Xt[keep])
    # oracle greedy: picks join the baseline, same seed, dimension (n_b + 1) m
    alive = np.ones(len(X), bool)
    picks, vals = [], []
    for _ in range(2):
        Xb = np.vstack([Xt[keep]] + [X[i][None, :] for i in picks])
        z = no.sobol_normal_base_samples_nd(S, len(Xb) + 1, m, seed)
        orc = no.NEHVIOracle(models, signs, Xb, ref, z)
        v = np.full(len(X), -np.inf)
        v[alive] = orc.values(X[alive])
        i = int(np.argmax(v))
        picks.append(i)
        vals.append(v[i])
        alive[i] = False
    assert res.indices == picks
    assert np.allclose(res.values, vals, rtol=0, atol=NEHVI_ATOL)


def test_pareto_recommendation_through_the_plugin_surface():
    """tests/test_surrogate.py:98-107 of the reference: composite (per-target) surrogates with a
    ParetoObjective and batch size 2; partial measurements are filtered per target."""
    from _baybe_shim import NumericalDiscreteParameter, NumericalTarget, ParetoObjective, SearchSpace
    from baybe_amd.recommenders import HipBotorchRecommender

    rng = np.random.default_rng(2)
    vals = np.arange(8) / 7.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
    exp = space.discrete.exp_rep
    meas = exp.iloc[rng.choice(len(exp), 18, replace=False)].copy()
    T = _targets(meas[["x0", "x1", "x2"]].to_numpy(float), rng)
    meas["t1"], meas["t2"] = T[:, 0], -T[:, 1]
    meas.loc[meas.index[0], "t2"] = np.nan  # partial measurement
    obj = ParetoObjective([NumericalTarget("t1"), NumericalTarget("t2", minimize=True)])
    rec = HipBotorchRecommender()
    got = rec.recommend(2, space, obj, meas)
    assert len(got) == 2 and len(set(got.index)) == 2
    stats = rec._surrogate_model.posterior_stats(exp.iloc[:10])
    assert list(stats.columns) == ["t1_mean", "t1_std", "t2_mean", "t2_std"]
    acq = rec.acquisition_values(exp.iloc[:50], space, obj, meas)
    assert isinstance(acq, pd.Series) and np.isfinite(acq.to_numpy()).all()


@pytest.mark.parametrize("m,nb,S", [(1, 5, 16), (2, 12, 64), (3, 31, 128), (3, 130, 64), (4, 24, 64), (3, 1, 8)])
def test_device_box_decompositions_equal_the_host_form(m, nb, S):
    """``bbh_cells_build_dev`` (one wavefront per sample, bound lists in LDS) against ``bbh_cells_create`` (host C++) on the same
    samples: the same cells in the same order - offsets and lower corners bitwise, log side lengths to rounding (the device takes
    log(len), the host log(min(up, 1e10) - lo): the same expression).  Ties, duplicated points, points below the reference point
    and dominated points are in the samples."""
    import ctypes as C

    import torch

    from baybe_amd import _lib, engine
    from baybe_amd.box_decomposition import pack_cells_native

    rng = np.random.default_rng(100 * m + nb)
    Y = rng.standard_normal((S, nb, m))
    Y[:, : nb // 3] = np.round(Y[:, : nb // 3], 1)  # ties in single coordinates
    if nb >= 4:
        Y[:, 1] = Y[:, 0]  # an exact duplicate
        Y[::3, 2] = -5.0  # a point below the reference point in every sample of a third
    ref = np.full(m, -0.8)
    g = engine.HipGP(0)
    lib = _lib.load_library()
    Yd = torch.from_numpy(Y).cuda()
    total, over = C.c_int64(), C.c_int64()
    g._check(lib.bbh_cells_build_dev(g._h, Yd.data_ptr(), S, nb, m, ref.ctypes.data_as(_lib.c_double_p), C.byref(total), C.byref(over)),
             "bbh_cells_build_dev")
    off_h, lo_h, ll_h = pack_cells_native(Y, ref)
    if over.value:  # (only four objectives can exceed the bound capacity; then the product takes the host form)
        assert m == 4
        return
    assert total.value == off_h[-1]
    off = np.zeros(S + 1, dtype=np.int64)
    lo, ll = np.zeros((total.value, m)), np.zeros((total.value, m))
    g._check(lib.bbh_cells_read_dev(g._h, off.ctypes.data_as(_lib.c_int64_p), lo.ctypes.data_as(_lib.c_double_p),
                                    ll.ctypes.data_as(_lib.c_double_p)), "bbh_cells_read_dev")
    assert np.array_equal(off, off_h)
    assert np.array_equal(lo, lo_h)
    assert np.allclose(ll, ll_h, rtol=0, atol=1e-14)
    g.close()


@pytest.mark.parametrize("n,seed,atol_cells,atol_scores", [(24, 0, 1e-10, 1e-9), (30, 7, 1e-9, 1e-8)])
def test_device_setup_equals_the_host_setup(n, seed, atol_cells, atol_scores):
    """The device set-up of a selection step (samples through the extended factor, weight columns L^-T [t; z], device box
    decompositions) against the round-4 host set-up (baseline posterior -> host Cholesky -> host samples / decompositions ->
    full target columns): the same pruned baseline, the same cells, the same scores.  The second case has three duplicated
    training rows (replicate measurements): the baseline's joint covariance is singular there and an extended model with two
    noise-free rows at one location has a rounding-noise pivot (the first device form returned scores off by O(1) there) - a
    repeated point enters the model and the decompositions once, its copies keep their base-sample columns (``_unique_rows``)."""
    import torch

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point

    m = 3
    X, Xt, Y, signs, engines, models = _setup(m, n=n, N=400, seed=seed)
    assert (len(np.unique(Xt, axis=0)) < len(Xt)) == (n == 30)
    ref = compute_ref_point(Y)
    Xd = torch.from_numpy(X).cuda()
    out = {}
    for mode in ("device", "host"):
        hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=64, prune_baseline=True)
        hv.device_setup = mode == "device"
        hv.prepare(21, prune_seed=22)
        assert hv._cells_on_device == (mode == "device")
        out[mode] = (hv._pruned.copy(), hv.cells(), hv.score(Xd).cpu().numpy())
    assert np.array_equal(out["device"][0], out["host"][0])  # the same pruned baseline
    (off_d, lo_d, ll_d), (off_h, lo_h, ll_h) = out["device"][1], out["host"][1]
    assert np.array_equal(off_d, off_h)
    if atol_cells is not None:
        assert np.allclose(lo_d, lo_h, rtol=0, atol=atol_cells), np.abs(lo_d - lo_h).max()
        assert np.allclose(ll_d, ll_h, rtol=0, atol=100 * atol_cells)
    sd, sh = out["device"][2], out["host"][2]
    dup = np.array([(np.abs(Xt - x).sum(1) < 1e-12).any() for x in X])
    assert np.allclose(sd[~dup], sh[~dup], rtol=0, atol=atol_scores), np.abs(sd - sh)[~dup].max()
    assert int(np.argmax(sd)) == int(np.argmax(sh))


# ---- round 6: base samples drawn on the device, fits side by side -------------------------------------------------------------------
@pytest.mark.parametrize("S,q,seed", [(9, 1, 3), (128, 96, 1234), (2048, 768, 4321), (512, 6, 99)])
def test_device_base_samples_equal_the_host_draw(S, q, seed):
    """``bbh_sobol_normal_dev`` (generator and scrambling on the host, points and the normal transform one thread per value on the
    device) against ``sobol_normal_base_samples`` (torch's engine + torch.erfinv): the same values up to the last bits of erfinv."""
    from baybe_amd import engine

    gp = engine.HipGP(0)
    got = gp.sobol_normal_dev(S, q, seed).cpu().numpy()
    want = engine.sobol_normal_base_samples(S, q, seed)
    assert got.shape == want.shape
    from conftest import record_deviation

    record_deviation(f"device_base_samples_rel_{S}x{q}", float(np.abs(got / want - 1).max()), 1e-14)
    assert np.allclose(got, want, rtol=1e-14, atol=0)
    gp.close()


def test_pruning_with_the_device_draw_equals_the_host_draw(monkeypatch):
    """The 2048-sample pruning draw never exists on the host (``bbh_nehvi_samples_dev`` reads the device draw through per-target
    offsets, repeated baseline rows skipped): same kept points as with the host draw, with and without repeated measurements."""
    from baybe_amd.nehvi import HipNEHVI, compute_ref_point

    X, Xt, Y, signs, engines, _ = _setup(3, n=40, seed=3)
    ref = compute_ref_point(Y * signs[None, :])
    for Xb in (Xt, np.vstack([Xt, Xt[[3, 7, 7]]])):
        kept = []
        for draw in (True, False):
            hv = HipNEHVI(engines, signs, Xb, ref, n_mc_samples=32, prune_baseline=True)
            hv.device_draw = draw
            kept.append(hv.prune_points(Xb, 4321))
            assert ("base_samples" in hv.last_prune_ms)
        assert kept[0].shape == kept[1].shape and np.array_equal(kept[0], kept[1])
        assert 0 < len(kept[0]) <= len(Xb)


def test_fits_side_by_side_equal_fits_in_sequence():
    """``engine.fit_side_by_side`` (what ``HipCompositeImpl.fit`` runs: one host thread and one private stream per target, the
    tile-dataflow launches entered in the device's residency ledger) ends at the hyper-parameters of the same fits run one after the
    other - every evaluation is a function of its own handle's data only."""
    from baybe_amd import engine, gp_spec

    rng = np.random.default_rng(5)
    d, n = 6, 300  # np = 320: the tile-dataflow factorisation + dataflow tail (64 < np <= 1024)
    Xt = rng.integers(0, 11, size=(n, d)) / 10.0
    Y = _targets(Xt, rng)
    engines = []
    for o in range(3):
        g = engine.HipGP(0)
        g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, Y[:, o])
        engines.append(g)
    seq = [g.fit() for g in engines]
    par = engine.fit_side_by_side([g.fit for g in engines], device=0)
    for a, b in zip(seq, par):
        assert a.nfev == b.nfev and a.fun == b.fun
        assert np.array_equal(a.params.lengthscale, b.params.lengthscale) and a.params.noise == b.params.noise
    for g in engines:
        g.close()


def test_installed_columns_are_invalidated_not_freed_by_a_refit():
    """a same-shape ``set_model`` keeps the (N-independent) column buffers of ``bbh_set_mean_columns`` and only
    invalidates what they hold - ``bbh_posterior_columns`` refuses until new columns are installed, and the re-installed ones give what a
    fresh handle gives (the refit path has no ``hipFree`` and with it no device-wide synchronisation)."""
    import torch

    from baybe_amd import engine, gp_spec
    from baybe_amd.engine import HipError

    rng = np.random.default_rng(4)
    d, n, S, N = 4, 90, 24, 700
    X = rng.random((N, d))
    Xt = rng.random((n, d))
    y = np.sin(3 * Xt[:, 0]) + Xt[:, 1:].sum(1) + 0.05 * rng.standard_normal(n)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    params = gp_spec.GPParams(np.full(d, 0.6), 0.01, 0.1)
    Y = y[:, None] + 0.1 * rng.standard_normal((n, S))
    g = engine.HipGP(0)
    g.set_model(spec, Xt, y)
    g.factorize(params)
    g.set_mean_columns(Y)
    want = g.posterior_columns(torch.from_numpy(X).cuda()).cpu().numpy()
    # a refit of the same shape: new measurements, same padded size
    y2 = y + 0.01
    g.set_model(spec, Xt, y2)
    g.factorize(params)
    with pytest.raises(HipError):
        g.posterior_columns(torch.from_numpy(X).cuda())  # the old columns belong to the old model
    g.set_mean_columns(Y)
    got = g.posterior_columns(torch.from_numpy(X).cuda()).cpu().numpy()
    fresh = engine.HipGP(0)
    fresh.set_model(spec, Xt, y2)
    fresh.factorize(params)
    fresh.set_mean_columns(Y)
    ref = fresh.posterior_columns(torch.from_numpy(X).cuda()).cpu().numpy()
    assert np.array_equal(got, ref) and got.shape == want.shape == (N, S)
    g.close()
    fresh.close()
