"""Selection on the device (``csrc/bbh_select.hip``): q' = 1 qLogEI with sample slices, chunk keys, the one-pass top-k - against the
oracle's ``qlogei_q1`` / ``topk_first_index`` (oracle/gp_oracle.py: what ``optimize_acqf_discrete`` does with the scores of a step,
baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126): descending scores, ties to the lower index, NaN never wins."""

import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    from baybe_amd import engine

    g = engine.HipGP(0)
    yield g
    g.close()


def _ref_topk(scores, k):
    """(values, indices) of the k best non-NaN scores, first index on ties; (-inf, -1) beyond their number."""
    from oracle import gp_oracle as go

    valid = np.nonzero(~np.isnan(scores))[0]
    order = valid[go.topk_first_index(scores[valid], k)]
    vals, idx = np.full(k, -np.inf), np.full(k, -1, dtype=np.int64)
    vals[: len(order)], idx[: len(order)] = scores[order], order
    return vals, idx


@pytest.mark.parametrize("N", [1, 5, 63, 64, 65, 1000, 4097, 100_000, 524_288, 524_289, 1_000_003])
def test_topk_equals_the_oracle_ranking(gp, N):
    import torch

    rng = np.random.default_rng(N)
    for variant in ("continuous", "ties", "nan_and_inf", "sorted_descending"):
        s = rng.standard_normal(N)
        if variant == "ties":
            s = np.round(s, 1)  # ~60 distinct values: long runs of equal scores across chunks
        elif variant == "nan_and_inf":
            s[rng.random(N) < 0.3] = np.nan
            s[rng.random(N) < 0.3] = -np.inf
        elif variant == "sorted_descending":
            s = -np.sort(-s)
        sd = torch.from_numpy(s).cuda()
        for k in (1, 8, 15, 64):
            k = min(k, N)
            vals, idx = gp.topk(sd, k)
            rv, ri = _ref_topk(s, k)
            assert np.array_equal(idx, ri), (N, variant, k)
            assert np.array_equal(vals, rv)
        v, i = gp.argmax(sd)
        rv, ri = _ref_topk(s, 1)
        assert i == ri[0] and (v == rv[0])


def test_topk_on_flat_scores_takes_the_in_kernel_rounds(gp):
    """All scores equal, k = 64: every element of the first 64 chunks ties with the threshold element up to its index - the
    candidate list cannot hold them and the kernel falls back to its k rounds over the selected chunks."""
    import torch

    for N in (5000, 100_000, 2_000_000):
        sd = torch.full((N,), 0.25, dtype=torch.float64, device="cuda")
        vals, idx = gp.topk(sd, 64)
        assert idx.tolist() == list(range(64)) and (vals == 0.25).all()
        sd[:40] = float("nan")
        sd[50] = 1.0
        vals, idx = gp.topk(sd, 64)
        assert idx.tolist() == [50] + list(range(40, 50)) + list(range(51, 104))
    sd = torch.full((300,), float("nan"), dtype=torch.float64, device="cuda")
    vals, idx = gp.topk(sd, 3)
    assert idx.tolist() == [-1, -1, -1] and np.isneginf(vals).all()
    sd[7] = -math.inf
    vals, idx = gp.topk(sd, 3)
    assert idx.tolist() == [7, -1, -1]


@pytest.mark.parametrize("S", [4, 7, 128, 512, 1000, 1024, 2048])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_sliced_qlogei_equals_the_oracle(gp, S, sign):
    """The sample-sliced kernel (S <= 1024; one thread per candidate beyond) on means / variances spanning every branch: improvement
    certain, impossible, marginal (samples inside the softplus transition -750 <= t <= 20), zero and negative variances."""
    import torch

    from oracle import gp_oracle as go

    rng = np.random.default_rng(S)
    z = go.sobol_normal_base_samples(S, 1, 11)[:, 0]
    N = 20_011
    mu = rng.standard_normal(N) * 0.7
    var = np.exp(rng.uniform(-30, 2, N))
    var[:50] = 0.0
    var[50:100] = -1e-13
    var[100:150] = -5.0  # the jitter ladder ends at sd = 0
    mu[150:400] = 0.3 + rng.uniform(-3e-5, 3e-5, 250)  # |objective - best_f| of a few tau: the transition region
    var[150:400] = np.exp(rng.uniform(-26, -18, 250))
    bf = 0.3 * sign
    so = go.qlogei_q1(mu, var, z, bf, sign)
    m, v = torch.from_numpy(mu).cuda(), torch.from_numpy(var).cuda()
    s = gp.qlogei(m, v, z, bf, sign).cpu().numpy()
    finite = np.isfinite(so)
    assert np.array_equal(np.isfinite(s), finite)
    err = np.abs(s[finite] - so[finite]).max()
    assert err < 1e-9, err
    alive = torch.from_numpy((rng.random(N) < 0.5).astype(np.uint8)).cuda()
    sc, vals, idx = gp.qlogei_topk(m, v, z, bf, sign, 8, alive)
    sc = sc.cpu().numpy()
    live = alive.cpu().numpy().astype(bool)
    assert np.isneginf(sc[~live]).all() and np.array_equal(sc[live], s[live])  # the same kernel, masked
    rv, ri = _ref_topk(sc, 8)
    assert np.array_equal(idx, ri) and np.array_equal(vals, rv)


def test_forms_agree_under_the_ab_switches():
    """BBH_Q1_SLICED=0 / BBH_SELECT=0 (handles read the switches when created): the former kernels give the same picks and the
    same scores to 1e-12."""
    import torch

    from baybe_amd import engine
    from oracle import gp_oracle as go

    rng = np.random.default_rng(5)
    N = 300_007
    mu, var = rng.standard_normal(N), np.exp(rng.uniform(-12, 1, N))
    z = go.sobol_normal_base_samples(512, 1, 3)[:, 0]
    m, v = torch.from_numpy(mu).cuda(), torch.from_numpy(var).cuda()
    out = {}
    for tag, env in (("new", {}), ("old", {"BBH_Q1_SLICED": "0", "BBH_SELECT": "0"})):
        keep = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            g = engine.HipGP(0)
            s, vals, idx = g.qlogei_topk(m, v, z, 0.4, 1.0, 15)
            out[tag] = (s.cpu().numpy(), vals, idx, g.argmax(s))
            g.close()
        finally:
            for k, val in keep.items():
                os.environ.pop(k, None) if val is None else os.environ.__setitem__(k, val)
    assert np.allclose(out["new"][0], out["old"][0], rtol=0, atol=1e-12)
    assert np.array_equal(out["new"][2], out["old"][2]) and out["new"][3][1] == out["old"][3][1]


def test_speculative_hit_then_miss_on_the_device():
    """The two-optima problem of ``tests/test_speculation_cpu.py`` (step 2 a hit, step 3 a miss handing over to its own pass) through
    libbaybe_hip: same picks as the oracle and as ``speculate=False``, values to 1e-8."""
    from baybe_amd import engine, gp_spec
    from oracle import gp_oracle as go
    from test_speculation_cpu import two_optima

    X, Xt, y = two_optima()
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(1, np.zeros(1), np.ones(1)), Xt, y)
    g.factorize(gp_spec.GPParams(np.array([0.12]), 1e-4, 0.0))
    om = go.fit_gp(go.GPSpec.baybe_default(1, np.zeros(1), np.ones(1)), Xt, y, params=go.GPParams(np.array([0.12]), 1e-4, 0.0))
    for q in (2, 3, 5):
        ref = go.optimize_acqf_discrete_qlogei(om, X, q, seed=5)
        on = g.greedy_qlogei(X, q, seed=5)
        off = g.greedy_qlogei(X, q, seed=5, speculate=False)
        assert on.indices == ref.indices == off.indices
        assert np.allclose(on.values, ref.values, rtol=0, atol=1e-8) and np.allclose(off.values, ref.values, rtol=0, atol=1e-8)
    assert abs(X[on.indices[0], 0] - 0.2) < 0.05 and abs(X[on.indices[1], 0] - 0.8) < 0.05
    g.close()


def test_slice_row_hint_makes_shard_scores_bit_identical():
    """the joint q'-batch kernel picks its number of sample slices from the candidate count, so a shard of another size
    adds a candidate's partial sums in another order (last-ulp differences, which can flip exact ties across ranks).  With the GLOBAL
    row count as the hint (``bbh_set_slice_rows``, ``RowShard.reproducible``) every shard reproduces the unsharded scores bit for bit."""
    import torch

    from _problems import make_problem
    from baybe_amd import engine, gp_spec

    X, Xt, y = make_problem(60_000, 6, 40, seed=3)
    g = engine.HipGP(0)
    spec = gp_spec.GPSpec.baybe_default(6, np.zeros(6), np.ones(6))
    g.set_model(spec, Xt, y)
    g.factorize(gp_spec.initial_params(spec))
    z = engine.sobol_normal_base_samples(512, 3, 7)
    Xd = torch.from_numpy(X).cuda()
    pend = X[[5, 17]]

    def scores(rows):
        mean, var = g.posterior(rows)
        g.set_pending(pend)
        cross = g.cross_cov(rows)
        s = g.qlogei_pending(mean, var, cross, z, 0.1).cpu().numpy()
        g.set_pending(None)
        return s

    full = scores(Xd)
    g.set_slice_rows(len(X))
    parts = [scores(Xd[a:b]) for a, b in ((0, 7_001), (7_001, 30_000), (30_000, 60_000))]
    g.set_slice_rows(0)
    assert np.array_equal(np.concatenate(parts), full)
    g.close()


def test_timing_can_be_restricted_to_kernel_families():
    """``bbh_timing_enable(h, 2 * mask)``: HIP events only around the named families (an event between two back-to-back kernels costs
    the stream ~5 us, so bench.py's timed region brackets the dominant kernel only); 1 = every family, 0 = none."""
    import torch

    from _problems import make_problem
    from baybe_amd import engine, gp_spec

    X, Xt, y = make_problem(20000, 5, 40, seed=2)
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(5, np.zeros(5), np.ones(5)), Xt, y)
    g.fit()
    Xd = torch.from_numpy(X).cuda()
    z = engine.sobol_normal_base_samples(128, 1, 1)[:, 0]

    def one_step():
        m, v = g.posterior(Xd)
        g.qlogei_topk(m, v, z, g.best_f(1.0), 1.0, 4)

    fams = ("posterior", "q1", "select")
    for want, on in ((("posterior",), {"posterior"}), (("q1", "select"), {"q1", "select"}), (None, set(fams))):
        g.timing(True, want)
        for f in fams:
            g.timing_read(reset=True, family=f)
        one_step()
        one_step()
        got = {f: g.timing_read(reset=True, family=f) for f in fams}
        for f in fams:
            assert (got[f][1] > 0) == (f in on), (want, got)
            assert (got[f][0] > 0.0) == (f in on)
    g.timing(False)
    one_step()
    assert all(g.timing_read(reset=True, family=f)[1] == 0 for f in fams)
    g.close()
