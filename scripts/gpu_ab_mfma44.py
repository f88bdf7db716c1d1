"""A/B: variance GEMM on v_mfma_f64_4x4x4_4b (BBH_MFMA44=1) vs v_mfma_f64_16x16x4 (=0), two handles in
one process, interleaved rounds; also checks that both give the same posterior."""
import os, sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
for (N, d, n) in ((1_000_000, 20, 512), (100_000, 15, 256), (200_000, 8, 100), (100_000, 15, 1024)):
    X, Xt, y = synth_problem(N, d, n, 0)
    gps = {}
    for flag in ("1", "0"):
        os.environ["BBH_MFMA44"] = flag
        g = engine.HipGP(0)
        g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
        g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
        gps[flag] = g
    Xd = torch.from_numpy(X).cuda()
    outs = {}
    for flag, g in gps.items():
        outs[flag] = g.posterior(Xd); g.posterior(Xd)
    dm = (outs["1"][0] - outs["0"][0]).abs().max().item(); dv = ((outs["1"][1] - outs["0"][1]).abs() / outs["0"][1]).max().item()
    t = {"1": [], "0": []}
    for rnd in range(5):
        for flag, g in gps.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): g.posterior(Xd)
            torch.cuda.synchronize(); t[flag].append((time.perf_counter() - t0) / 10 * 1e3)
    W = n * n + 2 * n * d + 16 * n + 16 * 512
    print(f"N={N} d={d} n={n}: 4x4x4 {np.median(t['1']):.3f} ms ({N*W/np.median(t['1'])/1e9:.1f} TF alg) | 16x16x4 {np.median(t['0']):.3f} ms | speedup {np.median(t['0'])/np.median(t['1']):.3f}x | mean absdiff {dm:.2e} var reldiff {dv:.2e}")
