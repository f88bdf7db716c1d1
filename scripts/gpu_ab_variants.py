"""A/B of the fused-posterior variants in one process: BBH_MFMA44 x BBH_LDS_R (interleaved rounds)."""
import os, sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
combos = [("0", "0"), ("1", "0"), ("0", "1"), ("1", "1")]
for (N, d, n) in ((1_000_000, 20, 512), (100_000, 15, 256), (100_000, 15, 1024)):
    X, Xt, y = synth_problem(N, d, n, 0)
    gps = {}
    for m44, ldsr in combos:
        os.environ["BBH_MFMA44"] = m44; os.environ["BBH_LDS_R"] = ldsr
        g = engine.HipGP(0)
        g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
        g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
        gps[(m44, ldsr)] = g
    Xd = torch.from_numpy(X).cuda()
    outs = {k: g.posterior(Xd) for k, g in gps.items()}
    ref = outs[("0", "0")]
    t = {k: [] for k in gps}
    for rnd in range(4):
        for k, g in gps.items():
            g.posterior(Xd); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): g.posterior(Xd)
            torch.cuda.synchronize(); t[k].append((time.perf_counter() - t0) / 8 * 1e3)
    W = n * n + 2 * n * d + 16 * n + 16 * 512
    for k in combos:
        dm = (outs[k][0] - ref[0]).abs().max().item(); dv = ((outs[k][1] - ref[1]).abs() / ref[1]).max().item()
        ms = np.median(t[k])
        print(f"N={N} n={n} mfma44={k[0]} lds_r={k[1]}: {ms:.3f} ms ({N*W/ms/1e9:.1f} TF alg)  mean absdiff {dm:.1e} var reldiff {dv:.1e}")
