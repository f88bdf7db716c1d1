"""Windowed (BBH_COOP=0) against cooperative (BBH_COOP=2) form of the fused posterior kernel for the small models BayBE
campaigns live in (n <= 256), over candidate counts from 1e4 to 1e6: where the cooperative form's 16-candidate workgroups
beat the tail of the windowed form's 64-candidate workgroups.  Kernel times from HIP events (bbh_timing)."""
import math, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec


def handle(flag, d, Xt, y):
    os.environ["BBH_COOP"] = flag
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    g.timing(True)
    return g


for (d, n) in ((15, 256), (15, 192), (15, 128), (10, 64), (6, 32)):
    Xall, Xt, y = synth_problem(1_000_000, d, n, 0)
    gs = {f: handle(f, d, Xt, y) for f in ("0", "2")}
    for N in (10_000, 30_000, 100_000, 200_000, 400_000, 1_000_000):
        Xd = torch.from_numpy(Xall[:N]).cuda()
        t = {}
        variants = (("0", "1"), ("2", "0"), ("2", "1"))  # (handle, BBH_COOP_SMALL): windowed, eight-round coop, four-round coop
        for f, sm in variants:
            os.environ["BBH_COOP_SMALL"] = sm
            for _ in range(3): gs[f].posterior(Xd)
            torch.cuda.synchronize(); gs[f].timing_read(reset=True)
        for rnd in range(3):
            for f, sm in variants:
                os.environ["BBH_COOP_SMALL"] = sm
                for _ in range(10): gs[f].posterior(Xd)
                torch.cuda.synchronize()
                ms, cnt = gs[f].timing_read(reset=True)
                t.setdefault((f, sm), []).append(ms / cnt)
        fl = N * (n * n + 2 * n * d + 16 * n)
        a, b, c = (np.median(t[v]) for v in variants)
        print(f"d={d} n={n} N={N}: windowed {a*1e3:.1f} us ({fl / (a * 1e-3) / 78.6e12:.3f})  cooperative 8 rounds {b*1e3:.1f} us ({fl / (b * 1e-3) / 78.6e12:.3f})"
              f"  4 rounds / 4 workgroups per CU {c*1e3:.1f} us ({fl / (c * 1e-3) / 78.6e12:.3f})", flush=True)
    for g in gs.values(): g.close()
