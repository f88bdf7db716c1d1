"""A/B on the MI355X: the register-resident small-model form of the fused posterior kernel (csrc/bbh_small.h, n <= 64) against the
cooperative form (BBH_SMALL=0) over candidate counts from 1e4 to 1e6.  Kernel times from HIP events (bbh_timing); fraction of the
fp64 peak (78.6 TFLOP/s) on the pass's own algorithmic flops n^2 + 2 n d + 16 n per candidate."""
import math, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec


def handle(small, d, Xt, y):
    os.environ["BBH_SMALL"] = small
    g = engine.HipGP(0)
    g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
    g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
    g.timing(True)
    return g


for (d, n) in ((15, 128), (15, 100), (10, 80), (10, 64), (14, 48), (6, 32), (3, 20), (6, 16)):
    Xall, Xt, y = synth_problem(1_000_000, d, n, 0)
    gs = {f: handle(f, d, Xt, y) for f in ("1", "0")}
    for N in (10_000, 100_000, 1_000_000):
        Xd = torch.from_numpy(Xall[:N]).cuda()
        t = {}
        for f in gs:
            for _ in range(3): gs[f].posterior(Xd)
            torch.cuda.synchronize(); gs[f].timing_read(reset=True)
        for rnd in range(3):
            for f in gs:
                for _ in range(10): gs[f].posterior(Xd)
                torch.cuda.synchronize()
                ms, cnt = gs[f].timing_read(reset=True)
                t.setdefault(f, []).append(ms / cnt)
        fl = N * (n * n + 2 * n * d + 16 * n)
        a, b = np.median(t["1"]), np.median(t["0"])
        print(f"d={d} n={n} N={N}: register-resident [{gs['1'].posterior_kernel_form()}] {a*1e3:.1f} us ({fl / (a * 1e-3) / 78.6e12:.3f})   "
              f"cooperative [{gs['0'].posterior_kernel_form()}] {b*1e3:.1f} us ({fl / (b * 1e-3) / 78.6e12:.3f})", flush=True)
    for g in gs.values(): g.close()
os.environ.pop("BBH_SMALL", None)
