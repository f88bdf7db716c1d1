"""Times the fused posterior kernel of several builds of the library in one call (same box): one child process
per library (BAYBE_AMD_LIB) and round, rounds interleaved.   python scripts/gpu_time_libs.py [lib.so[:ENV=VAL,...] ...]"""
import math, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import math, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from bench import synth_problem
from baybe_amd import engine, gp_spec
N, d, n = 1_000_000, 20, 512
X, Xt, y = synth_problem(N, d, n, 0)
g = engine.HipGP(0)
g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y)
g.factorize(gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2) - 3) * math.sqrt(d)), math.exp(-5.0), 0.0))
Xd = torch.from_numpy(X).cuda()
m, v = g.posterior(Xd); g.posterior(Xd)
t = []
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.posterior(Xd)
    torch.cuda.synchronize(); t.append((time.perf_counter() - t0) / 10 * 1e3)
print("RESULT %%.4f %%.4f %%.17g %%.17g" %% (np.median(t), min(t), float(m.sum()), float(v.sum())))
''' % str(ROOT)

specs = sys.argv[1:] or [str(ROOT / "baybe_amd" / "libbaybe_hip.so")]
res = {s: [] for s in specs}
for rnd in range(2):
    for s in specs:
        lib, _, envs = s.partition(":")
        env = dict(os.environ, BAYBE_AMD_LIB=str(Path(lib).resolve()))
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("RESULT")]
        if not line:
            print(s, "FAILED", out.stderr[-400:]); continue
        res[s].append(line[0].split()[1:])
for s, r in res.items():
    meds = [float(x[0]) for x in r]
    fl = 1e6 * 299008
    print(f"{s.replace(str(ROOT), ''):56s} median ms {meds}  best {min(meds):.3f} ms = {fl / (min(meds) * 1e-3) / 78.6e12:.3f} of peak  "
          f"checksums {r[0][2] if r else None} {r[0][3] if r else None}", flush=True)
