"""Does the tile-dataflow factorisation ever give up on an idle device?  500 factorisations at np = 320 and np = 512, max / median time."""
import sys, time, math, os
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["BBH_TILE_TRACE"] = "1"
import numpy as np
from bench import synth_problem
from baybe_amd import engine, gp_spec
for n, d in ((287, 15), (512, 20)):
    X, Xt, y = synth_problem(4096, d, n, 0)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    params = gp_spec.GPParams(np.full(d, math.exp(math.sqrt(2.0) - 3.0) * math.sqrt(d)), math.exp(-5.0), 0.0)
    gs = [engine.HipGP(0) for _ in range(3)]
    ts = []
    for it in range(300):
        for g in gs:
            t0 = time.perf_counter(); g.set_model(spec, Xt, y); t1 = time.perf_counter(); g.factorize(params); t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
    ts = np.array(ts) * 1e3
    print(f"n={n}: set_model median {np.median(ts[:,0]):.3f} max {ts[:,0].max():.3f} ms;  factorize median {np.median(ts[:,1]):.3f} max {ts[:,1].max():.3f} ms; "
          f"slow factorizes (> 3 ms): {np.nonzero(ts[:,1] > 3)[0].tolist()[:10]}", flush=True)
