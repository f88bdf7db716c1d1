import os, sys, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ["BBH_TILE_TRACE"] = "1"
import numpy as np
from _problems import make_problem, fixed_theta
from baybe_amd import engine, gp_spec
d, n = 8, 300
X, Xt, y = make_problem(3000, d, n, seed=61)
spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
ls, nz, _ = fixed_theta(d)
p = gp_spec.GPParams(np.full(d, ls), nz, 0.1)
for flow in ("1", "0"):
    for mode, env in (("tiles", {"BBH_POTRF_TILES": "1"}), ("steps", {"BBH_POTRF_TILES": "0"}), ("fallback", {"BBH_POTRF_TILES": "1", "BBH_TILE_SPIN": "0"})):
        for k in ("BBH_POTRF_TILES", "BBH_TILE_SPIN"):
            os.environ.pop(k, None)
        os.environ.update(env); os.environ["BBH_FIT_FLOW"] = flow
        g = engine.HipGP(0); g.set_model(spec, Xt, y)
        vals = [g.data_term(p)[0] for _ in range(3)]
        print(flow, mode, [repr(v) for v in vals], flush=True)
        g.close()
