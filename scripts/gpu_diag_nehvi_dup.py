import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from test_nehvi_gpu import _setup
from baybe_amd.nehvi import HipNEHVI, compute_ref_point
m = 3
X, Xt, Y, signs, engines, models = _setup(m, n=30, N=400, seed=7)
ref = compute_ref_point(Y)
Xd = torch.from_numpy(X).cuda()
res = {}
for mode in ("device", "host"):
    hv = HipNEHVI(engines, signs, Xt, ref, n_mc_samples=64, prune_baseline=True)
    hv.device_setup = mode == "device"
    hv.prepare(21, prune_seed=22)
    Xb = hv.X_b_current
    print(mode, "baseline", len(Xb), "unique", len(np.unique(Xb, axis=0)), "jitter", [o.ext.jitter for o in hv.outputs])
    off, lo, ll = hv.cells()
    sc = hv.score(Xd).cpu().numpy()
    cols = [o.ext.posterior_columns(Xd[:5]).cpu().numpy() for o in hv.outputs]
    res[mode] = (off, lo, sc, cols)
    print("  scores head", sc[:6])
for o in range(m):
    print("target", o, "max |columns dev - host| on 5 rows", np.abs(res["device"][3][o] - res["host"][3][o]).max(), "scale", np.abs(res["host"][3][o]).max())
print("cells sorted diff", np.abs(np.sort(res["device"][1], axis=0) - np.sort(res["host"][1], axis=0)).max())
