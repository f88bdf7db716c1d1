"""A/B of the qLogNEHVI cell kernel: (1 + u^2)^(-tau_max) factors in packed single precision (default) against the double-precision
sequence (BBH_NEHVI_PK=0), BASELINE configs[4] shape; the two score vectors are compared with each other and with the log-domain
kernel."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench
from baybe_amd import engine, gp_spec
from baybe_amd.nehvi import HipNEHVI, compute_ref_point

N, d, n, m = 100_000, 15, 256, 3
X, Xt, y = bench.synth_problem(N, d, n, 0)
ys = bench.synth_pareto_targets(Xt)
engines = []
for yo in ys:
    g = engine.HipGP(0); g.set_model(gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, np.asarray(yo)); g.fit(maxiter=30); engines.append(g)
ref = compute_ref_point(np.stack([np.asarray(v) for v in ys], 1))
Xd = torch.from_numpy(X).cuda()
for S in (512, 128):
    hv = HipNEHVI(engines, np.ones(m), Xt, ref, n_mc_samples=S, prune_baseline=True)
    hv.prepare(1234, prune_seed=99)
    out = {}
    for mode in ("0", "1"):
        os.environ["BBH_NEHVI_PK"] = mode
        sc = hv.score(Xd); torch.cuda.synchronize()
        for o in hv.outputs: o.ext.timing(True)
        for o in hv.outputs: o.ext.timing_read(True, "nehvi")
        for _ in range(5): sc = hv.score(Xd)
        torch.cuda.synchronize()
        tot = [o.ext.timing_read(True, "nehvi") for o in hv.outputs]
        ms, cnt = sum(t[0] for t in tot), sum(t[1] for t in tot)
        out[mode] = (ms / max(cnt, 1), sc.cpu().numpy())
    os.environ.pop("BBH_NEHVI_PK")
    os.environ["BBH_NEHVI_LOG"] = "1"; slog = hv.score(Xd).cpu().numpy(); os.environ.pop("BBH_NEHVI_LOG")
    a, b = out["0"][1], out["1"][1]
    fin = np.isfinite(a) & np.isfinite(b)
    dev = np.abs(a - b)[fin]
    head = np.argsort(-a, kind="stable")[:1000]
    print(f"S={S}: cell kernel fp64 tail {out['0'][0]:.3f} ms, packed fp32 tail {out['1'][0]:.3f} ms;  |pk - fp64| max {dev.max():.3e}  99% {np.quantile(dev, 0.99):.3e}  "
          f"median {np.median(dev):.3e}; over the top-1000 max {np.abs(a - b)[head].max():.3e};  |pk - log-domain| max {np.abs(b - slog)[fin & np.isfinite(slog) & (slog > -30)].max():.3e};  "
          f"same top-16: {np.array_equal(np.argsort(-a, kind='stable')[:16], np.argsort(-b, kind='stable')[:16])}", flush=True)
