import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from _replay import load_traces
from _problems import oracle_spec, oracle_params
from baybe_amd import engine, gp_spec
from oracle import gp_oracle as go
meta, data = load_traces()
c = meta["task"][0]; k = c["key"]
Xt, y = data[k + "_meas_x"], data[k + "_meas_y"][:, 0]
d = Xt.shape[1]
b = data[k + "_bounds"]
spec = gp_spec.GPSpec.baybe_default(d, b[0], b[1], task_idx=c["task_idx"], n_tasks=c["n_tasks"])
g = engine.HipGP(0); g.set_model(spec, Xt, y); fi = g.fit()
ospec = oracle_spec(spec)
Xn, ys = go.normalize_inputs(ospec, Xt), go.standardize_targets(y)[0]
fo = go.fit_hyperparameters(ospec, Xn, ys)
f_at, g_at = go.fit_objective(ospec, go.pack_raw(ospec, oracle_params(spec, fi.params)), Xn, ys)
print("device fit fun", repr(fi.fun), "nfev", fi.nfev, "| oracle objective at the device end point", repr(f_at), "|grad|", np.abs(g_at).max())
print("oracle fit fun", repr(fo.fun), "nfev", fo.nfev)
print("device ls", fi.params.lengthscale, "noise", fi.params.noise, "\noracle ls", fo.params.lengthscale, "noise", fo.params.noise)
print("device B", fi.params.task_B().round(4).tolist(), "oracle B", fo.params.task_B().round(4).tolist())
