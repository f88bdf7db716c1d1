"""Timeline of one one-launch fit evaluation (BBH_FLOW_TRACE=1): per role start / (row heads: D arrived) / end in microseconds."""
import os, sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["BBH_FLOW_TRACE"] = "1"
import numpy as np
from bench import synth_problem
from baybe_amd import engine, gp_spec, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
d = 20
X, Xt, y = synth_problem(4096, d, n, 0)
spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
g = engine.HipGP(0); g.set_model(spec, Xt, y)
th = gp_spec.theta_from_params(spec, gp_spec.initial_params(spec))
for _ in range(5): g._data_term_theta(th)
lib = _lib.load_library()
lib.bbh_flow_trace_read.restype = C.c_int
lib.bbh_flow_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
st = np.zeros((1024, 8), dtype=np.int64); roles = np.zeros(1024, dtype=np.int32)
nr = lib.bbh_flow_trace_read(g._h, st.ctypes.data, roles.ctypes.data, 1024)
st, roles = st[:nr], roles[:nr]
t0 = st[:, 0].min()
us = lambda v: (v - t0) / 100.0
names = ["RH", "L", "XT", "MT", "VEC", "QV", "QT", "GT"]
nbk_ = (n + 63) // 64
print(f"n={n}: {nr} roles, span {us(st[:, 2].max()):.1f} us")
for k in range(nr):
    ty, I, J = roles[k] & 15, (roles[k] >> 4) & 31, (roles[k] >> 9) & 31
    if ty == 0 or (ty == 1 and J == I - 2) or (ty == 2 and I == nbk_ - 1) or (ty == 3 and J == 0) or ty == 4:
        extra = (f" D at {us(st[k, 1]):7.1f} loaded {us(st[k, 7]):7.1f} factor {us(st[k, 4]):7.1f} .. {us(st[k, 5]):7.1f} published {us(st[k, 6]):7.1f}"
                 if ty == 0 and st[k, 1] else (f" factor {us(st[k, 4]):7.1f} .. {us(st[k, 5]):7.1f} published {us(st[k, 6]):7.1f}" if ty == 0 else ""))
        print(f"  ticket {k:3d} {names[ty]:>3s}({I},{J}) wg {st[k, 3]:3d}: start {us(st[k, 0]):7.1f}  end {us(st[k, 2]):7.1f}{extra}")
