"""Compact timeline of the dataflow tail kernel (BBH_FIT_FLOW=1, BBH_FLOW_TRACE=1): per role type start/end statistics."""
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
ran = st[:, 0] > 0  # (roles of the table that another launch ran carry no stamps)
st, roles = st[ran], roles[ran]
t0 = st[:, 0].min()
us = lambda v: (v - t0) / 100.0
names = ["RH", "L", "XT", "MT", "VEC", "QV", "QT", "GT"]
ty = roles & 15
print(f"n={n}: {nr} roles, span {us(st[:, 2].max()):.1f} us")
for k in range(8):
    m = ty == k
    if m.any():
        e = us(st[m, 2]); s_ = us(st[m, 0])
        print(f"  {names[k]:>3s}: {m.sum():3d} roles, start {s_.min():6.1f}..{s_.max():6.1f}, end min {e.min():6.1f} median {np.median(e):6.1f} max {e.max():6.1f}")
m = ty == 3
I = (roles >> 4) & 31; J = (roles >> 9) & 31
for i in sorted(set(I[m])):
    mm = m & (I == i)
    print(f"    MT row {i}: ends {np.round(np.sort(us(st[mm, 2])), 1).tolist()}")

m = ty == 7
if m.any():
    d = lambda a, b: np.median((st[m, a] - st[m, b]) / 100.0)
    print(f"  GT medians: role start -> flag seen {d(1, 0):.1f}; loads {d(4, 1):.1f}; pass 1 {d(5, 4):.1f}; pass 2 + row sums {d(6, 5):.1f}; row write + fence + count {d(7, 6):.1f}; end {d(2, 7):.1f}")
    late = us(st[m, 1])
    print(f"  GT flag seen: min {late.min():.1f} median {np.median(late):.1f} max {late.max():.1f}")
for k, nm in ((0, "RH"), (2, "XT")):
    m = ty == k
    if m.any():
        for i in sorted(set(I[m])):
            mm = m & (I == i)
            print(f"    {nm} row {i}: ends {np.round(np.sort(us(st[mm, 2])), 1).tolist()}")
