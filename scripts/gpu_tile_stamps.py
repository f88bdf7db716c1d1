"""Clock stamps inside the Gram-building tile-dataflow factorisation (BBH_TILE_STAMPS=1): where a tile's first microseconds go."""
import os, sys, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["BBH_TILE_STAMPS"] = "1"
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
lib.bbh_tiles_trace_read.restype = C.c_int
lib.bbh_tiles_trace_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
st = np.zeros((512, 8), dtype=np.int64)
nt = lib.bbh_tiles_trace_read(g._h, st.ctypes.data, 512)
st = st[:nt]
t0 = st[:, 0].min()
us = lambda v: (v - t0) / 100.0
nbk = (n + 63) // 64
print(f"n={n}: {nt} tiles")
for k in range(nbk):
    r = [us(st[k, j]) for j in range(8)]
    print(f"  row head {k}: entry {r[0]:6.1f} tiles ready {r[1]:6.1f} | D seen {r[2]:6.1f} loaded {r[3]:6.1f} panel {r[4]:6.1f} factor {r[5]:6.1f} .. {r[6]:6.1f} published {r[7]:6.1f}"
          + (f"   [wait->seen {r[2] - prev:4.1f}, load {r[3] - r[2]:4.1f}, panel {r[4] - r[3]:4.1f}, update {r[5] - r[4]:4.1f}, factor {r[6] - r[5]:4.1f}, publish {r[7] - r[6]:4.1f}; polls {int(st[k, 1])}]" if k else ""))
    prev = r[7]
