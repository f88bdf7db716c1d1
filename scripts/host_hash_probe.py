"""Content key of a 1e6 x 20 comp rep on the GPU box's host (bbh_content_key on native threads)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, pandas as pd
import baybe_amd.recommenders as R
df = pd.DataFrame(np.random.default_rng(0).integers(0, 11, size=(1_000_000, 20)) / 10.0, columns=[f"x{j}" for j in range(20)]).copy()
R._frame_content_hash(df)
ts = []
for _ in range(9):
    t0 = time.perf_counter(); k = R._frame_content_hash(df); ts.append((time.perf_counter() - t0) * 1e3)
print(f"frame key: median {np.median(ts):.2f} ms min {min(ts):.2f}")
