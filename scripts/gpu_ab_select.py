"""A/B on the MI355X: the selection tail of a step (q' = 1 qLogEI + top-8 to the host) in its two forms - sample-sliced kernel +
one-pass selection (csrc/bbh_select.hip) against one thread per candidate + k-round top-k - at the row counts of BASELINE configs[1],
the 8-GPU shard of configs[2] and configs[2] itself.  Prints one JSON line per (form, rows)."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from baybe_amd import engine


def run(tag, env):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    g = engine.HipGP(0)
    rng = np.random.default_rng(0)
    z = engine.sobol_normal_base_samples(512, 1, 1234)[:, 0]
    for N in (10_000, 100_000, 125_000, 1_000_000):
        m = torch.from_numpy(rng.standard_normal(N) * 0.3).cuda()
        v = torch.from_numpy(np.exp(rng.uniform(-8, -1, N))).cuda()
        buf = torch.empty(N, dtype=torch.float64, device="cuda")
        for _ in range(5):
            g.qlogei_topk(m, v, z, 0.5, 1.0, 8, scores=buf)
        g.timing(True)
        for fam in ("q1", "select"):
            g.timing_read(reset=True, family=fam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            _, vals, idx = g.qlogei_topk(m, v, z, 0.5, 1.0, 8, scores=buf)
        wall = (time.perf_counter() - t0) / reps * 1e3
        q1 = g.timing_read(family="q1")
        sel = g.timing_read(family="select")
        g.timing(False)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.argmax(buf)
        amax = (time.perf_counter() - t0) / reps * 1e3
        print(json.dumps({"form": tag, "rows": N, "wall_ms_per_call": wall, "q1_kernel_ms": q1[0] / max(q1[1], 1),
                          "select_ms": sel[0] / reps, "select_launches_per_call": sel[1] / reps, "argmax_call_ms": amax,
                          "top": idx[:3].tolist()}))
    g.close()
    for k, val in keep.items():
        os.environ.pop(k, None) if val is None else os.environ.__setitem__(k, val)


run("sliced+select", {})
run("sliced+select, results via device buffer + copy", {"BBH_SELECT_MAPPED": "0"})
run("per-candidate+rounds", {"BBH_Q1_SLICED": "0", "BBH_SELECT": "0"})
