"""Golden fixture for BASELINE configs[3] (ICM over 4 tasks, n = 1024, LOO criterion): the ORACLE's complete hyper-parameter
fit - scipy L-BFGS-B with scipy's defaults over ``oracle.fit_objective`` (torch autograd), no iteration cap, from the
deterministic start of ``initial_params`` - about 1100 evaluations of 0.3 - 0.5 s.  Too slow to repeat inside the GPU suite
(ten minutes of the 90-minute GPU budget when it was tried), so its end point is stored: objective value, raw and natural
hyper-parameters.  The GPU test fits the same model on the device and compares.   python tests/golden/make_golden_cfg4_fit.py
"""

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from _problems import make_tl_problem, oracle_spec  # noqa: E402
from baybe_amd import gp_spec  # noqa: E402  (model description only - pure Python, no device code)
from oracle import gp_oracle as go  # noqa: E402


def main():
    N, dnum, T, per_task = 100_000, 15, 4, 256
    _, Xt, y = make_tl_problem(N, dnum, per_task, T=T, seed=0)
    d = dnum + 1
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=dnum, n_tasks=T)
    ospec = oracle_spec(spec)
    ystd, _, _ = go.standardize_targets(y)
    Xn = go.normalize_inputs(ospec, Xt)
    t0 = time.time()
    fo = go.fit_hyperparameters(ospec, Xn, ystd)
    print(f"oracle ICM fit: fun {fo.fun:.12f} nit {fo.nit} nfev {fo.nfev} {fo.message} in {time.time() - t0:.0f} s", flush=True)
    p = fo.params
    np.savez_compressed(Path(__file__).resolve().parent / "cfg4_oracle_fit.npz", fun=fo.fun, nit=fo.nit, nfev=fo.nfev,
                        raw=go.pack_raw(ospec, p), lengthscale=p.lengthscale, noise=p.noise, mean=p.mean,
                        task_W=p.task_W, task_v=p.task_v, task_B=p.task_B())


if __name__ == "__main__":
    main()
