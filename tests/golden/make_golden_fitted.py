"""Golden fixtures for BASELINE configs[1] and configs[2] (one GPU's shard) in the mode BayBE actually runs: the ORACLE FITS the
hyper-parameters (``go.fit_hyperparameters``: scipy L-BFGS-B on the autograd objective - what ``fit_gpytorch_mll`` does,
/root/reference/baybe/surrogates/gaussian_process/core.py:331-341) and then ranks the FULL candidate set under ITS OWN theta
(optimize_acqf_discrete, /root/reference/baybe/recommenders/pure/bayesian/botorch/discrete.py:120-126): top-64 of the q = 1
ranking, a strided sample of all scores, checksums, every step of the greedy batch of 5, the posterior on a strided sample.
The GPU test (tests/test_fitted_parity_gpu.py) runs the DEVICE's own fit and asserts the same indices.

    python tests/golden/make_golden_fitted.py cfg2      # 1e5 x 15, n = 256:  about 3 minutes of CPU here
    python tests/golden/make_golden_fitted.py cfg3      # 1e6 x 20, n = 512:  about 25 minutes of CPU here (offline)
"""

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import gp_oracle as go  # noqa: E402

Q, SEED = 5, 1234


def workload(which):
    if which == "cfg2":
        from _problems import make_problem

        return make_problem(100_000, 15, 256, seed=0), 64, 997
    import bench  # only synth_problem: the bench workload generator; no device code is touched

    return bench.synth_problem(1_000_000, 20, 512, 0), 256, 997


def main(which):
    (X, Xt, y), stride, pstride = workload(which)
    N, d = X.shape
    spec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    ystd, _, _ = go.standardize_targets(y)
    t0 = time.time()
    fit = go.fit_hyperparameters(spec, go.normalize_inputs(spec, Xt), ystd)
    print(f"{which}: oracle fit {time.time() - t0:.0f} s, fun {fit.fun:.12f}, nfev {fit.nfev}, status {fit.status} {fit.message}", flush=True)
    model = go.fit_gp(spec, Xt, y, params=fit.params)
    t0 = time.time()
    res = go.optimize_acqf_discrete_qlogei(model, X, Q, seed=SEED, keep_scores=True)
    print(f"oracle greedy q={Q} over all {N} rows under its own theta: {time.time() - t0:.0f} s", res.indices, res.values, flush=True)
    s0 = res.first_scores
    order = go.topk_first_index(s0, 64)
    mu, var = model.posterior(X[::pstride])
    np.savez_compressed(
        Path(__file__).resolve().parent / f"{which}_fitted_ranking.npz",
        N=N, d=d, n=len(y), q=Q, seed=SEED,
        lengthscale=np.asarray(fit.params.lengthscale, dtype=float), noise=float(fit.params.noise), mean=float(fit.params.mean),
        fit_fun=fit.fun, fit_nfev=fit.nfev,
        greedy_idx=np.array(res.indices), greedy_val=np.array(res.values),
        greedy_second_idx=np.array([r[0] for r in res.runner_up]), greedy_second_val=np.array([r[1] for r in res.runner_up]),
        top_idx=order, top_val=s0[order], best_f=go.best_f_from_model(model),
        sample_stride=stride, sample_scores=s0[::stride].copy(), score_sum=float(np.sum(s0)), score_abs_sum=float(np.abs(s0).sum()),
        gaps_top=-np.diff(s0[order]),
        post_rows=np.arange(N)[::pstride], post_mean=mu, post_var=var,
    )


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
