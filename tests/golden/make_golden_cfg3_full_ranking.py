"""Golden fixture for BASELINE configs[2] (one GPU's shard) at full size: the ORACLE scores *all* 1e6 candidates
(d = 20, n_train = 512, Matérn-5/2, fixed-theta, qLogEI S = 512) — every greedy step of optimize_acqf_discrete(q = 5)
over the full remaining set, and the full q = 1 ranking.  The workload is exactly ``bench.synth_problem(1e6, 20, 512, 0)``
(rank 0's grid of bench.py), so ``bench.py``'s ``extra.greedy_q5_indices`` must equal ``greedy_idx`` here.
~20-30 minutes of CPU (8 cores).    python tests/golden/make_golden_cfg3_full_ranking.py
"""

import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402  (only synth_problem: the bench workload generator; no device code is touched)
from oracle import gp_oracle as go  # noqa: E402

N, d, n, q, SEED = 1_000_000, 20, 512, 5, 1234


def main():
    X, Xt, y = bench.synth_problem(N, d, n, 0)
    ls = math.exp(math.sqrt(2.0) - 3.0) * math.sqrt(d)
    model = go.fit_gp(go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y,
                      params=go.GPParams(np.full(d, ls), math.exp(-5.0), 0.0))
    t0 = time.time()
    res = go.optimize_acqf_discrete_qlogei(model, X, q, seed=SEED, keep_scores=True)
    print(f"oracle greedy q={q} over all {N} rows: {time.time() - t0:.0f} s", res.indices, res.values, flush=True)
    s0 = res.first_scores
    order = go.topk_first_index(s0, 64)
    mu, var = model.posterior(X[::997])
    np.savez_compressed(
        Path(__file__).resolve().parent / "cfg3_full_ranking.npz",
        N=N, d=d, n=n, q=q, seed=SEED, greedy_idx=np.array(res.indices), greedy_val=np.array(res.values),
        top_idx=order, top_val=s0[order], best_f=go.best_f_from_model(model),
        # every 256th q=1 score and checksums of all of them: the whole ranking is pinned, not only its head
        sample_scores=s0[::256].copy(), score_sum=float(np.sum(s0)), score_abs_sum=float(np.abs(s0).sum()),
        gap_top=float(s0[order[0]] - s0[order[1]]), min_gap_top16=float(np.min(-np.diff(s0[order[:16]]))),
        post_rows=np.arange(N)[::997], post_mean=mu, post_var=var,
    )


if __name__ == "__main__":
    main()
