"""Golden fixture for BASELINE configs[1] at full size: the ORACLE scores *all* 1e5 candidates (d = 15,
n_train = 256, Matérn-5/2, fixed-theta, qLogEI S = 512) — every greedy step of optimize_acqf_discrete(q = 5)
over the full remaining set, and the full q = 1 ranking.  ~10 minutes of CPU here; the GPU test
(tests/test_gpu_parity.py::test_cfg2_full_set_ranking_matches_the_oracle) compares the device's selection with
the stored indices, which no 4k-row sample could pin.   python tests/golden/make_golden_full_ranking.py
"""

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from _problems import fixed_theta, make_problem  # noqa: E402
from oracle import gp_oracle as go  # noqa: E402

N, d, n, q, SEED = 100_000, 15, 256, 5, 1234


def main():
    X, Xt, y = make_problem(N, d, n, seed=0)
    ls, nz, c = fixed_theta(d)
    model = go.fit_gp(go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d)), Xt, y, params=go.GPParams(np.full(d, ls), nz, c))
    t0 = time.time()
    res = go.optimize_acqf_discrete_qlogei(model, X, q, seed=SEED, keep_scores=True)
    print(f"oracle greedy q={q} over all {N} rows: {time.time() - t0:.0f} s", res.indices, res.values)
    s0 = res.first_scores
    order = go.topk_first_index(s0, 64)
    np.savez_compressed(
        Path(__file__).resolve().parent / "cfg2_full_ranking.npz",
        N=N, d=d, n=n, q=q, seed=SEED, greedy_idx=np.array(res.indices), greedy_val=np.array(res.values),
        top_idx=order, top_val=s0[order], best_f=go.best_f_from_model(model),
        # every 64th q=1 score and a checksum of all of them: the whole ranking is pinned, not only its head
        sample_scores=s0[::64].copy(), score_sum=float(np.sum(s0)), score_abs_sum=float(np.abs(s0).sum()),
        gap_top=float(s0[order[0]] - s0[order[1]]), min_gap_top16=float(np.min(-np.diff(s0[order[:16]]))),
    )


if __name__ == "__main__":
    main()
