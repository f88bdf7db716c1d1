"""Generates tests/golden/*.npz from the CPU oracle (oracle/gp_oracle.py).

The reference (BayBE -> BoTorch/GPyTorch) cannot be imported in this environment and its own
tests hold no golden vectors for this path, so these fixtures pin the ORACLE
(regression pins) and give the GPU parity tests fixed inputs/outputs that travel to the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from _problems import fixed_theta, make_problem, make_tl_problem  # noqa: E402
from oracle import gp_oracle as go  # noqa: E402

OUT = Path(__file__).resolve().parent


def case_single(name, N, d, n, kernel, q, minimize, seed, fit):
    X, Xt, y = make_problem(N, d, n, seed=seed, minimize=minimize)
    sign = -1.0 if minimize else 1.0
    spec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), kernel=kernel)
    if fit:
        model = go.fit_gp(spec, Xt, y)
    else:
        ls, nz, c = fixed_theta(d)
        prm = go.GPParams(np.full(d, ls) * (0.8 + 0.4 * np.random.default_rng(seed + 7).random(d)), nz, 0.05)
        model = go.fit_gp(spec, Xt, y, params=prm)
    prm = model.params
    mean, var = model.posterior(X)
    bf = go.best_f_from_model(model, sign)
    z1 = go.sobol_normal_base_samples(512, 1, 1234)[:, 0]
    scores = go.qlogei_q1(mean, var, z1, bf, sign)
    pend = X[[5]]
    gr = go.optimize_acqf_discrete_qlogei(model, X, q, seed=4321, sign=sign, X_pending=pend)
    p0 = go.initial_params(spec)
    dt = go.data_term(spec, p0, model.Xn, model.ystd)
    np.savez_compressed(
        OUT / f"{name}.npz", X=X, Xt=Xt, y=y, kernel=kernel, sign=sign, fit=fit,
        ls=prm.lengthscale, noise=prm.noise, mean_const=prm.mean,
        post_mean=mean, post_var=var, best_f=bf, z1=z1, scores=scores, pend=pend,
        greedy_idx=np.array(gr.indices), greedy_val=np.array(gr.values), q=q,
        dt_value=dt.value, dt_grad=np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls]),
    )
    print(name, "argmax", int(np.argmax(scores)), "greedy", gr.indices)


def case_tl(name, N, dnum, n_per_task, T, seed):
    X, Xt, y = make_tl_problem(N, dnum, n_per_task, T=T, seed=seed)
    d = dnum + 1
    spec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d), task_idx=dnum, n_tasks=T)
    prm = go.initial_params(spec)
    rng = np.random.default_rng(seed + 3)
    prm.task_W = 0.4 + 0.6 * rng.random((T, T))
    prm.task_v = 0.2 + 0.3 * rng.random(T)
    prm.mean = -0.1
    model = go.fit_gp(spec, Xt, y, params=prm)
    mean, var = model.posterior(X)
    dt = go.data_term(spec, prm, model.Xn, model.ystd)
    np.savez_compressed(
        OUT / f"{name}.npz", X=X, Xt=Xt, y=y, T=T, ls=prm.lengthscale, noise=prm.noise, mean_const=prm.mean,
        task_W=prm.task_W, task_v=prm.task_v, post_mean=mean, post_var=var, dt_value=dt.value,
        dt_grad=np.concatenate([[dt.g_noise, dt.g_mean, dt.g_outputscale], dt.g_ls, dt.g_task_B.reshape(-1)]),
    )
    print(name, "loo", dt.value)


if __name__ == "__main__":
    case_single("cfg1_plumbing", 1000, 3, 20, "matern52", 3, False, 0, True)
    case_single("small_matern52_max", 1500, 6, 70, "matern52", 4, False, 1, False)
    case_single("small_matern52_min", 1500, 6, 70, "matern52", 3, True, 2, False)
    case_single("small_rbf", 1200, 4, 100, "rbf", 2, False, 3, False)
    case_single("small_matern32", 1200, 5, 130, "matern32", 2, False, 4, False)
    case_single("small_matern12", 1200, 5, 50, "matern12", 2, False, 5, False)
    case_single("mid_320", 2000, 20, 320, "matern52", 2, False, 6, False)
    case_tl("tl_4tasks", 1500, 5, 40, 4, 8)
