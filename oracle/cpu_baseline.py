"""Reference-SHAPED CPU scoring path (torch-CPU fp64) — the timed ``cpu_baseline`` of bench.py.

TEST/BENCH INFRASTRUCTURE (see oracle/__init__.py).  Same maths as ``gp_oracle`` but with the
*shape of computation* BoTorch has on this path: candidates in
chunks of 2048 as q=1 t-batches, the training inputs expanded and concatenated per chunk as
ExactGP.__call__ does, batched pairwise distances through ||a||^2 + ||b||^2 - 2ab with mean
centring (gpytorch MaternKernel.forward), the cached root K* L^-T variance GEMM
(fast_pred_var), an S x b MC tensor pushed through log_fatplus and logmeanexp, then argmax.
Labelled "restated reference CPU path" — it is NOT BayBE/BoTorch itself (not importable here).
"""

from __future__ import annotations

import math
import time

import numpy as np
import torch

from oracle import gp_oracle as go

SQRT5 = math.sqrt(5.0)


class ReferenceShapedScorer:
    def __init__(self, model: go.GPModel, z: np.ndarray, best_f: float, sign: float = 1.0):
        assert model.spec.kernel == "matern52" and model.spec.task_idx is None
        self.m = model
        f64 = torch.float64
        self.lo = torch.tensor(model.spec.lo, dtype=f64)
        self.rng = torch.tensor(model.spec.hi - model.spec.lo, dtype=f64)
        self.Xn = torch.tensor(model.Xn, dtype=f64)  # [n, d]
        self.ls = torch.tensor(model.params.lengthscale, dtype=f64)
        self.alpha = torch.tensor(model.alpha, dtype=f64)
        L = torch.tensor(model.L, dtype=f64)
        self.Rinv = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=f64), upper=False).T.contiguous()  # L^-T
        self.c = float(model.params.mean)
        self.ybar, self.ysd = float(model.ybar), float(model.ysd)
        self.z = torch.tensor(z, dtype=f64).reshape(-1, 1)  # [S, 1]
        self.best_f, self.sign = float(best_f), float(sign)

    @torch.no_grad()
    def score_chunk(self, Xc: torch.Tensor) -> torch.Tensor:
        b, n = Xc.shape[0], self.Xn.shape[0]
        Xq = ((Xc - self.lo) / self.rng).unsqueeze(-2)  # b x 1 x d  (Normalize)
        full = torch.cat([self.Xn.expand(b, n, -1), Xq], dim=-2)  # b x (n+1) x d  (ExactGP.__call__)
        train, test = full[:, :n, :], full[:, n:, :]
        mean_c = train.mean(dim=-2, keepdim=True)  # gpytorch centring
        x1 = (test - mean_c) / self.ls
        x2 = (train - mean_c) / self.ls
        x1n = x1.pow(2).sum(-1, keepdim=True)
        x2n = x2.pow(2).sum(-1, keepdim=True)
        sq = (x1n + x2n.transpose(-1, -2) - 2.0 * x1 @ x2.transpose(-1, -2)).clamp_min_(1e-30)  # b x 1 x n
        dist = sq.sqrt()
        Ks = (1.0 + SQRT5 * dist + (5.0 / 3.0) * sq) * torch.exp(-SQRT5 * dist)
        mu = self.c + (Ks @ self.alpha.unsqueeze(-1)).squeeze(-1)  # b x 1
        root = Ks @ self.Rinv  # b x 1 x n
        var = (1.0 - root.pow(2).sum(-1)).clamp_min(0.0)  # b x 1
        mu = self.ybar + self.ysd * mu
        sd = self.ysd * var.sqrt()
        samples = mu.T + sd.T * self.z  # S x b
        t = (self.sign * samples - self.best_f) / go.TAU_RELU
        li = math.log(go.TAU_RELU) + torch.log(torch.nn.functional.softplus(t) + 0.1 / (1.0 + t * t))
        return torch.logsumexp(li, dim=0) - math.log(li.shape[0])

    @torch.no_grad()
    def score(self, X: np.ndarray, chunk: int = go.MAX_BATCH_SIZE) -> np.ndarray:
        Xt = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64))
        return torch.cat([self.score_chunk(c) for c in Xt.split(chunk)]).numpy()


def time_cpu_baseline(model: go.GPModel, X: np.ndarray, z: np.ndarray, best_f: float, budget_s: float = 15.0,
                      chunk: int = go.MAX_BATCH_SIZE):
    """Times chunked scoring on rows of X until ~budget_s of CPU work; returns
    (candidates_per_s, candidates_scored, threads)."""
    sc = ReferenceShapedScorer(model, z, best_f)
    Xt = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64))
    sc.score_chunk(Xt[:chunk])  # warm-up (MKL thread pool, allocator)
    done, t0 = 0, time.perf_counter()
    best = (-math.inf, -1)
    for s in range(0, Xt.shape[0], chunk):
        v = sc.score_chunk(Xt[s : s + chunk])
        i = int(torch.argmax(v))
        if float(v[i]) > best[0]:
            best = (float(v[i]), s + i)
        done += min(chunk, Xt.shape[0] - s)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done / dt, done, torch.get_num_threads()
