"""Row sharding of the discrete candidate set over the GPUs of one node.

Candidates are independent given the (replicated, deterministic) model state, so the path
partitions rows contiguously — rank r scores rows [start_r, stop_r) — and needs exactly one
exchange per selection step: an all-gather of every rank's local winner
``[score, global row index, row(d)]`` (d + 2 doubles).  Every rank then applies the same
first-index tie-break, so all ranks agree on the pick without a broadcast; global index order
equals shard order, hence ties resolve exactly as a single-GPU argmax would.
The collective runs through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests); the payload is a few hundred bytes, i.e. latency-bound.
"""

from __future__ import annotations

import math

import numpy as np


def shard_bounds(N: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split, the first N % world ranks get one extra row."""
    base, rem = divmod(int(N), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class RowShard:
    def __init__(self, N_total: int, rank: int | None = None, world: int | None = None, group=None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.N_total = int(N_total)
        self.start, self.stop = shard_bounds(N_total, self.rank, self.world)

    @property
    def n_local(self) -> int:
        return self.stop - self.start

    def owns(self, gidx: int) -> bool:
        return self.start <= gidx < self.stop

    def to_local(self, gidx: int) -> int:
        return int(gidx) - self.start

    def to_global(self, lidx: int) -> int:
        return int(lidx) + self.start

    def _all_gather(self, payload):
        import torch

        backend = self._dist.get_backend(self.group)
        if backend == "gloo" and payload.is_cuda:  # CPU tests / single-GPU dry runs: stage through the host
            payload = payload.cpu()
        out = torch.empty((self.world, payload.numel()), dtype=payload.dtype, device=payload.device)
        self._dist.all_gather_into_tensor(out, payload.reshape(1, -1).contiguous(), group=self.group)
        return out

    @staticmethod
    def pick_winner(gathered: np.ndarray) -> int:
        """Row of ``gathered`` ([world, 2 + d]: score, global index, row) that wins: highest
        score, ties -> lowest global index; ranks without candidates (index < 0) never win."""
        best = -1
        for r in range(gathered.shape[0]):
            v, gi = gathered[r, 0], gathered[r, 1]
            if gi < 0 or math.isnan(v):
                continue
            if best < 0 or v > gathered[best, 0] or (v == gathered[best, 0] and gi < gathered[best, 1]):
                best = r
        return best

    def global_argmax(self, val: float, lidx: int, X_local):
        """All-gather the local winners; returns (score, global index, row[d]) of the global one."""
        import torch

        d = X_local.shape[1]
        dev = X_local.device
        payload = torch.empty(2 + d, dtype=torch.float64, device=dev)
        if lidx is None or lidx < 0 or self.n_local == 0:
            payload.fill_(0.0)
            payload[0] = -math.inf
            payload[1] = -1.0
        else:
            payload[0] = val
            payload[1] = float(self.to_global(lidx))
            payload[2:] = X_local[lidx, :d]
        g = self._all_gather(payload).cpu().numpy()
        w = self.pick_winner(g)
        if w < 0:
            raise RuntimeError("no rank has a candidate left")
        return float(g[w, 0]), int(g[w, 1]), g[w, 2:].copy()

    def global_topk(self, vals: np.ndarray, lidx: np.ndarray, k: int, device=None):
        """Merge per-rank top-k lists (descending) into the global top-k (ties -> lower index)."""
        import torch

        buf = np.full((k, 2), -math.inf)
        buf[:, 1] = -1.0
        m = min(k, len(vals))
        buf[:m, 0] = vals[:m]
        buf[:m, 1] = np.asarray(lidx[:m], dtype=np.float64) + self.start
        payload = torch.from_numpy(buf.reshape(-1))
        if device is not None:
            payload = payload.to(device)
        g = self._all_gather(payload).cpu().numpy().reshape(-1, 2)
        g = g[g[:, 1] >= 0]
        order = np.lexsort((g[:, 1], -g[:, 0]))[:k]
        return g[order, 0], g[order, 1].astype(np.int64)
