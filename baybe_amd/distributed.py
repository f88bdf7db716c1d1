"""Row sharding of the discrete candidate set over the GPUs of one node.

Candidates are independent given the (replicated, deterministic) model state, so the path
partitions rows contiguously — rank r scores rows [start_r, stop_r) — and needs exactly one
exchange per selection step: an all-gather of every rank's local winner
``[score, global row index, row(d)]`` (d + 2 doubles).  Every rank then applies the same
first-index tie-break, so all ranks agree on the pick without a broadcast; global index order
equals shard order, hence ties resolve exactly as a single-GPU argmax would.
The collective runs through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" for the CPU tests); the payload is a few hundred bytes, i.e. latency-bound.  The library's
own RCCL entry points (``bbh_comm_init`` / ``bbh_allgather_topk`` / ``bbh_allgather_argmax``, payload built
on the device) do the same exchange without torch once ``RowShard.bind_rccl(engine)`` has run.

Agreement does not rest on coincidence: whatever a rank derives from its private state - fitted
hyper-parameters (a fit retry re-samples start points from the priors), MC sampler seeds - is taken
from rank 0 (``RowShard.agree``) before it is used.
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
        self.use_rccl = False  # True: selections go through the library's own communicator (bbh_allgather_*)
        # True: the joint q'-batch kernels add a candidate's MC partial sums in the order of the UNSHARDED pass (slice count from the
        # global row count): scores bit-identical to a single-device run whatever the shard sizes, so exact ties resolve alike;
        # default off - per-shard slice counts fill the device better, and scores then agree to the last ulp or two only
        self.reproducible = False

    def bind_rccl(self, engine) -> None:
        """Create the library-side RCCL communicator of ``engine`` (a ``HipGP`` handle) for this shard layout: rank 0's
        ncclUniqueId travels through ``agree`` (one torch.distributed broadcast), then every rank joins."""
        uid = None
        if self.rank == 0:
            try:
                uid = engine.comm_unique_id()
            except Exception as ex:  # noqa: BLE001  (RCCL not loadable): every rank must learn of it - the others wait in `agree`
                uid = ("error", str(ex))
        uid = self.agree(uid)
        if isinstance(uid, tuple):
            raise RuntimeError(f"rank 0 could not create an ncclUniqueId: {uid[1]}")
        engine.comm_init(self.rank, self.world, uid)
        self.use_rccl = True

    def rccl_bound(self, engine) -> bool:
        return self.use_rccl and getattr(engine, "_comm", None) == (self.rank, self.world)

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

    def agree(self, value):
        """Rank 0's ``value`` on every rank (one broadcast of a pickled object: hyper-parameters, seeds)."""
        if self.world == 1:
            return value
        box = [value if self.rank == 0 else None]
        src = self._dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self._dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def _payload_to_backend(self, payload_host: np.ndarray, device):
        """One host array -> one tensor where the backend wants it (a single H2D copy for RCCL)."""
        import torch

        t = torch.from_numpy(np.ascontiguousarray(payload_host, dtype=np.float64))
        if self._dist.get_backend(self.group) != "gloo" and device is not None and torch.device(device).type == "cuda":
            t = t.to(device)
        return t

    def global_argmax(self, val: float, lidx: int, X_local):
        """All-gather the local winners; returns (score, global index, row[d]) of the global one."""
        d = X_local.shape[1]
        payload = np.zeros(2 + d)
        payload[0], payload[1] = -math.inf, -1.0
        if lidx is not None and lidx >= 0 and self.n_local > 0:
            payload[0], payload[1] = val, float(self.to_global(lidx))
            payload[2:] = X_local[lidx, :d].cpu().numpy()  # one D2H of the winner's row
        g = self._all_gather(self._payload_to_backend(payload, X_local.device)).cpu().numpy()
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
        g = self._all_gather(self._payload_to_backend(buf.reshape(-1), device)).cpu().numpy().reshape(-1, 2)
        return self.merge_topk(g, k)

    def global_topk_rows(self, vals: np.ndarray, lidx: np.ndarray, k: int, X_local, d: int):
        """Global top-k WITH the comp-rep rows of its members: every rank contributes its local top-k as (score, global index,
        row [d]) - one all-gather of k (2 + d) doubles per rank - and every rank returns the same (values [m], global indices [m],
        rows [m, d]), m <= k, ties to the lower global index.  Used once per greedy batch, for the speculative columns."""
        buf = np.zeros((k, 2 + d))
        buf[:, 0], buf[:, 1] = -math.inf, -1.0
        m = min(k, len(vals))
        if m:
            import torch

            li = np.asarray(lidx[:m], dtype=np.int64)
            ok = li >= 0
            buf[:m, 0] = np.where(ok, vals[:m], -math.inf)
            buf[:m, 1] = np.where(ok, li + self.start, -1)
            if ok.any():
                rows = X_local[torch.as_tensor(li[ok], device=X_local.device), :d].cpu().numpy()  # one D2H of <= k rows
                buf[:m][ok, 2:] = rows
        g = self._all_gather(self._payload_to_backend(buf.reshape(-1), X_local.device)).cpu().numpy().reshape(-1, 2 + d)
        g = g[g[:, 1] >= 0]
        order = np.lexsort((g[:, 1], -g[:, 0]))[:k]
        return g[order, 0], g[order, 1].astype(np.int64), g[order, 2:].copy()

    @staticmethod
    def merge_topk(gathered: np.ndarray, k: int):
        """Rows (score, global index) of all ranks -> the k best, ties to the lower global index."""
        g = gathered[gathered[:, 1] >= 0]
        order = np.lexsort((g[:, 1], -g[:, 0]))[:k]
        return g[order, 0], g[order, 1].astype(np.int64)
