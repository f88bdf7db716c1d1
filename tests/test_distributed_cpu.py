"""N>1 path on CPU: world_size-2 gloo, one exchange per selection step."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from baybe_amd.distributed import RowShard, shard_bounds


def test_shard_bounds_cover_everything_contiguously():
    for N in (0, 1, 7, 8, 1000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pick_winner_tie_breaks_on_lowest_global_index():
    g = np.array([[1.0, 40, 0.0], [1.0, 7, 0.0], [0.5, 1, 0.0], [-np.inf, -1, 0.0]])
    assert RowShard.pick_winner(g) == 1
    assert RowShard.pick_winner(np.array([[-np.inf, -1, 0.0]])) == -1
    assert RowShard.pick_winner(np.array([[-np.inf, 3, 0.0], [-np.inf, -1, 0.0]])) == 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, d, q, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        X = rng.integers(0, 4, size=(N, d)).astype(np.float64)
        scores = rng.integers(0, 6, size=N).astype(np.float64)  # many exact ties
        sh = RowShard(N, rank, world)
        Xl = torch.from_numpy(X[sh.start:sh.stop])
        sl = scores[sh.start:sh.stop].copy()
        picks = []
        for _ in range(q):  # greedy: global argmax, drop the row on its owner
            if len(sl):
                li = int(np.argmax(sl))
                val = float(sl[li])
            else:
                li, val = -1, -np.inf
            v, gi, row = sh.global_argmax(val, li, Xl)
            assert np.array_equal(row, X[gi])
            if sh.owns(gi):
                sl[sh.to_local(gi)] = -np.inf
            picks.append(gi)
        k = 5
        order = np.argsort(-scores[sh.start:sh.stop], kind="stable")[:k]
        tv, ti = sh.global_topk(scores[sh.start:sh.stop][order], order, k)
        if rank == 0:
            out.put((picks, tv.tolist(), ti.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [101, 3])
def test_global_selection_equals_single_process(N):
    world, d, q = 2, 3, min(4, N)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, d, q, out)) for r in range(world)]
    for p in procs:
        p.start()
    picks, tv, ti = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    rng.integers(0, 4, size=(N, d))
    scores = rng.integers(0, 6, size=N).astype(np.float64)
    ref, s = [], scores.copy()
    for _ in range(q):
        i = int(np.argmax(s))
        ref.append(i)
        s[i] = -np.inf
    assert picks == ref
    k = min(5, N)
    order = np.argsort(-scores, kind="stable")[:k]
    assert ti[:k] == order.tolist() and np.allclose(tv[:k], scores[order])
