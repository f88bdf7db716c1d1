"""Row-sharded selection end to end on the device: two ranks (both on cuda:0, gloo standing in for
RCCL) each score half of the candidates and agree, through one all-gather per step, on exactly the
batch a single process selects.  Also: a campaign loop in the style of the
reference's run_iterations fixture (tests/conftest.py:945-976)."""

import os
import socket

import numpy as np
import pandas as pd
import pytest

from _problems import fixed_theta, make_problem

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(gp, d, Xt, y):
    from baybe_amd import gp_spec

    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    ls, nz, _ = fixed_theta(d)
    gp.set_model(spec, Xt, y)
    gp.factorize(gp_spec.GPParams(np.full(d, ls), nz, 0.0))


def _worker(rank, world, port, N, d, n, q, out):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from baybe_amd import engine
        from baybe_amd.distributed import RowShard

        X, Xt, y = make_problem(N, d, n, seed=12)
        gp = engine.HipGP(0)
        _model(gp, d, Xt, y)
        sh = RowShard(N, rank, world)
        Xl = torch.from_numpy(X[sh.start:sh.stop]).cuda()
        res = gp.greedy_qlogei(Xl, q, seed=21, X_pending=X[:1], shard=sh)
        m, v = gp.posterior(Xl)
        z = engine.sobol_normal_base_samples(512, 1, 21)[:, 0]
        s = gp.qlogei(m, v, z, gp.best_f())
        k = 6
        vals, idx = gp.topk(s, min(k, len(Xl)))
        tv, ti = sh.global_topk(vals, idx, k, device=Xl.device)
        if rank == 0:
            out.put((res.indices, res.values, tv.tolist(), ti.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_greedy_equals_single_process():
    import torch
    import torch.multiprocessing as mp

    from baybe_amd import engine

    N, d, n, q, world = 20001, 6, 80, 3, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, d, n, q, out)) for r in range(world)]
    for p in procs:
        p.start()
    idx, vals, tv, ti = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    X, Xt, y = make_problem(N, d, n, seed=12)
    gp = engine.HipGP(0)
    _model(gp, d, Xt, y)
    ref = gp.greedy_qlogei(X, q, seed=21, X_pending=X[:1])
    assert idx == ref.indices and np.allclose(vals, ref.values, rtol=0, atol=1e-12)
    m, v = gp.posterior(X)
    s = gp.qlogei(m, v, engine.sobol_normal_base_samples(512, 1, 21)[:, 0], gp.best_f())
    rv, ri = gp.topk(s, 6)
    assert ti == ri.tolist() and np.allclose(tv, rv, rtol=0, atol=1e-12)


def _nehvi_setup(N, d, n):
    from baybe_amd import engine, gp_spec

    X, Xt, y = make_problem(N, d, n, seed=5)
    Y = np.stack([y, -((Xt - 0.7) ** 2).sum(1)], 1)
    engines = []
    for o in range(2):
        g = engine.HipGP(0)
        _model(g, d, Xt, Y[:, o])
        engines.append(g)
    return X, Xt, Y, engines


def _nehvi_worker(rank, world, port, N, d, n, out):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from baybe_amd.distributed import RowShard
        from baybe_amd.nehvi import HipNEHVI, compute_ref_point

        X, Xt, Y, engines = _nehvi_setup(N, d, n)
        sh = RowShard(N, rank, world)
        hv = HipNEHVI(engines, np.ones(2), Xt, compute_ref_point(Y), n_mc_samples=32)
        res = hv.greedy(torch.from_numpy(X[sh.start:sh.stop]).cuda(), 2, seed=4, prune_seed=6, shard=sh)
        if rank == 0:
            out.put((res.indices, res.values))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_nehvi_equals_single_process():
    import torch
    import torch.multiprocessing as mp

    from baybe_amd.nehvi import HipNEHVI, compute_ref_point

    N, d, n, world = 5001, 4, 30, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nehvi_worker, args=(r, world, port, N, d, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    idx, vals = out.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    X, Xt, Y, engines = _nehvi_setup(N, d, n)
    hv = HipNEHVI(engines, np.ones(2), Xt, compute_ref_point(Y), n_mc_samples=32)
    ref = hv.greedy(torch.from_numpy(X).cuda(), 2, seed=4, prune_seed=6)
    assert idx == ref.indices and np.allclose(vals, ref.values, rtol=0, atol=1e-12)


def test_campaign_iterations_loop():
    """recommend -> measure -> add, several rounds; nothing is recommended twice, the candidate cache
    follows the shrinking candidate set, batch sizes vary."""
    import torch

    from _baybe_shim import Campaign, NumericalDiscreteParameter, NumericalTarget, SearchSpace, SingleTargetObjective
    from baybe_amd.recommenders import HipBotorchRecommender

    rng = np.random.default_rng(0)
    vals = np.arange(9) / 8.0
    space = SearchSpace.from_product([NumericalDiscreteParameter(f"x{i}", vals) for i in range(3)])
    exp = space.discrete.exp_rep

    def f(df):
        Xv = df[["x0", "x1", "x2"]].to_numpy(float)
        return -((Xv - 0.3) ** 2).sum(1) + 0.02 * rng.standard_normal(len(Xv))

    camp = Campaign(space, SingleTargetObjective(NumericalTarget("y")), HipBotorchRecommender())
    start = exp.iloc[rng.choice(len(exp), 6, replace=False)].copy()
    start["y"] = f(start)
    camp.add_measurements(start)
    seen = set(start.index)
    torch.manual_seed(3)
    best = [start["y"].max()]
    for it, bs in enumerate((2, 3, 1, 2)):
        rec = camp.recommend(bs)
        assert len(rec) == bs and not (set(rec.index) & seen)
        seen |= set(rec.index)
        rec = rec.copy()
        rec["y"] = f(rec)
        camp.add_measurements(rec)
        best.append(max(best[-1], rec["y"].max()))
    assert best[-1] >= best[0]
    assert len(camp.measurements) == 6 + 2 + 3 + 1 + 2


def test_bench_starts_its_own_ranks_and_reports_one_json_line():
    """``python bench.py --gpus 2`` without torchrun (the driver could not start it): bench.py spawns one
    process per rank itself; BENCH_SINGLE_DEVICE=1 puts both ranks on cuda:0 with gloo standing in for RCCL."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["BENCH_SINGLE_DEVICE"] = "1"
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "150000",
           "--cpu-budget", "0", "--greedy", "3"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    # more than one rank: --rows is the GLOBAL grid, split over the ranks (BASELINE configs[2] is one row-sharded grid)
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "strong" and rec["unit"] == "candidates/s"
    assert rec["config"]["global_rows"] == 150000 and rec["value"] > 0 and rec["dtype"] == "f64"
    assert rec["roofline"]["bound"] == "mfma" and 0 < rec["roofline"]["frac"] < 1
    assert len(set(rec["extra"]["greedy_q3_indices"])) == 3
    assert rec["extra"]["roofline_pending_kernels"]["launches"] == 2


def test_bench_under_torchrun_as_the_driver_launches_it():
    """The driver's command for N > 1: ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`` (ranks from RANK / LOCAL_RANK / WORLD_SIZE) - here with both ranks on
    the one device (BENCH_SINGLE_DEVICE=1, gloo).  One JSON line from rank 0: the strong-scaled configs[2] step (a 1e6-row grid split
    over the ranks) whose merged top indices equal the single-process ones, and in ``extra.weak_cfg3`` the weak-scaled figure."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["BENCH_SINGLE_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--cpu-budget", "0", "--greedy", "0"]  # (default rows: rank 0's grid and training rows are the single-process ones)
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    # the headline for N > 1 IS BASELINE configs[2]: the 1e6-row grid row-sharded; the weak figure sits in extra
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "strong" and rec["config"]["global_rows"] == 1_000_000
    assert rec["config"]["workload"].startswith("1000000 x 20 discrete grid row-sharded over 2 GPUs (500000 rows")
    weak = rec["extra"]["weak_cfg3"]
    assert weak["global_rows"] == 2000000 and weak["rows_per_rank"] == 1000000 and weak["scaling"] == "weak" and weak["value"] > 0
    strong = {"top_indices": rec["extra"]["top_indices"]}
    single = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-budget", "0", "--greedy", "0",
                             "--e2e", "0"], cwd=root, env={k: v for k, v in env.items() if k != "BENCH_SINGLE_DEVICE"},
                            capture_output=True, text=True, timeout=900)
    assert single.returncode == 0, single.stderr[-2000:]
    srec = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][-1])
    assert strong["top_indices"] == srec["extra"]["top_indices"]


def test_library_rccl_entry_points_on_a_single_rank_communicator():
    """``bbh_comm_init`` / ``bbh_allgather_topk`` / ``bbh_allgather_argmax`` (include/baybe_hip.h) with world = 1 - the
    GPU box has one device: communicator set-up, device-side payload, ncclAllGather, read-back and merge all run; the
    result must be the local selection shifted by the row offset."""
    import torch

    from baybe_amd import engine

    N, d, n = 5000, 6, 40
    X, Xt, y = make_problem(N, d, n, seed=4)
    gp = engine.HipGP(0)
    _model(gp, d, Xt, y)
    Xd = torch.from_numpy(X).cuda()
    m, v = gp.posterior(Xd)
    s = gp.qlogei(m, v, engine.sobol_normal_base_samples(512, 1, 3)[:, 0], gp.best_f())
    gp.comm_init(0, 1, gp.comm_unique_id())
    vals, idx = gp.topk(s, 7)
    gv, gi = gp.allgather_topk(s, 1000, 7)
    assert np.array_equal(gi, idx + 1000) and np.array_equal(gv, vals)
    val, i = gp.argmax(s)
    gval, gidx, row = gp.allgather_argmax(s, 1000, Xd)
    assert gidx == i + 1000 and gval == val and np.array_equal(row, X[i])
    ev, ei = gp.allgather_topk(s[:3], 0, 5)  # fewer rows than k: padded with (-inf, -1)
    assert ei[3:].tolist() == [-1, -1] and np.isneginf(ev[3:]).all() and set(ei[:3]) == {0, 1, 2}
    gp.close()


@pytest.mark.parametrize("collective", ["torch", "rccl"])
def test_bench_sharded_path_over_rccl_with_one_rank(collective):
    """The transport the 8-GPU run uses, on the one GPU there is: ``init_process_group("nccl")`` with world size 1 and
    the sharded code path forced (BENCH_FORCE_COLLECTIVE=1) - agreement broadcasts, the per-step all-gather on device
    tensors (torch.distributed, or the library's own communicator with BBH_COLLECTIVE=rccl), barrier, max-reduce of the
    time.  The selection must equal the single-process one."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_SINGLE_DEVICE")}
    cmd = [sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--rows", "120000", "--cpu-budget", "0",
           "--greedy", "3"]
    recs = []
    for env in (dict(base), dict(base, BENCH_FORCE_COLLECTIVE="1", BBH_COLLECTIVE=collective, MASTER_PORT=str(_free_port()))):
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        recs.append(json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]))
    plain, sharded = recs
    assert plain["config"]["collective"] == "none"
    assert sharded["config"]["collective"] == ("rccl (library)" if collective == "rccl" else "torch.distributed nccl")
    assert sharded["extra"]["greedy_q3_indices"] == plain["extra"]["greedy_q3_indices"] and sharded["n_gpus"] == 1


def test_bench_cfg5_shards_like_the_default_configuration():
    """BASELINE configs[4] (qLogNEHVI, 3 targets) is an 8-GPU configuration: ``bench.py --config cfg5`` through the sharded code path
    (one-rank RCCL communicator, ``BENCH_FORCE_COLLECTIVE=1``) - replicated set-up, rank-local scores, one all-gather of the per-shard
    top-k - must select the rows of the single-process run."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_SINGLE_DEVICE")}
    cmd = [sys.executable, str(root / "bench.py"), "--config", "cfg5", "--steps", "2", "--warmup", "1", "--cpu-budget", "0"]
    recs = []
    for env in (dict(base), dict(base, BENCH_FORCE_COLLECTIVE="1", BBH_COLLECTIVE="rccl", MASTER_PORT=str(_free_port()))):
        out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        recs.append(json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]))
    plain, sharded = recs
    assert plain["config"]["collective"] == "none" and sharded["config"]["collective"] == "rccl (library)"
    assert sharded["extra"]["top_indices"] == plain["extra"]["top_indices"] and len(set(plain["extra"]["top_indices"])) == 8


def test_bench_two_gpus_over_the_library_communicator():
    """``bench.py --gpus 2`` on two real devices: one rank per GPU, the per-step exchange through the library's own RCCL
    communicator (the default for more than one rank: payload built on the device, one ncclAllGather over xGMI, one
    read-back).  Skipped where fewer than two devices are visible (the gpurun box has one); the driver's 8-GPU run takes
    this path."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible HIP devices")
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "BENCH_SINGLE_DEVICE",
                                                             "BBH_COLLECTIVE")}
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "200000",
           "--cpu-budget", "0", "--greedy", "3"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["collective"] == "rccl (library)" and rec["config"]["global_rows"] == 200000
    single = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "1", "--warmup", "1", "--rows", "200000", "--cpu-budget",
                             "0", "--greedy", "0", "--strong"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert single.returncode == 0 and rec["value"] > 0
