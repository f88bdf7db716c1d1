#!/usr/bin/env python
"""bench.py — candidates scored per second on the GP recommend() hot path (MI355X).

A "step" is one full scoring pass of the hot path over the rank's resident candidate shard:
fused posterior (Normalize -> K(X*,X) -> mean / variance -> un-Standardize) -> qLogEI (S = 512
Sobol base samples) -> local top-k -> (N > 1) one all-gather of the per-shard top-k -> global
top-k on the host.  Inputs are resident in HBM when the timed region starts; the GP is
factorised once before timing (fit is reported separately in ``extra``).

Workload at every N: the configuration BASELINE.json's metric is quoted on — a 1e6-row grid per
GPU (weak scaling: the global grid has N * 1e6 rows), d = 20, n_train = 512, Matérn-5/2 ARD,
fixed-theta mode (prior modes of the BAYBE preset), qLogEI, fp64.

  python bench.py                       # 1 GPU, the metric's own configuration (BASELINE configs[2], one GPU's shard)
  python bench.py --config cfg2|cfg4|cfg5   # the other BASELINE configurations, each with its own roofline record (cfg2 / cfg5 also
                                        #   with --gpus N: row shards like the default):
                                        #   cfg2 = configs[1] 1e5 x 15, n = 256;  cfg4 = configs[3] ICM over 4 tasks, 1e5 x (15 + task),
                                        #   n = 1024;  cfg5 = configs[4] qLogNEHVI, 3 targets, 1e5 x 15, n = 256, S = 512 (one GPU)
  python bench.py --gpus N              # spawns one rank per GPU itself (torch.multiprocessing, 127.0.0.1 rendezvous)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W      # or under torchrun: RANK / WORLD_SIZE from the env
"""

from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet: FP64 matrix = FP64 vector = 78.6 TFLOP/s
TOPK = 8


def synth_problem(rows: int, d: int, n: int, rank: int):
    """Deterministic synthetic workload: a grid with 11 levels per dimension in [0,1] per rank; the
    training inputs are n rows of rank 0's candidate grid (``default_rng(1).choice``), identical on every rank."""
    def grid(r):
        return np.random.default_rng(1000 + r).integers(0, 11, size=(rows, d)) / 10.0

    X0 = grid(0)
    Xt = X0[np.random.default_rng(1).choice(rows, n, replace=False)]
    y = -((Xt - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * Xt[:, 0]) + 0.05 * np.random.default_rng(2).standard_normal(n)
    return (X0 if rank == 0 else grid(rank)), Xt, y


def synth_tl_problem(rows: int, dnum: int, per_task: int, T: int):
    """configs[3]: T tasks, task column last and INT-coded, ``per_task`` measurements each with
    y_t = (1 - 0.1 t) y + 0.2 t; candidates = ``rows`` grid rows of the active task 0."""
    rng = np.random.default_rng(0)
    X = np.hstack([rng.integers(0, 11, size=(rows, dnum)) / 10.0, np.zeros((rows, 1))])
    parts, ys = [], []
    for t in range(T):
        xt = np.random.default_rng(10 + t).integers(0, 11, size=(per_task, dnum)) / 10.0
        y = -((xt - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * xt[:, 0])
        ys.append((1 - 0.1 * t) * y + 0.2 * t + 0.05 * np.random.default_rng(20 + t).standard_normal(per_task))
        parts.append(np.hstack([xt, np.full((per_task, 1), float(t))]))
    return X, np.vstack(parts), np.concatenate(ys)


def synth_pareto_targets(Xt: np.ndarray):
    """configs[4]: f1 = -|x - 0.25|^2, f2 = -|x - 0.75|^2, f3 = -sum |x_j - 0.5|, + 0.05 noise each."""
    f = [-((Xt - 0.25) ** 2).sum(1), -((Xt - 0.75) ** 2).sum(1), -np.abs(Xt - 0.5).sum(1)]
    return [fo + 0.05 * np.random.default_rng(30 + o).standard_normal(len(Xt)) for o, fo in enumerate(f)]


def _latest_profile(cfg_name, what):
    """Newest committed ``profiles/rNN_<cfg>_<what>.json`` (the round number decides), as (path name, parsed content) or (None, None)."""
    import re

    best = None
    for fn in (ROOT / "profiles").glob(f"r*_{cfg_name}_{what}.json"):
        m = re.fullmatch(rf"r(\d+)_{re.escape(cfg_name)}_{what}\.json", fn.name)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), fn)
    if best is None:
        return None, None
    try:
        return best[1].name, json.loads(best[1].read_text())
    except Exception:  # noqa: BLE001
        return None, None


def issue_frac_of(cfg_name, kernel_prefix):
    """{kernel: issue_frac} from the newest committed PMC summary of this configuration (profiles/rNN_<cfg>_issue.json), or None."""
    _, ij = _latest_profile(cfg_name, "issue")
    try:
        got = {k: round(v["issue_frac"], 4) for k, v in ij["kernels"].items() if k.startswith(kernel_prefix)}
        return got or None
    except Exception:  # noqa: BLE001
        return None


def traffic_of_kernel(cfg_name, kernel_prefix):
    """(bytes per launch, source file) of the dominant kernel from the newest committed PMC traffic record of this configuration
    (profiles/rNN_<cfg>_traffic.json: rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE in separate --pmc passes, scripts/profile_config.sh)."""
    name, tj = _latest_profile(cfg_name, "traffic")
    try:
        cands = [(k, v) for k, v in tj.items() if isinstance(v, dict) and kernel_prefix in k and "hbm_bytes_per_launch" in v]
        k, v = max(cands, key=lambda kv: kv[1]["hbm_bytes_per_launch"])
        return v["hbm_bytes_per_launch"], f"profiles/{name}: {k}"
    except Exception:  # noqa: BLE001
        return None, None


class _BenchSpace:
    """What ``HipBotorchRecommender.recommend`` reads from ``baybe.searchspace.SearchSpace`` for a purely discrete, all-numerical
    space (attribute names of searchspace/core.py, searchspace/discrete.py): the N x d grid is both representations."""

    class _Discrete:
        n_subsets, parameters = 0, ()

        def __init__(self, frame):
            self.exp_rep = self.comp_rep = frame
            self.mask_keep = np.ones(len(frame), dtype=bool)

        def get_candidates(self):
            return self.exp_rep.loc[self.mask_keep], self.comp_rep.loc[self.mask_keep]

    class _Continuous:
        is_empty = True

    def __init__(self, X, task_idx=None, n_tasks=1):
        import pandas as pd

        cols = [f"x{j}" for j in range(X.shape[1])]
        self.discrete, self.continuous, self.parameters = self._Discrete(pd.DataFrame(X, columns=cols)), self._Continuous(), ()
        self.comp_rep_columns, self.task_idx, self.n_tasks = tuple(cols), task_idx, n_tasks
        hi = np.ones(len(cols))
        if task_idx is not None:
            hi[task_idx] = float(max(n_tasks - 1, 1))
        self.scaling_bounds = pd.DataFrame([np.zeros(len(cols)), hi], index=["min", "max"], columns=cols)

    def transform(self, df, allow_extra=False):
        return df[list(self.comp_rep_columns)]


def time_recommend_e2e(X, Xt, ys, d, batch, task=None, acquisition_function=None, measure=None):
    """``recommend(batch)`` of the plug-in recommender on the bench's own grid and measurements, wall clock in ms: the first call
    (comp rep hashed and uploaded, hyper-parameters fitted), a call with unchanged measurements (resident matrix, cached fit: the
    hot path plus the pandas boundary) and a call after one more measurement (refit).  ``ys``: one target column (array) or several
    (list: ParetoObjective -> replicated surrogate + qLogNEHVI, whose baseline pruning runs in EVERY call because the reference builds
    a new acquisition function per call, acqfs.py:477-484); ``task`` = (task column, number of tasks) for the transfer-learning form;
    ``measure(rows) -> [target columns]``: the synthetic experiment that "measures" a recommended batch (default: the targets' means)."""
    import pandas as pd
    import torch
    from types import SimpleNamespace

    from baybe_amd.recommenders import HipBotorchRecommender

    space = _BenchSpace(X) if task is None else _BenchSpace(X, task[0], task[1])
    cols = list(space.comp_rep_columns)
    meas = pd.DataFrame(Xt, columns=cols)
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    names = ["y"] if len(ys) == 1 else [f"y{o}" for o in range(len(ys))]
    for nm, yo in zip(names, ys):
        meas[nm] = yo
    objective = SimpleNamespace(targets=tuple(SimpleNamespace(name=nm, minimize=False, transformation=None) for nm in names),
                                is_multi_output=len(names) > 1)
    rec = HipBotorchRecommender() if acquisition_function is None else HipBotorchRecommender(acquisition_function=acquisition_function)
    out = {}

    def timed(label, m):
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = rec.recommend(batch, space, objective, m)
        torch.cuda.synchronize()
        out[label] = (time.perf_counter() - t0) * 1e3
        return got

    def measured(got):
        if measure is None:
            return got.assign(**{nm: float(np.mean(yo)) for nm, yo in zip(names, ys)})
        vals = measure(got[cols].to_numpy(dtype=np.float64))
        return got.assign(**{nm: np.asarray(v, dtype=np.float64) for nm, v in zip(names, vals)})

    timed("first_call_upload_and_fit", meas)
    timed("unchanged_measurements", meas)
    got = timed("unchanged_measurements_again", meas)
    more = pd.concat([meas, measured(got)], ignore_index=True)
    timed("after_new_measurements_refit", more)
    got = timed("unchanged_measurements_after_refit", more)
    more2 = pd.concat([more, measured(got)], ignore_index=True)
    timed("after_new_measurements_refit_again", more2)
    out["batch_size"] = batch
    out["rows"], out["n_train"], out["targets"] = int(X.shape[0]), int(len(Xt)), len(names)
    return out


CONFIGS = {  # BASELINE.json configs -> (rows per GPU, d, n_train); the default is the configuration the metric is quoted on
    "cfg3": (1_000_000, 20, 512),
    "cfg2": (100_000, 15, 256),
    "cfg4": (100_000, 15, 1024),
    "cfg5": (100_000, 15, 256),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3",
                    help="BASELINE.json configuration (default: the one the metric is quoted on)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=None, help="candidate rows per GPU (default: the configuration's)")
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--n-train", type=int, default=None)
    ap.add_argument("--mc-samples", type=int, default=512)
    ap.add_argument("--strong", action="store_true", help="fixed global grid of --rows rows, split over the GPUs (the default for more "
                                                          "than one GPU: BASELINE configs[2] is ONE 1e6-row grid row-sharded)")
    ap.add_argument("--weak", action="store_true", help="--rows rows PER GPU (weak scaling; for more than one GPU the headline is "
                                                        "otherwise the strong-scaled grid and the weak figure sits in extra)")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--fit", nargs="?", type=int, const=1, default=-1,
                    help="time a device hyper-parameter fit outside the timed region (extra.fit_ms); default: on for single-task "
                         "configurations on one GPU (second of two fits: the first pays the allocations), --fit 0 switches it off")
    ap.add_argument("--greedy", type=int, default=5, help="also time a greedy batch of this size (extra; 0 = skip)")
    ap.add_argument("--e2e", type=int, default=-1, help="time recommend() through the plug-in surface (extra.recommend_e2e_ms); default: on "
                                                        "for the single-target configurations on one GPU, --e2e 0 switches it off")
    args = ap.parse_args(argv)
    rows, d, n = CONFIGS[args.config]
    args.rows = rows if args.rows is None else args.rows
    args.d = d if args.d is None else args.d
    args.n_train = n if args.n_train is None else args.n_train
    return args


def _spawned_rank(local_rank: int, world: int, port: int, argv):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    run(parse_args(argv))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: become the launcher - one process per GPU, rendezvous on 127.0.0.1
        import socket

        import torch.multiprocessing as mp

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        mp.spawn(_spawned_rank, args=(args.gpus, port, sys.argv[1:]), nprocs=args.gpus, join=True)
        return
    run(args)


def run(args):
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # BENCH_SINGLE_DEVICE=1: dry run of the N > 1 code path on ONE GPU (all ranks on cuda:0, gloo
    # instead of RCCL) — used to test the sharded path where only one device is available.
    single_dev = os.environ.get("BENCH_SINGLE_DEVICE", "0") == "1"
    # BENCH_FORCE_COLLECTIVE=1: initialise the process group and run the sharded code path (RowShard, all-gather per
    # step, agreement broadcasts) even with one rank - the only way to drive the RCCL ("nccl") transport on a 1-GPU box
    dist_on = world > 1 or os.environ.get("BENCH_FORCE_COLLECTIVE", "0") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if single_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from baybe_amd import engine, gp_spec
    from baybe_amd.distributed import RowShard, shard_bounds

    d, n, S = args.d, args.n_train, args.mc_samples
    if world > 1 and not args.weak:
        # BASELINE configs[2] / configs[4] and north_star's ">= 6 x further at 8 GPUs" are ONE grid row-sharded over the GPUs: for N > 1
        # the headline is that strong-scaled configuration; the weak-scaled figure goes to extra.weak_<cfg>
        args.strong = True
    if args.strong:
        a, b = shard_bounds(args.rows, rank, world)
        rows_local, total_rows = b - a, args.rows
    else:
        rows_local, total_rows = args.rows, args.rows * world
    cfg = args.config
    if cfg == "cfg4" and dist_on:
        raise SystemExit(f"--config {cfg} is a single-GPU record (its multi-GPU form shards exactly like the default configuration)")
    extra = {}
    nehvi = None
    fit_start_raw = None
    if cfg == "cfg4":  # BASELINE configs[3]: transfer learning, ICM over 4 tasks, LOO criterion at fit time
        T = 4
        X, Xt, y = synth_tl_problem(rows_local, d, n // T, T)
        spec = gp_spec.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=T)
        params = gp_spec.initial_params(spec)  # fixed-theta mode: prior modes, deterministic task factors
    elif dist_on and rank != 0 and not args.strong:
        # the training rows are rows of rank 0's grid: rank 0 sends them instead of every rank regenerating that grid
        X = np.random.default_rng(1000 + rank).integers(0, 11, size=(rows_local, d)) / 10.0
        Xt = y = None
    else:
        if args.strong:  # one global grid: every rank derives the same measurements from it and keeps its own rows
            Xg, Xt, y = synth_problem(args.rows, d, n, 0)
            X = np.ascontiguousarray(Xg[a:b])
            del Xg
        else:
            X, Xt, y = synth_problem(rows_local, d, n, rank)
    if dist_on and not args.strong and cfg != "cfg4":
        box = [(Xt, y)]
        dist.broadcast_object_list(box, src=0, device=torch.device("cpu") if single_dev else torch.device("cuda", local_rank))
        Xt, y = box[0]
    ls = math.exp(math.sqrt(2.0) - 3.0) * math.sqrt(d)
    if cfg != "cfg4":
        spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        params = gp_spec.GPParams(np.full(d, ls), math.exp(-5.0), 0.0)

    if cfg == "cfg5":  # BASELINE configs[4]: ParetoObjective -> qLogNEHVI over 3 independent GPs (one GPU's share)
        from baybe_amd.nehvi import HipNEHVI, compute_ref_point

        ys = synth_pareto_targets(Xt)
        engines = []
        for yo in ys:
            g = engine.HipGP(local_rank)
            g.set_model(spec, Xt, yo)
            engines.append(g)
        if args.fit == 1 or (args.fit < 0 and world == 1):  # the three targets' fits, one host thread and one stream each (CompositeSurrogate)
            for _ in range(2):  # (what HipCompositeImpl.fit does: engine.fit_side_by_side - a private stream per fit)
                t0 = time.perf_counter()
                fis = engine.fit_side_by_side([g.fit for g in engines], device=local_rank)
                extra["fit_ms"] = (time.perf_counter() - t0) * 1e3
            extra["fit_nfev"] = [f.nfev for f in fis]
            t0 = time.perf_counter()
            for g in engines:
                g.fit()
            extra["fit_ms_sequential"] = (time.perf_counter() - t0) * 1e3
            fit_start_raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        for g in engines:
            g.factorize(params)
        ref = compute_ref_point(np.stack(ys, axis=1))  # acquisition/_builder.py:301-317, acqfs.py:406-426
        nehvi = HipNEHVI(engines, np.ones(len(ys)), Xt, ref, n_mc_samples=S, prune_baseline=True, device=local_rank)
        torch.manual_seed(0)
        t0 = time.perf_counter()
        nehvi.prepare(1234, prune_seed=4321)  # first call of an acquisition function: baseline pruning (2048 samples, once) + one step's set-up
        torch.cuda.synchronize()
        extra["nehvi_first_setup_ms"] = (time.perf_counter() - t0) * 1e3  # (also pays the first allocations of the extended models)
        extra["nehvi_prune_ms"] = nehvi.last_setup_ms.get("prune")
        extra["nehvi_prune_parts_ms"] = {k: round(v, 3) for k, v in getattr(nehvi, "last_prune_ms", {}).items()}
        ts_setup = []
        for _ in range(5):  # the per-selection-step set-up: extended models, baseline samples, box decompositions (device)
            t0 = time.perf_counter()
            nehvi.prepare(1234, prune_seed=4321)
            torch.cuda.synchronize()
            ts_setup.append((time.perf_counter() - t0) * 1e3)
        extra["nehvi_setup_ms"] = float(np.median(ts_setup))
        # pruning as every later recommend() pays it (the acquisition function is rebuilt per call, acqfs.py:477-484): a second object's
        # first prepare - without the process's one-off costs (self-check of the native generator, the engine's direction numbers)
        nehvi2 = HipNEHVI(engines, np.ones(len(ys)), Xt, ref, n_mc_samples=S, prune_baseline=True, device=local_rank)
        torch.cuda.synchronize()
        nehvi2.prepare(1234, prune_seed=4321)
        torch.cuda.synchronize()
        extra["nehvi_prune_ms_steady"] = nehvi2.last_setup_ms.get("prune")
        extra["nehvi_prune_parts_ms_steady"] = {k: round(v, 3) for k, v in getattr(nehvi2, "last_prune_ms", {}).items()}
        del nehvi2
        extra["nehvi_setup_parts_ms"] = {k: round(v, 3) for k, v in nehvi.last_setup_ms.items()}
        extra["nehvi_baseline_points"] = int(len(nehvi.X_b_current))
        extra["nehvi_cells_per_sample"] = float(nehvi.n_cells) / S
        gp = nehvi.outputs[0].ext
        timed_engines = [o.ext for o in nehvi.outputs]
        best_f = z = None
    else:
        gp = engine.HipGP(local_rank)
        gp.set_model(spec, Xt, y)
        if args.fit == 1 or (args.fit < 0 and world == 1):
            for _ in range(2 if spec.n_tasks == 1 else 1):  # (second of two fits: the first pays the allocations; the ICM fit is ~1000 evaluations)
                t0 = time.perf_counter()
                fi = gp.fit()
                extra["fit_ms"] = (time.perf_counter() - t0) * 1e3
            extra["fit_nfev"] = fi.nfev
            extra["fit_ms_per_evaluation"] = extra["fit_ms"] / max(fi.nfev, 1)
            fit_start_raw = gp_spec.pack_raw(spec, gp_spec.initial_params(spec))
        t0 = time.perf_counter()
        gp.factorize(params)
        torch.cuda.synchronize()
        extra["factorize_ms"] = (time.perf_counter() - t0) * 1e3
        best_f = gp.best_f()
        z = engine.sobol_normal_base_samples(S, 1, 1234)[:, 0]
        timed_engines = [gp]
    Xd = torch.from_numpy(X).cuda()
    shard = RowShard(total_rows, rank, world) if dist_on else None
    if shard is not None and not args.strong:
        shard.start, shard.stop = rank * rows_local, (rank + 1) * rows_local

    # The per-step exchange.  With more than one rank the library's own communicator is the default (bbh_allgather_topk: payload
    # built on the device from the device-side top-k, one ncclAllGather over xGMI on the handle's stream, one read-back = one
    # synchronisation per step); BBH_COLLECTIVE=torch selects torch.distributed (host-staged payload, three synchronisations),
    # which is also the fallback when the communicator cannot be set up on every rank.
    want = os.environ.get("BBH_COLLECTIVE", "rccl" if world > 1 else "torch")
    use_rccl = False
    if shard is not None and want == "rccl" and not single_dev:
        ok = 1
        try:
            shard.bind_rccl(gp)
        except Exception as ex:  # noqa: BLE001
            ok = 0
            extra["rccl_bind_error"] = str(ex)[:200]
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks or none
        use_rccl = bool(flag.item())
        if not use_rccl:
            shard.use_rccl = False  # (a rank whose own set-up succeeded must not take the library path alone)

    def make_step(Xs, row_start):
        """One selection step over the resident rows ``Xs`` (global rows ``row_start ...``): three launches - posterior, qLogEI with
        sample slices (which leaves the chunk keys), selection - whose k results land in host-mapped memory; output buffers are the
        step loop's own."""
        bufs = [torch.empty(Xs.shape[0], dtype=torch.float64, device=Xs.device) for _ in range(3)]

        def step():
            # posterior and qLogEI as two kernels: measured 1.5 % faster than the single fused posterior+qLogEI kernel
            # (scripts/gpu_ab_fused_acq.py: 5.64 vs 5.73 ms/step): the epilogue's VALU work costs the shared fp64 pipe the same
            # either way, but a separate launch runs it at full occupancy and leaves the fused kernel's LDS to the kernel-value cache
            if nehvi is not None:  # extended-model variance pass + S conditional means per target, then the cell kernel
                scores = nehvi.score(Xs)
                if shard is None:
                    return gp.topk(scores, TOPK)
                if use_rccl:  # (BASELINE configs[4] is an 8-GPU configuration: row shards, the same all-gather of per-shard top-k)
                    return gp.allgather_topk(scores, row_start, TOPK)
                vals, idx = gp.topk(scores, TOPK)
                return shard.global_topk(vals, idx, TOPK, device=Xs.device)
            mean, var = gp.posterior(Xs, out=(bufs[0], bufs[1]))
            if shard is None:
                _, vals, idx = gp.qlogei_topk(mean, var, z, best_f, 1.0, TOPK, scores=bufs[2])
                return vals, idx
            scores = gp.qlogei(mean, var, z, best_f, 1.0)
            if use_rccl:
                return gp.allgather_topk(scores, row_start, TOPK)
            vals, idx = gp.topk(scores, TOPK)
            return shard.global_topk(vals, idx, TOPK, device=Xs.device)

        return step

    step = make_step(Xd, shard.start if shard is not None else 0)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The device comes out of idle at ramping clocks: the posterior kernel of the first launches after set-up runs 6.0, 5.5,
    # 5.2, 5.0, 4.85, 4.76 ms before it settles at 4.73 (rocprofv3 trace of this command, profiles/r02_bench_kernel_stats.csv).
    # A fixed number of untimed passes before the W warm-up steps takes the ramp out of the timed region whatever W is.
    # Round 4: at least 8 passes AND at least 100 ms of them - on a 125 000-row shard eight passes are 6 ms and the ramp reached
    # through the whole timed region (rocprofv3 trace of that run: the posterior kernel fell from 0.69 to 0.64 ms over the 20 timed
    # steps and is 0.597 ms in back-to-back launches, profiles/r04_cfg3_125k_bench_posterior_launches.csv, r04_ramp_probe.log).
    def pre_warm(fn, first):  # the same number of passes on every rank (a pass ends in a collective)
        t_pw = time.perf_counter()
        for _ in range(first):
            fn()
        el = time.perf_counter() - t_pw
        more = int(min(2000, math.ceil(max(0.0, 0.1 - el) / max(el / first, 1e-6))))
        if dist_on:
            mt = torch.tensor([more], dtype=torch.int64, device="cpu" if single_dev else "cuda")
            dist.all_reduce(mt, op=dist.ReduceOp.MAX)
            more = int(mt.item())
        for _ in range(more):
            fn()
        return first + more

    fence()  # the first collective of the process group: torch's communicator comes up here (100s of ms of idle device with RCCL), not
             # between the warm passes and the timed region
    extra["pre_warm_steps"] = pre_warm(step, 8)
    for _ in range(args.warmup):
        step()
    FAMILIES = ("posterior", "cross", "pending", "columns", "nehvi", "q1", "select")
    DOM = "nehvi" if nehvi is not None else "posterior"  # the family of the dominant kernel (the roofline's)

    def timers(on, fams=None):
        for g in timed_engines:
            g.timing(on, fams)
            for fam in FAMILIES:
                g.timing_read(reset=True, family=fam)

    def read_timers():
        acc = {fam: [0.0, 0] for fam in FAMILIES}
        for g in timed_engines:
            for fam in FAMILIES:
                ms, cnt = g.timing_read(reset=True, family=fam)
                acc[fam][0] += ms
                acc[fam][1] += cnt
        return acc

    # Timed region: HIP events bracket the dominant kernel only.  An event between two back-to-back kernels costs the stream ~5 us
    # (rocprofv3 trace: 10 us gaps posterior -> qLogEI -> selection with every family's events on, none without), which is 3 % of a
    # 0.65 ms step; the other families are timed in K more steps below.
    timers(True, (DOM,))
    fence()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        vals, idx = step()  # ends with the top-k on the host: a step is complete when it returns
        step_ms.append((time.perf_counter() - ts) * 1e3)
    fence()
    dt = time.perf_counter() - t0
    dom_timed = read_timers()[DOM]
    # Instrumented steps: every family's events (and, for qLogNEHVI, the targets' passes one after the other instead of overlapped on
    # their own streams - HIP-event times of one family would otherwise include the others' work).
    seq = nehvi is not None and getattr(nehvi, "concurrent", False) and nehvi.m > 1
    if seq:
        nehvi.concurrent = False
    timers(True)
    step()
    read_timers()
    fence()
    t_ins = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    extra["ms_per_step_instrumented"] = (time.perf_counter() - t_ins) / args.steps * 1e3
    fam_step = read_timers()
    fam_step[DOM] = dom_timed
    timers(False)
    if seq:
        nehvi.concurrent = True
    extra["device_ms_note"] = (f"'{DOM}': HIP events over the timed region (the only events in it); the other families: HIP events over "
                               f"{args.steps} further steps with every family instrumented"
                               + (", the targets' passes in sequence" if seq else ""))
    fused_ms, fused_launches = fam_step["posterior"]
    extra["top_indices"] = [int(i) for i in np.asarray(idx).ravel()[:TOPK]]  # the last step's top-k (global row numbers)
    extra["ms_per_step_median"] = float(np.median(step_ms))
    extra["ms_per_step_mean"] = float(np.mean(step_ms))
    extra["device_ms_per_step"] = {fam: v[0] / args.steps for fam, v in fam_step.items() if v[1]}
    form = timed_engines[0].posterior_kernel_form()
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if single_dev else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # value and ms_per_step: the K timed steps against the fenced wall clock (max over ranks), as the contract asks - value = rows /
    # ms_per_step holds exactly; the median step sits in extra.ms_per_step_median
    ms_per_step = dt / args.steps * 1e3
    value = total_rows * args.steps / dt

    # ---- extra (qLogNEHVI): what a selection step costs WITH its set-up, and a greedy batch ------------------------------------
    if nehvi is not None:
        def selection_step():
            nehvi.prepare(1234, prune_seed=4321)
            return step()

        selection_step()
        fence()
        ts_sel = []
        for _ in range(max(5, args.steps // 2)):
            ts = time.perf_counter()
            selection_step()
            ts_sel.append((time.perf_counter() - ts) * 1e3)
        fence()
        sel = float(np.median(ts_sel))
        if dist_on:
            mt = torch.tensor([sel], dtype=torch.float64, device="cpu" if single_dev else "cuda")
            dist.all_reduce(mt, op=dist.ReduceOp.MAX)
            sel = float(mt.item())
        extra["ms_per_selection_step"] = sel  # set-up (replicated on every rank) + scoring pass + top-k exchange
        extra["selection_steps_per_s_candidates"] = total_rows / (sel * 1e-3)
        if args.greedy > 1:
            nehvi.greedy(Xd, args.greedy, seed=1234, prune_seed=4321, shard=shard)
            fence()
            t0 = time.perf_counter()
            gres = nehvi.greedy(Xd, args.greedy, seed=1234, prune_seed=4321, shard=shard)
            fence()
            extra[f"greedy_q{args.greedy}_ms"] = (time.perf_counter() - t0) * 1e3
            extra[f"greedy_q{args.greedy}_indices"] = gres.indices

    # ---- extra: the OTHER scaling mode of the same configuration next to the headline (N > 1 only) ----------------------------------
    # headline strong (default for N > 1: the configuration's grid split over the ranks) -> extra.weak_cfg3: a full grid per rank;
    # headline weak (--weak) -> extra.strong_cfg3: BASELINE configs[2]'s 1e6 rows split over the ranks
    if dist_on and cfg == "cfg3" and nehvi is None and (world > 1 or not args.strong):
        other_strong = not args.strong
        g_rows = CONFIGS["cfg3"][0]
        if other_strong:
            a_s, b_s = shard_bounds(g_rows, rank, world)
            Xs = torch.from_numpy(np.ascontiguousarray(synth_problem(g_rows, d, n, 0)[0][a_s:b_s])).cuda()
        else:
            a_s, b_s = rank * g_rows, (rank + 1) * g_rows
            Xs = torch.from_numpy(np.ascontiguousarray(synth_problem(g_rows, d, n, rank)[0])).cuda()
        keep = (shard.start, shard.stop, shard.N_total)
        shard.start, shard.stop = a_s, b_s
        shard.N_total = g_rows if other_strong else g_rows * world
        sstep = make_step(Xs, a_s)
        pre_warm(sstep, 3)
        fence()
        ts_ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts = time.perf_counter()
            svals, sidx = sstep()
            ts_ms.append((time.perf_counter() - ts) * 1e3)
        fence()
        sdt = time.perf_counter() - t0
        tt = torch.tensor([sdt, float(np.median(ts_ms))], dtype=torch.float64, device="cpu" if single_dev else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        shard.start, shard.stop, shard.N_total = keep
        tot = g_rows if other_strong else g_rows * world
        extra["strong_cfg3" if other_strong else "weak_cfg3"] = {
            "global_rows": tot, "rows_per_rank": b_s - a_s, "ms_per_step": float(tt[1].item()),
            "value": tot * args.steps / float(tt[0].item()), "unit": "candidates/s", "scaling": "strong" if other_strong else "weak",
            "top_indices": [int(i) for i in sidx]}
        del Xs

    # ---- extra: one greedy batch (optimize_acqf_discrete, q = --greedy) on the same shard, timed per kernel family ----
    pending_roofline = None
    if args.greedy > 1 and nehvi is None:
        gp.timing(True)
        for fam in ("posterior", "cross", "pending"):
            gp.timing_read(reset=True, family=fam)
        gp.greedy_qlogei(Xd, args.greedy, S=S, seed=1234, best_f=best_f, shard=shard)  # warm-up (instantiations, LDS limits)
        for fam in ("posterior", "cross", "pending"):
            gp.timing_read(reset=True, family=fam)
        fence()
        t0 = time.perf_counter()
        gres = gp.greedy_qlogei(Xd, args.greedy, S=S, seed=1234, best_f=best_f, shard=shard)
        fence()
        g_wall = (time.perf_counter() - t0) * 1e3
        fam_ms = {fam: gp.timing_read(reset=True, family=fam) for fam in ("posterior", "cross", "pending")}
        gp.timing(False)
        extra[f"greedy_q{args.greedy}_ms"] = g_wall
        extra[f"greedy_q{args.greedy}_device_ms"] = {k: v[0] for k, v in fam_ms.items()}
        extra[f"greedy_q{args.greedy}_indices"] = gres.indices
        # Roofline of the joint q'-batch kernels (bbh_qlogei_pending_q_kernel<Q>), steps 2..q: bound by the fp64 vector pipe.
        # ALGORITHMIC flops by the maths of BoTorch's qLogEI (not by this kernel's instruction tally), with exp / log priced at 20
        # flops and a division at 8: per candidate, MC sample and step with q' = Q points - affine map y = m + L z: Q (Q + 1);
        # per point t = (y - best_f) / tau (2), fatplus = softplus(t) + 0.1 / (1 + t^2) (exp + log + 11 = 51), log (20), fatmax term
        # (alpha / (alpha + (M - li) / tau))^alpha (14); per sample max / sum / tau log(sum) (25) and the exp + accumulate of the
        # log-mean-exp (22): Q (Q + 1) + 87 Q + 47; plus the joint Cholesky factor, Q^3 / 3 flops per candidate and step.
        flops_p = sum(S * (Q * (Q + 1) + 87 * Q + 47) + Q * Q * Q // 3 for Q in range(2, args.greedy + 1))
        p_ms, p_n = fam_ms["pending"]
        if p_n:
            ach = rows_local * flops_p / (p_ms * 1e-3) / 1e12
            pending_roofline = {
                "kernel": "bbh_qlogei_pending_q_kernel<2..%d>" % args.greedy, "bound": "fp64 vector pipe (VALU issue)",
                # NOT a roofline fraction: the reference formulation's operation count (exp / log at 20 flops, a division at 8) over the
                # fp64 peak; the kernel reaches the same values (1e-10) with series and partly packed single precision, so it can exceed 1
                "throughput_vs_fp64_formulation": ach / FP64_MFMA_PEAK_TFLOPS,
                # the utilisation statement: issued wave-instructions x cycles of their class over SIMD-cycles, from the PMC passes of
                # scripts/profile_config.sh (SQ_INSTS_VALU, GRBM_GUI_ACTIVE) and the kernels' static class mix (profiles/r04_isa_class_mix.json)
                "issue_frac": issue_frac_of(cfg, "bbh_qlogei_pending_q_kernel"),
                "flops_per_candidate": flops_p, "flops_formula": "sum over q' = Q of S (Q (Q + 1) + 87 Q + 47) + Q^3 / 3",
                "launches": p_n, "total_ms": p_ms,
                "hbm_bytes_per_candidate_algorithmic": sum(8 * (2 + (Q - 1)) + 8 for Q in range(2, args.greedy + 1)),
            }

    # ---- extra: end-to-end recommend() through the plug-in surface ("report additionally ... end-to-end recommend()") ----
    if (args.e2e == 1 or (args.e2e < 0 and world == 1)) and not dist_on:
        q_e2e = args.greedy if args.greedy > 0 else 5
        if nehvi is not None:  # configs[4]: three fits + baseline pruning + the greedy batch, per call
            from baybe_amd.acquisition import qLogNoisyExpectedHypervolumeImprovement

            extra["recommend_e2e_ms"] = time_recommend_e2e(X, Xt, ys, d, q_e2e, measure=synth_pareto_targets,
                                                           acquisition_function=qLogNoisyExpectedHypervolumeImprovement(n_mc_samples=S))
        elif cfg == "cfg4":  # configs[3]: the ICM / LOO fit is the call (the new rows are measurements of the active task 0)
            extra["recommend_e2e_ms"] = time_recommend_e2e(
                X, Xt, y, d + 1, q_e2e, task=(d, 4),
                measure=lambda R: [-((R[:, :d] - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * R[:, 0])])
        else:
            extra["recommend_e2e_ms"] = time_recommend_e2e(
                X, Xt, y, d, q_e2e, measure=lambda R: [-((R - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * R[:, 0])])

    # ---- roofline of the dominant kernel, algorithmic flops ----
    # PMC-derived L2<->fabric bytes per launch of the dominant kernel for this exact workload: this round's record if one is committed
    # (profiles/rNN_<cfg>_traffic.json, newest round), collected offline by scripts/profile_config.sh
    prof_cfg = cfg + ("_125k" if (cfg == "cfg3" and rows_local == 125_000) else "")
    standard_rows = rows_local == (125_000 if prof_cfg.endswith("125k") else CONFIGS[cfg][0]) and d == CONFIGS[cfg][1] and n == CONFIGS[cfg][2]

    kernel_names = {"cooperative": "bbh_coop_posterior_kernel", "windowed": "bbh_fused_posterior_kernel",
                    "cooperative-2sweep": "bbh_coop2_posterior_kernel"}
    if nehvi is None:
        flops_per_cand = n * n + 2 * n * d + 16 * n + 16 * S
        avg_ms = fused_ms / max(fused_launches, 1)
        achieved = rows_local * flops_per_cand / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        q1_ms = fam_step["q1"][0] / max(fam_step["q1"][1], 1)
        roofline = {
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            # scripts/mfma_valu_overlap_probe.hip: a pure v_mfma_f64_16x16x4_f64 stream reaches 77.8 TFLOP/s, and
            # fp64 VALU work issued next to it adds its time (one shared DP pipe) - the peak is for MFMA + VALU
            "peak_mfma_stream_microbench": 77.8,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": traffic_of_kernel(prof_cfg, kernel_names.get(form, form))[0] if standard_rows else None,
            "traffic_source": traffic_of_kernel(prof_cfg, kernel_names.get(form, form))[1] if standard_rows else None,
            "traffic_unit": "L2<->fabric bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes); "
                            "algorithmic = %d (read every candidate row once, write mean + variance)" % (rows_local * (8 * d + 16)),
            "kernel": kernel_names.get(form, form),
            "kernel_form": form,
            "avg_launch_ms": avg_ms,
            "launches": fused_launches,
            "flops_per_candidate": flops_per_cand,
            "flops_formula": "n^2 + 2 n d + 16 n + 16 S (SURVEY.md 8d: triangular solve, scaled distances, kernel function + mean, MC)",
            # two stricter readings of the same launch times: the 16 S term is the qLogEI kernel's work, not this kernel's
            "frac_own_flops": rows_local * (flops_per_cand - 16 * S) / (avg_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if avg_ms > 0 else 0.0,
            "frac_both_kernels": rows_local * flops_per_cand / ((avg_ms + q1_ms) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if avg_ms > 0 else 0.0,
            "qlogei_kernel_avg_ms": q1_ms,
            "hbm_GBps_algorithmic": rows_local * (8 * d + 16) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
        }
    else:
        # qLogNEHVI pass = per target (m): variance pass of the model extended by the nb baseline points (n_e = n + nb), S conditional
        # means (mean-only pass against S target columns), then one cell kernel.  Algorithmic fp64 flops per candidate:
        #   variance   m (n_e^2 + 2 n_e d + 16 n_e)                      [bbh_coop_posterior_kernel / bbh_fused_posterior_kernel]
        #   columns    m (2 n_e S + 2 n_e d + 16 n_e)                    [bbh_fused_columns_kernel: K* A, A = K^-1 Y, n_e x S]
        #   cells      S (2 m + C (108 m + 23) + 23), C = cells per sample [bbh_qlognehvi_kernel] - by the maths of
        #              acqfs.py:477-484 / BoTorch's qLogNEHVI with exp / log priced at 20 flops and a division at 8:
        #              per cell and target log_fatplus (2 + 51 + 20) and fatmin against the side length (34 + 1), per cell one
        #              exp and 3 flops of the log-sum-exp, per sample the same once more
        m_t = len(nehvi.outputs)
        n_e = n + extra["nehvi_baseline_points"]
        C = extra["nehvi_cells_per_sample"]
        parts = {
            "variance": (m_t * (n_e * n_e + 2 * n_e * d + 16 * n_e), fam_step["posterior"]),
            "columns": (m_t * (2 * n_e * S + 2 * n_e * d + 16 * n_e), fam_step["columns"]),
            "cells": (S * (2 * m_t + C * (108 * m_t + 23) + 23), fam_step["nehvi"]),
        }
        recs = {}
        for name, (fl, (ms, cnt)) in parts.items():
            per_step = ms / args.steps
            recs[name] = {"flops_per_candidate": fl, "device_ms_per_step": per_step, "launches_per_step": cnt / args.steps,
                          "achieved": rows_local * fl / (per_step * 1e-3) / 1e12 if per_step > 0 else 0.0}
            recs[name]["frac"] = recs[name]["achieved"] / FP64_MFMA_PEAK_TFLOPS
        # the cell kernel is VALU-issue-bound and a third of its instructions are packed single precision: its figure on the fp64
        # formulation's operation count is a throughput, its utilisation is the issue fraction
        recs["cells"]["throughput_vs_fp64_formulation"] = recs["cells"].pop("frac")
        recs["cells"]["issue_frac"] = issue_frac_of(cfg, "bbh_qlognehvi_lin_kernel")
        dom = max(recs, key=lambda k: recs[k]["device_ms_per_step"])
        roofline = {
            "bound": "mfma",  # the fp64 pipe: matrix and vector fp64 share it and have the same peak (78.6 TFLOP/s)
            "achieved": recs[dom]["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": recs[dom].get("frac"),  # measured in this run, or null: the cell kernel is VALU-issue-bound, its utilisation is ...
            "issue_frac_offline": {"value": (recs[dom].get("issue_frac") or {}).get("bbh_qlognehvi_lin_kernel<3, true>"),
                                   "source": f"profiles/{_latest_profile(cfg, 'issue')[0]} (SQ_INSTS_VALU x static class mix over SIMD cycles, offline PMC passes)"},
            "traffic": traffic_of_kernel(cfg, "bbh_qlognehvi")[0] if standard_rows else None,
            "traffic_source": traffic_of_kernel(cfg, "bbh_qlognehvi")[1] if standard_rows else None,
            "kernel": {"variance": kernel_names.get(form, form), "columns": "bbh_coop_columns_kernel", "cells": "bbh_qlognehvi_lin_kernel"}[dom],
            "dominant_part": dom, "parts": recs,
            "note": "cells: operation count by the maths of qLogNEHVI (exp / log at 20 flops, a division at 8) against the fp64 roof; "
                    "about a third of the cell kernel's instructions execute as packed single precision (DESIGN.md 4.4), so its fraction "
                    "is throughput on that count, not a utilisation bound",
            "whole_pass": {"flops_per_candidate": sum(v[0] for v in parts.values()),
                           "frac": rows_local * sum(v[0] for v in parts.values())
                                   / (sum(r["device_ms_per_step"] for r in recs.values()) * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "n_extended": n_e, "cells_per_sample": C,
            "hbm_GBps_algorithmic": rows_local * (8 * d + 8) / (ms_per_step * 1e-3) / 1e9,
        }
        flops_per_cand = roofline["whole_pass"]["flops_per_candidate"]

    out = {
        "metric": {"cfg3": "acquisition candidates scored/sec (1e6 discrete grid, n_train=512, qLogEI)",
                   "cfg2": "acquisition candidates scored/sec (1e5 discrete grid, d=15, n_train=256, qLogEI)",
                   "cfg4": "acquisition candidates scored/sec (1e5 candidates, ICM 4 tasks, n_train=1024, qLogEI)",
                   "cfg5": "acquisition candidates scored/sec (1e5 discrete grid, 3 targets, qLogNEHVI, 512 MC samples)"}[cfg],
        "value": value,
        "unit": "candidates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        # the timed step of the single-target configurations (posterior + q' = 1 qLogEI + selection) is fp64 throughout; off the headline
        # two kernel families evaluate their smoothing factors in packed single precision (A/B against the
        # fp64 form: max |delta score| 1.7e-9; BBH_NEHVI_PK=0 at run time / -DBBH_PENDING_PK=0 at build time select the fp64 forms)
        "dtype_note": ("fp64 + packed-fp32 smoothing factors (the fat-min / fat-max factors of the qLogNEHVI cell kernel; posterior, "
                       "conditional means and the log-sum-exp are fp64)" if cfg == "cfg5" else
                       "fp64 (timed step); extra.greedy_q*: the joint q' >= 2 qLogEI kernels are fp64 + packed-fp32 smoothing factors"),
        "data": "synthetic",
        "config": {
            "workload": {
                "cfg3": (f"{total_rows} x {d} discrete grid row-sharded over {world} GPUs ({rows_local} rows on rank 0), " if (args.strong and world > 1)
                         else f"{rows_local} x {d} discrete grid per GPU, ") + f"n_train={n}, Matern-5/2 ARD, qLogEI S={S}, fixed-theta, top-{TOPK} to host",
                "cfg2": f"{rows_local} x {d} discrete grid per GPU, n_train={n}, Matern-5/2 ARD, qLogEI S={S}, fixed-theta, top-{TOPK} to host "
                        f"(BASELINE configs[1])",
                "cfg4": f"{rows_local} x ({d} + task) candidates of the active task, transfer-learning GP (ICM over 4 tasks), n_train={n}, "
                        f"qLogEI S={S}, fixed-theta, top-{TOPK} to host (BASELINE configs[3])",
                "cfg5": (f"{total_rows} x {d} discrete grid row-sharded over {world} GPUs ({rows_local} rows on rank 0)" if (args.strong and world > 1)
                         else f"{rows_local} x {d} discrete grid") + f", ParetoObjective of 3 targets -> qLogNEHVI, n_train={n}, S={S} MC samples, "
                        f"pruned baseline, top-{TOPK} to host (BASELINE configs[4], one GPU's share); the timed step is the SCORING PASS - a "
                        f"selection step adds {extra.get('nehvi_setup_ms', float('nan')):.2f} ms of set-up: extra.ms_per_selection_step = "
                        f"{extra.get('ms_per_selection_step', float('nan')):.2f} ms",
            }[cfg],
            "baseline_config": cfg,
            "global_rows": total_rows,
            "parallelism": f"row-shard x{world}",
            "collective": "none" if not dist_on else ("rccl (library)" if use_rccl else "torch.distributed " + dist.get_backend()),
        },
        "roofline": roofline,
        "extra": extra,
    }
    if pending_roofline is not None:
        out["extra"]["roofline_pending_kernels"] = pending_roofline

    if rank == 0 and world == 1 and args.cpu_budget > 0 and cfg in ("cfg3", "cfg2"):
        from oracle import cpu_baseline as cb
        from oracle import gp_oracle as go

        ospec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        om = go.fit_gp(ospec, Xt, y, params=go.GPParams(np.full(d, ls), math.exp(-5.0), 0.0))
        # the restated reference path at torch's default thread count (what a user gets) and at smaller pools (the
        # b x 1 x n t-batches of 2048 do not feed 128 threads); the best one is the reported baseline
        import torch as _t

        default_threads = _t.get_num_threads()
        sweep = sorted({default_threads, *[c for c in (8, 16, 32) if c < default_threads]})
        by_threads = {}
        for th in sweep:
            _t.set_num_threads(th)
            cps_t, scored_t, _ = cb.time_cpu_baseline(om, X, z, go.best_f_from_model(om), budget_s=args.cpu_budget / len(sweep))
            by_threads[th] = (cps_t, scored_t)
        _t.set_num_threads(default_threads)
        threads = max(by_threads, key=lambda k: by_threads[k][0])
        cps, scored = by_threads[threads]
        out["cpu_baseline"] = {
            "value": cps,
            "unit": "candidates/s",
            "cores": threads,
            "kind": "port",
            "sample": f"first {scored} rows of the same grid, chunks of 2048 (restated reference CPU path, torch-CPU fp64, "
                      f"{os.cpu_count()} logical CPUs; best of the thread counts tried)",
            "by_threads": {str(k): v[0] for k, v in by_threads.items()},
            "default_threads": default_threads,
        }
        out["extra"]["speedup_vs_cpu_baseline"] = value / cps
    if rank == 0 and world == 1 and args.cpu_budget > 0 and fit_start_raw is not None:
        # the fit objective on the host cores beside the device fit: evaluations of the oracle's objective (torch-CPU fp64, autograd
        # gradient - what fit_gpytorch_mll evaluates per L-BFGS-B step) at the fit's starting point, a bounded sample
        import torch as _t
        from oracle import gp_oracle as go

        if cfg == "cfg4":
            ospec = go.GPSpec.baybe_default(d + 1, np.zeros(d + 1), np.ones(d + 1), task_idx=d, n_tasks=4)
            y_fit = y
        else:
            ospec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
            y_fit = ys[0] if cfg == "cfg5" else y
        Xn = go.normalize_inputs(ospec, Xt)
        ystd = go.standardize_targets(np.asarray(y_fit))[0]
        raw0 = go.pack_raw(ospec, go.initial_params(ospec))
        go.fit_objective(ospec, raw0, Xn, ystd)
        t0 = time.perf_counter()
        evals = 0
        while evals < 20 and time.perf_counter() - t0 < min(5.0, args.cpu_budget / 3):
            go.fit_objective(ospec, raw0, Xn, ystd)
            evals += 1
        per_eval = (time.perf_counter() - t0) / max(evals, 1) * 1e3
        nfev = out["extra"].get("fit_nfev")
        nfev_tot = int(np.sum(nfev)) if nfev is not None else 0
        out["extra"]["cpu_fit"] = {
            "ms_per_evaluation": per_eval, "evaluations_timed": evals, "threads": _t.get_num_threads(),
            "cpu_fit_ms": per_eval * nfev_tot, "note": "oracle fit objective (value + autograd gradient) x the device fit's evaluation count"
                                                       + (" (three targets)" if cfg == "cfg5" else ""),
        }
    # The JSON line is the LAST thing on rank 0's stdout: RCCL prints a version banner through C stdio (buffered when stdout is a pipe, so
    # it used to come out at exit, BEHIND the line); the process group goes first, C stdio is flushed, then the line.
    if dist_on:
        try:
            dist.barrier()
        except Exception:  # noqa: BLE001
            pass
        dist.destroy_process_group()
    if rank == 0:
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
