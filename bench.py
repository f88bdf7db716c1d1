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

  python bench.py                       # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
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
    """Deterministic synthetic grid: 11 levels per dimension in [0,1]; the training
    set is identical on every rank (drawn from rank 0's stream), the shard differs per rank."""
    X0 = np.random.default_rng(0).integers(0, 11, size=(max(n * 4, 4096), d)) / 10.0
    idx = np.random.default_rng(1).choice(X0.shape[0], n, replace=False)
    Xt = X0[idx]
    y = -((Xt - 0.5) ** 2).sum(1) + 0.1 * np.sin(2 * np.pi * Xt[:, 0]) + 0.05 * np.random.default_rng(2).standard_normal(n)
    X = np.random.default_rng(1000 + rank).integers(0, 11, size=(rows, d)) / 10.0
    return X, Xt, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000, help="candidate rows per GPU")
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--n-train", type=int, default=512)
    ap.add_argument("--mc-samples", type=int, default=512)
    ap.add_argument("--strong", action="store_true", help="fixed global grid of --rows rows, split over the GPUs")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--fit", action="store_true", help="also time a device hyper-parameter fit (extra)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # BENCH_SINGLE_DEVICE=1: dry run of the N > 1 code path on ONE GPU (all ranks on cuda:0, gloo
    # instead of RCCL) — used to test the sharded path where only one device is available.
    single_dev = os.environ.get("BENCH_SINGLE_DEVICE", "0") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from baybe_amd import engine, gp_spec
    from baybe_amd.distributed import RowShard, shard_bounds

    d, n, S = args.d, args.n_train, args.mc_samples
    if args.strong:
        a, b = shard_bounds(args.rows, rank, world)
        rows_local, total_rows = b - a, args.rows
    else:
        rows_local, total_rows = args.rows, args.rows * world
    X, Xt, y = synth_problem(rows_local, d, n, rank if not args.strong else 0)
    if args.strong:
        X = synth_problem(args.rows, d, n, 0)[0][a:b]

    gp = engine.HipGP(local_rank)
    spec = gp_spec.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
    gp.set_model(spec, Xt, y)
    extra = {}
    if args.fit:
        t0 = time.perf_counter()
        fi = gp.fit()
        extra["fit_ms"] = (time.perf_counter() - t0) * 1e3
        extra["fit_nfev"] = fi.nfev
    ls = math.exp(math.sqrt(2.0) - 3.0) * math.sqrt(d)
    params = gp_spec.GPParams(np.full(d, ls), math.exp(-5.0), 0.0)
    t0 = time.perf_counter()
    gp.factorize(params)
    torch.cuda.synchronize()
    extra["factorize_ms"] = (time.perf_counter() - t0) * 1e3
    best_f = gp.best_f()
    z = engine.sobol_normal_base_samples(S, 1, 1234)[:, 0]
    Xd = torch.from_numpy(X).cuda()
    shard = RowShard(total_rows, rank, world) if world > 1 else None
    if shard is not None and not args.strong:
        shard.start, shard.stop = rank * rows_local, (rank + 1) * rows_local

    def step():
        # two kernels: measured 1.5 % faster than the single fused posterior+qLogEI kernel
        # (scripts/gpu_ab_fused_acq.py: 5.64 vs 5.73 ms/step): the epilogue's VALU work costs the shared fp64
        # pipe the same either way, but a separate launch runs it at full occupancy and leaves the fused
        # kernel's LDS to the kernel-value cache
        mean, var = gp.posterior(Xd)
        scores = gp.qlogei(mean, var, z, best_f, 1.0)
        vals, idx = gp.topk(scores, TOPK)
        if shard is not None:
            vals, idx = shard.global_topk(vals, idx, TOPK, device=Xd.device)
        return vals, idx

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gp.timing(True)
    gp.timing_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals, idx = step()
    fence()
    dt = time.perf_counter() - t0
    fused_ms, fused_launches = gp.timing_read(reset=True)
    gp.timing(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if single_dev else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ms_per_step = dt / args.steps * 1e3
    value = total_rows * args.steps / dt

    # ---- roofline of the dominant kernel (fused posterior), algorithmic flops ----
    flops_per_cand = n * n + 2 * n * d + 16 * n + 16 * S
    avg_ms = fused_ms / max(fused_launches, 1)
    achieved = rows_local * flops_per_cand / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    traffic = None
    try:  # PMC-derived HBM bytes per launch for this exact workload (collected offline, profiles/)
        tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        traffic = tj["workloads"].get(f"{rows_local}x{d}_n{n}", {}).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        traffic = None
    roofline = {
        "bound": "mfma",
        "achieved": achieved,
        "peak": FP64_MFMA_PEAK_TFLOPS,
        # scripts/mfma_valu_overlap_probe.hip: a pure v_mfma_f64_16x16x4_f64 stream reaches 77.8 TFLOP/s, and
        # fp64 VALU work issued next to it adds its time (one shared DP pipe) - the peak is for MFMA + VALU
        "peak_mfma_stream_microbench": 77.8,
        "unit": "TFLOP/s",
        "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
        "traffic": traffic,
        "traffic_unit": "L2<->fabric bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json; algorithmic = %d; "
                        "the excess is scratch traffic of 76 spilled VGPRs - the kernel is fp64-pipe bound)" % (rows_local * (8 * d + 16)),
        "kernel": "bbh_fused_posterior_kernel",
        "avg_launch_ms": avg_ms,
        "launches": fused_launches,
        "flops_per_candidate": flops_per_cand,
        "hbm_GBps_algorithmic": rows_local * (8 * d + 16) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
    }

    out = {
        "metric": "acquisition candidates scored/sec (1e6 discrete grid, n_train=512, qLogEI)",
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
        "data": "synthetic",
        "config": {
            "workload": f"{rows_local} x {d} discrete grid per GPU, n_train={n}, Matern-5/2 ARD, qLogEI S={S}, "
                        f"fixed-theta, top-{TOPK} to host",
            "global_rows": total_rows,
            "parallelism": f"row-shard x{world}",
        },
        "roofline": roofline,
        "extra": extra,
    }

    if rank == 0 and world == 1 and args.cpu_budget > 0:
        from oracle import cpu_baseline as cb
        from oracle import gp_oracle as go

        ospec = go.GPSpec.baybe_default(d, np.zeros(d), np.ones(d))
        om = go.fit_gp(ospec, Xt, y, params=go.GPParams(np.full(d, ls), math.exp(-5.0), 0.0))
        cps, scored, threads = cb.time_cpu_baseline(om, X, z, go.best_f_from_model(om), budget_s=args.cpu_budget)
        out["cpu_baseline"] = {
            "value": cps,
            "unit": "candidates/s",
            "cores": threads,
            "kind": "port",
            "sample": f"first {scored} rows of the same grid, chunks of 2048 (restated reference CPU path, torch-CPU fp64, "
                      f"{os.cpu_count()} logical CPUs)",
        }
        out["extra"]["speedup_vs_cpu_baseline"] = value / cps
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
