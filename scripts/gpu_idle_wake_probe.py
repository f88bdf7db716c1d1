"""What the first device operation after an idle gap costs (round 6: set_model's 4 KB uploads took 8 - 35 ms in campaign-like loops, 40 us
back to back).  After sleeping for a few idle times: (a) a tiny kernel + synchronise, (b) an H2D copy of 4 KB from pinned memory
(hipMemcpyAsync: the copy engine) + synchronise, (c) the same D2H, (d) a kernel reading the pinned buffer through its device mapping."""
import sys, time
import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev, dtype=torch.float64)
pin = torch.zeros(512, dtype=torch.float64).pin_memory()
out_pin = torch.zeros(512, dtype=torch.float64).pin_memory()
torch.cuda.synchronize()


def timed(fn, idle):
    ts = []
    for _ in range(7):
        time.sleep(idle)
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def kernel():
    x.add_(1.0)


def h2d():
    x[:512].copy_(pin, non_blocking=True)


def d2h():
    out_pin.copy_(x[:512], non_blocking=True)


for idle in (0.0, 0.002, 0.01, 0.05, 0.3):
    print(f"idle {idle * 1e3:6.1f} ms:  " + "   ".join(f"{name} median {m:7.3f} (min {lo:6.3f}, max {hi:7.3f}) ms"
          for name, (m, lo, hi) in (("kernel", timed(kernel, idle)), ("H2D 4 KB", timed(h2d, idle)), ("D2H 4 KB", timed(d2h, idle)))), flush=True)

# ---- round 6, second question: does freeing host memory that a pageable copy touched stall the next device operation? -----------------
import numpy as np

print("host-memory experiments: median / max ms of a tiny kernel + synchronise after ...")


def after(prep, label, reps=9):
    ts = []
    for _ in range(reps):
        prep()
        t0 = time.perf_counter(); kernel(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"  {label:70s} median {ts[len(ts) // 2]:8.3f}  max {ts[-1]:8.3f}", flush=True)


after(lambda: None, "nothing")
after(lambda: np.empty(4_000_000).fill(1.0), "allocating, touching and freeing 32 MB of host memory")


def pageable_copy(n):
    a = np.ones(n)
    t = torch.from_numpy(a).to(dev)  # pageable H2D copy
    torch.cuda.synchronize()
    del t, a  # the host buffer goes back to the OS (mmap threshold exceeded)


after(lambda: pageable_copy(50_000), "a pageable H2D copy of 0.4 MB, buffer freed")
after(lambda: pageable_copy(420_000), "a pageable H2D copy of 3.4 MB, buffer freed")
after(lambda: pageable_copy(4_000_000), "a pageable H2D copy of 32 MB, buffer freed")


def pageable_d2h(n):
    t = torch.ones(n, device=dev, dtype=torch.float64)
    a = t.cpu().numpy()
    del a, t


after(lambda: pageable_d2h(420_000), "a D2H copy of 3.4 MB into pageable memory, buffer freed")
keep = []


def pageable_copy_kept(n):
    a = np.ones(n)
    t = torch.from_numpy(a).to(dev)
    torch.cuda.synchronize()
    keep.append(a)
    del t


after(lambda: pageable_copy_kept(420_000), "a pageable H2D copy of 3.4 MB, buffer KEPT")

# ---- third question: does a BURST of real work followed by a short idle gap make the next launch start late (clock / power-state transition)? ----
a = torch.randn(4096, 4096, device=dev, dtype=torch.float64)
print("burst-then-gap experiments: ms until a tiny kernel + synchronise completes (median / max of 9), after a burst of fp64 matmuls and a host-side gap")
for burst_ms, gap_ms in ((0, 3), (5, 0), (5, 1), (5, 3), (5, 10), (20, 3), (20, 30), (100, 3), (100, 100)):
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < burst_ms:
            (a @ a).sum().item()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) * 1e3 < gap_ms:  # busy host, idle device (as pandas work between device calls)
            pass
        t0 = time.perf_counter(); kernel(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"  burst {burst_ms:4d} ms, gap {gap_ms:4d} ms: median {ts[len(ts) // 2]:8.3f}  max {ts[-1]:8.3f}", flush=True)
