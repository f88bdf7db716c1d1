"""Box decomposition of the non-dominated region (host-side set-up of qLogNEHVI).

Per Monte-Carlo sample the baseline's Pareto front defines the region a candidate can still
improve; it is cut into disjoint boxes so that the hypervolume improvement of a point is a sum of
clipped box volumes.  BoTorch does this in Python on the CPU as well
(``FastNondominatedPartitioning``, one partitioning per MC sample for m > 2); it is O(S P^2 m) set-up
work, not per-candidate work.  Algorithm: local upper bounds of the negated problem, incrementally
(Lacour, Klamroth & Fonseca 2017, Alg. 1, general position), each bound u contributing the box
``z_1 < u_1, max_{k<j} z^k(u)_j <= z_j < u_j``.
"""

from __future__ import annotations

import numpy as np

UPPER_CLAMP = 1e10  # BoTorch clamps cell upper bounds at 1e10 (double) before taking log lengths


def pareto_mask(Y: np.ndarray) -> np.ndarray:
    """Boolean mask of the non-dominated rows of Y (maximisation); vectorised O(P^2 m)."""
    Y = np.asarray(Y, dtype=np.float64)
    ge = (Y[None, :, :] >= Y[:, None, :]).all(-1)
    gt = (Y[None, :, :] > Y[:, None, :]).any(-1)
    return ~(ge & gt).any(1)


def nondominated_cells(Y: np.ndarray, ref: np.ndarray):
    """(lower [K,m], upper [K,m]) of disjoint boxes tiling {y >= ref, y not dominated by Y}."""
    ref = np.asarray(ref, dtype=np.float64)
    m = ref.shape[0]
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, m)
    if len(Y):
        Y = np.unique(Y[pareto_mask(Y)], axis=0)
        Y = Y[(Y > ref).all(1)]
    U = (-ref)[None, :].copy()  # [nU, m] local upper bounds, minimisation space
    Z = np.full((1, m, m), -np.inf)  # Z[u, k] = defining point of u in dimension k
    Z[0, np.arange(m), np.arange(m)] = -ref
    for p in -Y:
        hit = (p[None, :] < U).all(1)
        if not hit.any():
            continue
        keepU, keepZ = [U[~hit]], [Z[~hit]]
        Uh, Zh = U[hit], Z[hit]
        for j in range(m):
            others = [k for k in range(m) if k != j]
            ok = (Zh[:, others, j] < p[j]).all(1)
            if ok.any():
                Uj = Uh[ok].copy()
                Uj[:, j] = p[j]
                Zj = Zh[ok].copy()
                Zj[:, j, :] = p
                keepU.append(Uj)
                keepZ.append(Zj)
        U, Z = np.concatenate(keepU), np.concatenate(keepZ)
    lb = np.full_like(U, -np.inf)
    for j in range(1, m):
        lb[:, j] = Z[:, :j, j].max(axis=1)
    ok = (lb < U).all(1)
    return -U[ok], -lb[ok]


def pack_cells(obj_b: np.ndarray, ref: np.ndarray):
    """Decompose every MC sample's baseline objective values obj_b [S, n_b, m].

    Returns (cell_off [S+1] int64, cell_lo [K,m], cell_loglen [K,m]) in the layout ``bbh_qlognehvi``
    expects: ``cell_loglen = log(min(upper, 1e10) - lower)``."""
    S, _, m = obj_b.shape
    off = np.zeros(S + 1, dtype=np.int64)
    los, lls = [], []
    for s in range(S):
        lo, up = nondominated_cells(obj_b[s], ref)
        off[s + 1] = off[s] + len(lo)
        if len(lo):
            los.append(lo)
            lls.append(np.log(np.minimum(up, UPPER_CLAMP) - lo))
    if los:
        return off, np.ascontiguousarray(np.concatenate(los)), np.ascontiguousarray(np.concatenate(lls))
    return off, np.zeros((0, m)), np.zeros((0, m))


def pack_cells_native(obj_b: np.ndarray, ref: np.ndarray):
    """``pack_cells`` through the library's host code (``bbh_cells_create`` / ``_get`` / ``_destroy``,
    csrc/bbh_cells.hip): the same algorithm and visiting order in C++, ~100x faster than the numpy form above,
    which stays as the readable statement of the algorithm and as its cross-check in the tests."""
    import ctypes as C

    from baybe_amd import _lib

    lib = _lib.load_library()
    obj_b = np.ascontiguousarray(obj_b, dtype=np.float64)
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    S, n, m = obj_b.shape
    handle, total = C.c_void_p(), C.c_int64()
    rc = lib.bbh_cells_create(obj_b.ctypes.data_as(_lib.c_double_p), S, n, m, ref.ctypes.data_as(_lib.c_double_p),
                              C.byref(handle), C.byref(total))
    if rc != 0:
        raise ValueError("bbh_cells_create: bad arguments (1 <= m <= 4)")
    try:
        K = int(total.value)
        off = np.zeros(S + 1, dtype=np.int64)
        lo, ll = np.zeros((K, m)), np.zeros((K, m))
        lib.bbh_cells_get(handle, off.ctypes.data_as(_lib.c_int64_p), lo.ctypes.data_as(_lib.c_double_p),
                          ll.ctypes.data_as(_lib.c_double_p))
    finally:
        lib.bbh_cells_destroy(handle)
    return off, lo, ll
