"""DIMACS minimum-cost-flow files (include/psn_builder.h): export a network, import one as
residual CSR, and optionally hand a network to an operator-supplied CPU solver."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

from . import _lib


def write(path: str, n: int, row_off, head, cap, cost, supply, comment: str = "libpsn export"):
    lib = _lib.load_library()
    row_off = np.ascontiguousarray(row_off, np.int64); head = np.ascontiguousarray(head, np.int32)
    cap = np.ascontiguousarray(cap, np.int32); cost = np.ascontiguousarray(cost, np.int32)
    supply = np.ascontiguousarray(supply, np.int64)
    _lib.check(lib.psn_dimacs_write(path.encode(), int(n), int(head.shape[0]), row_off.ctypes.data, head.ctypes.data,
                                    cap.ctypes.data, cost.ctypes.data, supply.ctypes.data, comment.encode()))


def read(path: str):
    """-> dict(n, m, tail, head, cap, cost, supply, base_cost, low) with 0-based ids; cap is the
    capacity above the lower bound `low`, which is already routed (supply, base_cost)."""
    lib = _lib.load_library()
    n, m = C.c_int32(), C.c_int64()
    _lib.check(lib.psn_dimacs_size(path.encode(), C.byref(n), C.byref(m)))
    n, m = n.value, m.value
    tail, head, cap, cost, low = (np.empty(m, np.int32) for _ in range(5))
    supply = np.empty(n, np.int64)
    base = C.c_int64()
    _lib.check(lib.psn_dimacs_read_lower(path.encode(), n, m, tail.ctypes.data, head.ctypes.data, cap.ctypes.data, cost.ctypes.data,
                                         supply.ctypes.data, C.byref(base), low.ctypes.data))
    return {"n": n, "m": m, "tail": tail, "head": head, "cap": cap, "cost": cost, "supply": supply, "base_cost": base.value,
            "low": low}


def write_changes(path: str, arc_changes=(), supply_changes=(), comment: str = "libpsn change records"):
    """arc_changes: iterable of (tail, head, cap, cost[, k]) with 0-based ids, k = which of the parallel
    arcs tail -> head in problem-file order; supply_changes: iterable of (vertex, supply)."""
    with open(path, "w") as f:
        f.write(f"c {comment}\n")
        for v, sup in supply_changes:
            f.write(f"n {int(v) + 1} {int(sup)}\n")
        for rec in arc_changes:
            t, h, cp, cs = rec[:4]
            k = f" {int(rec[4])}" if len(rec) > 4 else ""
            f.write(f"x {int(t) + 1} {int(h) + 1} 0 {int(cp)} {int(cs)}{k}\n")


def read_changes(path: str, problem: dict):
    """Change records resolved against a problem read by read(): -> dict(arc, cap, cost, vertex,
    supply) where arc[i] indexes the problem's arc list.  A record for an arc that had a lower bound
    in the problem file is refused (PsnError, PSN_EINVAL)."""
    lib = _lib.load_library()
    na, ns = C.c_int64(), C.c_int64()
    _lib.check(lib.psn_dimacs_changes_size(path.encode(), C.byref(na), C.byref(ns)))
    arc = np.empty(na.value, np.int64); cap = np.empty(na.value, np.int32); cost = np.empty(na.value, np.int32)
    vertex = np.empty(ns.value, np.int32); supply = np.empty(ns.value, np.int64)
    tail = np.ascontiguousarray(problem["tail"], np.int32); head = np.ascontiguousarray(problem["head"], np.int32)
    low = np.ascontiguousarray(problem.get("low", np.zeros(int(problem["m"]), np.int32)), np.int32)
    _lib.check(lib.psn_dimacs_read_changes_lower(path.encode(), int(problem["n"]), int(problem["m"]), tail.ctypes.data, head.ctypes.data,
                                                 low.ctypes.data, arc.ctypes.data, cap.ctypes.data, cost.ctypes.data, vertex.ctypes.data,
                                                 supply.ctypes.data))
    return {"arc": arc, "cap": cap, "cost": cost, "vertex": vertex, "supply": supply}


def csr_from_arcs(n: int, tail, head, cap, cost):
    """-> (row_off, head, cap, cost, rev, fwd_pos) residual CSR of an arc list."""
    lib = _lib.load_library()
    tail = np.ascontiguousarray(tail, np.int32); head = np.ascontiguousarray(head, np.int32)
    cap = np.ascontiguousarray(cap, np.int32); cost = np.ascontiguousarray(cost, np.int32)
    m = int(tail.shape[0])
    row_off = np.empty(n + 1, np.int64)
    r_head, r_cap, r_cost, r_rev = (np.empty(2 * m, np.int32) for _ in range(4))
    fwd = np.empty(m, np.int64)
    _lib.check(lib.psn_csr_from_arcs(int(n), m, tail.ctypes.data, head.ctypes.data, cap.ctypes.data, cost.ctypes.data,
                                     row_off.ctypes.data, r_head.ctypes.data, r_cap.ctypes.data, r_cost.ctypes.data,
                                     r_rev.ctypes.data, fwd.ctypes.data))
    return row_off, r_head, r_cap, r_cost, r_rev, fwd


def run_external_solver(path: str, timeout_s: float = 600.0):
    """If the environment names a DIMACS min-cost-flow solver (PSN_EXT_SOLVER: an executable that
    reads the problem on stdin and prints an "s <cost>" line), run it on `path`.
    -> (cost, seconds) or None when no solver is configured.  Nothing is ever fetched."""
    exe = os.environ.get("PSN_EXT_SOLVER")
    if not exe:
        return None
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        out = subprocess.run([exe], stdin=f, capture_output=True, timeout=timeout_s, check=True).stdout.decode(errors="replace")
    dt = time.perf_counter() - t0
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 2 and parts[0] == "s":
            return int(parts[-1]), dt
    raise RuntimeError("external solver printed no 's <cost>' line")
