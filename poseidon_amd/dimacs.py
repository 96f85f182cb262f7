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
    """-> dict(n, m, tail, head, cap, cost, supply, base_cost) with 0-based ids."""
    lib = _lib.load_library()
    n, m = C.c_int32(), C.c_int64()
    _lib.check(lib.psn_dimacs_size(path.encode(), C.byref(n), C.byref(m)))
    n, m = n.value, m.value
    tail, head, cap, cost = (np.empty(m, np.int32) for _ in range(4))
    supply = np.empty(n, np.int64)
    base = C.c_int64()
    _lib.check(lib.psn_dimacs_read(path.encode(), n, m, tail.ctypes.data, head.ctypes.data, cap.ctypes.data, cost.ctypes.data,
                                   supply.ctypes.data, C.byref(base)))
    return {"n": n, "m": m, "tail": tail, "head": head, "cap": cap, "cost": cost, "supply": supply, "base_cost": base.value}


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
