"""ORACLE (test infrastructure): ctypes loader for oracle/psn_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The shared object is built by ``oracle/Makefile`` (plain gcc).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libpsn_oracle.so")


class Stats(C.Structure):
    _fields_ = [("pushes", C.c_int64), ("relabels", C.c_int64), ("arc_scans", C.c_int64),
                ("global_updates", C.c_int64), ("refines", C.c_int64), ("solve_ms", C.c_double)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "psn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.psn_oracle_fnv1a64.restype = C.c_uint64
        _lib.psn_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_int64, C.c_uint64]
        _lib.psn_oracle_check_eps.restype = C.c_int64
    return _lib


def _p(a, dtype):
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def _m2(g) -> int:
    """Residual arc count: 2m, unless the network carries self-paired placeholder arcs."""
    return int(g.row_off[-1])


def fnv1a64(*arrays) -> int:
    h = 0xCBF29CE484222325
    for a in arrays:
        a = np.ascontiguousarray(a)
        h = lib().psn_oracle_fnv1a64(a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes), C.c_uint64(h))
    return int(h)


def csr_checksum(g) -> int:
    return fnv1a64(g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)


def eval_costs(cl):
    E, N, K = cl.req_cpu.shape[0], cl.n_machines, cl.k_slots
    cost = np.empty((E, N, K), np.int32)
    cap = np.empty((E, N, K), np.int32)
    lib().psn_oracle_eval_costs(C.c_int32(E), C.c_int32(N), C.c_int32(K),
                                _p(cl.req_cpu, np.int64), _p(cl.req_mem, np.int64),
                                _p(cl.cap_cpu, np.int64), _p(cl.cap_mem, np.int64),
                                _p(cl.used_cpu, np.int64), _p(cl.used_mem, np.int64),
                                _p(cost, np.int32), _p(cap, np.int32))
    return cost, cap


def ssp(g):
    """-> (rc, total_cost, residual_cap, potentials)"""
    cap = g.cap.copy()
    tc = C.c_int64(0)
    pot = np.zeros(g.n, np.int64)
    rc = lib().psn_oracle_ssp(C.c_int32(g.n), C.c_int64(_m2(g)), _p(g.row_off, np.int64),
                              _p(g.head, np.int32), _p(cap, np.int32), _p(g.cost, np.int32),
                              _p(g.rev, np.int32), _p(g.supply, np.int64), C.byref(tc),
                              _p(pot, np.int64))
    return rc, tc.value, cap, pot


CS_EPS_PERCENTILE, CS_PRICE_REFINE, CS_COVER_SCALE = 1, 2, 4
# The timed CPU baseline: the plain textbook variant (the one the parity tests use as the oracle)
# and the tuned ones (the GPU solver's first epsilon, price refinement before each later phase,
# alpha = 16 or the GPU solver's 8).  Arc fixing and push look-ahead are not implemented.
BASELINE_VARIANTS = {
    "plain (eps from max cost, alpha 12)": {},
    "tuned (eps from 90th-percentile cost, alpha 16, price refinement)": {"alpha": 16, "flags": CS_EPS_PERCENTILE | CS_PRICE_REFINE},
    "tuned + cost scale from the cycle-length bound, alpha 16 (as the GPU solver)": {"alpha": 16, "flags": CS_EPS_PERCENTILE | CS_PRICE_REFINE | CS_COVER_SCALE},
    "tuned + cost scale from the cycle-length bound, alpha 8": {"alpha": 8, "flags": CS_EPS_PERCENTILE | CS_PRICE_REFINE | CS_COVER_SCALE},
}


def use_native_build() -> str:
    """For the timed CPU baseline: compile the oracle for the host this process runs on
    (-O3 -march=native) into a temporary directory and use that copy from now on.  Falls back to
    the shipped build when no compiler is at hand.  Returns the flags in use."""
    global _lib, _native_flags
    import shutil
    import tempfile
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc:
        out = os.path.join(tempfile.mkdtemp(prefix="psn_oracle_"), "libpsn_oracle_native.so")
        flags = ["-O3", "-march=native", "-fPIC", "-shared"]
        try:
            subprocess.check_call([cc] + flags + ["-o", out, os.path.join(_HERE, "psn_oracle.c")])
            lib()  # make sure the default is loadable first (keeps error messages sane)
            _lib = None
            globals()["_SO"] = out
            lib()
            _native_flags = os.path.basename(cc) + " " + " ".join(flags) + " (built on this host)"
        except (subprocess.CalledProcessError, OSError):
            pass
    return build_flags()


_native_flags = None


def build_flags() -> str:
    if _native_flags:
        return _native_flags
    try:
        return open(os.path.join(_HERE, "build", "flags.txt")).read().strip()
    except OSError:
        return "unknown"


def cost_scaling(g, alpha: int = 12, flags: int = 0):
    """-> (rc, total_cost, residual_cap, scaled_prices, Stats)"""
    cap = g.cap.copy()
    tc = C.c_int64(0)
    price = np.zeros(g.n, np.int64)
    st = Stats()
    rc = lib().psn_oracle_cost_scaling_ex(C.c_int32(g.n), C.c_int64(_m2(g)), _p(g.row_off, np.int64),
                                          _p(g.head, np.int32), _p(cap, np.int32),
                                          _p(g.cost, np.int32), _p(g.rev, np.int32),
                                          _p(g.supply, np.int64), C.c_int32(alpha), C.c_int32(flags), C.byref(tc),
                                          _p(price, np.int64), C.byref(st))
    return rc, tc.value, cap, price, st


def check(g, cap_final, price_hint=None, scale: int | None = None):
    """Certificate.  -> (rc, total_cost, potentials); rc 0 = feasible and optimal,
    1 = conservation, 2 = bounds/structure, 3 = not optimal.  price_hint: scaled prices to start
    the checker's own potentials from (scaled by `scale`, default n + 1); only a starting point."""
    tc = C.c_int64(0)
    pot = np.zeros(g.n, np.int64)
    cap_final = np.ascontiguousarray(cap_final, dtype=np.int32)
    if price_hint is not None and scale is not None and scale != g.n + 1:
        price_hint = np.floor(np.asarray(price_hint, np.float64) / float(scale)).astype(np.int64) * (g.n + 1)
    hint = None if price_hint is None else _p(np.ascontiguousarray(price_hint, np.int64), np.int64)
    rc = lib().psn_oracle_check(C.c_int32(g.n), C.c_int64(_m2(g)), _p(g.row_off, np.int64),
                                _p(g.head, np.int32), _p(g.cap, np.int32), _p(cap_final, np.int32),
                                _p(g.cost, np.int32), _p(g.rev, np.int32), _p(g.supply, np.int64),
                                hint, C.byref(tc), _p(pot, np.int64))
    return rc, tc.value, pot


def cycle_bound(g) -> int:
    """Upper bound on the length of a simple cycle of g (vertex-cover argument)."""
    lib().psn_oracle_cycle_bound.restype = C.c_int64
    return int(lib().psn_oracle_cycle_bound(C.c_int32(g.n), _p(g.row_off, np.int64), _p(g.head, np.int32)))


def check_eps(g, cap_final, price, eps: int = 1, scale: int | None = None) -> int:
    """Number of residual arcs violating eps-optimality under prices scaled by `scale` (default
    n + 1); -1 when eps is too large for that scale to imply optimality."""
    cap_final = np.ascontiguousarray(cap_final, dtype=np.int32)
    price = np.ascontiguousarray(price, dtype=np.int64)
    if scale is not None:
        lib().psn_oracle_check_eps_scaled.restype = C.c_int64
        return int(lib().psn_oracle_check_eps_scaled(C.c_int32(g.n), C.c_int64(_m2(g)), _p(g.row_off, np.int64),
                                                     _p(g.head, np.int32), _p(cap_final, np.int32),
                                                     _p(g.cost, np.int32), _p(price, np.int64), C.c_int64(eps),
                                                     C.c_int64(scale)))
    return int(lib().psn_oracle_check_eps(C.c_int32(g.n), C.c_int64(_m2(g)), _p(g.row_off, np.int64),
                                          _p(g.head, np.int32), _p(cap_final, np.int32),
                                          _p(g.cost, np.int32), _p(price, np.int64), C.c_int64(eps)))
