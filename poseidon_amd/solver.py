"""Solver context: thin object wrapper over the psn_* C ABI (one context per GPU)."""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from .network import Cluster, Network, _vp


@dataclasses.dataclass
class SolveResult:
    total_cost: int
    status: int
    refines: int
    arc_scans: int
    pushes: int
    relabels: int
    vertex_visits: int
    gu_arc_scans: int
    gu_relaxations: int
    sat_arc_scans: int
    rounds: int
    gu_iterations: int
    global_updates: int
    launches: int
    solve_ms: float
    gu_arc_skips: int = 0
    arc_skips: int = 0
    warm_started: int = 0
    mirrored_replays: int = 0
    replays: int = 0
    first_units: int = 0
    first_eps_log2: int = 0
    last_phase_replays: int = 0
    warm_declined: int = 0

    @property
    def algorithmic_bytes(self) -> int:
        """Bytes the sweeps had to move: 20 B per arc scan (head, cap, cost, gathered
        price), 36 B per push, 32 B per vertex visit, 8 B per relabel."""
        return (20 * (self.arc_scans + self.gu_arc_scans + self.sat_arc_scans) + 36 * self.pushes
                + 32 * self.vertex_visits + 8 * self.relabels)


class Solver:
    def __init__(self, device: int = 0, alpha_log2: int = 0, rounds_per_batch: int = 0,
                 eager: bool = False, time_kernels: bool = False, eps_from_max: bool = False):
        self._lib = _lib.load_library()
        self._ctx = C.c_void_p(None)
        flags = (_lib.PSN_CFG_EAGER if eager or time_kernels else 0) | (_lib.PSN_CFG_KTIME if time_kernels else 0) \
            | (_lib.PSN_CFG_EPS_FROM_MAX if eps_from_max else 0)
        cfg = _lib.Config(device, alpha_log2, rounds_per_batch, flags)
        rc = self._lib.psn_create(C.byref(cfg), C.byref(self._ctx))
        if rc != _lib.PSN_OK:
            self._ctx = C.c_void_p(None)
            _lib.check(rc)
        self.n = 0
        self.m2 = 0
        self.layout = None

    def close(self):
        if self._ctx:
            self._lib.psn_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        _lib.check(rc, self._ctx)

    def upload_csr(self, n, row_off, head, cap, cost, rev, supply):
        row_off = np.ascontiguousarray(row_off, np.int64)
        head = np.ascontiguousarray(head, np.int32)
        cap = np.ascontiguousarray(cap, np.int32)
        cost = np.ascontiguousarray(cost, np.int32)
        rev = np.ascontiguousarray(rev, np.int32)
        supply = np.ascontiguousarray(supply, np.int64)
        m2 = int(head.shape[0])
        self._chk(self._lib.psn_upload_csr(self._ctx, int(n), m2, _vp(row_off), _vp(head), _vp(cap),
                                           _vp(cost), _vp(rev), _vp(supply)))
        self.n, self.m2 = int(n), m2

    def upload(self, net: Network):
        self.upload_csr(net.n, net.row_off, net.head, net.cap, net.cost, net.rev, net.supply)
        self.layout = net.layout

    def build_network(self, cl: Cluster, unsched_cost: int | None = None):
        """Build the scheduling network of `cl` on the device (psn_build_network_device) and make it
        the resident network.  Returns a Network that carries the layout and ec_arc_base only (the
        arc arrays exist on the device; fetch them with costs() / residual() if needed)."""
        from .network import UNSCHED_COST
        lay = cl.layout
        n = 1 + lay.n_machines + lay.n_classes + lay.n_jobs + lay.n_tasks
        m2 = 2 * (2 * lay.n_tasks + lay.n_classes * lay.n_machines * lay.k_slots + lay.n_machines + lay.n_jobs)
        base = np.empty(lay.n_classes, np.int64)
        tc = np.ascontiguousarray(cl.task_class, np.int32)
        tj = np.ascontiguousarray(cl.task_job, np.int32)
        self._chk(self._lib.psn_build_network_device(self._ctx, C.byref(lay), int(cl.slots),
                                                     int(UNSCHED_COST if unsched_cost is None else unsched_cost),
                                                     _vp(tc), _vp(tj), _vp(base)))
        self.n, self.m2, self.layout = n, m2, lay
        empty32 = np.empty(0, np.int32)
        return Network(lay, n, m2, np.empty(0, np.int64), empty32, empty32, empty32, empty32, np.empty(0, np.int64), base)

    def structure(self):
        """(head, rev) of the resident network, read back from the device."""
        head = np.empty(self.m2, np.int32)
        rev = np.empty(self.m2, np.int32)
        self._chk(self._lib.psn_get_structure(self._ctx, _vp(head), _vp(rev)))
        return head, rev

    def rows(self):
        """(row_off, supply) of the resident network."""
        row_off = np.empty(self.n + 1, np.int64)
        supply = np.empty(self.n, np.int64)
        self._chk(self._lib.psn_get_rows(self._ctx, _vp(row_off), _vp(supply)))
        return row_off, supply

    def eval_costs(self, cl: Cluster, net: Network):
        self._chk(self._lib.psn_eval_costs(self._ctx, C.byref(cl.layout), _vp(cl.cap_cpu), _vp(cl.cap_mem),
                                           _vp(cl.used_cpu), _vp(cl.used_mem), _vp(cl.req_cpu),
                                           _vp(cl.req_mem), _vp(net.ec_arc_base)))

    def update_caps(self, arc_index, new_cap):
        arc_index = np.ascontiguousarray(arc_index, np.int64)
        new_cap = np.ascontiguousarray(new_cap, np.int32)
        self._chk(self._lib.psn_update_caps(self._ctx, arc_index.shape[0], _vp(arc_index), _vp(new_cap)))

    def update_costs(self, arc_index, new_cost):
        arc_index = np.ascontiguousarray(arc_index, np.int64)
        new_cost = np.ascontiguousarray(new_cost, np.int32)
        self._chk(self._lib.psn_update_costs(self._ctx, arc_index.shape[0], _vp(arc_index), _vp(new_cost)))

    def set_admission_mask(self, layout, allowed):
        """allowed: bool [E, N] (class e may use machine m) or None to lift the mask."""
        if allowed is None:
            self._chk(self._lib.psn_set_admission_mask(self._ctx, C.byref(layout), None))
            return
        allowed = np.asarray(allowed, bool)
        E, N = allowed.shape
        words = (N + 31) // 32
        padded = np.zeros((E, words * 32), bool)
        padded[:, :N] = allowed
        mask = np.packbits(padded.reshape(E, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(E, words)
        mask = np.ascontiguousarray(mask)
        self._chk(self._lib.psn_set_admission_mask(self._ctx, C.byref(layout), _vp(mask)))

    def update_supplies(self, vertex, new_supply):
        vertex = np.ascontiguousarray(vertex, np.int32)
        new_supply = np.ascontiguousarray(new_supply, np.int64)
        self._chk(self._lib.psn_update_supplies(self._ctx, vertex.shape[0], _vp(vertex), _vp(new_supply)))

    def solve(self, warm_start: bool = False) -> SolveResult:
        r = _lib.Result()
        rc = self._lib.psn_solve(self._ctx, _lib.PSN_WARM_START if warm_start else 0, C.byref(r))
        self._chk(rc)
        return SolveResult(*(getattr(r, f[0]) for f in _lib.Result._fields_ if not f[0].startswith("reserved")))

    def flow(self) -> np.ndarray:
        out = np.empty(self.m2, np.int32)
        self._chk(self._lib.psn_get_flow(self._ctx, _vp(out)))
        return out

    def residual(self) -> np.ndarray:
        out = np.empty(self.m2, np.int32)
        self._chk(self._lib.psn_get_residual(self._ctx, _vp(out)))
        return out

    def prices(self) -> np.ndarray:
        out = np.empty(self.n, np.int64)
        self._chk(self._lib.psn_get_prices(self._ctx, _vp(out)))
        return out

    def price_scale(self) -> int:
        """The factor costs are multiplied by: prices() are 1-optimal for cost * price_scale()."""
        v = C.c_int64(0)
        self._chk(self._lib.psn_price_scale(self._ctx, C.byref(v)))
        return int(v.value)

    def costs(self):
        cost = np.empty(self.m2, np.int32)
        cap = np.empty(self.m2, np.int32)
        self._chk(self._lib.psn_get_costs(self._ctx, _vp(cost), _vp(cap)))
        return cost, cap

    def placements(self, layout=None) -> np.ndarray:
        layout = layout or self.layout
        out = np.empty(layout.n_tasks, np.int32)
        self._chk(self._lib.psn_get_placements(self._ctx, C.byref(layout), _vp(out)))
        return out

    def placements_into(self, device_ptr: int, layout=None):
        """Placement extraction straight into caller-owned device memory (int32[P])."""
        layout = layout or self.layout
        self._chk(self._lib.psn_get_placements_device(self._ctx, C.byref(layout), C.c_void_p(device_ptr)))

    def kernel_times(self):
        """{class name: (total device ms, launches)} of the last solve (time_kernels=True)."""
        ms = (C.c_double * len(_lib.KT_NAMES))()
        calls = (C.c_uint64 * len(_lib.KT_NAMES))()
        self._chk(self._lib.psn_kernel_times(self._ctx, ms, calls))
        return {nm: (ms[i], int(calls[i])) for i, nm in enumerate(_lib.KT_NAMES)}

    def placements_device_ptr(self):
        p = C.c_void_p(None)
        cnt = C.c_int64(0)
        self._chk(self._lib.psn_placements_device_ptr(self._ctx, C.byref(p), C.byref(cnt)))
        return p.value, cnt.value
