"""Cluster state and flow-network construction (host side, over the C++ builder).

Mirrors what the reference accumulates from TaskSubmitted / NodeAdded
(/root/reference/pkg/k8sclient/podwatcher.go:377-410,
/root/reference/pkg/k8sclient/nodewatcher.go:292-339): per task a CPU request in
millicores and a memory request in Kb, per machine a capacity in the same units.
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib

CONFIGS = {0: (100, 1_000), 1: (1_000, 10_000), 2: (10_000, 100_000)}
N_CLASSES = 64
K_SLOTS = 8
SLOTS = 12
UNSCHED_COST = 2_560_000


def _vp(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


@dataclasses.dataclass
class Cluster:
    layout: _lib.Layout
    cap_cpu: np.ndarray
    cap_mem: np.ndarray
    used_cpu: np.ndarray
    used_mem: np.ndarray
    req_cpu: np.ndarray
    req_mem: np.ndarray
    task_class: np.ndarray
    task_job: np.ndarray
    slots: int = SLOTS


@dataclasses.dataclass
class Network:
    layout: _lib.Layout
    n: int
    m2: int
    row_off: np.ndarray
    head: np.ndarray
    cap: np.ndarray
    cost: np.ndarray
    rev: np.ndarray
    supply: np.ndarray
    ec_arc_base: np.ndarray

    @property
    def m(self) -> int:
        return self.m2 // 2

    def checksum(self) -> int:
        lib = _lib.load_library()
        h = 0xCBF29CE484222325
        for a in (self.row_off, self.head, self.cap, self.cost, self.rev, self.supply):
            h = lib.psn_fnv1a64(_vp(a), a.nbytes, h)
        return int(h)


def make_layout(n_machines: int, n_tasks: int) -> _lib.Layout:
    return _lib.Layout(n_machines, N_CLASSES, (n_tasks + 49) // 50, n_tasks, K_SLOTS)


def synth_cluster(config_id: int, seed: int, n_machines: int | None = None,
                  n_tasks: int | None = None) -> Cluster:
    if n_machines is None or n_tasks is None:
        n_machines, n_tasks = CONFIGS[config_id]
    lay = make_layout(n_machines, n_tasks)
    N, P, E = n_machines, n_tasks, N_CLASSES
    cl = Cluster(lay, np.empty(N, np.int64), np.empty(N, np.int64), np.empty(N, np.int64),
                 np.empty(N, np.int64), np.empty(E, np.int64), np.empty(E, np.int64),
                 np.empty(P, np.int32), np.empty(P, np.int32))
    _lib.check(_lib.load_library().psn_synth_cluster(
        config_id, seed, C.byref(lay), _vp(cl.cap_cpu), _vp(cl.cap_mem), _vp(cl.used_cpu),
        _vp(cl.used_mem), _vp(cl.req_cpu), _vp(cl.req_mem), _vp(cl.task_class), _vp(cl.task_job)))
    return cl


def build_network(cl: Cluster) -> Network:
    lib = _lib.load_library()
    n = C.c_int32(0)
    m2 = C.c_int64(0)
    _lib.check(lib.psn_network_size(C.byref(cl.layout), C.byref(n), C.byref(m2)))
    n, m2 = n.value, m2.value
    net = Network(cl.layout, n, m2, np.empty(n + 1, np.int64), np.empty(m2, np.int32),
                  np.empty(m2, np.int32), np.empty(m2, np.int32), np.empty(m2, np.int32),
                  np.empty(n, np.int64), np.empty(cl.layout.n_classes, np.int64))
    _lib.check(lib.psn_build_network(C.byref(cl.layout), cl.slots, UNSCHED_COST, _vp(cl.task_class),
                                     _vp(cl.task_job), _vp(net.row_off), _vp(net.head), _vp(net.cap),
                                     _vp(net.cost), _vp(net.rev), _vp(net.supply), _vp(net.ec_arc_base)))
    return net
