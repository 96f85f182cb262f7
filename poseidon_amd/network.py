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


# ---- churn rounds of the incremental configuration (psn-synth-v1, configs[3])
_MAGIC = 0x506F736569646F6E
_GAMMA = 0x9E3779B97F4A7C15
_M64 = 0xFFFFFFFFFFFFFFFF


def _splitmix64(seed: int):
    """Generator of splitmix64 outputs started at `seed`."""
    x = seed & _M64
    while True:
        x = (x + _GAMMA) & _M64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        yield z ^ (z >> 31)


@dataclasses.dataclass
class ChurnRound:
    """What one round of churn does to a cluster: tasks that leave and what replaces them, machines
    that report new usage, machines that go away (indices into the cluster's arrays)."""
    tasks_replaced: np.ndarray
    new_class: np.ndarray
    new_job: np.ndarray
    machines_restat: np.ndarray
    new_used_cpu: np.ndarray
    new_used_mem: np.ndarray
    machines_out: np.ndarray


def churn_round(cl: Cluster, config_id: int, seed: int, round_no: int = 1) -> ChurnRound:
    """5 % of the tasks leave and as many arrive with fresh (class, job) draws, 5 % of the machines
    report fresh usage (0..600 permille of capacity), 1 % of the machines go away.  Stream seed
    MAGIC ^ (config_id << 32) ^ seed ^ (0xC3 << 56) ^ (round_no << 40); draws in that order, repeats
    among the chosen tasks / machines rejected."""
    N, P, J, E = cl.layout.n_machines, cl.layout.n_tasks, cl.layout.n_jobs, cl.layout.n_classes
    g = _splitmix64(_MAGIC ^ (config_id << 32) ^ seed ^ (0xC3 << 56) ^ (round_no << 40))

    def distinct(count, modulus):
        seen, out = set(), []
        while len(out) < count:
            x = next(g) % modulus
            if x not in seen:
                seen.add(x)
                out.append(x)
        return np.array(out, dtype=np.int64)

    tasks = distinct(P // 20, P)
    new_class = np.empty(tasks.shape[0], np.int32)
    new_job = np.empty(tasks.shape[0], np.int32)
    for i in range(tasks.shape[0]):
        new_class[i] = next(g) % E
        new_job[i] = next(g) % J
    mach = distinct(N // 20, N)
    nuc = np.empty(mach.shape[0], np.int64)
    num = np.empty(mach.shape[0], np.int64)
    for i, m in enumerate(mach):
        nuc[i] = int(cl.cap_cpu[m]) * (next(g) % 601) // 1000
        num[i] = int(cl.cap_mem[m]) * (next(g) % 601) // 1000
    out = distinct(N // 100, N)
    return ChurnRound(tasks, new_class, new_job, mach, nuc, num, out)
