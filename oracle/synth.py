"""ORACLE (test infrastructure, never shipped, never measured as the product).

numpy restatement of the synthetic cluster / flow-network family ``psn-synth-v1``
and of the CPU/memory arc-cost model.  It is written independently of the C++
builder in ``poseidon_amd/csrc`` so that the two can be compared array by array.

Parity status: **parity unpinned**.  The reference repository holds no flow-graph
builder, no cost model and no solver (Firmament is an absent submodule,
/root/reference/.gitmodules:1-3; the solver runs in a separate container,
/root/reference/deploy/firmament-deployment.yaml:29-30), so none of its tests pin
these arrays.  Units follow the reference's own conversions: CPU in millicores
(/root/reference/pkg/k8sclient/podwatcher.go:382-386) and memory in Kb
(/root/reference/pkg/k8sclient/nodewatcher.go:114-117); one PU per machine
(/root/reference/pkg/k8sclient/nodewatcher.go:316-336) is collapsed into the
machine vertex.

Vertex ids: 0 sink | 1..N machines | N+1..N+E task classes (EC) |
then J unscheduled aggregators | then P tasks.
Forward arcs, in global order:
  type 1  task -> EC(shape)          cap 1      cost 0            (by task)
  type 2  task -> unsched(job)       cap 1      cost C_UNSCHED    (by task)
  type 3  EC e -> machine m, slot k  cap 0/1    cost model        (by e, m, k)
  type 4  machine -> sink            cap slots  cost 0            (by machine)
  type 5  unsched j -> sink          cap |job|  cost 0            (by j)
Residual CSR: arcs grouped by tail; inside a tail ordered by (type, forward index)
of the underlying forward arc, whether the entry is the forward arc or the
reverse residual of an incoming arc.
"""
from __future__ import annotations

import dataclasses
import numpy as np

MAGIC = 0x506F736569646F6E  # "Poseidon"
GAMMA = 0x9E3779B97F4A7C15
C_UNSCHED = 2_560_000
CPU_CHOICES = np.array([4000, 8000, 16000, 32000, 64000], dtype=np.int64)
GIB_PER_CORE = np.array([2, 4, 8], dtype=np.int64)
REQ_CPU = np.array([10, 50, 100, 250, 500, 1000, 2000, 4000], dtype=np.int64)
REQ_MEM_MI = np.array([10, 64, 128, 256, 512, 1024, 2048, 4096], dtype=np.int64)
SLOTS = 12
K_SLOTS = 8
E_SHAPES = 64

CONFIGS = {  # config_id -> (N machines, P pods)
    0: (100, 1_000),
    1: (1_000, 10_000),
    2: (10_000, 100_000),
}


def splitmix64_stream(seed: int, count: int) -> np.ndarray:
    """First ``count`` outputs of splitmix64 started at ``seed`` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


@dataclasses.dataclass
class Cluster:
    n_machines: int
    n_pods: int
    n_jobs: int
    cap_cpu: np.ndarray
    cap_mem: np.ndarray
    used_cpu: np.ndarray
    used_mem: np.ndarray
    req_cpu: np.ndarray   # [E]
    req_mem: np.ndarray   # [E]
    pod_shape: np.ndarray  # [P] int32
    pod_job: np.ndarray    # [P] int32
    slots: int = SLOTS
    k_slots: int = K_SLOTS
    machine_out: np.ndarray | None = None  # [N] bool: machine taken out (machine -> sink capacity 0)
    allowed: np.ndarray | None = None      # [E, N] bool: label selectors of class e admit machine m


def make_cluster(config_id: int, seed: int, n_machines: int | None = None,
                 n_pods: int | None = None) -> Cluster:
    """Draw order: per machine (cpu, GiB/core, used_cpu permille, used_mem permille),
    then per pod (shape, job).  Every draw is ``next() % k``."""
    if n_machines is None or n_pods is None:
        n_machines, n_pods = CONFIGS[config_id]
    N, P = int(n_machines), int(n_pods)
    J = (P + 49) // 50
    s = (MAGIC ^ (config_id << 32) ^ seed) & 0xFFFFFFFFFFFFFFFF
    r = splitmix64_stream(s, 4 * N + 2 * P)
    rm = r[: 4 * N].reshape(N, 4)
    cap_cpu = CPU_CHOICES[(rm[:, 0] % np.uint64(5)).astype(np.int64)]
    g = GIB_PER_CORE[(rm[:, 1] % np.uint64(3)).astype(np.int64)]
    cap_mem = (cap_cpu // 1000) * g * 1_048_576
    u = (rm[:, 2] % np.uint64(601)).astype(np.int64)
    v = (rm[:, 3] % np.uint64(601)).astype(np.int64)
    used_cpu = cap_cpu * u // 1000
    used_mem = cap_mem * v // 1000
    rp = r[4 * N:].reshape(P, 2)
    pod_shape = (rp[:, 0] % np.uint64(E_SHAPES)).astype(np.int32)
    pod_job = (rp[:, 1] % np.uint64(J)).astype(np.int32)
    e = np.arange(E_SHAPES)
    req_cpu = REQ_CPU[e // 8]
    req_mem = REQ_MEM_MI[e % 8] * 1024
    return Cluster(N, P, J, cap_cpu, cap_mem, used_cpu, used_mem, req_cpu, req_mem,
                   pod_shape, pod_job)


@dataclasses.dataclass
class Churn:
    """One round of cluster churn (the incremental configuration): which pods left and what
    replaced them, which machines report new usage, which machines went away."""
    pods_replaced: np.ndarray    # [P//20] pod indices, distinct
    new_shape: np.ndarray
    new_job: np.ndarray
    machines_restat: np.ndarray  # [N//20] machine indices, distinct
    new_used_cpu: np.ndarray
    new_used_mem: np.ndarray
    machines_out: np.ndarray     # [N//100] machine indices, distinct


def _distinct(draw, count, modulus):
    seen, out = set(), []
    while len(out) < count:
        x = int(draw() % modulus)
        if x not in seen:
            seen.add(x)
            out.append(x)
    return np.array(out, dtype=np.int64)


def churn(cl: Cluster, config_id: int, seed: int, round_no: int = 1):
    """5 % of the pods leave and as many arrive with fresh (shape, job) draws, 5 % of the
    machines get fresh usage, 1 % of the machines are taken out.  Stream seed:
    MAGIC ^ (config_id << 32) ^ seed ^ (0xC3 << 56) ^ (round_no << 40); draw order: replaced pods
    (rejecting repeats), then (shape, job) per replaced pod, then re-measured machines, then
    (used_cpu permille, used_mem permille) per such machine, then removed machines.
    Returns (new Cluster, Churn)."""
    N, P, J = cl.n_machines, cl.n_pods, cl.n_jobs
    s = (MAGIC ^ (config_id << 32) ^ seed ^ (0xC3 << 56) ^ (round_no << 40)) & 0xFFFFFFFFFFFFFFFF
    stream = iter(splitmix64_stream(s, 4 * (P // 20 + N // 20 + N // 100) + 4096).tolist())
    draw = lambda: next(stream)
    pods = _distinct(draw, P // 20, P)
    new_shape = np.empty(pods.shape[0], np.int32)
    new_job = np.empty(pods.shape[0], np.int32)
    for i in range(pods.shape[0]):
        new_shape[i] = draw() % E_SHAPES
        new_job[i] = draw() % J
    mach = _distinct(draw, N // 20, N)
    nuc = np.empty(mach.shape[0], np.int64)
    num = np.empty(mach.shape[0], np.int64)
    for i, m in enumerate(mach):
        nuc[i] = int(cl.cap_cpu[m]) * (draw() % 601) // 1000
        num[i] = int(cl.cap_mem[m]) * (draw() % 601) // 1000
    out = _distinct(draw, N // 100, N)
    new = dataclasses.replace(cl, pod_shape=cl.pod_shape.copy(), pod_job=cl.pod_job.copy(),
                              used_cpu=cl.used_cpu.copy(), used_mem=cl.used_mem.copy(),
                              machine_out=(np.zeros(N, bool) if cl.machine_out is None else cl.machine_out.copy()))
    new.pod_shape[pods] = new_shape
    new.pod_job[pods] = new_job
    new.used_cpu[mach] = nuc
    new.used_mem[mach] = num
    new.machine_out[out] = True
    return new, Churn(pods, new_shape, new_job, mach, nuc, num, out)


def eval_costs(cl: Cluster):
    """CPU/memory arc-cost model for the EC -> machine slot arcs.

    Returns (cost[E,N,K] int32, cap[E,N,K] int32).  Pure int64 arithmetic:
      fit  = min((cap_cpu-used_cpu)//req_cpu, (cap_mem-used_mem)//req_mem)
      cap  = 1 if k < fit else 0
      cost = 1000*(used_cpu+(k+1)*req_cpu)//cap_cpu + 1000*(used_mem+(k+1)*req_mem)//cap_mem
    """
    K = cl.k_slots
    rc = cl.req_cpu[:, None, None]
    rm = cl.req_mem[:, None, None]
    cc = cl.cap_cpu[None, :, None]
    cm = cl.cap_mem[None, :, None]
    uc = cl.used_cpu[None, :, None]
    um = cl.used_mem[None, :, None]
    k1 = np.arange(1, K + 1, dtype=np.int64)[None, None, :]
    fit = np.minimum((cc - uc) // rc, (cm - um) // rm)
    cap = (k1 - 1 < fit).astype(np.int32)
    if cl.allowed is not None:
        cap = cap * cl.allowed[:, :, None].astype(np.int32)
    cost = (1000 * (uc + k1 * rc)) // cc + (1000 * (um + k1 * rm)) // cm
    return cost.astype(np.int32), cap


@dataclasses.dataclass
class Graph:
    n: int
    m: int                 # forward arcs
    row_off: np.ndarray    # int64[n+1]
    head: np.ndarray       # int32[2m]
    cap: np.ndarray        # int32[2m] residual capacity (reverse arcs start at 0)
    cost: np.ndarray       # int32[2m] (reverse = -cost)
    rev: np.ndarray        # int32[2m]
    supply: np.ndarray     # int64[n]
    fwd_pos: np.ndarray    # int64[m] position of forward arc i in the residual arrays
    n_machines: int = 0
    n_ec: int = 0
    n_jobs: int = 0
    n_pods: int = 0


def _csr_from_forward(n, tail, head, cap, cost, typ, supply):
    """Residual CSR from a forward arc list, ordering entries inside a tail by
    (type, forward index)."""
    m = tail.shape[0]
    idx = np.arange(m, dtype=np.int64)
    # entry list: forward entries then reverse entries
    e_tail = np.concatenate([tail, head]).astype(np.int64)
    e_head = np.concatenate([head, tail]).astype(np.int32)
    e_cap = np.concatenate([cap, np.zeros(m, dtype=np.int32)]).astype(np.int32)
    e_cost = np.concatenate([cost, -cost]).astype(np.int32)
    e_typ = np.concatenate([typ, typ]).astype(np.int64)
    e_fidx = np.concatenate([idx, idx])
    order = np.lexsort((e_fidx, e_typ, e_tail))
    pos = np.empty(2 * m, dtype=np.int64)
    pos[order] = np.arange(2 * m, dtype=np.int64)
    row_off = np.zeros(n + 1, dtype=np.int64)
    np.add.at(row_off, e_tail + 1, 1)
    row_off = np.cumsum(row_off)
    rev = np.empty(2 * m, dtype=np.int32)
    rev[pos[:m]] = pos[m:].astype(np.int32)
    rev[pos[m:]] = pos[:m].astype(np.int32)
    return Graph(n, m, row_off, e_head[order], e_cap[order], e_cost[order], rev,
                 supply.astype(np.int64), pos[:m].copy())


def build_graph(cl: Cluster) -> Graph:
    N, P, J, E, K = cl.n_machines, cl.n_pods, cl.n_jobs, E_SHAPES, cl.k_slots
    n = 1 + N + E + J + P
    v_m = 1
    v_ec = 1 + N
    v_un = 1 + N + E
    v_t = 1 + N + E + J
    t = np.arange(P, dtype=np.int64)
    cost3, cap3 = eval_costs(cl)
    e3, m3, _k3 = np.meshgrid(np.arange(E), np.arange(N), np.arange(K), indexing="ij")
    job_size = np.bincount(cl.pod_job, minlength=J).astype(np.int32)
    tail = np.concatenate([v_t + t, v_t + t, (v_ec + e3).ravel(),
                           v_m + np.arange(N), v_un + np.arange(J)])
    head = np.concatenate([v_ec + cl.pod_shape.astype(np.int64),
                           v_un + cl.pod_job.astype(np.int64), (v_m + m3).ravel(),
                           np.zeros(N, dtype=np.int64), np.zeros(J, dtype=np.int64)])
    sink_cap = np.full(N, cl.slots, np.int32)
    if cl.machine_out is not None:
        sink_cap[cl.machine_out] = 0
    cap = np.concatenate([np.ones(P, np.int32), np.ones(P, np.int32), cap3.ravel(),
                          sink_cap, job_size])
    cost = np.concatenate([np.zeros(P, np.int32), np.full(P, C_UNSCHED, np.int32),
                           cost3.ravel(), np.zeros(N, np.int32), np.zeros(J, np.int32)])
    typ = np.concatenate([np.full(P, 1), np.full(P, 2), np.full(E * N * K, 3),
                          np.full(N, 4), np.full(J, 5)])
    supply = np.zeros(n, dtype=np.int64)
    supply[v_t:] = 1
    supply[0] = -P
    g = _csr_from_forward(n, tail, head, cap, cost, typ, supply)
    g.n_machines, g.n_ec, g.n_jobs, g.n_pods = N, E, J, P
    return g


def graph_from_arcs(n, tail, head, cap, cost, supply) -> Graph:
    """Residual CSR for an arbitrary forward arc list (all arcs one type)."""
    tail = np.asarray(tail, dtype=np.int64)
    head = np.asarray(head, dtype=np.int64)
    return _csr_from_forward(n, tail, head, np.asarray(cap, np.int32),
                             np.asarray(cost, np.int32),
                             np.ones(tail.shape[0], np.int64), np.asarray(supply, np.int64))


def fnv1a64(*arrays: np.ndarray) -> int:
    """FNV-1a-64 over the concatenated little-endian bytes of the arrays."""
    h = 0xCBF29CE484222325
    prime = 0x100000001B3
    mask = 0xFFFFFFFFFFFFFFFF
    for a in arrays:
        for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
            h = ((h ^ b) * prime) & mask
    return h


def csr_checksum(g: Graph) -> int:
    return fnv1a64(g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
