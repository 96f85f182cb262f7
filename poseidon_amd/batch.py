"""Batches of independent flow networks (what-if / rescheduling scenarios) over several
GPUs of one node: one process per GPU, network b goes to rank b mod G, every rank solves
its share with its own solver context, and the only exchange is one all-gather of the
int32 placement vectors (plus one of the int64 costs).  The collective runs on whatever
backend the process group was created with: RCCL ("nccl") for device tensors on MI355X,
gloo for host tensors in CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard(n_graphs: int, world: int, rank: int) -> List[int]:
    """Graph ids solved by `rank` (round robin: equal-sized graphs are balanced this way)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    return list(range(rank, n_graphs, world))


def padded_share(n_graphs: int, world: int) -> int:
    return (n_graphs + world - 1) // world


def graph_order(n_graphs: int, world: int) -> List[int]:
    """Row of a rank-major gathered buffer (world blocks of padded_share rows, rank r's row i being
    graph r + i * world) that holds graph b, for b = 0 .. n_graphs - 1."""
    share = padded_share(n_graphs, world)
    return [(b % world) * share + b // world for b in range(n_graphs)]


def all_gather_placements(local: torch.Tensor, n_graphs: int, world: int, rank: int, group=None) -> torch.Tensor:
    """local: int32 [len(shard), P] placements of this rank's graphs (graph-id order).
    Returns int32 [n_graphs, P] in graph-id order on every rank."""
    if local.dtype != torch.int32 or local.dim() != 2:
        raise ValueError("placements must be int32 [graphs, tasks]")
    share = padded_share(n_graphs, world)
    P = local.shape[1]
    send = torch.full((share, P), -2, dtype=torch.int32, device=local.device)
    send[: local.shape[0]] = local
    if world == 1:
        return send[:n_graphs].clone()
    recv = torch.empty((world * share, P), dtype=torch.int32, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    # rank r holds graphs r, r+G, ...: recv[r*share + i] is graph r + i*G
    recv = recv.view(world, share, P).transpose(0, 1).reshape(share * world, P)
    return recv[:n_graphs].contiguous()


def all_gather_costs(local: Sequence[int], n_graphs: int, world: int, rank: int, device="cpu", group=None) -> List[int]:
    share = padded_share(n_graphs, world)
    send = torch.zeros(share, dtype=torch.int64, device=device)
    if len(local):
        send[: len(local)] = torch.tensor(list(local), dtype=torch.int64, device=device)
    if world == 1:
        return [int(x) for x in send[:n_graphs]]
    recv = torch.empty(world * share, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, share).t().reshape(-1)
    return [int(x) for x in recv[:n_graphs]]


def solve_batch(solvers, warm_start: bool = False, workers: int = 0):
    """Solve the networks resident in `solvers` (distinct Solver objects, each uploaded and
    priced) concurrently through psn_solve_batch; returns one SolveResult per solver."""
    import ctypes as C

    from . import _lib
    from .solver import SolveResult

    lib = _lib.load_library()
    n = len(solvers)
    ctxs = (C.c_void_p * n)(*[s._ctx for s in solvers])
    out = (_lib.Result * n)()
    rc = lib.psn_solve_batch(ctxs, n, _lib.PSN_WARM_START if warm_start else 0, workers, out)
    if rc != _lib.PSN_OK:
        bad = next((s for s, r in zip(solvers, out) if r.status != _lib.PSN_OK), solvers[0] if solvers else None)
        _lib.check(rc, bad._ctx if bad is not None else None)
    return [SolveResult(*(getattr(r, f[0]) for f in _lib.Result._fields_ if not f[0].startswith("reserved"))) for r in out]
