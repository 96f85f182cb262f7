#!/usr/bin/env python3
"""Batch of independent what-if networks (psn-synth-v1 config 1, seed = graph id + 1): every
rank keeps its share resident, the timed region prices and solves all of them
(psn_solve_batch) and extracts placements; placements are all-gathered over RCCL.

  python tools/bench_batch.py --graphs 256 --workers 8 [--config 1]
  python -m torch.distributed.run --nproc-per-node G ... tools/bench_batch.py --graphs 256
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch

import poseidon_amd as pa
from poseidon_amd import batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=256)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=4, help="graphs per rank checked against the CPU oracle")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    mine = batch.shard(args.graphs, world, rank)
    clusters, nets, solvers = [], [], []
    t0 = time.perf_counter()
    for b in mine:
        cl = pa.synth_cluster(args.config, b + 1)
        net = pa.build_network(cl)
        s = pa.Solver(device=local_rank)
        s.upload(net)
        clusters.append(cl); nets.append(net); solvers.append(s)
    upload_s = time.perf_counter() - t0
    P = clusters[0].layout.n_tasks
    place = torch.empty((len(mine), P), dtype=torch.int32, device="cuda")

    parts = {"price": 0.0, "solve": 0.0, "extract": 0.0, "gather": 0.0}

    def step():
        t = time.perf_counter()
        for s, cl, net in zip(solvers, clusters, nets):
            s.eval_costs(cl, net)
        t1 = time.perf_counter(); parts["price"] += t1 - t
        res = batch.solve_batch(solvers, workers=args.workers)
        t2 = time.perf_counter(); parts["solve"] += t2 - t1
        for i, s in enumerate(solvers):
            s.placements_into(place[i].data_ptr())
        t3 = time.perf_counter(); parts["extract"] += t3 - t2
        allp = batch.all_gather_placements(place, args.graphs, world, rank)
        parts["gather"] += time.perf_counter() - t3
        return res, allp

    step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    for k in parts: parts[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, allp = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    el = time.perf_counter() - t0
    if dist:
        t = torch.tensor([el], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); el = float(t.item())
    costs = batch.all_gather_costs([r.total_cost for r in res], args.graphs, world, rank, device="cuda")
    if args.check:
        from oracle import capi, synth
        for i in range(min(args.check, len(mine))):
            rc, tc, _, _, _ = capi.cost_scaling(synth.build_graph(synth.make_cluster(args.config, mine[i] + 1)))
            assert rc == 0 and tc == res[i].total_cost, (mine[i], tc, res[i].total_cost)
    if rank == 0:
        m = nets[0].m
        print(json.dumps({"metric": "batched MCMF scheduling solves", "graphs": args.graphs, "n_gpus": world, "workers_per_gpu": args.workers,
                          "graphs_per_s": round(args.graphs * args.steps / el, 2), "arcs_per_s": round(m * args.graphs * args.steps / el, 1),
                          "ms_per_batch": round(el / args.steps * 1e3, 2), "ms_per_batch_parts": {k: round(v / args.steps * 1e3, 1) for k, v in parts.items()}, "mean_single_solve_ms": round(float(np.mean([r.solve_ms for r in res])), 2),
                          "config": f"psn-synth-v1 config {args.config}, seed = graph id + 1, m = {m} forward arcs per graph",
                          "upload_s_not_timed": round(upload_s, 2), "placed_total": int((allp >= 0).sum().item()),
                          "cost_checksum": int(sum(costs) % (1 << 61)), "oracle_checked_per_rank": min(args.check, len(mine))}), flush=True)
    for s in solvers:
        s.close()
    if dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
