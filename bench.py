#!/usr/bin/env python3
"""Benchmark of the scheduling hot path on MI355X: arc pricing (K1) + cost-scaling
min-cost-flow solve + placement extraction (K5) on the 10k-machine / 100k-pod synthetic
network (BASELINE.json configs[2]), inputs resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full pass (K1 + cold solve + K5) over one network per GPU; with N > 1
every rank solves its own independent network (seed = 1 + rank, weak scaling) and the
placements are all-gathered over RCCL.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import poseidon_amd as pa

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E


def algorithmic_bytes(r, kind):
    """Algorithmic bytes per unit of work: 20 B per arc scan (head 4 + cap 4 + cost 4 +
    gathered 8-byte price or distance), 36 B per push, 32 B per vertex visit, 8 B per relabel."""
    if kind == "gu_expand":
        return 20 * r.gu_arc_scans
    if kind == "push":
        return 20 * r.arc_scans + 36 * r.pushes + 32 * r.vertex_visits
    return r.algorithmic_bytes


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in KiB per dispatch; FETCH_SIZE doubled, as gfx950 tallies 128-byte
    read requests as 64 bytes).  None when no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_write.csv")))
    if not files:
        return None, None
    for row in csv.DictReader(open(files[-1])):
        if row["kernel"].endswith(kernel):
            f, w = float(row["FETCH_SIZE_per_dispatch"]), float(row["WRITE_SIZE_per_dispatch"])
            return int((2 * f + w) * 1024), os.path.basename(files[-1])
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, help="psn-synth-v1 config id (2 = 10k machines / 100k pods)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-graphs", type=int, default=64,
                    help="what-if networks (config 1) per GPU for the batched-throughput figure reported beside the headline; 0: skip")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libpsn has no CPU path")
    # PSN_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, collectives go through host tensors); the measured configuration is nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("PSN_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where collective operands live
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    seed = 1 + rank
    cl = pa.synth_cluster(args.config, seed)
    net = pa.build_network(cl)
    P = cl.layout.n_tasks
    solver = pa.Solver(device=local_rank)
    t0 = time.perf_counter()
    solver.upload(net)
    upload_s = time.perf_counter() - t0
    place = torch.empty(P, dtype=torch.int32, device="cuda")
    gathered = torch.empty(world * P, dtype=torch.int32, device=cdev) if world > 1 else None

    def step():
        solver.eval_costs(cl, net)             # K1: prices the class->machine slot arcs, resets the flow
        r = solver.solve()                     # cost-scaling push-relabel, cold start
        solver.placements_into(place.data_ptr())  # K5
        if world > 1:
            dist.all_gather_into_tensor(gathered, place if backend == "nccl" else place.cpu())
        return r

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    res = []
    for _ in range(args.steps):
        res.append(step())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        costs = torch.tensor([res[-1].total_cost], dtype=torch.int64, device=cdev)
        allc = [torch.empty_like(costs) for _ in range(world)]
        dist.all_gather(allc, costs)
        all_costs = [int(c.item()) for c in allc]
        placed_all = int((gathered >= 0).sum().item())
    else:
        all_costs = [res[-1].total_cost]
        placed_all = int((place >= 0).sum().item())

    # ---- beside the headline: the batch of independent what-if networks (1 000 machines / 10 000 pods each),
    # a fixed number per GPU, solved in lockstep groups; placements all-gathered like the headline's
    batch_out = None
    if args.batch_graphs > 0:
        from poseidon_amd import batch as pbatch
        B = args.batch_graphs
        bcl, bnet, bsol = [], [], []
        for b in range(B):
            c1 = pa.synth_cluster(1, 1 + rank * B + b)
            n1 = pa.build_network(c1)
            s1 = pa.Solver(device=local_rank)
            s1.upload(n1)
            bcl.append(c1); bnet.append(n1); bsol.append(s1)
        P1 = bcl[0].layout.n_tasks
        bplace = torch.empty((B, P1), dtype=torch.int32, device="cuda")
        bgath = torch.empty((world * B, P1), dtype=torch.int32, device=cdev) if world > 1 else None

        def bstep():
            for s1, c1, n1 in zip(bsol, bcl, bnet):
                s1.eval_costs(c1, n1)
            rs = pbatch.solve_batch(bsol)
            for i, s1 in enumerate(bsol):
                s1.placements_into(bplace[i].data_ptr())
            if world > 1:
                dist.all_gather_into_tensor(bgath, bplace if backend == "nccl" else bplace.cpu())
            return rs

        bstep()
        fence()
        tb = time.perf_counter()
        bsteps = 2
        for _ in range(bsteps):
            brs = bstep()
        fence()
        bel = time.perf_counter() - tb
        if world > 1:
            t = torch.tensor([bel], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bel = float(t.item())
        batch_out = {"workload": f"psn-synth-v1 config 1 ({bnet[0].m} forward arcs each), {B} networks per GPU, lockstep groups",
                     "graphs": B * world, "graphs_per_s": round(B * world * bsteps / bel, 1),
                     "arcs_per_s": round(bnet[0].m * B * world * bsteps / bel, 1), "ms_per_batch": round(bel / bsteps * 1e3, 2),
                     "cost_checksum_rank0": int(sum(r_.total_cost for r_ in brs) % (1 << 61))}
        for s1 in bsol:
            s1.close()

    ms_per_step = elapsed / args.steps * 1e3
    arcs_per_s = net.m * world * args.steps / elapsed
    r = res[-1]
    solve_ms = float(np.median([x.solve_ms for x in res]))

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: one extra solve with HIP-event timing of every launch
        roof = None
        with pa.Solver(device=local_rank, time_kernels=True) as ts:
            ts.upload(net)
            ts.eval_costs(cl, net)
            tr = ts.solve()
            kt = ts.kernel_times()
        dom = max(("push", "gu_expand"), key=lambda k: kt[k][0])
        dom_ms, dom_calls = kt[dom]
        if dom_ms > 0 and dom_calls > 0:
            bytes_total = algorithmic_bytes(tr, dom)
            achieved = bytes_total / (dom_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic("k_" + dom)
            roof = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_unit": "bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 from " + str(traffic_src),
                    "launches": dom_calls, "avg_launch_us": round(dom_ms * 1e3 / dom_calls, 2),
                    "algorithmic_bytes_per_launch": int(bytes_total / dom_calls),
                    "kernel_ms": {k: round(v[0], 2) for k, v in kt.items()}}
        # ---- CPU baseline: the oracle's single-thread cost-scaling solver on the same network
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # timed on rank 0 of the single-GPU run only
            from oracle import capi, synth
            ocl = synth.make_cluster(args.config, seed)
            og = synth.build_graph(ocl)
            times, ccost = [], None
            for _ in range(3 if args.config >= 2 else 5):
                rc, ccost, _, _, st = capi.cost_scaling(og)
                times.append(st.solve_ms)
            cpu_ms = float(np.median(times))
            cpu = {"value": round(og.m / (cpu_ms * 1e-3), 1), "unit": "arcs/s", "cores": 1, "kind": "port",
                   "solve_ms": round(cpu_ms, 1), "host_cores_total": os.cpu_count(), "optimal_cost": int(ccost),
                   "sample": f"full psn-synth-v1 config {args.config} seed {seed} network, single-thread cost-scaling "
                             f"push-relabel (oracle/psn_oracle.c), median of {len(times)} solves"}
            assert ccost == all_costs[0], (ccost, all_costs)
        out = {
            "metric": "MCMF scheduling solve throughput (forward arcs of the input network per second; K1 pricing + cold "
                      "cost-scaling solve + placement extraction, 10k-node/100k-pod graph)",
            "value": round(arcs_per_s, 1), "unit": "arcs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic" if backend == "nccl" else f"synthetic (rehearsal over {backend}: not a measurement)",
            "config": {"workload": f"psn-synth-v1 config {args.config}: {cl.layout.n_machines} machines / {P} pods, "
                                   f"n={net.n} vertices, m={net.m} forward arcs, one network per GPU (seed = 1 + rank)",
                       "graphs_per_step": world, "alpha": 16, "rounds_per_global_update": 12},
            "solve_ms": round(solve_ms, 3), "optimal_cost": all_costs, "tasks_placed": placed_all,
            "upload_s_not_timed": round(upload_s, 4),
            "counters": {"arc_scans": r.arc_scans, "gu_arc_scans": r.gu_arc_scans, "sat_arc_scans": r.sat_arc_scans,
                         "pushes": r.pushes, "relabels": r.relabels, "vertex_visits": r.vertex_visits, "rounds": r.rounds,
                         "gu_iterations": r.gu_iterations, "global_updates": r.global_updates, "refines": r.refines,
                         "launches": r.launches},
            "solve_algorithmic_GBps": round(r.algorithmic_bytes / (solve_ms * 1e-3) / 1e9, 1),
            "roofline": roof, "cpu_baseline": cpu, "what_if_batch": batch_out,
        }
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
