#!/usr/bin/env python3
"""Benchmark of the scheduling hot path on MI355X: arc pricing (K1) + cost-scaling
min-cost-flow solve + placement extraction (K5) on the 10k-machine / 100k-pod synthetic
network (BASELINE.json configs[2]), inputs resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full pass (K1 + cold solve + K5) over one network per GPU; with N > 1
every rank solves its own independent network (seed = 1 + rank, weak scaling) and the
placements are all-gathered over RCCL.  Rank 0 prints ONE JSON line.  Beside the headline:
  what_if_batch  BASELINE.json configs[4]: 256 independent 1k-machine / 10k-pod networks in
                 total, network b on rank b mod N (strong scaling), solved in lockstep groups,
                 placements all-gathered;
  c3             configs[3] on one GPU: scheduler-service rounds at the 1k and 10k sizes, cold
                 and warm after 5 % churn, with the parts of each round;
  roofline       the dominant kernel against the HBM roofline, in both currencies;
  cpu_baseline   the CPU cost-scaling solver on the same network, on the host cores.
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


def profile_files():
    """The latest committed profile set: (PMC pass summary, kernel-trace statistics of the same run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_write.csv")))
    if not files:
        return None, None
    stats = files[-1].replace("pmc_fetch_write.csv", "bench_kernel_stats.csv")
    return files[-1], (stats if os.path.exists(stats) else None)


def check_profile_matches_library():
    """The committed counter pass must come from a build that launches the kernels this build
    launches: a profile of another solver would put another run's bytes over this run's time.
    Raises SystemExit naming what differs (PSN_BENCH_ALLOW_STALE_PROFILE=1: report it instead)."""
    import csv
    pmc, _ = profile_files()
    if pmc is None:
        return "no profile committed"
    want = set(pa.load_library().psn_kernel_list().decode().split(","))
    have = {row["kernel"].split("::")[-1] for row in csv.DictReader(open(pmc)) if row["kernel"]}
    missing = sorted(want - have)
    if missing:
        msg = f"{os.path.basename(pmc)} has no counters for {missing}: it was taken from another build of the solver (psn_version {pa.load_library().psn_version()})"
        if os.environ.get("PSN_BENCH_ALLOW_STALE_PROFILE") != "1":
            raise SystemExit("bench.py: stale profile: " + msg + "; run tools/measure.sh, commit the summaries under profiles/ (or set PSN_BENCH_ALLOW_STALE_PROFILE=1)")
        return msg
    return None


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in KiB per dispatch; FETCH_SIZE doubled, as gfx950 tallies 128-byte
    read requests as 64 bytes), with where the figure comes from: the profile's file, its dispatch
    count for the kernel and, from the kernel trace of the same profiled run, that run's launch
    count and average duration.  None when no profile is committed."""
    import csv
    pmc, stats = profile_files()
    if pmc is None:
        return None, None
    src = {"file": os.path.basename(pmc)}
    traffic = None
    for row in csv.DictReader(open(pmc)):
        if row["kernel"].endswith(kernel):
            f, w = float(row["FETCH_SIZE_per_dispatch"]), float(row["WRITE_SIZE_per_dispatch"])
            traffic = int((2 * f + w) * 1024)
            src["pmc_dispatches"] = int(float(row["dispatches"]))
    if stats:
        for row in csv.DictReader(open(stats)):
            if row["Name"].split("(")[0].endswith(kernel):
                src["trace_file"] = os.path.basename(stats)
                src["trace_launches"] = int(row["Calls"])
                src["trace_avg_launch_us"] = round(float(row["AverageNs"]) / 1e3, 2)
    return traffic, src


def machine_uuid(i):
    return f"00000000-0000-4000-8000-{i:012d}"


def service_rounds(config_id, seed, device, churn_rounds=2):
    """configs[3]: the cluster fed to the scheduler service RPC by RPC, one cold Schedule, then
    rounds of 5 % churn applied in place and warm-started.  Every task is re-optimised in every
    round (reschedule_running), so each round's optimum is that of the churned cluster built from
    scratch; the last churned cluster is also solved cold in a fresh context for the comparison."""
    from poseidon_amd import network
    from poseidon_amd.firmament import ChangeType
    cl = pa.synth_cluster(config_id, seed)
    N, P = cl.layout.n_machines, cl.layout.n_tasks
    task_class, task_job = cl.task_class.copy(), cl.task_job.copy()
    used_cpu, used_mem = cl.used_cpu.copy(), cl.used_mem.copy()
    out_mask = np.zeros(N, bool)
    rounds = []

    def part(st, wall_ms, n_deltas, moved=None):
        d = {"round_ms": round(st.total_ms, 2), "wall_ms": round(wall_ms, 2), "build_ms": round(st.build_ms, 2), "upload_ms": round(st.upload_ms, 2),
             "update_ms": round(st.update_ms, 2), "solve_ms": round(st.solve_ms, 2), "extract_ms": round(st.extract_ms, 2),
             "warm": int(st.last_round_warm), "warm_declined": int(st.solve_warm_declined), "deltas": n_deltas, "cost": int(st.total_cost),
             "patched_arcs": int(st.patched_arcs), "cap_updates": int(st.cap_updates), "supply_updates": int(st.supply_updates),
             # why a warm solve costs what it does: the epsilon its first phase ran at, the units of excess on the
             # work list when its first round started, and the replays (one global update + 14 rounds each) it took
             "solve": {"device_ms": round(st.solve_device_ms, 2), "first_eps_log2": int(st.solve_first_eps_log2), "first_units": int(st.solve_first_units),
                       "phases": int(st.solve_refines), "replays": int(st.solve_replays), "last_phase_replays": int(st.solve_last_phase_replays),
                       "rounds": int(st.solve_rounds), "global_updates": int(st.solve_global_updates), "pushes": int(st.solve_pushes)}}
        if moved is not None:
            d["migrated_or_preempted"] = moved
        return d

    with pa.SchedulerService(device=device, reschedule_running=True) as s:
        for m in range(N):
            s.NodeAdded(machine_uuid(m), int(cl.cap_cpu[m]), int(cl.cap_mem[m]))
            s.AddNodeStats(machine_uuid(m), int(used_cpu[m]), int(used_mem[m]))
        uids = np.arange(1, P + 1, dtype=np.int64)
        for t in range(P):
            e = int(task_class[t])
            s.TaskSubmitted(int(uids[t]), f"job-{int(task_job[t]):06d}", int(cl.req_cpu[e]), int(cl.req_mem[e]))
        t0 = time.perf_counter()
        d_uid, d_type, _ = s.schedule_arrays()  # the reply as arrays: wall time of the call = the round
        rounds.append(("cold", part(s.stats(), (time.perf_counter() - t0) * 1e3, int(d_uid.shape[0]))))
        allocs_before = pa.load_library().psn_debug_alloc_count()
        nxt = P + 1
        for rnd in range(1, churn_rounds + 1):
            ch = network.churn_round(cl, config_id, seed, rnd)
            for i, t in enumerate(ch.tasks_replaced):
                s.TaskRemoved(int(uids[t]))
                uids[t] = nxt + i
                task_class[t], task_job[t] = ch.new_class[i], ch.new_job[i]
                e = int(ch.new_class[i])
                s.TaskSubmitted(int(uids[t]), f"job-{int(ch.new_job[i]):06d}", int(cl.req_cpu[e]), int(cl.req_mem[e]))
            nxt += ch.tasks_replaced.shape[0]
            for i, m in enumerate(ch.machines_restat):
                if not out_mask[m]:
                    s.AddNodeStats(machine_uuid(int(m)), int(ch.new_used_cpu[i]), int(ch.new_used_mem[i]))
                used_cpu[m], used_mem[m] = ch.new_used_cpu[i], ch.new_used_mem[i]
            for m in ch.machines_out:
                if not out_mask[m]:
                    s.NodeRemoved(machine_uuid(int(m)))
                out_mask[m] = True
            t0 = time.perf_counter()
            d_uid, d_type, _ = s.schedule_arrays()
            wall = (time.perf_counter() - t0) * 1e3
            moved = int(np.isin(d_type, (int(ChangeType.MIGRATE), int(ChangeType.PREEMPT))).sum())
            rounds.append((f"churn{rnd}", part(s.stats(), wall, int(d_uid.shape[0]), moved)))
        # one more round in which only pods change (5 % replaced, no machine re-priced, none taken out): no arc
        # cost moves, so the kept prices are violated nowhere but at the arcs of the new tasks
        ch = network.churn_round(cl, config_id, seed, churn_rounds + 1)
        for i, t in enumerate(ch.tasks_replaced):
            s.TaskRemoved(int(uids[t]))
            uids[t] = nxt + i
            task_class[t], task_job[t] = ch.new_class[i], ch.new_job[i]
            e = int(ch.new_class[i])
            s.TaskSubmitted(int(uids[t]), f"job-{int(ch.new_job[i]):06d}", int(cl.req_cpu[e]), int(cl.req_mem[e]))
        t0 = time.perf_counter()
        d_uid, d_type, _ = s.schedule_arrays()
        wall = (time.perf_counter() - t0) * 1e3
        rounds.append(("churn_pods_only", part(s.stats(), wall, int(d_uid.shape[0]), int(np.isin(d_type, (int(ChangeType.MIGRATE), int(ChangeType.PREEMPT))).sum()))))
        allocs_warm = pa.load_library().psn_debug_alloc_count() - allocs_before
    # the last churned cluster from scratch: same optimum, and what a cold solve of it costs
    cl2 = pa.network.Cluster(cl.layout, cl.cap_cpu, cl.cap_mem, used_cpu, used_mem, cl.req_cpu, cl.req_mem, task_class, task_job, cl.slots)
    net2 = pa.build_network(cl2)
    E, K = cl.layout.n_classes, cl.layout.k_slots
    with pa.Solver(device=device) as sv:
        sv.upload(net2)
        sv.eval_costs(cl2, net2)
        gone = np.nonzero(out_mask)[0]
        if gone.shape[0]:
            arcs = np.array([net2.row_off[1 + m] + E * K for m in gone], np.int64)  # machine -> sink
            sv.update_caps(arcs, np.zeros(arcs.shape[0], np.int32))
        rc = sv.solve()
    out = {name: p for name, p in rounds}
    out["cold_solve_of_last_churned_cluster_ms"] = round(rc.solve_ms, 2)
    out["warm_equals_cold_cost"] = bool(rc.total_cost == rounds[-1][1]["cost"])
    out["allocations_during_warm_rounds"] = int(allocs_warm)
    # churn rounds (applied in place; "warm" says whether the solver resumed from the kept state or declined it) against
    # the cold solve of the same cluster
    warm_ms = float(np.mean([p["solve"]["device_ms"] for name, p in rounds[1:1 + churn_rounds]]))
    out["warm_over_cold_solve"] = round(warm_ms / max(rc.solve_ms, 1e-9), 3)
    out["churn_rounds_that_declined_the_kept_state"] = int(sum(p["warm_declined"] for name, p in rounds[1:1 + churn_rounds]))
    out["cold_solve_of_last_churned_cluster"] = {"device_ms": round(rc.solve_ms, 2), "first_eps_log2": rc.first_eps_log2, "first_units": rc.first_units,
                                                  "phases": rc.refines, "replays": rc.replays, "last_phase_replays": rc.last_phase_replays,
                                                  "rounds": rc.rounds, "global_updates": rc.global_updates, "pushes": rc.pushes}
    out["why_warm_is_not_cheaper"] = ("a solve is its replays (a global update over the whole network + 14 rounds, about half a millisecond at "
                                      "10 000 machines); re-priced machines with free slots are genuinely cheaper or dearer by hundreds of cost "
                                      "units, so the kept prices are violated by that much, the scaling resumes near the epsilon a cold solve "
                                      "starts at and runs the same phases; it moves a third of the pushes in as many replays or more.  A kept "
                                      "state whose first epsilon is within a power of two of a cold solve's is therefore declined and the "
                                      "round solved cold on the resident network (warm_declined).  With only pods replaced (churn_pods_only) "
                                      "no cost moves and the first epsilon is small")
    return out, (cl2, out_mask)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, help="psn-synth-v1 config id (2 = 10k machines / 100k pods)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-graphs", type=int, default=256,
                    help="what-if networks (config 1) IN TOTAL for the batched figure (network b on rank b mod N); 0: skip")
    ap.add_argument("--batch-check", type=int, default=4, help="networks per rank of the batch checked against the CPU oracle")
    ap.add_argument("--no-c3", action="store_true", help="skip the scheduler-service rounds (configs[3])")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libpsn has no CPU path")
    stale_profile = check_profile_matches_library() if int(os.environ.get("RANK", "0")) == 0 else None
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
    solver.close()

    # ---- beside the headline: the batch of independent what-if networks (1 000 machines / 10 000 pods each),
    # 256 in total, network b on rank b mod N, solved in lockstep groups; placements all-gathered into graph order
    batch_out = None
    if args.batch_graphs > 0:
        from poseidon_amd import batch as pbatch
        B = args.batch_graphs
        mine = pbatch.shard(B, world, rank)
        bcl, bnet, bsol = [], [], []
        for b in mine:
            c1 = pa.synth_cluster(1, 1 + b)
            n1 = pa.build_network(c1)
            s1 = pa.Solver(device=local_rank)
            s1.upload(n1)
            bcl.append(c1); bnet.append(n1); bsol.append(s1)
        P1 = 10_000
        m1 = bnet[0].m if bnet else 533_200
        bplace = torch.empty((len(mine), P1), dtype=torch.int32, device="cuda")

        bparts = {"price_ms": 0.0, "solve_ms": 0.0, "extract_ms": 0.0, "gather_ms": 0.0}

        def bstep():
            t_a = time.perf_counter()
            for s1, c1, n1 in zip(bsol, bcl, bnet):
                s1.eval_costs(c1, n1)
            t_b = time.perf_counter()
            rs = pbatch.solve_batch(bsol) if bsol else []
            t_c = time.perf_counter()
            for i, s1 in enumerate(bsol):
                s1.placements_into(bplace[i].data_ptr())
            torch.cuda.synchronize()
            t_d = time.perf_counter()
            allp = pbatch.all_gather_placements(bplace if backend == "nccl" else bplace.cpu(), B, world, rank)
            torch.cuda.synchronize()
            t_e = time.perf_counter()
            for key, dt in (("price_ms", t_b - t_a), ("solve_ms", t_c - t_b), ("extract_ms", t_d - t_c), ("gather_ms", t_e - t_d)):
                bparts[key] += dt * 1e3
            return rs, allp

        bstep()
        fence()
        for key in bparts:
            bparts[key] = 0.0
        tb = time.perf_counter()
        bsteps = 3
        for _ in range(bsteps):
            brs, ballp = bstep()
        fence()
        bel = time.perf_counter() - tb
        if world > 1:
            t = torch.tensor([bel], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bel = float(t.item())
        bcosts = pbatch.all_gather_costs([r_.total_cost for r_ in brs], B, world, rank, device=cdev)
        # this rank's share of a batch step, part by part (host clock; the gather includes waiting for the slowest rank)
        my_parts = torch.tensor([bparts[k_] / bsteps for k_ in ("price_ms", "solve_ms", "extract_ms", "gather_ms")], dtype=torch.float64, device=cdev)
        if world > 1:
            all_parts = [torch.empty_like(my_parts) for _ in range(world)]
            dist.all_gather(all_parts, my_parts)
        else:
            all_parts = [my_parts]
        per_rank = [{"rank": i_, "networks": len(pbatch.shard(B, world, i_)), "price_ms": round(float(p_[0]), 2), "solve_ms": round(float(p_[1]), 2),
                     "extract_ms": round(float(p_[2]), 2), "gather_ms": round(float(p_[3]), 2)} for i_, p_ in enumerate(all_parts)]
        # the checker: some of this rank's networks against the CPU oracle (outside every timed region)
        checked = 0
        if args.batch_check > 0:
            from oracle import capi, synth
            for i in range(min(args.batch_check, len(mine))):
                rc_, tc_, _, _, _ = capi.cost_scaling(synth.build_graph(synth.make_cluster(1, 1 + mine[i])))
                assert rc_ == 0 and tc_ == brs[i].total_cost, (mine[i], tc_, brs[i].total_cost)
                checked += 1
        if world > 1:
            t = torch.tensor([checked], dtype=torch.int64, device=cdev)
            dist.all_reduce(t)
            checked = int(t.item())
        batch_out = {"workload": f"BASELINE configs[4]: {B} psn-synth-v1 config 1 networks in total ({m1} forward arcs each, seed = graph id + 1), "
                                 f"network b on rank b mod {world}, lockstep groups, placements all-gathered",
                     "graphs": B, "n_gpus": world, "scaling": "strong", "graphs_per_s": round(B * bsteps / bel, 1),
                     "arcs_per_s": round(m1 * B * bsteps / bel, 1), "ms_per_batch": round(bel / bsteps * 1e3, 2),
                     "placed_total": int((ballp >= 0).sum().item()), "cost_checksum": int(sum(bcosts) % (1 << 61)),
                     "oracle_checked": checked, "per_rank": per_rank}
        for s1 in bsol:
            s1.close()

    ms_per_step = elapsed / args.steps * 1e3
    arcs_per_s = net.m * world * args.steps / elapsed
    r = res[-1]
    solve_all = [x.solve_ms for x in res]
    solve_ms = float(np.median(solve_all))

    if rank == 0:
        # ---- roofline of the dominant kernel: one extra solve with HIP-event timing of every launch
        roof = None
        with pa.Solver(device=local_rank, time_kernels=True) as ts:
            ts.upload(net)
            ts.eval_costs(cl, net)
            tr = ts.solve()
            kt = ts.kernel_times()
        def sweep_roof(kind):
            """One sweep kernel against the HBM roofline: the contract's bytes per unit of work over
            its event-timed launches, what the PMC pass of the committed profile saw, and the
            contract's figure with the arcs of which only the 4-byte capacity was read counted at 4 B."""
            ms, calls = kt[kind]
            if not (ms > 0 and calls > 0):
                return None
            bytes_total = algorithmic_bytes(tr, kind)
            achieved = bytes_total / (ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic("k_" + kind)
            avg_us = ms * 1e3 / calls
            scans, skips = (tr.gu_arc_scans, tr.gu_arc_skips) if kind == "gu_expand" else (tr.arc_scans, tr.arc_skips)
            real = bytes_total - 16 * skips
            return {"bound": "hbm", "kernel": "k_" + kind, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "traffic_unit": "bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024 of the committed counter pass",
                    "traffic_source": traffic_src, "traffic_source_problem": stale_profile,
                    "launches": calls, "avg_launch_us": round(avg_us, 2), "kernel_total_ms": round(ms, 2),
                    "algorithmic_bytes_per_launch": int(bytes_total / calls),
                    # both currencies: the 20-bytes-per-arc-scan rule, and what HBM saw (PMC pass of the committed profile)
                    "achieved_contract": round(achieved, 1),
                    "achieved_hbm": round(traffic / (avg_us * 1e-6) / 1e9, 1) if traffic else None,
                    "arc_scans": scans, "arc_skips_read_at_4B": skips,
                    "achieved_skips_at_4B": round(real / (ms * 1e-3) / 1e9, 1)}

        # the dominant kernel is the sweep with the larger total over the solve; the other one rides beside it
        sweeps = sorted(("push", "gu_expand"), key=lambda kd: -kt[kd][0])
        roof = sweep_roof(sweeps[0])
        if roof is not None:
            roof["kernel_ms"] = {k: round(v[0], 2) for k, v in kt.items()}
            roof["other_sweep"] = sweep_roof(sweeps[1])
        # ---- configs[3]: scheduler-service rounds, cold and warm after churn
        c3 = None
        if not args.no_c3 and world == 1:
            c3 = {}
            for cfg_id in (1, 2):
                c3[f"config{cfg_id}"], _ = service_rounds(cfg_id, 1, local_rank)
        # ---- CPU baseline: the oracle's single-thread cost-scaling solver on the same network
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # timed on rank 0 of the single-GPU run only
            from oracle import capi, synth
            try:
                os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})  # one core, as a sequential solver gets
                pinned = True
            except (AttributeError, OSError):
                pinned = False
            capi.use_native_build()
            ocl = synth.make_cluster(args.config, seed)
            og = synth.build_graph(ocl)
            rows = {}
            for name, kw in capi.BASELINE_VARIANTS.items():
                times, ccost = [], None
                for _ in range(3 if args.config >= 2 else 5):
                    rc, ccost, _, _, st = capi.cost_scaling(og, **kw)
                    assert rc == 0
                    times.append(st.solve_ms)
                assert ccost == all_costs[0], (name, ccost, all_costs)
                rows[name] = {"solve_ms": round(float(np.median(times)), 1), "arcs_per_s": round(og.m / (float(np.median(times)) * 1e-3), 1),
                              "pushes": int(st.pushes), "relabels": int(st.relabels), "global_updates": int(st.global_updates)}
            best = min(rows, key=lambda k: rows[k]["solve_ms"])
            cpu = {"value": rows[best]["arcs_per_s"], "unit": "arcs/s", "cores": 1, "kind": "port",
                   "solve_ms": rows[best]["solve_ms"],
                   "variant": best + "; not given to the CPU port: the GPU solver's early phase ends (a sequential solver's last units "
                              "of a phase cost a few hundred operations each, not a pair of launches) and its replays on the reversed network",
                   "variants": rows, "pinned_to_one_core": pinned,
                   "host_cores_total": os.cpu_count(), "optimal_cost": int(all_costs[0]), "build": capi.build_flags(),
                   "sample": f"full psn-synth-v1 config {args.config} seed {seed} network, single-thread cost-scaling "
                             f"push-relabel (oracle/psn_oracle.c), median of 3 solves per variant, the faster variant reported"}
        out = {
            "metric": "MCMF scheduling solve throughput (forward arcs of the input network per second; K1 pricing + cold "
                      "cost-scaling solve + placement extraction, 10k-node/100k-pod graph)",
            "value": round(arcs_per_s, 1), "unit": "arcs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic" if backend == "nccl" else f"synthetic (rehearsal over {backend}: not a measurement)",
            "config": {"workload": f"psn-synth-v1 config {args.config}: {cl.layout.n_machines} machines / {P} pods, "
                                   f"n={net.n} vertices, m={net.m} forward arcs, one network per GPU (seed = 1 + rank)",
                       "graphs_per_step": world, "alpha": 16, "rounds_per_global_update": 14},
            "solve_ms": round(solve_ms, 3), "solve_ms_min": round(min(solve_all), 3), "solve_ms_max": round(max(solve_all), 3),
            "solve_ms_p10_p50_p90": [round(float(np.percentile(solve_all, q)), 3) for q in (10, 50, 90)],
            "optimal_cost": all_costs, "tasks_placed": placed_all,
            "upload_s_not_timed": round(upload_s, 4),
            "counters": {"arc_scans": r.arc_scans, "gu_arc_scans": r.gu_arc_scans, "gu_arc_skips": r.gu_arc_skips, "arc_skips": r.arc_skips, "sat_arc_scans": r.sat_arc_scans,
                         "pushes": r.pushes, "relabels": r.relabels, "vertex_visits": r.vertex_visits, "rounds": r.rounds,
                         "gu_iterations": r.gu_iterations, "global_updates": r.global_updates, "refines": r.refines,
                         "launches": r.launches, "replays": r.replays, "mirrored_replays": r.mirrored_replays,
                         "last_phase_replays": r.last_phase_replays},
            "solve_algorithmic_GBps": round(r.algorithmic_bytes / (solve_ms * 1e-3) / 1e9, 1),
            "roofline": roof, "cpu_baseline": cpu, "what_if_batch": batch_out, "c3": c3,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
