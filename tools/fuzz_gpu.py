#!/usr/bin/env python3
"""Differential soak test of the GPU solver against the CPU oracle (not part of the test suite:
it runs for minutes).  Random generic graphs, hub-heavy graphs with parallel arcs that exercise the
split-vertex paths (segment bounds, claims), small synthetic clusters with random shapes, each solved
cold several times (the push order is timing-dependent, the optimum is not), then perturbed and
warm-started.

  python tools/fuzz_gpu.py [seconds] [seed]
  python tools/fuzz_gpu.py [seconds] [seed] groups     lockstep groups (psn_solve_batch) of 2..12 random networks,
                                                       infeasible ones among them, cold and then warm
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa
from oracle import capi, synth


def generic(rng):
    n = int(rng.integers(2, 400))
    m = int(rng.integers(1, 10 * n))
    tail = rng.integers(0, n, m); head = rng.integers(0, n, m)
    keep = tail != head
    tail, head = tail[keep], head[keep]
    m = tail.shape[0]
    cap = rng.integers(0, int(rng.choice([2, 6, 1000])) + 1, m)
    cost = rng.integers(0, int(rng.choice([1, 20, 100000])) + 1, m)
    if rng.random() < 0.3:
        pi = rng.integers(0, 50, n)
        cost = cost + pi[tail] - pi[head]
    f = rng.integers(0, cap + 1)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f); np.add.at(supply, head, -f)
    return synth.graph_from_arcs(n, tail, head, cap, cost, supply)


def hubby(rng):
    """Bipartite-ish: a few hubs with thousands of (parallel) arcs, like task classes over machines."""
    nh = int(rng.integers(1, 5)); nm = int(rng.integers(150, 500)); k = int(rng.integers(1, 9)); nt = int(rng.integers(10, 400))
    n = 1 + nh + nm + nt  # 0 sink | hubs | machines | tasks
    tail, head, cap, cost = [], [], [], []
    for h in range(nh):
        for mch in range(nm):
            base = int(rng.integers(0, 2000))
            for s in range(k):
                tail.append(1 + h); head.append(1 + nh + mch); cap.append(int(rng.integers(0, 2))); cost.append(base + s * int(rng.integers(0, 50)))
    for mch in range(nm):
        tail.append(1 + nh + mch); head.append(0); cap.append(int(rng.integers(0, 4))); cost.append(0)
    for t in range(nt):
        v = 1 + nh + nm + t
        tail.append(v); head.append(1 + int(rng.integers(0, nh))); cap.append(1); cost.append(0)
        tail.append(v); head.append(0); cap.append(1); cost.append(int(rng.integers(1000, 100000)))
    supply = np.zeros(n, np.int64)
    supply[1 + nh + nm:] = 1
    supply[0] = -nt
    return synth.graph_from_arcs(n, np.array(tail), np.array(head), np.array(cap), np.array(cost), supply)


def cluster(rng):
    cl = synth.make_cluster(0, int(rng.integers(1, 1 << 30)), int(rng.integers(5, 300)), int(rng.integers(5, 3000)))
    if rng.random() < 0.5:
        cl.allowed = rng.random((64, cl.n_machines)) < rng.random()
    return synth.build_graph(cl)


def groups(budget, rng):
    from poseidon_amd import batch
    t_end = time.time() + budget
    t_say = time.time()
    count = {"groups": 0, "members": 0, "infeasible": 0, "warm": 0}
    pool = [pa.Solver() for _ in range(12)]
    while time.time() < t_end:
        if time.time() - t_say > 30:
            print("  ...", count, flush=True)
            t_say = time.time()
        k = int(rng.integers(2, 13))
        gs, want = [], []
        for i in range(k):
            kind = rng.choice(["generic", "hubby", "cluster"], p=[0.5, 0.3, 0.2])
            g = {"generic": generic, "hubby": hubby, "cluster": cluster}[kind](rng)
            rc, tc, _, _ = capi.ssp(g)
            gs.append(g); want.append(tc if rc == 0 else None)
            pool[i].upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
        for warm in (False, True):
            if warm:  # perturb capacities of the feasible members
                for i, g in enumerate(gs):
                    if want[i] is None or not (g.cap > 0).any():
                        continue
                    live = np.nonzero(g.cap > 0)[0]
                    idx = rng.choice(live, size=min(int(rng.integers(1, 20)), live.shape[0]), replace=False)
                    newcap = rng.integers(0, 8, idx.shape[0]).astype(np.int32)
                    pool[i].update_caps(idx.astype(np.int64), newcap)
                    g.cap[idx] = newcap
                    rc, tc, _, _ = capi.ssp(g)
                    want[i] = tc if rc == 0 else None
            try:
                res = batch.solve_batch(pool[:k], warm_start=warm, workers=int(rng.integers(1, 4)))
                assert all(w is not None for w in want), "an infeasible member went unreported"
            except pa.PsnError as e:
                assert e.code == -4 and any(w is None for w in want), (e, want)
                res = None
            for i, g in enumerate(gs):
                if want[i] is None:
                    count["infeasible"] += 1
                    continue
                crc, ctc, _ = capi.check(g, pool[i].residual(), pool[i].prices(), scale=pool[i].price_scale())
                assert crc == 0 and ctc == want[i], (i, crc, ctc, want[i])
                if res is not None:
                    assert res[i].total_cost == want[i]
            count["warm" if warm else "groups"] += 1
            count["members"] += k
    for s in pool:
        s.close()
    print("fuzz groups ok:", count, flush=True)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if "groups" in sys.argv[3:]:
        return groups(budget, rng)
    t_end = time.time() + budget
    count = {"generic": 0, "hubby": 0, "cluster": 0, "warm": 0}
    t_say = time.time()
    with pa.Solver() as s, pa.Solver(eager=True) as se:
        while time.time() < t_end:
            if time.time() - t_say > 30:
                print("  ...", count, flush=True)
                t_say = time.time()
            kind = rng.choice(["generic", "hubby", "cluster"], p=[0.5, 0.3, 0.2])
            g = {"generic": generic, "hubby": hubby, "cluster": cluster}[kind](rng)
            rc, want, _, _ = capi.ssp(g)
            for sv in (s, s, se):
                sv.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
                try:
                    r = sv.solve()
                except pa.PsnError as e:
                    assert rc != 0 and e.code == -4, (kind, rc, e)
                    continue
                assert rc == 0 and r.total_cost == want, (kind, g.n, rc, want, r.total_cost)
                crc, ctc, _ = capi.check(g, sv.residual(), sv.prices(), scale=sv.price_scale())
                assert crc == 0 and ctc == want, (kind, crc)
            count[kind] += 1
            if rc == 0 and rng.random() < 0.5:  # perturb capacities, warm start
                m2 = g.head.shape[0]
                k = int(rng.integers(1, max(2, m2 // 10)))
                idx = rng.choice(np.nonzero(g.cap > 0)[0], size=min(k, int((g.cap > 0).sum())), replace=False) if (g.cap > 0).any() else np.zeros(0, np.int64)
                if idx.shape[0]:
                    newcap = rng.integers(0, 8, idx.shape[0]).astype(np.int32)
                    s.update_caps(idx.astype(np.int64), newcap)
                    g2 = synth.Graph(g.n, g.m, g.row_off, g.head, g.cap.copy(), g.cost, g.rev, g.supply, g.fwd_pos)
                    g2.cap[idx] = newcap
                    rc2, want2, _, _ = capi.ssp(g2)
                    try:
                        r2 = s.solve(warm_start=True)
                        assert rc2 == 0 and r2.total_cost == want2, ("warm", kind, rc2, want2, r2.total_cost)
                    except pa.PsnError as e:
                        assert rc2 != 0 and e.code == -4, ("warm", kind, rc2, e)
                    count["warm"] += 1
    print("fuzz ok:", count, flush=True)


if __name__ == "__main__":
    main()
