"""Generates tests/golden/psn_synth_v1.json.

For every (config, seed) of the synthetic family psn-synth-v1 it records the network
sizes, the FNV-1a-64 checksum of the residual CSR, the optimal cost found by the two C
oracles (successive shortest paths where affordable, cost scaling always, both passing
the optimality certificate) and, for the sizes it can finish, the optimal cost found by
an independent third-party solver that ships in this container: networkx 3.4.2
`network_simplex`.  No reference solver exists to generate vectors from (Firmament is
absent from /root/reference), so these numbers pin the oracle against independent
mathematics, not against the reference: parity with the reference stays unpinned.

Run from the repo root:  python tests/golden/make_golden.py        (--all: with the slow parts below, about an hour)
then, optionally (tens of minutes):  python tests/golden/make_golden.py --nx-c1-only
which adds the networkx cost of config 1 / seed 1 to the existing file;
  python tests/golden/make_golden.py --ssp-c2
adds the successive-shortest-paths optimum of the config 2 networks (about seven minutes each,
single thread), so that the full-size golden cost rests on two C oracles that agree;
  python tests/golden/make_golden.py --c4
writes psn_c4_costs.json: the cost-scaling optimum (certificate checked) and the CSR checksum of
the 256 config 1 networks of the what-if batch (network b uses seed b + 1).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from oracle import capi, synth


def nx_cost(g):
    import networkx as nx
    G = nx.MultiDiGraph()
    for v in range(g.n):
        G.add_node(v, demand=int(-g.supply[v]))
    tail = np.repeat(np.arange(g.n), np.diff(g.row_off))
    for a in g.fwd_pos:
        if g.cap[a] > 0:
            G.add_edge(int(tail[a]), int(g.head[a]), capacity=int(g.cap[a]), weight=int(g.cost[a]))
    cost, _ = nx.network_simplex(G)
    return int(cost)


def patch_nx_c1(path):
    g = synth.build_graph(synth.make_cluster(1, 1))
    t = time.time()
    cost = nx_cost(g)
    with open(path) as f:
        out = json.load(f)
    for e in out["entries"]:
        if e["config"] == 1 and e["seed"] == 1:
            assert e["optimal_cost"] == cost, (e["optimal_cost"], cost)
            e["networkx_network_simplex"] = cost
            e["networkx_s"] = round(time.time() - t, 1)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("networkx config 1 seed 1:", cost)


def _c4_member(seed):
    g = synth.build_graph(synth.make_cluster(1, seed))
    rc, tc, cap, price, _ = capi.cost_scaling(g)
    assert rc == 0
    crc, ctc, _ = capi.check(g, cap, price)
    assert crc == 0 and ctc == tc
    return seed, tc, "%016x" % capi.csr_checksum(g)


def make_c4(path, graphs=256, procs=4):
    import multiprocessing as mp
    with mp.Pool(procs) as pool:
        rows = pool.map(_c4_member, range(1, graphs + 1), chunksize=4)
    out = {"family": "psn-synth-v1", "config": 1, "graphs": graphs, "seed_of_graph_b": "b + 1",
           "oracle": "oracle/psn_oracle.c cost scaling, optimality certificate checked",
           "optimal_cost": [r[1] for r in rows], "csr_fnv1a64": [r[2] for r in rows]}
    assert [r[0] for r in rows] == list(range(1, graphs + 1))
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, "checksum of costs", sum(out["optimal_cost"]))


def patch_ssp_c2(path):
    with open(path) as f:
        out = json.load(f)
    for e in out["entries"]:
        if e["config"] != 2 or "ssp" in e:
            continue
        g = synth.build_graph(synth.make_cluster(2, e["seed"]))
        t = time.time()
        rc, tcs, caps, _ = capi.ssp(g)
        assert rc == 0 and capi.check(g, caps)[0] == 0
        assert tcs == e["optimal_cost"], (tcs, e["optimal_cost"])
        e["ssp"] = tcs
        e["ssp_s"] = round(time.time() - t, 1)
        print("config 2 seed", e["seed"], "ssp", tcs, e["ssp_s"], "s", flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "psn_synth_v1.json")
    if "--nx-c1-only" in sys.argv:
        return patch_nx_c1(path)
    if "--ssp-c2" in sys.argv:
        return patch_ssp_c2(path)
    if "--c4" in sys.argv:
        return make_c4(os.path.join(os.path.dirname(path), "psn_c4_costs.json"))
    nx_c1 = False
    out = {"family": "psn-synth-v1", "entries": []}
    for cfg in (0, 1, 2):
        for seed in (1, 2, 3):
            cl = synth.make_cluster(cfg, seed)
            g = synth.build_graph(cl)
            e = {"config": cfg, "seed": seed, "n": g.n, "m": g.m, "n_machines": cl.n_machines, "n_pods": cl.n_pods,
                 "csr_fnv1a64": "%016x" % capi.csr_checksum(g)}
            t = time.time()
            rc, tc, cap, price, st = capi.cost_scaling(g)
            assert rc == 0
            crc, ctc, _ = capi.check(g, cap, price)
            assert crc == 0 and ctc == tc
            e["cost_scaling"] = tc
            e["cost_scaling_s"] = round(time.time() - t, 2)
            if cfg <= 1:
                rc, tcs, caps, _ = capi.ssp(g)
                assert rc == 0 and capi.check(g, caps)[0] == 0
                e["ssp"] = tcs
                assert tcs == tc
            if cfg == 0 or (cfg == 1 and nx_c1):
                t = time.time()
                e["networkx_network_simplex"] = nx_cost(g)
                e["networkx_s"] = round(time.time() - t, 1)
                assert e["networkx_network_simplex"] == tc
            e["optimal_cost"] = tc
            print(e, flush=True)
            out["entries"].append(e)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    if "--all" in sys.argv:  # the slow parts as well: everything the committed file holds
        patch_ssp_c2(path)
        patch_nx_c1(path)
        make_c4(os.path.join(os.path.dirname(path), "psn_c4_costs.json"))


if __name__ == "__main__":
    main()
