"""CPU tests of the oracle: the two C solvers, the certificate and the numpy builder agree
with each other, with the committed golden vectors (cross-checked once against networkx
network_simplex, see tests/golden/make_golden.py) and on adversarial small cases."""
import json
import os

import numpy as np
import pytest

from oracle import capi, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "psn_synth_v1.json")


def _golden():
    with open(GOLD) as f:
        return json.load(f)["entries"]


def random_graph(rng, n, m, maxcap=5, maxcost=20, neg=False):
    tail = rng.integers(0, n, m)
    head = rng.integers(0, n, m)
    keep = tail != head
    tail, head = tail[keep], head[keep]
    m = tail.shape[0]
    cap = rng.integers(1, maxcap + 1, m)
    cost = rng.integers(0, maxcost + 1, m)
    if neg:  # negative arc costs but no negative cycle: shift by a random potential
        pi = rng.integers(0, 10, n)
        cost = cost + pi[tail] - pi[head]
    f = rng.integers(0, cap + 1)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f)
    np.add.at(supply, head, -f)
    return synth.graph_from_arcs(n, tail, head, cap, cost, supply)


def test_splitmix_known_answers():
    # splitmix64 reference outputs for seed 0 (Vigna's test vector)
    out = synth.splitmix64_stream(0, 3)
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_cost_model_numpy_vs_c():
    cl = synth.make_cluster(0, 2)
    c1, p1 = synth.eval_costs(cl)
    c2, p2 = capi.eval_costs(cl)
    assert np.array_equal(c1, c2) and np.array_equal(p1, p2)
    # fits: capacity 1 exactly for the first `fit` slots, convex cost in k
    assert (np.diff(c1, axis=2) >= 0).all()
    assert (np.diff(p1, axis=2) <= 0).all()
    # a request larger than what is left never fits (the reference's e2e expectation that an
    # over-committed pod stays pending, /root/reference/test/e2e/poseidon_integration.go:310-408)
    free_cpu = cl.cap_cpu - cl.used_cpu
    e_big = np.argmax(cl.req_cpu)
    assert (p1[e_big][free_cpu < cl.req_cpu[e_big]] == 0).all()


@pytest.mark.parametrize("entry", [e for e in _golden() if e["config"] == 0], ids=lambda e: f"c{e['config']}s{e['seed']}")
def test_golden_small(entry):
    cl = synth.make_cluster(entry["config"], entry["seed"])
    g = synth.build_graph(cl)
    assert (g.n, g.m) == (entry["n"], entry["m"])
    assert "%016x" % capi.csr_checksum(g) == entry["csr_fnv1a64"]
    rc, tc, cap, pot = capi.ssp(g)
    assert rc == 0 and tc == entry["optimal_cost"] == entry["networkx_network_simplex"]
    assert capi.check(g, cap)[:2] == (0, tc)
    rc, tc2, cap2, price, st = capi.cost_scaling(g)
    assert rc == 0 and tc2 == tc
    assert capi.check(g, cap2, price)[:2] == (0, tc)
    assert capi.check_eps(g, cap2, price, 1) == 0


def test_golden_c1_cost_scaling():
    entry = [e for e in _golden() if e["config"] == 1 and e["seed"] == 1][0]
    g = synth.build_graph(synth.make_cluster(1, 1))
    assert "%016x" % capi.csr_checksum(g) == entry["csr_fnv1a64"]
    rc, tc, cap, price, st = capi.cost_scaling(g)
    assert rc == 0 and tc == entry["optimal_cost"]
    assert capi.check(g, cap, price)[:2] == (0, tc)


def test_random_graphs_ssp_equals_cost_scaling():
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(2, 40))
        g = random_graph(rng, n, int(rng.integers(1, 6 * n)), neg=trial % 3 == 0)
        rc1, tc1, cap1, _ = capi.ssp(g)
        rc2, tc2, cap2, price, _ = capi.cost_scaling(g)
        assert rc1 == 0 and rc2 == 0 and tc1 == tc2, (trial, rc1, rc2, tc1, tc2)
        assert capi.check(g, cap1)[0] == 0 and capi.check(g, cap2, price)[0] == 0


def test_certificate_rejects_wrong_answers():
    g = synth.build_graph(synth.make_cluster(0, 1, 20, 100))
    rc, tc, cap, price, _ = capi.cost_scaling(g)
    assert capi.check(g, cap, price)[0] == 0
    # break conservation: cancel one unit of flow on one arc only
    a = int(np.nonzero((g.cap - cap) > 0)[0][0])
    bad = cap.copy(); bad[a] += 1; bad[g.rev[a]] -= 1
    assert capi.check(g, bad)[0] == 1
    # bounds: negative residual capacity
    bad = cap.copy(); bad[a] = -1
    assert capi.check(g, bad)[0] == 2
    # feasible but not optimal: the all-unscheduled flow
    N, E, J, P = g.n_machines, g.n_ec, g.n_jobs, g.n_pods
    sub = cap.copy(); sub[:] = g.cap
    row = g.row_off
    for t in range(P):
        v = 1 + N + E + J + t
        a2 = row[v] + 1                      # task -> unsched
        sub[a2] -= 1; sub[g.rev[a2]] += 1
        j = g.head[a2]
        a3 = row[j + 1] - 1                  # unsched -> sink
        sub[a3] -= 1; sub[g.rev[a3]] += 1
    rc_sub, cost_sub, _ = capi.check(g, sub)
    assert rc_sub == 3 and cost_sub == P * synth.C_UNSCHED


def test_infeasible_and_trivial():
    # supply with no outgoing capacity
    g = synth.graph_from_arcs(3, [0], [1], [1], [1], [0, -1, 1])
    assert capi.ssp(g)[0] == -4
    # empty flow: all supplies zero
    g = synth.graph_from_arcs(3, [0, 1], [1, 2], [2, 2], [1, 1], [0, 0, 0])
    assert capi.ssp(g)[:2] == (0, 0)
    assert capi.cost_scaling(g)[:2] == (0, 0)


def test_tuned_baseline_variant_agrees_with_the_plain_oracle():
    """The timed CPU baseline has a tuned variant (first epsilon from the 90th-percentile cost,
    alpha 16, price refinement); it must reach the same optimum as the plain variant, on the
    synthetic family and on random graphs with negative costs (where the percentile start is off)."""
    rng = np.random.default_rng(11)
    graphs = [synth.build_graph(synth.make_cluster(0, s)) for s in (1, 2)]
    for _ in range(30):
        n = int(rng.integers(2, 60))
        m = int(rng.integers(1, 6 * n))
        tail, head = rng.integers(0, n, m), rng.integers(0, n, m)
        keep = tail != head
        tail, head = tail[keep], head[keep]
        cap = rng.integers(1, 6, tail.shape[0])
        pi = rng.integers(0, 10, n)
        cost = rng.integers(0, 30, tail.shape[0]) + pi[tail] - pi[head]
        f = rng.integers(0, cap + 1)
        supply = np.zeros(n, np.int64)
        np.add.at(supply, tail, f)
        np.add.at(supply, head, -f)
        graphs.append(synth.graph_from_arcs(n, tail, head, cap, cost, supply))
    for g in graphs:
        want = None
        for kw in capi.BASELINE_VARIANTS.values():
            rc, tc, cap, price, _ = capi.cost_scaling(g, **kw)
            assert rc == 0
            assert capi.check(g, cap, price)[:2] == (0, tc)
            want = tc if want is None else want
            assert tc == want


def _longest_simple_cycle(n, adj_pairs):
    """Brute force: the largest number of arcs of a simple cycle in the undirected multigraph given
    as a list of vertex pairs (a pair listed twice is two parallel arcs: a cycle of 2)."""
    nbr = {v: {} for v in range(n)}
    for u, v in adj_pairs:
        nbr[u][v] = nbr[u].get(v, 0) + 1
        nbr[v][u] = nbr[v].get(u, 0) + 1
    best = 2 if any(c >= 2 for d in nbr.values() for c in d.values()) else 0

    def walk(start, v, seen, length):
        nonlocal best
        for w in nbr[v]:
            if w == start and length >= 2:
                best = max(best, length + 1)
            elif w > start and w not in seen:
                seen.add(w)
                walk(start, w, seen, length + 1)
                seen.discard(w)

    for s in range(n):
        walk(s, s, {s}, 0)
    return best


def test_cycle_length_bound_holds_on_small_graphs():
    """The cost scale rests on an upper bound for the number of arcs of a simple residual cycle:
    min(n, 2 |cover|, 4 |cover minus its independent subset|).  Checked against the longest simple
    cycle found by exhaustive search on small random graphs of several shapes (sparse, dense,
    bipartite with hubs and leaves like a scheduling network, with parallel arcs)."""
    rng = np.random.default_rng(5)
    tight = 0
    for trial in range(400):
        n = int(rng.integers(2, 10))
        kind = trial % 4
        if kind == 0:
            m = int(rng.integers(1, 3 * n))
            tail, head = rng.integers(0, n, m), rng.integers(0, n, m)
        elif kind == 1:
            pairs = [(u, v) for u in range(n) for v in range(u + 1, n) if rng.random() < 0.7]
            tail, head = np.array([p[0] for p in pairs], int), np.array([p[1] for p in pairs], int)
        elif kind == 2:  # leaves -> hubs -> middle -> sink, the shape of a scheduling network
            hubs = max(1, n // 4)
            tail, head = [], []
            for v in range(hubs + 1, n):
                tail += [v, v]; head += [1 + int(rng.integers(0, hubs)), 0 if rng.random() < 0.5 else 1 + int(rng.integers(0, hubs))]
            for h in range(1, hubs + 1):
                tail.append(h); head.append(0)
            tail, head = np.array(tail, int), np.array(head, int)
        else:
            m = int(rng.integers(1, 2 * n))
            tail, head = rng.integers(0, n, m), rng.integers(0, n, m)
            tail, head = np.concatenate([tail, tail[: m // 2]]), np.concatenate([head, head[: m // 2]])  # parallel arcs
        keep = tail != head
        tail, head = tail[keep], head[keep]
        if tail.shape[0] == 0:
            continue
        m = tail.shape[0]
        g = synth.graph_from_arcs(n, tail, head, np.ones(m, np.int64), np.ones(m, np.int64), np.zeros(n, np.int64))
        bound = capi.cycle_bound(g)
        longest = _longest_simple_cycle(n, list(zip(tail.tolist(), head.tolist())))
        assert longest <= bound, (trial, n, tail.tolist(), head.tolist(), longest, bound)
        tight += longest == bound
    assert tight > 0  # the bound is attained on some of them
    # the scheduling networks: classes and the sink are what is left outside the independent subset
    assert capi.cycle_bound(synth.build_graph(synth.make_cluster(1, 1))) == 4 * 65


def test_every_golden_cost_rests_on_two_oracles_that_agree():
    """Each committed optimum was found by the two C oracles independently (successive shortest
    paths and cost scaling; at the 10 000-machine size the former takes seven minutes per network
    and was run once, by tests/golden/make_golden.py --ssp-c2), and the small ones by networkx too."""
    for e in _golden():
        assert e["ssp"] == e["cost_scaling"] == e["optimal_cost"], (e["config"], e["seed"])
        if e["config"] == 0:
            assert e["networkx_network_simplex"] == e["optimal_cost"]
    assert sorted((e["config"], e["seed"]) for e in _golden()) == [(c, s) for c in (0, 1, 2) for s in (1, 2, 3)]
