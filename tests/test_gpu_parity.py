"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on
the same seeded inputs and against the committed golden costs.  Integer work: every
comparison is bit-exact."""
import json
import os

import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import _lib
from oracle import capi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "psn_synth_v1.json")


def _gold(cfg, seed):
    return [e for e in json.load(open(GOLD))["entries"] if e["config"] == cfg and e["seed"] == seed][0]


@pytest.fixture(scope="module")
def solver():
    with pa.Solver(device=0) as s:
        yield s


def random_graph(rng, n, m, maxcap=5, maxcost=20, neg=False):
    tail = rng.integers(0, n, m)
    head = rng.integers(0, n, m)
    keep = tail != head
    tail, head = tail[keep], head[keep]
    m = tail.shape[0]
    cap = rng.integers(1, maxcap + 1, m)
    cost = rng.integers(0, maxcost + 1, m)
    if neg:
        pi = rng.integers(0, 10, n)
        cost = cost + pi[tail] - pi[head]
    f = rng.integers(0, cap + 1)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f)
    np.add.at(supply, head, -f)
    return synth.graph_from_arcs(n, tail, head, cap, cost, supply)


def solve_and_certify(s, g):
    s.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
    r = s.solve()
    resid = s.residual()
    price = s.prices()
    crc, ctc, _ = capi.check(g, resid, price, scale=s.price_scale())
    assert crc == 0, f"certificate failed with code {crc}"
    assert ctc == r.total_cost
    assert capi.check_eps(g, resid, price, 1, scale=s.price_scale()) == 0
    flow = s.flow()
    assert np.array_equal(flow, g.cap - resid)
    return r


def test_random_generic_graphs_vs_ssp(solver):
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(2, 80))
        g = random_graph(rng, n, int(rng.integers(1, 8 * n)), neg=trial % 4 == 0)
        r = solve_and_certify(solver, g)
        rc, tc, _, _ = capi.ssp(g)
        assert rc == 0 and tc == r.total_cost, (trial, tc, r.total_cost)


def test_random_graphs_with_hubs_and_large_caps(solver):
    """Star-like graphs: a few vertices above the 1024-arc item size, large capacities."""
    rng = np.random.default_rng(9)
    for trial in range(4):
        n = 600
        hubs = rng.integers(0, n, 3)
        m = 9000
        tail = np.where(rng.random(m) < 0.7, hubs[rng.integers(0, 3, m)], rng.integers(0, n, m))
        head = rng.integers(0, n, m)
        keep = tail != head
        tail, head = tail[keep], head[keep]
        cap = rng.integers(1, 1000, tail.shape[0])
        cost = rng.integers(0, 5000, tail.shape[0])
        f = rng.integers(0, cap + 1)
        supply = np.zeros(n, np.int64)
        np.add.at(supply, tail, f)
        np.add.at(supply, head, -f)
        g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
        assert (np.diff(g.row_off) > 1024).any()
        r = solve_and_certify(solver, g)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0 and tc == r.total_cost


@pytest.mark.parametrize("n_big,m", [(60, 100_000), (150, 400_000)])
def test_more_split_vertices_than_sweep_workgroups_or_hub_slots(solver, n_big, m):
    """Many vertices above the item size in a network small enough for the smallest sweep grid
    (fewer workgroups than hubs: a workgroup looks after several), and more split vertices
    than the 128 hub slots (the rest go through the staged append)."""
    rng = np.random.default_rng(n_big)
    n = 3000
    big = rng.choice(n, n_big, replace=False)
    tail = np.where(rng.random(m) < 0.85, big[rng.integers(0, n_big, m)], rng.integers(0, n, m))
    head = rng.integers(0, n, m)
    keep = tail != head
    tail, head = tail[keep], head[keep]
    cap = rng.integers(1, 50, tail.shape[0])
    cost = rng.integers(0, 3000, tail.shape[0])
    f = np.where(rng.random(tail.shape[0]) < 0.3, rng.integers(0, cap + 1), 0)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f)
    np.add.at(supply, head, -f)
    g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
    assert (np.diff(g.row_off) > 1152).sum() >= n_big * 0.9
    r = solve_and_certify(solver, g)
    rc, tc, _, _, _ = capi.cost_scaling(g)
    assert rc == 0 and tc == r.total_cost


def test_trivial_and_edge_cases(solver):
    # all supplies zero: nothing to route
    g = synth.graph_from_arcs(3, [0, 1], [1, 2], [2, 2], [1, 1], [0, 0, 0])
    assert solve_and_certify(solver, g).total_cost == 0
    # single arc
    g = synth.graph_from_arcs(2, [0], [1], [5], [7], [3, -3])
    assert solve_and_certify(solver, g).total_cost == 21
    # no arcs at all, no supply
    g = synth.graph_from_arcs(4, [], [], [], [], [0, 0, 0, 0])
    assert solve_and_certify(solver, g).total_cost == 0
    # parallel arcs of different cost: cheapest first
    g = synth.graph_from_arcs(2, [0, 0, 0], [1, 1, 1], [1, 1, 1], [5, 1, 3], [2, -2])
    assert solve_and_certify(solver, g).total_cost == 4


def test_error_codes(solver):
    lib = pa.load_library()
    # infeasible: supply that cannot leave
    g = synth.graph_from_arcs(3, [0], [1], [1], [1], [0, -1, 1])
    solver.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
    with pytest.raises(pa.PsnError) as ei:
        solver.solve()
    assert ei.value.code == _lib.PSN_EINFEASIBLE
    # supplies that do not balance are rejected at upload
    with pytest.raises(pa.PsnError) as ei:
        solver.upload_csr(2, np.array([0, 1, 2]), [1, 0], [1, 0], [1, -1], [1, 0], [1, 0])
    assert ei.value.code == _lib.PSN_EINFEASIBLE
    # malformed CSR: rev not an involution
    with pytest.raises(pa.PsnError) as ei:
        solver.upload_csr(2, np.array([0, 1, 2]), [1, 0], [1, 0], [1, -1], [0, 0], [1, -1])
    assert ei.value.code == _lib.PSN_EINVAL
    # costs that would overflow int64 prices
    big = synth.graph_from_arcs(3, [0, 1], [1, 2], [1, 1], [2**31 - 1, 2**31 - 1], [1, 0, -1])
    n = 3_000_000
    with pytest.raises(pa.PsnError) as ei:
        row = np.zeros(n + 1, np.int64); row[1:] = 2; row[2:] = 4
        solver.upload_csr(n, row, big.head, big.cap, big.cost, big.rev, np.zeros(n, np.int64))
    assert ei.value.code in (_lib.PSN_EOVERFLOW, _lib.PSN_EINVAL)
    assert lib.psn_last_error(solver._ctx)


def test_malformed_networks_are_rejected_before_any_solve(solver):
    """The arcs are checked on the device after the copy, indices range-checked before use: every
    kind of damage is refused with its own message, the context is left without a network, and a
    sound upload afterwards works."""
    lib = pa.load_library()
    good = synth.build_graph(synth.make_cluster(0, 4, 20, 150))
    m2 = good.head.shape[0]
    cand = np.nonzero(good.rev > np.arange(m2))[0]  # its pair lies further on: the damaged arc is met first
    fwd = int(cand[cand.shape[0] // 2])

    def damaged(field, at, value):
        arr = {k: getattr(good, k).copy() for k in ("head", "cap", "cost", "rev")}
        arr[field][at] = value
        return arr

    cases = [("head", fwd, good.n, _lib.PSN_EINVAL, "head out of range"),
             ("head", fwd, -1, _lib.PSN_EINVAL, "head out of range"),
             ("rev", fwd, m2 + 5, _lib.PSN_EINVAL, "involution"),
             ("rev", fwd, -7, _lib.PSN_EINVAL, "involution"),
             ("rev", fwd, 2**31 - 1, _lib.PSN_EINVAL, "involution"),
             ("rev", fwd, fwd, _lib.PSN_EINVAL, "placeholder"),
             ("cost", fwd, int(good.cost[fwd]) + 1, _lib.PSN_EINVAL, "cost[rev[a]]"),
             ("cap", fwd, -3, _lib.PSN_EINVAL, "negative capacity"),
             ("cap", fwd, 2**31 - 1, None, None)]
    for field, at, value, code, text in cases:
        arr = damaged(field, at, value)
        if code is None:  # a capacity pair that overflows needs both sides large
            arr["cap"][good.rev[at]] = 2**31 - 1
            code, text = _lib.PSN_EOVERFLOW, "overflows"
        with pytest.raises(pa.PsnError) as ei:
            solver.upload_csr(good.n, good.row_off, arr["head"], arr["cap"], arr["cost"], arr["rev"], good.supply)
        assert ei.value.code == code, (field, value, ei.value.code)
        assert text in lib.psn_last_error(solver._ctx).decode(), (field, value, lib.psn_last_error(solver._ctx))
        with pytest.raises(pa.PsnError) as ei:  # nothing is resident after a rejected upload
            solver.solve()
        assert ei.value.code == _lib.PSN_ESTATE
    # the lowest damaged arc is the one reported
    arr = damaged("cap", m2 - 1, -1)
    arr["head"][0] = good.n
    with pytest.raises(pa.PsnError):
        solver.upload_csr(good.n, good.row_off, arr["head"], arr["cap"], arr["cost"], arr["rev"], good.supply)
    assert "head out of range" in lib.psn_last_error(solver._ctx).decode()
    rc, tc, _, _, _ = capi.cost_scaling(good)
    assert rc == 0 and solve_and_certify(solver, good).total_cost == tc


@pytest.mark.parametrize("cfg,seed", [(0, 1), (0, 2), (0, 3), (1, 1), (1, 2)])
def test_synthetic_clusters_vs_golden_and_oracle(solver, cfg, seed):
    cl = pa.synth_cluster(cfg, seed)
    net = pa.build_network(cl)
    solver.upload(net)
    solver.eval_costs(cl, net)
    # K1 parity: priced arcs equal the oracle's cost model bit for bit
    og = synth.build_graph(synth.make_cluster(cfg, seed))
    cost, ocap = solver.costs()
    assert np.array_equal(cost, og.cost) and np.array_equal(ocap, og.cap)
    r = solver.solve()
    gold = _gold(cfg, seed)
    assert r.total_cost == gold["optimal_cost"]
    # work counters: machine rows go through the push sweep's sparse step (most of their arcs carry no flow)
    assert 0 < r.arc_skips < r.arc_scans and r.gu_arc_skips <= r.gu_arc_scans
    resid, price = solver.residual(), solver.prices()
    assert capi.check(og, resid, price, scale=solver.price_scale())[:2] == (0, r.total_cost)
    # K5: every placed task sits on a machine that its class can use, within slot capacity
    place = solver.placements()
    N, K = cl.layout.n_machines, cl.layout.k_slots
    placed = place >= 0
    assert placed.sum() == int((og.cap - resid)[og.fwd_pos[:cl.layout.n_tasks]].sum())  # flow on task->class arcs
    load = np.bincount(place[placed], minlength=N)
    assert load.max() <= cl.slots
    c3, cap3 = synth.eval_costs(synth.make_cluster(cfg, seed))
    per = {}
    for t in np.nonzero(placed)[0]:
        key = (int(cl.task_class[t]), int(place[t]))
        per[key] = per.get(key, 0) + 1
    for (e, m), cnt in per.items():
        assert cnt <= int(cap3[e, m].sum()), "more tasks of a class on a machine than fit"
    # placements reproduce the objective: k tasks of class e on machine m use its k cheapest slots
    total = (~placed).sum() * synth.C_UNSCHED + sum(int(np.sort(c3[e, m][cap3[e, m] > 0])[:cnt].sum()) for (e, m), cnt in per.items())
    assert total == r.total_cost


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_full_size_network_certificate(seed):
    """10k machines / 100k pods: cost equals the golden value and the flow carries an
    optimality certificate (size-independent check, no CPU re-solve needed)."""
    cl = pa.synth_cluster(2, seed)
    net = pa.build_network(cl)
    with pa.Solver(device=0) as s:
        s.upload(net)
        s.eval_costs(cl, net)
        r = s.solve()
        resid, price = s.residual(), s.prices()
        pscale = s.price_scale()
        cost, ocap = s.costs()
        place = s.placements()
        # idempotence: a second cold solve gives the same cost
        s.eval_costs(cl, net)
        r2 = s.solve()
    assert r.total_cost == _gold(2, seed)["optimal_cost"] == r2.total_cost
    og = synth.build_graph(synth.make_cluster(2, seed))
    assert np.array_equal(cost, og.cost) and np.array_equal(ocap, og.cap)
    assert capi.check(og, resid, price, scale=pscale)[:2] == (0, r.total_cost)
    assert capi.check_eps(og, resid, price, 1, scale=pscale) == 0
    routed = int((og.cap - resid)[og.fwd_pos[:100000]].sum())  # flow on task -> class arcs
    assert int((place >= 0).sum()) == routed
    assert np.bincount(place[place >= 0], minlength=10000).max() <= cl.slots


def test_eager_and_graph_agree():
    cl = pa.synth_cluster(0, 2)
    net = pa.build_network(cl)
    costs = []
    for eager in (False, True):
        with pa.Solver(device=0, eager=eager) as s:
            s.upload(net)
            s.eval_costs(cl, net)
            costs.append(s.solve().total_cost)
    assert costs[0] == costs[1] == _gold(0, 2)["optimal_cost"]


def test_warm_start_after_capacity_change(solver):
    """Take machines out (machine->sink capacity 0) and re-solve warm: the result equals a
    cold solve of the changed network."""
    cl = pa.synth_cluster(0, 3)
    net = pa.build_network(cl)
    solver.upload(net)
    solver.eval_costs(cl, net)
    solver.solve()
    N, E, K = cl.layout.n_machines, cl.layout.n_classes, cl.layout.k_slots
    gone = np.arange(0, N, 7)
    arcs = np.array([net.row_off[1 + m] + E * K for m in gone], np.int64)  # machine -> sink
    solver.update_caps(arcs, np.zeros(arcs.shape[0], np.int32))
    rw = solver.solve(warm_start=True)
    og = synth.build_graph(synth.make_cluster(0, 3))
    og.cap[arcs] = 0
    rc, tc, _, _, _ = capi.cost_scaling(og)
    assert rc == 0 and rw.total_cost == tc
    assert capi.check(og, solver.residual(), solver.prices(), scale=solver.price_scale())[:2] == (0, tc)


def test_batch_of_independent_networks_matches_oracle():
    """Eight small what-if networks solved concurrently (psn_solve_batch): each cost equals the
    oracle's, and equals a plain sequential solve of the same context."""
    from poseidon_amd import batch
    solvers, want = [], []
    for b in range(8):
        cl = pa.synth_cluster(0, b + 1, 40 + 5 * b, 300 + 40 * b)
        net = pa.build_network(cl)
        s = pa.Solver()
        s.upload(net)
        s.eval_costs(cl, net)
        solvers.append(s)
        g = synth.build_graph(synth.make_cluster(0, b + 1, 40 + 5 * b, 300 + 40 * b))
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0
        want.append(tc)
    res = batch.solve_batch(solvers, workers=4)
    assert [r.total_cost for r in res] == want
    assert all(s.solve().total_cost == w for s, w in zip(solvers, want))
    with pytest.raises(_lib.PsnError):
        batch.solve_batch([solvers[0], solvers[0]])
    for s in solvers:
        s.close()


def test_lockstep_group_of_unlike_networks(monkeypatch):
    """psn_solve_batch solves contexts that share a device and configuration in lockstep (one
    launch per stage for the whole group).  Members of very different size and shape, one of them
    infeasible, one with nothing to route: every member gets its own optimum and certificate or
    its own error; a warm re-solve of the group after capacity changes equals cold solves; and
    group sizes 1 (no grouping), 3 (uneven split) and the default agree."""
    from poseidon_amd import batch
    rng = np.random.default_rng(77)
    graphs = [synth.build_graph(synth.make_cluster(1, 5)),                       # 1 000 machines, ~1 M residual arcs
              synth.build_graph(synth.make_cluster(0, 3)),
              random_graph(rng, 60, 400, neg=True),
              synth.graph_from_arcs(3, [0, 1], [1, 2], [2, 2], [1, 1], [0, 0, 0]),  # nothing to route
              random_graph(rng, 500, 6000, maxcap=50, maxcost=4000),
              synth.graph_from_arcs(2, [0], [1], [5], [7], [3, -3]),
              synth.build_graph(synth.make_cluster(0, 9, 30, 500))]
    want = []
    for g in graphs:
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0
        want.append(tc)
    solvers = []
    for g in graphs:
        s = pa.Solver()
        s.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
        solvers.append(s)
    bad = synth.graph_from_arcs(3, [0], [1], [1], [1], [0, -1, 1])             # supply that cannot leave
    sb = pa.Solver()
    sb.upload_csr(bad.n, bad.row_off, bad.head, bad.cap, bad.cost, bad.rev, bad.supply)
    for cap in ("1", "3", None):
        if cap is None:
            monkeypatch.delenv("PSN_BATCH_GROUP", raising=False)
        else:
            monkeypatch.setenv("PSN_BATCH_GROUP", cap)
        res = batch.solve_batch(solvers, workers=2)
        assert [r.total_cost for r in res] == want, cap
        for s, g, r in zip(solvers, graphs, res):
            assert capi.check(g, s.residual(), s.prices(), scale=s.price_scale())[:2] == (0, r.total_cost)
            assert capi.check_eps(g, s.residual(), s.prices(), 1, scale=s.price_scale()) == 0
    # one infeasible member: the call reports it, the other members are solved all the same
    with pytest.raises(_lib.PsnError) as ei:
        batch.solve_batch(solvers[:3] + [sb] + solvers[3:], workers=1)
    assert ei.value.code == _lib.PSN_EINFEASIBLE
    for s, g, w in zip(solvers, graphs, want):
        assert capi.check(g, s.residual(), s.prices(), scale=s.price_scale())[:2] == (0, w)
    # warm group solve after capacity changes on two members
    for i in (0, 4):
        g = graphs[i]
        fwd = np.nonzero(g.cap > 0)[0]
        idx = rng.choice(fwd, 20, replace=False)
        g.cap[idx] += 1
        solvers[i].update_caps(idx, g.cap[idx])
        rc, want[i], _, _, _ = capi.cost_scaling(g)
        assert rc == 0
    res = batch.solve_batch(solvers, warm_start=True, workers=2)
    assert [r.total_cost for r in res] == want
    for s in solvers + [sb]:
        s.close()


def test_cost_model_rejects_a_layout_it_cannot_price():
    """psn_eval_costs pairs every class row with the machine rows; arc bases that do not line
    up with the uploaded network are refused instead of pricing the wrong arcs."""
    cl = pa.synth_cluster(0, 1)
    net = pa.build_network(cl)
    with pa.Solver() as sv:
        sv.upload(net)
        good = net.ec_arc_base.copy()
        net.ec_arc_base = good - 8
        with pytest.raises(_lib.PsnError) as e:
            sv.eval_costs(cl, net)
        assert e.value.code == _lib.PSN_EINVAL
        net.ec_arc_base = good
        sv.upload(net)
        sv.eval_costs(cl, net)
        assert sv.solve().total_cost == 587046


@pytest.mark.parametrize("cfg,seed", [(0, 1), (1, 2)])
def test_network_built_on_the_device_equals_the_host_builder(cfg, seed):
    """psn_build_network_device writes, arc for arc, what psn_build_network lays out (and what the
    oracle's builder lays out: the arrays' FNV checksum is the committed golden one once priced)."""
    cl = pa.synth_cluster(cfg, seed)
    net = pa.build_network(cl)
    with pa.Solver(device=0) as s:
        dnet = s.build_network(cl)
        assert np.array_equal(dnet.ec_arc_base, net.ec_arc_base)
        head, rev = s.structure()
        row_off, supply = s.rows()
        cost, ocap = s.costs()
        assert np.array_equal(row_off, net.row_off) and np.array_equal(supply, net.supply)
        assert np.array_equal(head, net.head) and np.array_equal(rev, net.rev)
        assert np.array_equal(cost, net.cost) and np.array_equal(ocap, net.cap)
        assert np.array_equal(s.residual(), net.cap)
        s.eval_costs(cl, dnet)
        og = synth.build_graph(synth.make_cluster(cfg, seed))
        cost, ocap = s.costs()
        assert np.array_equal(cost, og.cost) and np.array_equal(ocap, og.cap)
        assert capi.fnv1a64(row_off, head, ocap, cost, rev, supply) == int(_gold(cfg, seed)["csr_fnv1a64"], 16)
        r = s.solve()
        assert r.total_cost == _gold(cfg, seed)["optimal_cost"]
        assert capi.check(og, s.residual(), s.prices(), scale=s.price_scale())[:2] == (0, r.total_cost)
        # malformed task tables are refused before anything is built
        bad = pa.synth_cluster(cfg, seed)
        bad.task_class = bad.task_class.copy(); bad.task_class[3] = 64
        with pytest.raises(pa.PsnError) as ei:
            s.build_network(bad)
        assert ei.value.code == _lib.PSN_EINVAL


@pytest.mark.parametrize("div", ["1", "3", "16"])
def test_phases_cut_short_still_end_at_the_optimum(solver, monkeypatch, div):
    """Phases other than the last may stop with some units of excess still on their way (by default
    1/16 of the supply, 176 units at most; here all, a third and a sixteenth of it without the upper
    bound, so that small networks exercise it): the leftovers are routed by the later phases and the
    result is the same optimum, certified."""
    monkeypatch.setenv("PSN_TAIL_DIV", div)
    rng = np.random.default_rng(int(div))
    for trial in range(12):
        n = int(rng.integers(20, 300))
        g = random_graph(rng, n, int(rng.integers(2 * n, 10 * n)), maxcap=int(rng.choice([1, 5, 50])), maxcost=2000, neg=trial % 3 == 0)
        r = solve_and_certify(solver, g)
        rc, tc, _, _ = capi.ssp(g)
        assert rc == 0 and tc == r.total_cost, (trial, tc, r.total_cost)
    e = _gold(0, 1)
    cl = pa.synth_cluster(0, 1)
    net = pa.build_network(cl)
    solver.upload(net)
    solver.eval_costs(cl, net)
    r = solver.solve()
    assert r.total_cost == e["optimal_cost"]
    og = synth.build_graph(synth.make_cluster(0, 1))
    assert capi.check(og, solver.residual(), solver.prices(), scale=solver.price_scale())[:2] == (0, r.total_cost)


def test_leaf_vertices_that_fetch_their_own_distance(solver):
    """Vertices of degree <= 4 whose neighbours all have a larger degree are left out of the
    frontier expansions of a price update and work their distance out themselves when the frontier
    runs dry.  Transportation networks whose many small sources AND many small sinks are such
    leaves around a few large transit vertices, with every route through a transit vertex and some
    routes through another leaf's transit neighbour only: optimum and certificate as for any graph."""
    rng = np.random.default_rng(77)
    for trial in range(10):
        n_mid = int(rng.integers(2, 6))
        n_src = int(rng.integers(20, 200))
        n_dst = int(rng.integers(20, 200))
        n = n_mid + n_src + n_dst
        tail, head, cap, cost = [], [], [], []
        for s_ in range(n_src):  # 1..3 arcs from each source leaf to transit vertices
            for m_ in rng.choice(n_mid, size=int(rng.integers(1, min(3, n_mid) + 1)), replace=False):
                tail.append(n_mid + s_); head.append(int(m_)); cap.append(int(rng.integers(1, 4))); cost.append(int(rng.integers(0, 500)))
        for d_ in range(n_dst):  # and from transit vertices to each sink leaf
            for m_ in rng.choice(n_mid, size=int(rng.integers(1, min(3, n_mid) + 1)), replace=False):
                tail.append(int(m_)); head.append(n_mid + n_src + d_); cap.append(int(rng.integers(1, 4))); cost.append(int(rng.integers(0, 500)))
        for a in range(n_mid):  # transit vertices among themselves
            for b in range(n_mid):
                if a != b:
                    tail.append(a); head.append(b); cap.append(1000); cost.append(int(rng.integers(1, 300)))
        tail, head, cap, cost = (np.asarray(x, np.int64) for x in (tail, head, cap, cost))
        # supplies from a random feasible flow: source -> transit -> sink paths
        supply = np.zeros(n, np.int64)
        out_cap = np.zeros(n, np.int64); in_cap = np.zeros(n, np.int64)
        np.add.at(out_cap, tail[tail >= n_mid], cap[tail >= n_mid])
        np.add.at(in_cap, head[head >= n_mid + n_src], cap[head >= n_mid + n_src])
        units = int(min(out_cap.sum(), in_cap.sum()) // 3)
        f = np.zeros(len(cap), np.int64)
        # greedy: fill source arcs and sink arcs up to `units` each, transit arcs carry the imbalance
        left = units
        for a in np.flatnonzero(tail >= n_mid):
            x = min(int(cap[a]), left); f[a] = x; left -= x
        left = units
        for a in np.flatnonzero(head >= n_mid + n_src):
            x = min(int(cap[a]), left); f[a] = x; left -= x
        np.add.at(supply, tail, f); np.add.at(supply, head, -f)
        # transit vertices are left unbalanced by the two greedy fills: move the difference to transit 0
        # over the transit arcs (capacity 1000 each, plenty)
        for m_ in range(1, n_mid):
            if supply[m_] != 0:
                supply[0] += supply[m_]; supply[m_] = 0
        g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
        rc, tc, _, _ = capi.ssp(g)
        solver.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
        if rc != 0:
            with pytest.raises(pa.PsnError):
                solver.solve()
            continue
        r = solve_and_certify(solver, g)
        assert tc == r.total_cost, (trial, tc, r.total_cost)


@pytest.mark.parametrize("always", [False, True])
def test_rewired_arcs_keep_or_remake_the_tables_made_from_the_heads(monkeypatch, always):
    """psn_patch_arcs makes the vertex cover, its independent subset and the other tables that follow
    from the heads again only when a patched arc breaks one of them.  Random graphs (a ring of wide,
    dear arcs keeps every supply routable) have the heads of two arcs exchanged - u -> v, x -> y become
    u -> y, x -> v, with their reverse arcs re-paired - several times in a row; after every exchange the
    solve (cold, and the kept state offered) ends at the oracle's optimum of the rewired graph with a
    certificate: by the rule, and with the tables made again after every patch."""
    if always:
        monkeypatch.setenv("PSN_PATCH_TABLES", "1")
    rng = np.random.default_rng(99)
    with pa.Solver(device=0) as s:
        for trial in range(6):
            n = int(rng.integers(12, 120))
            m = int(rng.integers(2 * n, 6 * n))
            tail = rng.integers(0, n, m)
            head = (tail + rng.integers(1, n, m)) % n
            cap = rng.integers(1, 6, m)
            cost = rng.integers(0, 40, m)
            f = rng.integers(0, cap + 1)
            supply = np.zeros(n, np.int64)
            np.add.at(supply, tail, f)
            np.add.at(supply, head, -f)
            ring = np.arange(n)
            tail = np.concatenate([tail, ring, (ring + 1) % n])
            head = np.concatenate([head, (ring + 1) % n, ring])
            cap = np.concatenate([cap, np.full(2 * n, 10_000)])
            cost = np.concatenate([cost, np.full(2 * n, 500)])
            g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
            s.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
            for step in range(4):
                r = s.solve(warm_start=step > 0)
                rc, tc, _, _ = capi.ssp(g)
                assert rc == 0 and r.total_cost == tc, (trial, step, always)
                assert capi.check(g, s.residual(), s.prices(), scale=s.price_scale())[:2] == (0, tc)
                # exchange the heads of two of the random arcs (forward arcs i and j of the list)
                i, j = (int(x) for x in rng.choice(m, 2, replace=False))
                a, b = int(g.fwd_pos[i]), int(g.fwd_pos[j])
                ra, rb = int(g.rev[a]), int(g.rev[b])
                u, v, x, y = int(tail[i]), int(g.head[a]), int(tail[j]), int(g.head[b])
                if u == y or x == v:
                    continue  # would make a self-loop
                ca, cb = int(g.cost[a]), int(g.cost[b])
                # a: u -> y paired with rb (in y's row, now y -> u); b: x -> v paired with ra (in v's row, now v -> x)
                idx = np.array([a, rb, b, ra], np.int64)
                nh = np.array([y, u, v, x], np.int32)
                nc = np.array([g.cap[a] + g.cap[ra], 0, g.cap[b] + g.cap[rb], 0], np.int32)  # original capacities, no flow
                nk = np.array([ca, -ca, cb, -cb], np.int32)
                nr = np.array([rb, a, ra, b], np.int32)
                assert s._lib.psn_patch_arcs(s._ctx, 4, idx.ctypes.data, nh.ctypes.data, nc.ctypes.data, nk.ctypes.data, nr.ctypes.data) == 0
                g.head[idx] = nh; g.cap[idx] = nc; g.cost[idx] = nk; g.rev[idx] = nr
                head[i], head[j] = y, v


@pytest.mark.parametrize("decline", [None, "-1", "40"])
def test_warm_start_that_is_no_head_start_is_declined(solver, monkeypatch, decline):
    """A kept state that the changes violate by about as much as an empty one (its first epsilon within
    a power of two of a cold solve's) is declined: the solve starts cold on the resident network and
    says so.  Slot costs of a solved cluster are changed wholesale (every third class -> machine arc by
    up to +-2000 of 0..2000) and the network re-solved with PSN_WARM_START: by the default rule, with the
    rule off (always resumes) and with a rule that always declines - the oracle's optimum of the changed
    network and a certificate every time; a small change (ten arcs by one unit) is resumed."""
    if decline is not None:
        monkeypatch.setenv("PSN_WARM_DECLINE", decline)
    cl = pa.synth_cluster(0, 2)
    net = pa.build_network(cl)
    solver.upload(net)
    solver.eval_costs(cl, net)
    r0 = solver.solve()
    assert r0.warm_declined == 0 and r0.warm_started == 0
    og = synth.build_graph(synth.make_cluster(0, 2))
    N, E, K = cl.layout.n_machines, cl.layout.n_classes, cl.layout.k_slots
    rng = np.random.default_rng(4)
    P = cl.layout.n_tasks
    slot = og.fwd_pos[2 * P:2 * P + E * N * K]  # the class -> machine slot arcs (after task -> class and task -> unscheduled)
    assert ((og.head[slot] >= 1) & (og.head[slot] <= N)).all()
    slot = slot[og.cap[slot] > 0]

    def change(arcs, delta):
        newc = np.maximum(0, og.cost[arcs] + delta).astype(np.int32)
        og.cost[arcs] = newc
        og.cost[og.rev[arcs]] = -newc
        solver.update_costs(arcs.astype(np.int64), newc)

    few = rng.choice(slot, 10, replace=False)
    change(few, np.ones(10, np.int64))
    r1 = solver.solve(warm_start=True)
    rc, tc, _, _, _ = capi.cost_scaling(og)
    assert rc == 0 and r1.total_cost == tc
    if decline != "40":
        assert r1.warm_started == 1 and r1.warm_declined == 0
    many = slot[::3]
    change(many, rng.integers(-2000, 2001, many.shape[0]))
    r2 = solver.solve(warm_start=True)
    rc, tc, _, _, _ = capi.cost_scaling(og)
    assert rc == 0 and r2.total_cost == tc
    assert r2.warm_started + r2.warm_declined == 1
    if decline == "-1":
        assert r2.warm_started == 1
    if decline == "40":
        assert r2.warm_declined == 1 and r1.warm_declined == 1
    if decline is None:
        assert r2.warm_declined == 1, (r2.first_eps_log2, r0.first_eps_log2)
    assert capi.check(og, solver.residual(), solver.prices(), scale=solver.price_scale())[:2] == (0, tc)


@pytest.mark.parametrize("shards", ["1", "0"])
def test_hub_purses_and_sharded_accumulators_on_and_off(monkeypatch, shards):
    """Networks from two million residual arcs keep their hubs' excess in per-hub purses during a push
    round and collect what the hubs receive in sharded accumulators that the relabel sweep folds back;
    smaller networks use the hubs' own words.  Both ways forced on networks of either kind: star-like
    random graphs whose hubs are sources, sinks and pass-through vertices (with replays on the reversed
    network forced, which turns every sink into a source), the 1 000-machine cluster, and - the other
    way round from its default - the full-size network; optimum and certificate as everywhere."""
    monkeypatch.setenv("PSN_HUB_SHARDS", shards)
    monkeypatch.setenv("PSN_MIRROR_DELTA", "40")
    rng = np.random.default_rng(321)
    with pa.Solver(device=0) as s:
        for trial in range(6):
            n = 500
            hubs = rng.integers(0, n, 4)
            m = 12000
            tail = np.where(rng.random(m) < 0.45, hubs[rng.integers(0, 4, m)], rng.integers(0, n, m))
            head = np.where(rng.random(m) < 0.45, hubs[rng.integers(0, 4, m)], rng.integers(0, n, m))
            keep = tail != head
            tail, head = tail[keep], head[keep]
            cap = rng.integers(1, int(rng.choice([2, 30, 1000])), tail.shape[0])
            cost = rng.integers(0, int(rng.choice([4, 5000])), tail.shape[0])
            f = rng.integers(0, cap + 1)
            supply = np.zeros(n, np.int64)
            np.add.at(supply, tail, f)
            np.add.at(supply, head, -f)
            g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
            assert (np.diff(g.row_off) > 1152).any()
            r = solve_and_certify(s, g)
            rc, tc, _, _, _ = capi.cost_scaling(g)
            assert rc == 0 and tc == r.total_cost, (trial, tc, r.total_cost)
            r2 = s.solve(warm_start=True)
            assert r2.total_cost == tc and r2.pushes == 0
        for cfg, seed in ((1, 2), (2, 2)) if shards == "0" else ((1, 2),):
            cl = pa.synth_cluster(cfg, seed)
            net = pa.build_network(cl)
            s.upload(net)
            s.eval_costs(cl, net)
            r = s.solve()
            assert r.total_cost == _gold(cfg, seed)["optimal_cost"]
            og = synth.build_graph(synth.make_cluster(cfg, seed))
            assert capi.check(og, s.residual(), s.prices(), scale=s.price_scale())[:2] == (0, r.total_cost)


@pytest.mark.parametrize("ratio", ["1", "3", "12"])
def test_replays_on_the_reversed_network_end_at_the_same_optimum(solver, monkeypatch, ratio):
    """Replays may work on the mirror image of the network (arcs reversed, supplies and prices
    negated, deficits pushed towards the excesses).  Forced on in every phase and with an eager, the
    default and a reluctant threshold: random graphs (negative costs, large capacities, hubs) and
    synthetic clusters reach the oracle's optimum with a certificate, the prices come back the way the
    network is stored, and a warm re-solve from the state a mirrored solve left finds nothing to do."""
    monkeypatch.setenv("PSN_MIRROR_DELTA", "40")
    monkeypatch.setenv("PSN_MIRROR_RATIO", ratio)
    rng = np.random.default_rng(77)
    mirrored = 0
    for i in range(24):
        n = int(rng.integers(5, 300))
        g = random_graph(rng, n, int(rng.integers(n, 12 * n)), maxcap=int(rng.choice([1, 4, 500])), maxcost=int(rng.choice([3, 50, 20000])),
                         neg=bool(i % 3 == 0))
        rc, tc, _, _ = capi.ssp(g)
        assert rc == 0
        r = solve_and_certify(solver, g)
        assert r.total_cost == tc
        mirrored += r.mirrored_replays
        if r.mirrored_replays:
            r2 = solver.solve(warm_start=True)
            assert r2.total_cost == tc and r2.warm_started == 1 and r2.pushes == 0
    for cfg, seed in ((0, 1), (0, 3), (1, 1)):
        cl = pa.synth_cluster(cfg, seed)
        net = pa.build_network(cl)
        solver.upload(net)
        solver.eval_costs(cl, net)
        r = solver.solve()
        assert r.total_cost == _gold(cfg, seed)["optimal_cost"]
        og = synth.build_graph(synth.make_cluster(cfg, seed))
        assert capi.check(og, solver.residual(), solver.prices(), scale=solver.price_scale())[:2] == (0, r.total_cost)
        mirrored += r.mirrored_replays
        assert r.mirrored_replays > 0 or ratio == "12"
    assert mirrored > 0
