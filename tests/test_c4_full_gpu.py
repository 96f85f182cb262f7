"""The what-if batch at its stated size: 256 independent 1 000-machine / 10 000-pod networks
(network b uses seed b + 1) priced, solved in lockstep groups and extracted on one GPU.  Every
optimum is compared with a committed golden list (tests/golden/psn_c4_costs.json, written by
tests/golden/make_golden.py --c4 from the CPU oracle with its certificate), 32 members spread over
the batch carry their own optimality certificate here, and the multi-GPU sharding rule
(network b on rank b mod G) reassembles the per-rank results into graph order for G = 2, 4, 8."""
import json
import os

import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import batch
from oracle import capi, synth

pytestmark = pytest.mark.gpu

B = 256
CERTIFIED = list(range(0, B, 8))  # 32 members


def _golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "psn_c4_costs.json")) as f:
        g = json.load(f)
    assert g["graphs"] == B and g["config"] == 1
    return g


def test_256_config1_networks_match_golden_costs():
    gold = _golden()
    solvers, clusters = [], []
    for b in range(B):
        cl = pa.synth_cluster(1, b + 1)
        net = pa.build_network(cl)
        s = pa.Solver()
        s.upload(net)
        s.eval_costs(cl, net)
        if b in CERTIFIED:  # builder + cost kernel emit the network the golden list was made from
            cost, ocap = s.costs()
            assert "%016x" % capi.fnv1a64(net.row_off, net.head, ocap, cost, net.rev, net.supply) == gold["csr_fnv1a64"][b]
        solvers.append(s); clusters.append(cl)
    res = batch.solve_batch(solvers)
    costs = [r.total_cost for r in res]
    assert all(r.status == 0 for r in res)
    assert costs == gold["optimal_cost"]
    place = np.stack([s.placements() for s in solvers])
    assert place.shape == (B, 10_000)
    for b in CERTIFIED:
        og = synth.build_graph(synth.make_cluster(1, b + 1))
        s = solvers[b]
        resid, price = s.residual(), s.prices()
        assert capi.check(og, resid, price, scale=s.price_scale())[:2] == (0, costs[b])
        assert capi.check_eps(og, resid, price, 1, scale=s.price_scale()) == 0
        routed = int((og.cap - resid)[og.fwd_pos[:10_000]].sum())  # flow on task -> class arcs
        assert int((place[b] >= 0).sum()) == routed
        assert np.bincount(place[b][place[b] >= 0], minlength=1000).max() <= clusters[b].slots
    # what the ranks of a G-GPU run would each hold, put back into graph order the way the
    # all-gather does (rank-major blocks of ceil(B / G) rows, row i of rank r = network r + i * G)
    for world in (2, 4, 8):
        per = (B + world - 1) // world
        gathered = np.full((world * per, 10_000), -2, np.int32)
        gcosts = np.zeros(world * per, np.int64)
        seen = []
        for rank in range(world):
            mine = batch.shard(B, world, rank)
            assert mine == [b for b in range(B) if b % world == rank]
            seen += mine
            gathered[rank * per:rank * per + len(mine)] = place[mine]
            gcosts[rank * per:rank * per + len(mine)] = np.array(costs, np.int64)[mine]
        assert sorted(seen) == list(range(B))
        order = batch.graph_order(B, world)
        assert np.array_equal(gathered[order], place)
        assert gcosts[order].tolist() == gold["optimal_cost"]
    for s in solvers:
        s.close()
