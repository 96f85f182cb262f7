"""DIMACS files through the HIP solver: a synthetic cluster written as a DIMACS problem, read
back, uploaded and solved on the GPU gives the golden optimum; a churn round written as change
records (capacities and costs of the re-priced machines' slot arcs, machines taken out, supplies)
and applied in place gives, warm-started, the oracle's optimum of the changed network."""
import dataclasses
import json
import os

import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import dimacs
from oracle import capi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "psn_synth_v1.json")


def test_dimacs_problem_and_change_records_on_the_gpu(tmp_path):
    cfg, seed = 0, 1
    cl = synth.make_cluster(cfg, seed)
    g = synth.build_graph(cl)
    path = str(tmp_path / "c0.min")
    dimacs.write(path, g.n, g.row_off, g.head, g.cap, g.cost, g.supply, comment="psn-synth-v1 config 0 seed 1")
    d = dimacs.read(path)
    assert d["n"] == g.n and d["m"] == int((g.cap > 0).sum()) and d["base_cost"] == 0
    row_off, head, cap, cost, rev, fwd = dimacs.csr_from_arcs(d["n"], d["tail"], d["head"], d["cap"], d["cost"])
    gold = [e for e in json.load(open(GOLD))["entries"] if e["config"] == cfg and e["seed"] == seed][0]["optimal_cost"]
    with pa.Solver(device=0) as s:
        s.upload_csr(d["n"], row_off, head, cap, cost, rev, d["supply"])
        r = s.solve()
        assert r.total_cost == gold == 587046
        # ---- one churn round as change records: every arc of the file whose capacity or cost differs
        churned, _ = synth.churn(cl, cfg, seed, 1)
        # the machine side of the churn (new usage, machines taken out) edits capacities and costs in place;
        # the task side would rewire arcs, which change records do not do
        cl2 = dataclasses.replace(cl, used_cpu=churned.used_cpu, used_mem=churned.used_mem, machine_out=churned.machine_out)
        g2 = synth.build_graph(cl2)  # same layout: only capacities and costs differ
        assert np.array_equal(g2.head, g.head) and np.array_equal(g2.row_off, g.row_off)
        tail_of = np.repeat(np.arange(g.n), np.diff(g.row_off))
        live = np.nonzero(g.cap > 0)[0]                      # arcs the problem file holds, in file order
        diff = live[(g2.cap[live] != g.cap[live]) | (g2.cost[live] != g.cost[live])]
        assert diff.shape[0] > 0
        # parallel arcs (slot arcs of one class and machine) are told apart by their position among the file's arcs
        recs = []
        for a in diff:
            same = live[(tail_of[live] == tail_of[a]) & (g.head[live] == g.head[a])]
            recs.append((tail_of[a], g.head[a], g2.cap[a], g2.cost[a], int(np.nonzero(same == a)[0][0])))
        # supplies: two tasks leave (their unit of supply goes, the sink takes two units fewer)
        P = cl.n_pods
        gone = [g.n - 1, g.n - 7]
        sup_changes = [(v, 0) for v in gone] + [(0, -(P - len(gone)))]
        chg = str(tmp_path / "c0.chg")
        dimacs.write_changes(chg, arc_changes=recs, supply_changes=sup_changes)
        c = dimacs.read_changes(chg, d)
        assert c["arc"].shape[0] == diff.shape[0]
        pos = fwd[c["arc"]]
        s.update_caps(pos, c["cap"])
        s.update_costs(pos, c["cost"])
        if c["vertex"].shape[0]:
            s.update_supplies(c["vertex"], c["supply"])
        rw = s.solve(warm_start=True)
        # the oracle on the changed problem; arcs absent from the file (no capacity before) cannot have come back
        og = synth.graph_from_arcs(d["n"], d["tail"], d["head"], d["cap"], d["cost"], d["supply"])
        ocap, ocost, osup = d["cap"].copy(), d["cost"].copy(), d["supply"].copy()
        ocap[c["arc"]], ocost[c["arc"]] = c["cap"], c["cost"]
        osup[c["vertex"]] = c["supply"]
        og2 = synth.graph_from_arcs(d["n"], d["tail"], d["head"], ocap, ocost, osup)
        rc, tc, _, _, _ = capi.cost_scaling(og2)
        assert rc == 0 and rw.total_cost == tc
        s.upload_csr(d["n"], *dimacs.csr_from_arcs(d["n"], d["tail"], d["head"], ocap, ocost)[:5], osup)
        assert s.solve().total_cost == tc  # and a cold solve of the changed problem agrees
        del og
