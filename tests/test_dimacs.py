"""DIMACS min-cost-flow import/export (host code, no GPU): round trips keep the optimum, a
hand-checked instance reads correctly, lower bounds are folded into supplies."""
import os
import stat

import numpy as np
import pytest

from poseidon_amd import _lib, dimacs
from oracle import capi, synth

TINY = """c four vertices; optimum 14: 2 units 1->2->4 (cost 2+1 each... see test), rest via 3
p min 4 5
n 1 4
n 4 -4
a 1 2 0 4 2
a 1 3 0 2 2
a 2 3 0 2 1
a 2 4 0 3 3
a 3 4 0 5 1
"""


def _solve(n, tail, head, cap, cost, supply):
    g = synth.graph_from_arcs(n, tail, head, cap, cost, supply)
    rc, tc, _, _, _ = capi.cost_scaling(g)
    assert rc == 0
    rc2, tc2, _, _ = capi.ssp(g)
    assert rc2 == 0 and tc2 == tc
    return tc


def test_hand_checked_instance(tmp_path):
    p = tmp_path / "tiny.min"
    p.write_text(TINY)
    d = dimacs.read(str(p))
    assert (d["n"], d["m"], d["base_cost"]) == (4, 5, 0)
    assert d["tail"].tolist() == [0, 0, 1, 1, 2] and d["head"].tolist() == [1, 2, 2, 3, 3]
    assert d["supply"].tolist() == [4, 0, 0, -4]
    # paths: 1-3-4 costs 3 (cap 2), 1-2-3-4 costs 4 (cap 2): 2*3 + 2*4 = 14
    assert _solve(4, d["tail"], d["head"], d["cap"], d["cost"], d["supply"]) == 14
    row_off, head, cap, cost, rev, fwd = dimacs.csr_from_arcs(4, d["tail"], d["head"], d["cap"], d["cost"])
    assert row_off.tolist() == [0, 2, 5, 8, 10] and (rev[rev] == np.arange(10)).all()
    assert (cost[rev] == -cost).all() and (cap[fwd] == d["cap"]).all() and (cap[rev[fwd]] == 0).all()


def test_lower_bounds_are_folded(tmp_path):
    p = tmp_path / "low.min"
    p.write_text("p min 3 3\nn 1 3\nn 3 -3\na 1 2 1 3 5\na 2 3 0 3 1\na 1 3 0 2 4\n")
    d = dimacs.read(str(p))
    assert d["base_cost"] == 5 and d["cap"].tolist() == [2, 3, 2] and d["supply"].tolist() == [2, 1, -3]
    # forced unit on 1->2 (5) continues 2->3 (1); the other two units take 1->3 (4 each): 5 + 1 + 8 = 14
    assert _solve(3, d["tail"], d["head"], d["cap"], d["cost"], d["supply"]) + d["base_cost"] == 14


def test_round_trip_keeps_the_optimum(tmp_path):
    g = synth.build_graph(synth.make_cluster(0, 1))
    p = str(tmp_path / "c0.min")
    dimacs.write(p, g.n, g.row_off, g.head, g.cap, g.cost, g.supply)
    d = dimacs.read(p)
    assert d["n"] == g.n and d["m"] == int((g.cap > 0).sum())
    assert _solve(d["n"], d["tail"], d["head"], d["cap"], d["cost"], d["supply"]) == 587046


def test_malformed_files_are_rejected(tmp_path):
    for text in ("p max 3 1\na 1 2 0 1 1\n", "p min 2 1\na 1 3 0 1 1\n", "p min 2 2\na 1 2 0 1 1\n", "p min 2 1\na 1 2 2 1 1\n"):
        p = tmp_path / "bad.min"
        p.write_text(text)
        with pytest.raises(_lib.PsnError):
            dimacs.read(str(p))
    with pytest.raises(_lib.PsnError):
        dimacs.read(str(tmp_path / "missing.min"))


def test_external_solver_hook(tmp_path, monkeypatch):
    p = tmp_path / "tiny.min"
    p.write_text(TINY)
    monkeypatch.delenv("PSN_EXT_SOLVER", raising=False)
    assert dimacs.run_external_solver(str(p)) is None
    exe = tmp_path / "fake_solver.sh"
    exe.write_text("#!/bin/sh\ncat > /dev/null\necho 'c done'\necho 's 14'\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PSN_EXT_SOLVER", str(exe))
    cost, secs = dimacs.run_external_solver(str(p))
    assert cost == 14 and secs >= 0
