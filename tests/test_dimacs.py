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


def test_supply_lines_after_arc_lines_keep_lower_bound_shifts(tmp_path):
    """An 'n' line that comes after the 'a' lines must not wipe out what lower bounds moved into the supplies."""
    a = tmp_path / "a.min"
    b = tmp_path / "b.min"
    a.write_text("p min 3 2\nn 1 5\nn 3 -5\na 1 2 2 6 1\na 2 3 0 6 1\n")
    b.write_text("p min 3 2\na 1 2 2 6 1\na 2 3 0 6 1\nn 1 5\nn 3 -5\n")
    da, db = dimacs.read(str(a)), dimacs.read(str(b))
    assert da["supply"].tolist() == db["supply"].tolist() == [3, 2, -5]
    assert da["base_cost"] == db["base_cost"] == 2 and da["cap"].tolist() == [4, 6]


def test_change_records_resolve_against_the_problem(tmp_path):
    p = tmp_path / "tiny.min"
    p.write_text(TINY)
    d = dimacs.read(str(p))
    ch = tmp_path / "tiny.chg"
    dimacs.write_changes(str(ch), arc_changes=[(1, 3, 1, 3), (0, 2, 2, 9)], supply_changes=[(0, 3), (3, -3)])
    c = dimacs.read_changes(str(ch), d)
    assert c["arc"].tolist() == [3, 1] and c["cap"].tolist() == [1, 2] and c["cost"].tolist() == [3, 9]
    assert c["vertex"].tolist() == [0, 3] and c["supply"].tolist() == [3, -3]
    cap, cost, supply = d["cap"].copy(), d["cost"].copy(), d["supply"].copy()
    cap[c["arc"]], cost[c["arc"]] = c["cap"], c["cost"]
    supply[c["vertex"]] = c["supply"]
    # 3 units: 1-2-3-4 (cost 4, cap 2), then 1-2-4 (cost 5, cap 1): 8 + 5 = 13; 1-3 now costs 9
    assert _solve(4, d["tail"], d["head"], cap, cost, supply) == 13
    # parallel arcs are told apart by their position in the problem file
    q = tmp_path / "par.min"
    q.write_text("p min 2 3\nn 1 2\nn 2 -2\na 1 2 0 1 5\na 1 2 0 1 7\na 1 2 0 1 9\n")
    dq = dimacs.read(str(q))
    chq = tmp_path / "par.chg"
    dimacs.write_changes(str(chq), arc_changes=[(0, 1, 1, 1, 2)])
    assert dimacs.read_changes(str(chq), dq)["arc"].tolist() == [2]
    # records that name no arc of the problem, add arcs, or carry a lower bound are refused
    for text in ("x 1 4 0 1 1\n", "a 1 2 0 1 1\n", "x 1 2 1 4 2\n", "x 1 2 0 4 2 7\n", "n 9 1\n"):
        bad = tmp_path / "bad.chg"
        bad.write_text(text)
        with pytest.raises(_lib.PsnError):
            dimacs.read_changes(str(bad), d)


def test_change_record_on_an_arc_with_a_folded_lower_bound_is_refused(tmp_path):
    """The reader stores capacity hi - lo for an arc with a lower bound and moves lo into the supplies and
    the base cost; an `x` record carries only a new upper bound and cost, so applying it to such an arc
    would give the wrong residual capacity and leave the base cost stale: refused.  The same record on an
    arc without a lower bound goes through."""
    p = tmp_path / "low.min"
    p.write_text("p min 3 3\nn 1 3\nn 3 -3\na 1 2 1 3 5\na 2 3 0 3 1\na 1 3 0 2 4\n")
    d = dimacs.read(str(p))
    assert d["low"].tolist() == [1, 0, 0]
    ok = tmp_path / "ok.chg"
    ok.write_text("x 2 3 0 2 1\n")
    assert dimacs.read_changes(str(ok), d)["arc"].tolist() == [1]
    bad = tmp_path / "bad.chg"
    bad.write_text("x 1 2 0 2 5\n")
    with pytest.raises(_lib.PsnError) as ei:
        dimacs.read_changes(str(bad), d)
    assert ei.value.code == _lib.PSN_EINVAL


def test_over_long_lines_are_refused_not_split(tmp_path):
    """A line longer than the readers' buffer would be cut and its tail parsed as a record of its own
    ("a 1 2 0 1 1" at the end of a long comment line); such files are refused instead."""
    pad = "c " + "x" * 600 + "\n"
    p = tmp_path / "long.min"
    p.write_text("p min 2 1\n" + pad + "n 1 1\nn 2 -1\na 1 2 0 1 1\n")
    with pytest.raises(_lib.PsnError):
        dimacs.read(str(p))
    # exactly at the limit (510 characters + newline) is fine
    fit = "c " + "y" * 508 + "\n"
    assert len(fit) == 511
    p.write_text("p min 2 1\n" + fit + "n 1 1\nn 2 -1\na 1 2 0 1 1\n")
    d = dimacs.read(str(p))
    assert d["m"] == 1
    ch = tmp_path / "long.chg"
    ch.write_text(pad + "x 1 2 0 1 1\n")
    with pytest.raises(_lib.PsnError):
        dimacs.read_changes(str(ch), d)
