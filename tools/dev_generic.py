"""Development aid: solve times on networks that are not scheduling networks (random sparse graphs,
grids, dense transportation problems), with the early phase ends on and off; costs checked against
the CPU oracle's cost scaling.   python tools/dev_generic.py [reps]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def feasible_supply(rng, n, tail, head, cap):
    f = rng.integers(0, cap + 1)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f)
    np.add.at(supply, head, -f)
    return supply


def cases(rng):
    from oracle import synth
    n, m = 50_000, 400_000
    t, h = rng.integers(0, n, m), rng.integers(0, n, m)
    k = t != h; t, h = t[k], h[k]
    cap = rng.integers(1, 20, t.shape[0]); cost = rng.integers(0, 10_000, t.shape[0])
    yield "random sparse 50k/400k", synth.graph_from_arcs(n, t, h, cap, cost, feasible_supply(rng, n, t, h, cap))
    w = 200
    idx = np.arange(w * w).reshape(w, w)
    t = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel(), idx[:, 1:].ravel(), idx[1:, :].ravel()])
    h = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel(), idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    cap = rng.integers(1, 10, t.shape[0]); cost = rng.integers(1, 1000, t.shape[0])
    yield "grid 200x200", synth.graph_from_arcs(w * w, t, h, cap, cost, feasible_supply(rng, w * w, t, h, cap))
    a, b = 300, 2000
    t = np.repeat(np.arange(a), b); h = a + np.tile(np.arange(b), a)
    cap = rng.integers(1, 4, t.shape[0]); cost = rng.integers(0, 100_000, t.shape[0])
    f = (rng.random(t.shape[0]) < 0.01) * cap
    supply = np.zeros(a + b, np.int64); np.add.at(supply, t, f); np.add.at(supply, h, -f)
    yield "transportation 300x2000 dense", synth.graph_from_arcs(a + b, t, h, cap, cost, supply)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    import poseidon_amd as pa
    from oracle import capi
    reps = int(sys.argv[2])
    rng = np.random.default_rng(3)
    with pa.Solver() as s:
        for name, g in cases(rng):
            s.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
            ms = []
            for _ in range(reps):
                r = s.solve()
                ms.append(r.solve_ms)
            rc, tc, _, _, _ = capi.cost_scaling(g)
            print(f"  {name}: mean {np.mean(ms):.2f} min {min(ms):.2f} max {max(ms):.2f} ms rounds {r.rounds} gus {r.global_updates} cost {'ok' if rc == 0 and tc == r.total_cost else 'MISMATCH'}", flush=True)
else:
    reps = sys.argv[1] if len(sys.argv) > 1 else "5"
    for label, env in (("phases end early (default)", {}), ("phases run to the end", {"PSN_TAIL_UNITS": "0"})):
        print(label, flush=True)
        subprocess.run([sys.executable, __file__, "child", reps], env={**os.environ, **env}, check=True)
