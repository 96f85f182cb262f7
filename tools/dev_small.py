"""Development driver: random small generic graphs through the C ABI vs the oracle."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa
from oracle import capi, synth


def random_graph(rng, n, m, maxcap=5, maxcost=20):
    tail = rng.integers(0, n, m)
    head = rng.integers(0, n, m)
    keep = tail != head
    tail, head = tail[keep], head[keep]
    m = tail.shape[0]
    cap = rng.integers(1, maxcap + 1, m)
    cost = rng.integers(-2 if False else 0, maxcost + 1, m)
    # feasible supplies: route random flow along the arcs
    f = rng.integers(0, cap + 1)
    supply = np.zeros(n, np.int64)
    np.add.at(supply, tail, f)
    np.add.at(supply, head, -f)
    return synth.graph_from_arcs(n, tail, head, cap, cost, supply)


def main():
    rng = np.random.default_rng(7)
    eager = len(sys.argv) > 1 and sys.argv[1] == "eager"
    with pa.Solver(eager=eager) as s:
        for trial in range(int(os.environ.get("TRIALS", "20"))):
            n = int(rng.integers(3, 60))
            m = int(rng.integers(n, 8 * n))
            g = random_graph(rng, n, m)
            s.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
            r = s.solve()
            rc, tc, _, _ = capi.ssp(g)
            resid = s.residual()
            crc, ctc, _ = capi.check(g, resid, s.prices())
            print(f"trial {trial}: n {g.n} m {g.m} gpu {r.total_cost} ssp {tc} (rc {rc}) cert {crc} rounds {r.rounds} ms {r.solve_ms:.2f}", flush=True)
            assert rc == 0 and tc == r.total_cost and crc == 0 and ctc == tc


if __name__ == "__main__":
    main()
