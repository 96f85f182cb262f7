"""Development aid: cold solves of clusters of other shapes than the benchmark's, early phase ends on and off."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

SHAPES = [(500, 20_000), (2_000, 40_000), (10_000, 10_000), (5_000, 100_000), (20_000, 200_000)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import poseidon_amd as pa
    for nm, npods in SHAPES:
        cl = pa.synth_cluster(0, 7, nm, npods)
        with pa.Solver() as s:
            net = s.build_network(cl)
            ms = []
            for _ in range(4):
                s.eval_costs(cl, net)
                r = s.solve()
                ms.append(r.solve_ms)
        print(f"  {nm} machines / {npods} pods: mean {np.mean(ms[1:]):.2f} min {min(ms[1:]):.2f} max {max(ms[1:]):.2f} ms rounds {r.rounds} gus {r.global_updates} cost {r.total_cost}", flush=True)
else:
    for label, env in (("phases end early (default)", {}), ("phases run to the end", {"PSN_TAIL_UNITS": "0"})):
        print(label, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, check=True)
