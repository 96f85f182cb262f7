"""Development aid: solve time of one synthetic config over solver parameters (alpha, rounds per update)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
for alpha in (3, 4, 5, 6):
    for rounds in (8, 12, 16):
        with pa.Solver(alpha_log2=alpha, rounds_per_batch=rounds) as s:
            s.upload(net)
            ms, rr, gi = [], [], []
            for _ in range(reps):
                s.eval_costs(cl, net)
                r = s.solve()
                ms.append(r.solve_ms); rr.append(r.rounds); gi.append(r.gu_iterations)
            print(f"cfg {cfg} alpha_log2 {alpha} rounds {rounds}: solve_ms median {np.median(ms):.2f} min {min(ms):.2f} max {max(ms):.2f} "
                  f"rounds {int(np.median(rr))} gu_iters {int(np.median(gi))} refines {r.refines} cost {r.total_cost}", flush=True)
