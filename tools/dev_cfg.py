"""Development aid: cold solves of a synthetic configuration, mean / min / max solve time and the
last solve's counters.   python tools/dev_cfg.py <config> [steps] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cl = pa.synth_cluster(cfg, seed)
net = pa.build_network(cl)
ms = []
with pa.Solver() as s:
    s.upload(net)
    for i in range(steps + 1):
        s.eval_costs(cl, net)
        r = s.solve()
        if i: ms.append(r.solve_ms)
print(f"cfg {cfg} seed {seed} {os.environ.get('PSN_TAIL_DIV', '-')}/{os.environ.get('PSN_TAIL_ITEMS', '-')}: mean {np.mean(ms):.2f} min {min(ms):.2f} max {max(ms):.2f} ms  cost {r.total_cost} rounds {r.rounds} gu_iters {r.gu_iterations} gus {r.global_updates} refines {r.refines}")
