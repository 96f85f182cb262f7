"""Development aid: distribution of the cold solve time of one synthetic config over many solves
(the solve is not reproducible to the round, so single runs do not separate small effects).
  python tools/dev_stats.py <cfg> <solves> [label]      env: ROUNDS, ALPHA, PSN_TAIL_UNITS, ..."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
label = sys.argv[3] if len(sys.argv) > 3 else ""
cl = pa.synth_cluster(cfg, int(os.environ.get("SEED", "1")))
net = pa.build_network(cl)
ms, rounds, gus, cost, mir, gui = [], [], [], set(), [], []
with pa.Solver(rounds_per_batch=int(os.environ.get("ROUNDS", "0")), alpha_log2=int(os.environ.get("ALPHA", "0"))) as s:
    s.upload(net)
    for i in range(reps + 2):
        s.eval_costs(cl, net)
        r = s.solve()
        if i >= 2:
            ms.append(r.solve_ms); rounds.append(r.rounds); gus.append(r.global_updates); cost.add(r.total_cost); mir.append(r.mirrored_replays); gui.append(r.gu_iterations)
ms = np.array(ms)
print(f"{label:28s} cfg {cfg} n {reps}: mean {ms.mean():6.2f} +- {ms.std() / np.sqrt(reps):4.2f}  median {np.median(ms):6.2f}  p10 {np.percentile(ms, 10):6.2f} "
      f"p90 {np.percentile(ms, 90):6.2f} max {ms.max():6.2f}  rounds {np.mean(rounds):6.0f}  gus {np.mean(gus):5.1f} gu_iters {np.mean(gui):5.0f} mirrored {np.mean(mir):4.1f}  costs {sorted(cost)}", flush=True)
