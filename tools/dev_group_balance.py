"""Development aid: replays a lockstep group needs against what its members need on their own."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa
from poseidon_amd import batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
solvers, clusters, nets = [], [], []
for b in range(B):
    cl = pa.synth_cluster(1, 1 + b)
    s = pa.Solver()
    net = s.build_network(cl)
    solvers.append(s); clusters.append(cl); nets.append(net)
own = []
for s, cl, net in zip(solvers, clusters, nets):
    s.eval_costs(cl, net)
    r = s.solve()
    own.append((r.launches / 41.0, r.solve_ms))
rep = np.array([x[0] for x in own]); ms = np.array([x[1] for x in own])
print(f"alone: replays mean {rep.mean():.1f} min {rep.min():.0f} max {rep.max():.0f}; solve_ms mean {ms.mean():.2f} sum {ms.sum():.1f}")
for s, cl, net in zip(solvers, clusters, nets):
    s.eval_costs(cl, net)
res = batch.solve_batch(solvers)
print(f"group of {B}: launches {res[0].launches} = {res[0].launches / 41.0:.1f} replays, solve_ms {res[0].solve_ms:.1f} ({res[0].solve_ms / B:.2f} per network)")
