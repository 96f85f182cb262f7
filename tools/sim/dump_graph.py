"""Dump a psn-synth-v1 graph as raw little-endian arrays for tools/sim/sync_sim.cpp."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from oracle import synth

cfg, seed, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cl = synth.make_cluster(cfg, seed)
g = synth.build_graph(cl)
with open(out, "wb") as f:
    np.array([g.n, 2 * g.m], np.int64).tofile(f)
    for a in (g.row_off, g.head, g.cap, g.cost, g.rev, g.supply):
        a.tofile(f)
print("wrote", out, g.n, g.m)
