"""Development aid: work-list sizes per round (lane items, wave items) of a cold solve of a cluster of a given shape."""
import os
import sys

os.environ["PSN_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C

import numpy as np

import poseidon_amd as pa

nm, npods = int(sys.argv[1]), int(sys.argv[2])
cl = pa.synth_cluster(0, 7, nm, npods)
with pa.Solver() as s:
    net = s.build_network(cl)
    s.eval_costs(cl, net)
    r = s.solve()
    buf = np.zeros((1 << 16, 4), np.int32); cnt = C.c_int64()
    s._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    s._lib.psn_debug_trace(s._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
    tr = buf[: cnt.value]
print("solve_ms", r.solve_ms, "rounds", r.rounds, "gus", r.global_updates, "pushes", r.pushes, "cost", r.total_cost)
cur, i = [], 0
for kind, k, a, b in tr:
    if kind == 1:
        if k == 0 and cur:
            print(i, cur); i += 1; cur = []
        cur.append((int(a), int(b)))
print(i, cur)
