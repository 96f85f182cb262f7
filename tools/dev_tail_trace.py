"""Development aid: work-list sizes per round in the last replays of the slowest of several cold solves."""
import ctypes as C
import os
import sys

os.environ["PSN_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
worst = None
with pa.Solver() as s:
    s.upload(net)
    s._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    for i in range(tries):
        s.eval_costs(cl, net)
        r = s.solve()
        buf = np.zeros((1 << 18, 4), np.int32); cnt = C.c_int64()
        s._lib.psn_debug_trace(s._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
        print("solve", i, "ms", round(r.solve_ms, 2), "rounds", r.rounds, "gus", r.global_updates, "records", cnt.value, flush=True)
        if worst is None or r.rounds > worst[0]: worst = (r.rounds, buf[: cnt.value].copy())
tr = worst[1]
reps, cur = [], []
for kind, k, a, b in tr:
    if kind == 1:
        if k == 0 and cur: reps.append(cur); cur = []
        cur.append(int(a) + int(b))
    elif kind in (2, 3):
        cur.append(-(int(a) + int(b)))
reps.append(cur)
print("replays with rounds:", len(reps))
for i, c in enumerate(reps[-45:]): print(len(reps) - 45 + i, "push:", [x for x in c if x >= 0], " gu:", [-x for x in c if x < 0])
