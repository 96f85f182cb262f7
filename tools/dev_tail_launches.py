"""Development aid: the sweep launches of the last replays of one cold solve (kind 1 push round,
2 frontier expansion, 3 pull sweep; k; lane-list size; item-list size), from the launch trace
(PSN_TRACE=1).  Shows what the last units of a solve do round by round.
  python tools/dev_tail_launches.py [cfg] [last_n_records]"""
import ctypes as C
import os
import sys

os.environ["PSN_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
with pa.Solver() as s:
    s.upload(net)
    for _ in range(3):
        s.eval_costs(cl, net)
        r = s.solve()
    buf = np.zeros((1 << 16, 4), np.int32)
    cnt = C.c_int64()
    s._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    s._lib.psn_debug_trace(s._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
print("solve_ms", r.solve_ms, "rounds", r.rounds, "gu_iters", r.gu_iterations, "records", cnt.value)
rec = buf[:cnt.value][-last:]
line = []
for kind, k, a, b in rec.tolist():
    if kind == 1 and k == 0:
        print(" ".join(line)); line = []
    line.append({1: "P", 2: "E", 3: "L"}[kind] + f"{k}:{a}+{b}")
print(" ".join(line))
