"""Development aid: the network of a keep-running churn round (service, config 2), exported and
solved on its own with the launch trace on: work-list sizes per round, replay by replay."""
import os
import sys

os.environ["PSN_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ctypes as C

import numpy as np

import poseidon_amd as pa
from oracle import synth
from svc_util import apply_churn, exported_graph, feed_cluster

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cl = synth.make_cluster(cfg, 1)
with pa.SchedulerService(device=0) as s:
    uids = feed_cluster(s, cl)
    s.Schedule()
    nxt = int(uids.max()) + 1
    cl, ch = synth.churn(cl, cfg, 1, 1)
    uids = apply_churn(s, cl, ch, uids, nxt)
    g, _ = exported_graph(s)
m2 = int(g.row_off[g.n]); print("network", g.n, m2, "supply", int(g.supply[g.supply > 0].sum()))
if len(sys.argv) > 2:
    import struct
    with open(sys.argv[2], "wb") as f:  # the simulator's input format
        f.write(struct.pack("qq", g.n, m2))
        for a, t in ((g.row_off, np.int64), (g.head, np.int32), (g.cap, np.int32), (g.cost, np.int32), (g.rev, np.int32), (g.supply, np.int64)):
            f.write(np.ascontiguousarray(a, t).tobytes())
with pa.Solver() as sv:
    sv.upload_csr(g.n, g.row_off, g.head, g.cap, g.cost, g.rev, g.supply)
    r = sv.solve()
    buf = np.zeros((1 << 16, 4), np.int32); cnt = C.c_int64()
    sv._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    sv._lib.psn_debug_trace(sv._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
    tr = buf[: cnt.value]
print("solve_ms", r.solve_ms, "rounds", r.rounds, "gus", r.global_updates, "cost", r.total_cost)
cur = []
i = 0
for kind, k, a, b in tr:
    if kind == 1:
        if k == 0 and cur:
            print(i, cur); i += 1; cur = []
        cur.append((int(a), int(b)))
print(i, cur)
