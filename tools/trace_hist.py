"""Development aid: histogram of work-list sizes per sweep launch (PSN_TRACE=1) for one solve."""
import ctypes as C
import os
import sys

os.environ["PSN_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
with pa.Solver() as s:
    s.upload(net)
    s.eval_costs(cl, net)
    r = s.solve()
    buf = np.zeros((1 << 20, 4), np.int32)
    cnt = C.c_int64()
    s._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    s._lib.psn_debug_trace(s._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
    tr = buf[: cnt.value]
print("solve_ms", r.solve_ms, "rounds", r.rounds, "gu_iters", r.gu_iterations, "records", len(tr))
edges = [0, 1, 4, 16, 64, 256, 1024, 4096, 16384, 1 << 30]
for kind, name in ((1, "push"), (2, "gu_expand")):
    x = tr[tr[:, 0] == kind]
    print(name, len(x))
    for col, lab in ((2, "lane items"), (3, "wave items")):
        h = np.histogram(x[:, col], bins=edges)[0]
        print("   ", lab, " ".join(f"<{edges[i + 1]}:{h[i]}" for i in range(len(h))))
    small = (x[:, 2] <= 256) & (x[:, 3] <= 16)
    mid = (x[:, 2] <= 4096) & (x[:, 3] <= 256) & ~small
    print("    launches with <=256 lane and <=16 wave items:", int(small.sum()), " with <=4096 lane and <=256 wave items:", int(mid.sum()))
