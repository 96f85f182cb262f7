"""Development aid (diagnostic build): cycle breakdown of the last light push round."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import poseidon_amd as pa
cl = pa.synth_cluster(2, 1); net = pa.build_network(cl)
with pa.Solver() as s:
    s.upload(net); s.eval_costs(cl, net); r = s.solve()
    buf = np.zeros((4096, 16), np.uint64)
    s._lib.psn_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert s._lib.psn_debug_stamps(s._ctx, buf.ctypes.data) == 0
b = buf[buf[:, 13] == 777].astype(np.int64)
print("solve_ms", r.solve_ms, "waves recorded", len(b), "nseg", b[0, 12])
busy = b[b[:, 5] > 0]
idle = b[b[:, 5] == 0]
cyc = 2300.0
def show(name, x): print(f"  {name:34s} mean {x.mean() / cyc:6.2f} us  p50 {np.percentile(x, 50) / cyc:6.2f}  p90 {np.percentile(x, 90) / cyc:6.2f}")
print("busy waves", len(busy), "idle waves", len(idle))
show("item list read", busy[:, 6]); show("price/excess/bounds", busy[:, 7]); show("arc rows", busy[:, 8]); show("head prices + keys", busy[:, 9])
show("claim + push part", busy[:, 10]); show("item tail", busy[:, 11]); show("items loop total", busy[:, 1] - busy[:, 0])
show("lane items loop", busy[:, 2] - busy[:, 1]); show("hub flush (barrier + atomics)", busy[:, 3] - busy[:, 2]); show("tally + drain", busy[:, 4] - busy[:, 3])
show("wave total", busy[:, 4] - busy[:, 0])
show("idle wave total", idle[:, 4] - idle[:, 0]); show("idle: hub flush", idle[:, 3] - idle[:, 2])
