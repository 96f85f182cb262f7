"""Development aid (diagnostic build with PSN_PROBE_APPEND): stage times of the list append in the relabel sweep."""
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
b = buf[buf[:, 13] == 4242].astype(np.int64)
print("solve_ms", r.solve_ms, "waves recorded", len(b))
cyc = 2300.0
for row in b:
    d = np.diff(row[:7]) / cyc
    print("nsplit %3d segs %5d | scan+sync %.2f stage+sync %.2f atomic+prefix+sync %.2f low/med stores %.2f segment stores %.2f drain %.2f | total %.2f us" % (row[7], row[8], *d, (row[6] - row[0]) / cyc))
