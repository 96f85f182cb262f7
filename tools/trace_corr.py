"""Run C2 with PSN_TRACE=1 under rocprofv3 and correlate frontier sizes with kernel durations."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["PSN_TRACE"] = "1"
import numpy as np
import poseidon_amd as pa
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cl = pa.synth_cluster(cfg, 1); net = pa.build_network(cl)
with pa.Solver() as s:
    s.upload(net); s.eval_costs(cl, net)
    r = s.solve()
    print("solve_ms", r.solve_ms, "cost", r.total_cost)
    buf = np.zeros((1 << 16, 4), np.int32); cnt = C.c_int64(0)
    s._lib.psn_debug_trace(s._ctx, buf.ctypes.data_as(C.c_void_p), C.c_int64(1 << 16), C.byref(cnt))
    np.save(os.path.join(os.environ.get("OUT", "."), "trace.npy"), buf[:cnt.value])
    print("records", cnt.value)
