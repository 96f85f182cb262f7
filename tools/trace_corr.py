"""Development aid: join PSN_TRACE records (work-list sizes per sweep launch) with a rocprofv3
kernel trace of the same process and print mean duration per work-size class.

  PSN_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tc -- python3 tools/trace_corr.py run gpurun_out/tc_trace.npy
  python3 tools/trace_corr.py report gpurun_out/tc gpurun_out/tc_trace.npy [rounds_per_replay]
"""
import csv
import glob
import os
import sys

import numpy as np

if sys.argv[1] == "run":
    os.environ["PSN_TRACE"] = "1"
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import ctypes as C
    import poseidon_amd as pa
    cl = pa.synth_cluster(int(os.environ.get("PSN_TC_CONFIG", "2")), 1)
    net = pa.build_network(cl)
    with pa.Solver() as s:
        s.upload(net); s.eval_costs(cl, net)
        r = s.solve()
        buf = np.zeros((1 << 20, 4), np.int32); cnt = C.c_int64()
        s._lib.psn_debug_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        s._lib.psn_debug_trace(s._ctx, buf.ctypes.data, buf.shape[0], C.byref(cnt))
        np.save(sys.argv[2], buf[: cnt.value])
    print("solve_ms", r.solve_ms, "records", cnt.value)
else:
    d, tr = sys.argv[2], np.load(sys.argv[3])
    R = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*_kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ti = 0; pk = 0; ek = 0; out = []
    for r in rows:
        nm = r["Kernel_Name"]
        if "k_push" in nm: kind, k = 1, pk; pk = (pk + 1) % R
        elif "k_gu_expand" in nm: kind, k = 2, ek; ek = (ek + 1) % 12
        else: continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if ti < len(tr) and (tr[ti][0] == kind or (kind == 2 and tr[ti][0] == 3)) and tr[ti][1] == k: out.append((kind, int(tr[ti][2]), int(tr[ti][3]), dur)); ti += 1
        else: out.append((kind, 0, 0, dur))
    print("matched", ti, "of", len(tr))
    edges = [0, 1, 16, 64, 256, 1024, 4096, 16384, 1 << 30]
    for kind, name in ((1, "push"), (2, "gu_expand")):
        print(name, " rows: lane items, cols: wave items", edges[1:])
        x = [o for o in out if o[0] == kind]
        for i in range(len(edges) - 1):
            line = f"  lane<{edges[i + 1]:>10}: "
            for j in range(len(edges) - 1):
                sel = [o[3] for o in x if edges[i] <= o[1] < edges[i + 1] and edges[j] <= o[2] < edges[j + 1]]
                line += f"{len(sel):5d}x{(np.mean(sel) if sel else 0):6.1f}us({(sum(sel) / 1e3):5.1f}ms) "
            print(line)
