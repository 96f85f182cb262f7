"""Development aid: time the frontier expansions of a global update, expansion by expansion, on the
state a finished solve left (library built with -DPSN_UBENCH, path in PSN_LIB; see tools/ubench/gu_bench.sh).

  PSN_LIB=bin_tmp/libpsn_ub.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ub -- \
      python3 tools/ubench/gu_bench.py run <config> <eps_shift>
  python3 tools/ubench/gu_bench.py report gpurun_out/ub
"""
import csv
import glob
import os
import sys

import numpy as np

N_EXP = 6
REPS = 8

if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import ctypes as C
    import poseidon_amd as pa
    cfg, shift = int(sys.argv[2]), int(sys.argv[3])
    cl = pa.synth_cluster(cfg, 1)
    net = pa.build_network(cl)
    with pa.Solver() as s:
        s.upload(net); s.eval_costs(cl, net)
        r = s.solve()
        sink = int(np.argmin(np.asarray(net.supply)))
        fn = s._lib.psn_debug_gu_bench
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        rc = fn(s._ctx, REPS, shift, sink, N_EXP)
        assert rc == 0, rc
    print("solve_ms", r.solve_ms, "sink", sink)
else:
    rows = list(csv.DictReader(open(sorted(glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv"))[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    res = {}
    i = 0
    while i < len(rows):
        if "k_ub_seed" not in rows[i]["Kernel_Name"]:
            i += 1
            continue
        for k in range(N_EXP):  # after the seed: k_gu_init, then the expansions
            r = rows[i + 2 + k]
            res.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        i += 2 + N_EXP
    for k in sorted(res):
        x = res[k]
        print(f"expansion {k}: n {len(x):3d} min {min(x):6.1f} median {np.median(x):6.1f} us")
