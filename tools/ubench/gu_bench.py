"""Development aid: time variants of the frontier-expansion sweep on the state a finished solve
left (library built with -DPSN_UBENCH, path in PSN_LIB).

  PSN_LIB=bin_tmp/libpsn_ub.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ub -- \
      python3 tools/ubench/gu_bench.py run <config> <eps_shift> <variants comma separated>
  python3 tools/ubench/gu_bench.py report gpurun_out/ub
"""
import csv
import glob
import os
import sys

import numpy as np

MAX_KT = 4
REPS = 5

if sys.argv[1] == "run":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import ctypes as C
    import poseidon_amd as pa
    cfg, shift = int(sys.argv[2]), int(sys.argv[3])
    variants = [int(x) for x in sys.argv[4].split(",")]
    cl = pa.synth_cluster(cfg, 1)
    net = pa.build_network(cl)
    with pa.Solver() as s:
        s.upload(net); s.eval_costs(cl, net)
        r = s.solve()
        sink = int(np.argmin(np.asarray(net.supply)))
        fn = s._lib.psn_debug_gu_bench
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        for v in variants:
            rc = fn(s._ctx, v, REPS, shift, sink, MAX_KT)
            assert rc == 0, rc
    print("solve_ms", r.solve_ms, "sink", sink)
else:
    rows = list(csv.DictReader(open(sorted(glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv"))[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # after each k_ub_seed: k_gu_init, kt baseline expansions, one k_ub_expand<V>
    res = {}
    i = 0
    while i < len(rows):
        if "k_ub_seed" not in rows[i]["Kernel_Name"]:
            i += 1
            continue
        j = i + 2
        base = []
        while "k_ub_expand" not in rows[j]["Kernel_Name"]:
            base.append((int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3)
            j += 1
        v = rows[j]["Kernel_Name"]
        v = v[v.index("<") + 1:v.index(">")]
        d = (int(rows[j]["End_Timestamp"]) - int(rows[j]["Start_Timestamp"])) / 1e3
        res.setdefault((v, len(base)), []).append(d)
        res.setdefault(("ship", len(base) - 1), []).append(base[-1])
        i = j + 1
    for key in sorted(res):
        x = res[key]
        print(f"variant {key[0]:>5} iteration {key[1]}: n {len(x):3d} min {min(x):6.1f} median {np.median(x):6.1f} us")
