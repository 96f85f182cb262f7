"""Development aid: mean cold solve time of several settings (environment variables the solver reads
at every solve) over a set of networks, several solves each; one table, settings as rows.
  python tools/dev_sweep.py <solves> <setting> [<setting> ...]     setting: NAME=VAL,NAME=VAL (or "base");
  ROUNDS= and ALPHA= are the solver's rounds per global update and log2 of its epsilon step
  NETS env: comma list of cfg:seed[:machines:pods]  (default 2:1,2:2,1:1,1:2,1:3,0:1,0:2,0:3)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

reps = int(sys.argv[1])
settings = sys.argv[2:]
nets = os.environ.get("NETS", "2:1,2:2,1:1,1:2,1:3,0:1,0:2,0:3").split(",")
rows = {s: [] for s in settings}
names = []
for spec in nets:
    f = [int(x) for x in spec.split(":")]
    cl = pa.synth_cluster(*f)
    net = pa.build_network(cl)
    names.append(spec)
    cost = None
    for st in settings:
        keys = []
        ctor = {}
        if st != "base":
            for kv in st.split(","):
                k, v = kv.split("=")
                if k == "ROUNDS":
                    ctor["rounds_per_batch"] = int(v)
                elif k == "ALPHA":
                    ctor["alpha_log2"] = int(v)
                else:
                    os.environ[k] = v
                    keys.append(k)
        with pa.Solver(**ctor) as sv:
            sv.upload(net)
            ms = []
            for i in range(reps + 1):
                sv.eval_costs(cl, net)
                r = sv.solve()
                if i:
                    ms.append(r.solve_ms)
                assert cost is None or cost == r.total_cost, (spec, st, cost, r.total_cost)
                cost = r.total_cost
            for k in keys:
                del os.environ[k]
            rows[st].append((float(np.mean(ms)), float(np.percentile(ms, 90))))
print("%-44s" % "setting (mean ms | p90)", " ".join("%13s" % n for n in names), "   rel")
base = rows[settings[0]]
for st in settings:
    rel = float(np.mean([a[0] / b[0] for a, b in zip(rows[st], base)]))
    print("%-44s" % st[:44], " ".join("%6.2f|%6.2f" % x for x in rows[st]), "  %.3f" % rel, flush=True)
