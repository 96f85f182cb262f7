"""Development aid: stop cold solves after a number of replays and say where the excess sits."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
stop = sys.argv[2] if len(sys.argv) > 2 else "50"
tries = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
N, P = cl.cap_cpu.shape[0], cl.task_class.shape[0]
E = 64
J = net.n - 1 - N - E - P
row = np.asarray(net.row_off, np.int64)
tail = np.repeat(np.arange(net.n, dtype=np.int64), np.diff(row))
head = np.asarray(net.head, np.int64)


def kind(v):
    if v == 0: return "sink"
    if v <= N: return "machine"
    if v <= N + E: return "class"
    if v <= N + E + J: return "unsched"
    return "task"


with pa.Solver() as s:
    s.upload(net)
    for i in range(tries):
        s.eval_costs(cl, net)
        cost, ocap = s.costs()
        os.environ["PSN_MAX_REPLAYS"] = stop
        try:
            r = s.solve()
            print("solve", i, "finished within the limit: rounds", r.rounds)
            continue
        except pa._lib.PsnError as e:
            pass
        finally:
            os.environ.pop("PSN_MAX_REPLAYS", None)
        cap = s.residual().astype(np.int64)
        price = s.prices()
        S = s.price_scale()
        flow = np.where(ocap > 0, ocap - cap, 0).astype(np.int64)  # forward arcs carry original capacity
        ex = np.asarray(net.supply, np.int64).copy()
        np.subtract.at(ex, tail, flow)
        np.add.at(ex, head, flow)
        act = np.nonzero(ex > 0)[0]
        dfc = np.nonzero(ex < 0)[0]
        print("solve", i, "stopped: active", len(act), "units", int(ex[act].sum()), "deficit vertices", len(dfc), "deficit", int(-ex[dfc].sum()))
        from collections import Counter
        print("   active by kind", dict(Counter(kind(v) for v in act)), " units by kind", {k: int(sum(ex[v] for v in act if kind(v) == k)) for k in ("machine", "class", "unsched", "task")})
        print("   deficits by kind", dict(Counter(kind(v) for v in dfc)))
        for v in act[:12]:
            a0, a1 = row[v], row[v + 1]
            live = np.nonzero(cap[a0:a1] > 0)[0] + a0
            rc = cost[live].astype(np.int64) * S + price[v] - price[head[live]]
            neg = live[rc < 0]
            print("    ", kind(v), v, "excess", int(ex[v]), "live arcs", len(live), "admissible", len(neg), "min reduced cost", int(rc.min()) if len(rc) else None,
                  "heads of live:", Counter(kind(h) for h in head[live]).most_common(3))
