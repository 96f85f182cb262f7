"""Development aid: stop cold solves after a number of replays and, for the units of excess that
are left, work out on the host the shortest residual paths to a deficit under the prices at that
moment (reduced costs + 1 per arc, as the global update measures them at epsilon 1): how many arcs
they have, where they end, and how many of them share an arc of capacity 1."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import dijkstra

import poseidon_amd as pa

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
stop = sys.argv[2] if len(sys.argv) > 2 else "45"
tries = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cl = pa.synth_cluster(cfg, 1)
net = pa.build_network(cl)
N, P, E = cl.cap_cpu.shape[0], cl.task_class.shape[0], 64
J = net.n - 1 - N - E - P
row = np.asarray(net.row_off, np.int64)
tail = np.repeat(np.arange(net.n, dtype=np.int64), np.diff(row))
head = np.asarray(net.head, np.int64)
kind = lambda v: "sink" if v == 0 else "machine" if v <= N else "class" if v <= N + E else "unsched" if v <= N + E + J else "task"

with pa.Solver() as s:
    s.upload(net)
    for i in range(tries):
        s.eval_costs(cl, net)
        cost, ocap = s.costs()
        os.environ["PSN_MAX_REPLAYS"] = stop
        try:
            r = s.solve()
            print("solve", i, "finished within the limit: rounds", r.rounds, flush=True)
            continue
        except pa._lib.PsnError:
            pass
        finally:
            os.environ.pop("PSN_MAX_REPLAYS", None)
        cap = s.residual().astype(np.int64)
        price = s.prices()
        S = s.price_scale()
        flow = np.where(ocap > 0, ocap - cap, 0).astype(np.int64)
        ex = np.asarray(net.supply, np.int64).copy()
        np.subtract.at(ex, tail, flow); np.add.at(ex, head, flow)
        act = np.nonzero(ex > 0)[0]; dfc = np.nonzero(ex < 0)[0]
        print("solve", i, "stopped: active", len(act), "units", int(ex[act].sum()), "deficit vertices", len(dfc), dict(Counter(kind(v) for v in dfc)), flush=True)
        live = np.nonzero(cap > 0)[0]
        rc = cost[live].astype(np.int64) * S + price[tail[live]] - price[head[live]]
        print("   residual arcs", len(live), "reduced cost min", int(rc.min()), "below 0:", int((rc < 0).sum()))
        w = np.maximum(rc, -1) + 1  # arc length at epsilon 1
        # reversed graph with a super source (vertex n) at the deficits: distance of v = its distance to the nearest deficit
        n = net.n
        rows = np.concatenate([head[live], np.full(len(dfc), n)]); cols = np.concatenate([tail[live], dfc])
        vals = np.concatenate([w.astype(np.float64) + 1e-9, np.full(len(dfc), 1e-9)])
        g = csr_matrix((vals, (rows, cols)), shape=(n + 1, n + 1))
        dist, pred = dijkstra(g, directed=True, indices=n, return_predecessors=True)
        hops, ends, used = [], [], Counter()
        for v in act:
            h, x = 0, v
            while pred[x] != n and pred[x] >= 0 and h < 1000:
                used[(int(pred[x]), int(x))] += 1  # arc x -> pred[x] of the original graph
                x = pred[x]; h += 1
            hops.append(h); ends.append(kind(x))
        print("   path arcs per active vertex:", sorted(Counter(hops).items()), " ends at", dict(Counter(ends)))
        print("   distance (units of epsilon) per active vertex: min %.0f median %.0f max %.0f" % (dist[act].min(), np.median(dist[act]), dist[act].max()))
        shared = [c for c in used.values() if c > 1]
        print("   arcs on these paths:", len(used), " used by more than one path:", len(shared), " most shared:", sorted(shared)[-5:])
