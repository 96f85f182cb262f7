"""Development driver: solve one synthetic config on the GPU, check against the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import poseidon_amd as pa
from oracle import capi, synth


def run(cfg, seed, eager=False, rounds=0, alpha=0, check=True, nm=None, np_=None, repeat=1):
    cl = pa.synth_cluster(cfg, seed, nm, np_)
    net = pa.build_network(cl)
    with pa.Solver(eager=eager, rounds_per_batch=rounds, alpha_log2=alpha, eps_from_max=bool(int(os.environ.get('EPSMAX', '0')))) as s:
        t = time.time(); s.upload(net); t_up = time.time() - t
        t = time.time(); s.eval_costs(cl, net); t_k1 = time.time() - t
        for _ in range(repeat):
            s.eval_costs(cl, net)
            r = s.solve()
        print(f"cfg {cfg} seed {seed} eager {eager} n {net.n} m {net.m}: cost {r.total_cost} solve_ms {r.solve_ms:.2f} "
              f"refines {r.refines} rounds {r.rounds} gus {r.global_updates} gu_iters {r.gu_iterations} "
              f"scans {r.arc_scans} gu_scans {r.gu_arc_scans} sat {r.sat_arc_scans} pushes {r.pushes} "
              f"relabels {r.relabels} visits {r.vertex_visits} launches {r.launches} upload_s {t_up:.3f} k1_s {t_k1:.4f}",
              flush=True)
        gbs = r.algorithmic_bytes / (r.solve_ms * 1e-3) / 1e9
        print(f"   arcs/s {net.m / (r.solve_ms * 1e-3):.3e}  algorithmic GB/s {gbs:.1f}  ({gbs / 8000:.4f} of 8 TB/s)", flush=True)
        if check:
            ocl = synth.make_cluster(cfg, seed, nm, np_)
            g = synth.build_graph(ocl)
            cost, ocap = s.costs()
            assert (cost == g.cost).all() and (ocap == g.cap).all(), "K1 mismatch"
            rc, tc, cap_o, price_o, st = capi.cost_scaling(g)
            resid = s.residual()
            crc, ctc, _ = capi.check(g, resid, s.prices())
            print(f"   oracle cost {tc} (cpu {st.solve_ms:.1f} ms)  certificate rc {crc} cost {ctc}  eps-viol {capi.check_eps(g, resid, s.prices())}", flush=True)
            assert tc == r.total_cost and crc == 0 and ctc == tc
            pl = s.placements()
            print("   placed", int((pl >= 0).sum()), "of", pl.shape[0])
    return r


if __name__ == "__main__":
    args = sys.argv[1:]
    cfg = int(args[0]) if args else 0
    seed = int(args[1]) if len(args) > 1 else 1
    eager = len(args) > 2 and args[2] == "eager"
    run(cfg, seed, eager=eager, check="nocheck" not in args, repeat=int(os.environ.get("REPEAT", "1")), rounds=int(os.environ.get("ROUNDS", "0")),
        alpha=int(os.environ.get("ALPHA", "0")))
