"""Development aid: lockstep group solve against one-at-a-time solves of the same networks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import poseidon_amd as pa
from poseidon_amd import batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cls, nets, sol = [], [], []
for b in range(n):
    cl = pa.synth_cluster(cfg, b + 1); net = pa.build_network(cl); s = pa.Solver(device=0); s.upload(net)
    cls.append(cl); nets.append(net); sol.append(s)
for rep in range(2):
    for s, cl, net in zip(sol, cls, nets): s.eval_costs(cl, net)
    t = time.perf_counter(); one = [s.solve() for s in sol]; t_one = time.perf_counter() - t
    for s, cl, net in zip(sol, cls, nets): s.eval_costs(cl, net)
    t = time.perf_counter(); grp = batch.solve_batch(sol, workers=1); t_grp = time.perf_counter() - t
    assert [r.total_cost for r in one] == [r.total_cost for r in grp]
    print(f"one at a time: {t_one * 1e3:.1f} ms total, solve_ms {[round(r.solve_ms, 1) for r in one]}, launches {[r.launches for r in one]}")
    print(f"group of {n}:    {t_grp * 1e3:.1f} ms total, solve_ms {grp[0].solve_ms:.1f}, launches {grp[0].launches}, rounds {[r.rounds for r in grp]}, gu_iters {[r.gu_iterations for r in grp]}")
