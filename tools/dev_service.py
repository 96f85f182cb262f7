"""Development driver: scheduler service rounds on a synthetic config (cold, idle, churned warm)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import poseidon_amd as pa
from oracle import synth
from svc_util import apply_churn, feed_cluster, oracle_cost


def show(tag, st, n_deltas, wall):
    print(f"{tag}: cost {st.total_cost} warm {st.last_round_warm} deltas {n_deltas} wall_ms {wall:.1f} | build {st.build_ms:.1f} "
          f"upload {st.upload_ms:.1f} update {st.update_ms:.1f} solve {st.solve_ms:.1f} extract {st.extract_ms:.1f} | "
          f"patched {st.patched_arcs} caps {st.cap_updates} supplies {st.supply_updates} n {st.n} m2 {st.m2}", flush=True)


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    check = "nocheck" not in sys.argv
    cl = synth.make_cluster(cfg, seed)
    with pa.SchedulerService(device=0, reschedule_running="resched" in sys.argv) as s:
        t = time.time(); uids = feed_cluster(s, cl); print(f"feed {time.time() - t:.2f} s")
        t = time.time(); d = s.Schedule(); show("cold ", s.stats(), len(d), (time.time() - t) * 1e3)
        if check:
            assert s.stats().total_cost == oracle_cost(cl)
        t = time.time(); d = s.Schedule(); show("idle ", s.stats(), len(d), (time.time() - t) * 1e3)
        nxt = int(uids.max()) + 1
        for r in range(1, rounds + 1):
            cl, ch = synth.churn(cl, cfg, seed, r)
            uids = apply_churn(s, cl, ch, uids, nxt)
            nxt += ch.pods_replaced.shape[0]
            t = time.time(); d = s.Schedule(); show(f"churn{r}", s.stats(), len(d), (time.time() - t) * 1e3)
            if check:
                assert s.stats().total_cost == oracle_cost(cl)


if __name__ == "__main__":
    main()
