"""Scheduler service on the GPU: Schedule rounds (cold, then warm after churn) against the
CPU oracle's optimum of the same cluster built from scratch; deltas follow
/root/reference/pkg/firmament/scheduling_delta.proto:26-35 and are what
/root/reference/cmd/poseidon/poseidon.go:36-68 consumes."""
import collections

import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import _lib
from poseidon_amd.firmament import ChangeType as CT, NodeReplyType as NR, TaskReplyType as TR
from oracle import capi, synth

from svc_util import apply_churn, exported_graph, feed_cluster, job_uuid, machine_uuid, oracle_cost

pytestmark = pytest.mark.gpu


def placement_cost(cl, uids, where):
    """Cost of a placement under the slot-cost model: tasks of class e on machine m take the
    machine's cheapest fitting slots (costs grow with the slot index); others pay the
    unscheduled cost.  Also checks machine slot limits and that every task fits."""
    cost3, cap3 = capi.eval_costs(cl)
    cnt = collections.Counter()
    per_machine = collections.Counter()
    total = 0
    for t in range(cl.n_pods):
        m = where.get(int(uids[t]), -1)
        if m < 0:
            total += synth.C_UNSCHED
            continue
        cnt[(int(cl.pod_shape[t]), m)] += 1
        per_machine[m] += 1
    for (e, m), c in cnt.items():
        assert c <= int(cap3[e, m].sum()), "more tasks of a class on a machine than fit"
        total += int(cost3[e, m, :c].sum())
    for m, c in per_machine.items():
        assert c <= cl.slots
        assert cl.machine_out is None or not cl.machine_out[m]
    return total


def machine_index(uuid):
    return int(uuid[-12:]) if uuid else -1


@pytest.mark.parametrize("cfg,seed", [(0, 1), (0, 2), (1, 1)])
def test_schedule_rounds_cold_then_warm(cfg, seed):
    cl = synth.make_cluster(cfg, seed)
    with pa.SchedulerService(device=0, reschedule_running=True) as s:
        assert s.Check()
        uids = feed_cluster(s, cl)
        deltas = s.Schedule()
        st = s.stats()
        assert st.total_cost == oracle_cost(cl) and st.last_round_warm == 0 and st.rebuilds == 1
        assert all(d.type == CT.PLACE for d in deltas) and len(deltas) == st.tasks_placed
        where = {d.task_id: machine_index(d.resource_id) for d in deltas}
        assert placement_cost(cl, uids, where) == st.total_cost
        # nothing changed: the next round reports nothing
        assert s.Schedule() == [] and s.stats().total_cost == st.total_cost
        # nothing was queued since the last round: the resident optimum stands, no solve is run
        idle = s.stats()
        assert idle.solve_ms == 0 and idle.last_round_warm == 1 and idle.rounds == st.rounds + 1
        nxt = int(uids.max()) + 1
        for rnd in (1, 2):
            old_uids = set(int(u) for u in uids)
            cl, ch = synth.churn(cl, cfg, seed, rnd)
            uids = apply_churn(s, cl, ch, uids, nxt)
            nxt += ch.pods_replaced.shape[0]
            deltas = s.Schedule()
            st = s.stats()
            # in place: the solver either resumed from the kept state or declined it as no head start - never a rebuild
            assert st.last_round_warm + st.solve_warm_declined == 1 and st.rebuilds == 1, "churn inside the head-room must be applied in place"
            assert st.total_cost == oracle_cost(cl)
            live = set(int(u) for u in uids)
            for d in deltas:
                assert d.task_id in live
                if d.task_id not in old_uids:
                    assert d.type == CT.PLACE
                if d.type == CT.PREEMPT:
                    assert d.resource_id == "" and where.get(d.task_id, -1) >= 0
                    where[d.task_id] = -1
                else:
                    if d.type == CT.MIGRATE:
                        assert where.get(d.task_id, -1) >= 0 and where[d.task_id] != machine_index(d.resource_id)
                    where[d.task_id] = machine_index(d.resource_id)
            for u in old_uids - live:
                where.pop(u, None)
            # tasks that sat on a machine that went away must have been moved or preempted
            for u, m in where.items():
                assert m < 0 or not cl.machine_out[m]
            assert placement_cost(cl, uids, where) == st.total_cost
            for u in list(live)[:50]:
                assert machine_index(s.task_placement(u)) == where.get(u, -1)


def test_churn_round_at_full_scale():
    """The incremental configuration at the 10k-machine / 100k-pod size: one cold round, one churn
    round applied in place and warm-started; both optima equal the oracle's from-scratch optimum."""
    cl = synth.make_cluster(2, 1)
    with pa.SchedulerService(device=0, reschedule_running=True) as s:
        uids = feed_cluster(s, cl)
        deltas = s.Schedule()
        st = s.stats()
        assert st.total_cost == 56359449 and len(deltas) == st.tasks_placed
        cl2, ch = synth.churn(cl, 2, 1, 1)
        uids = apply_churn(s, cl2, ch, uids, int(uids.max()) + 1)
        deltas = s.Schedule()
        st = s.stats()
        assert st.last_round_warm + st.solve_warm_declined == 1 and st.rebuilds == 1
        assert st.total_cost == oracle_cost(cl2)
        assert st.patched_arcs > 0 and st.patched_arcs < 40_000  # only what changed crossed PCIe
        new = set(int(u) for u in uids[ch.pods_replaced])
        assert all(d.type == CT.PLACE for d in deltas if d.task_id in new)


def test_warm_round_moves_few_tasks():
    """A warm round after 5 % churn must not reshuffle the cluster: deltas stay a small
    multiple of what changed."""
    cl = synth.make_cluster(1, 2)
    with pa.SchedulerService(device=0, reschedule_running=True) as s:
        uids = feed_cluster(s, cl)
        s.Schedule()
        cl2, ch = synth.churn(cl, 1, 2, 1)
        uids = apply_churn(s, cl2, ch, uids, int(uids.max()) + 1)
        deltas = s.Schedule()
        assert s.stats().total_cost == oracle_cost(cl2)
        moved = sum(d.type in (CT.MIGRATE, CT.PREEMPT) for d in deltas)
        assert moved <= cl.n_pods // 4, moved


def test_node_failure_and_task_lifecycle():
    cl = synth.make_cluster(0, 3)
    with pa.SchedulerService(device=0, emit_noop=True, reschedule_running=True) as s:
        uids = feed_cluster(s, cl)
        deltas = s.Schedule()
        assert len(deltas) == cl.n_pods  # with emit_noop every runnable task is reported
        where = {d.task_id: machine_index(d.resource_id) for d in deltas if d.type == CT.PLACE}
        victim = collections.Counter(where.values()).most_common(1)[0][0]
        on_victim = [u for u, m in where.items() if m == victim]
        assert s.NodeFailed(machine_uuid(victim)) == NR.NODE_FAILED_OK
        done = [int(u) for u in uids[:10] if where.get(int(u), -1) not in (-1, victim)]
        for u in done:
            assert s.TaskCompleted(u) == TR.TASK_COMPLETED_OK
        deltas = {d.task_id: d for d in s.Schedule()}
        assert not any(u in deltas for u in done), "completed tasks are no longer scheduled"
        for u in on_victim:  # their machine is gone: the previous placement does not count
            assert deltas[u].type in (CT.PLACE, CT.NOOP)
            if deltas[u].type == CT.PLACE:
                assert machine_index(deltas[u].resource_id) != victim
        # the node comes back; removing the completed tasks is still acknowledged
        assert s.NodeAdded(machine_uuid(victim), int(cl.cap_cpu[victim]), int(cl.cap_mem[victim])) == NR.NODE_ADDED_OK
        for u in done:
            assert s.TaskRemoved(u) == TR.TASK_REMOVED_OK
        s.Schedule()
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0 and tc == s.stats().total_cost


def test_patch_arcs_rejects_broken_pairs():
    cl = pa.synth_cluster(0, 1)
    net = pa.build_network(cl)
    with pa.Solver() as sv:
        sv.upload(net)
        lib = sv._lib
        a = int(net.row_off[-2])  # first arc of the last task
        idx = np.array([a], np.int64)
        one = lambda v: np.array([v], np.int32)
        # pairing it with an arc that does not point back must be caught on the device
        h, c, k, r = one(int(net.head[a])), one(1), one(0), one(int(net.rev[a]) + 1)
        rc = lib.psn_patch_arcs(sv._ctx, 1, idx.ctypes.data, h.ctypes.data, c.ctypes.data, k.ctypes.data, r.ctypes.data)
        assert rc == _lib.PSN_EINVAL and b"pair up" in lib.psn_last_error(sv._ctx)
        with pytest.raises(_lib.PsnError) as e:
            sv.solve()
        assert e.value.code == _lib.PSN_ESTATE  # the network has to be uploaded again
        sv.upload(net)
        idx = np.array([net.m2], np.int64)
        rc = lib.psn_patch_arcs(sv._ctx, 1, idx.ctypes.data, h.ctypes.data, c.ctypes.data, k.ctypes.data, r.ctypes.data)
        assert rc == _lib.PSN_EINVAL and b"out of range" in lib.psn_last_error(sv._ctx)
        # a consistent patch: park both arcs of the task and its reverse places, then solve
        ta = a
        pairs = [ta, ta + 1, int(net.rev[ta]), int(net.rev[ta + 1])]
        tail = np.repeat(np.arange(net.n), np.diff(net.row_off))
        idx = np.array(pairs, np.int64)
        h = np.array([tail[x] for x in pairs], np.int32)
        z = np.zeros(4, np.int32)
        r = np.array(pairs, np.int32)
        assert lib.psn_patch_arcs(sv._ctx, 4, idx.ctypes.data, h.ctypes.data, z.ctypes.data, z.ctypes.data, r.ctypes.data) == 0
        sv.update_supplies(np.array([net.n - 1, 0], np.int32), np.array([0, -(cl.layout.n_tasks - 1)], np.int64))
        sv.eval_costs(cl, net)
        res = sv.solve()
        # same as the cluster without its last pod
        ocl = synth.make_cluster(0, 1)
        ocl = synth.Cluster(ocl.n_machines, ocl.n_pods - 1, ocl.n_jobs, ocl.cap_cpu, ocl.cap_mem, ocl.used_cpu, ocl.used_mem,
                            ocl.req_cpu, ocl.req_mem, ocl.pod_shape[:-1], ocl.pod_job[:-1])
        assert res.total_cost == oracle_cost(ocl)


def test_admission_mask_parity_with_oracle():
    """K1 with a random class x machine mask: arrays bit-equal to the oracle's, optimum equal."""
    rng = np.random.default_rng(5)
    cl = pa.synth_cluster(0, 2)
    net = pa.build_network(cl)
    ocl = synth.make_cluster(0, 2)
    ocl.allowed = rng.random((64, ocl.n_machines)) < 0.3
    g = synth.build_graph(ocl)
    with pa.Solver() as sv:
        sv.upload(net)
        sv.set_admission_mask(cl.layout, ocl.allowed)
        sv.eval_costs(cl, net)
        cost, ocap = sv.costs()
        assert (cost == g.cost).all() and (ocap == g.cap).all()
        r = sv.solve()
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0 and r.total_cost == tc
        sv.set_admission_mask(cl.layout, None)  # lifted again
        sv.eval_costs(cl, net)
        assert sv.solve().total_cost == 558826


def test_node_selector_end_to_end():
    """A pod whose selector matches no node stays unscheduled; one whose selector matches a
    single node lands there (/root/reference/test/e2e/poseidon_integration.go:417-472)."""
    from poseidon_amd.service import IN_SET
    with pa.SchedulerService(device=0) as s:
        for m in range(6):
            assert s.NodeAdded(machine_uuid(m), 16000, 64 << 20) == NR.NODE_ADDED_OK
        assert s.NodeLabels(machine_uuid(4), {"kubernetes.io/hostname": "node-4"}) == NR.NODE_UPDATED_OK
        for uid in range(1, 21):
            assert s.TaskSubmitted(uid, job_uuid(0), 500, 1 << 20) == TR.TASK_SUBMITTED_OK
        assert s.TaskSubmitted(100, job_uuid(1), 500, 1 << 20, selectors=[(IN_SET, "kubernetes.io/hostname", ["node-4"])]) \
            == TR.TASK_SUBMITTED_OK
        assert s.TaskSubmitted(101, job_uuid(1), 500, 1 << 20, selectors=[(IN_SET, "kubernetes.io/hostname", ["node-9"])]) \
            == TR.TASK_SUBMITTED_OK
        g, _ = exported_graph(s)  # what this round solves (placed tasks leave the network afterwards)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        deltas = {d.task_id: d for d in s.Schedule()}
        assert rc == 0 and tc == s.stats().total_cost
        assert deltas[100].type == CT.PLACE and machine_index(deltas[100].resource_id) == 4
        assert 101 not in deltas and s.task_placement(101) == ""
        assert len(deltas) == 21
        # the label appears on a node later: the pending pod is placed by a warm round
        assert s.NodeLabels(machine_uuid(2), {"kubernetes.io/hostname": "node-9"}) == NR.NODE_UPDATED_OK
        deltas = {d.task_id: d for d in s.Schedule()}
        assert list(deltas) == [101], "the tasks that are running stay where they are and are not reported again"
        assert deltas[101].type == CT.PLACE and machine_index(deltas[101].resource_id) == 2
        # an over-committed pod (request larger than any node) stays unscheduled (:310-408)
        assert s.TaskSubmitted(102, job_uuid(2), 64000, 1 << 20) == TR.TASK_SUBMITTED_OK
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert s.Schedule() == []
        assert rc == 0 and tc == s.stats().total_cost == synth.C_UNSCHED


@pytest.mark.parametrize("seed", [21, 22])
def test_random_rpc_sequences_scheduled_on_the_gpu(seed):
    """Random RPC bursts with a Schedule after each: the round's cost (warm or after a new layout)
    equals the oracle's optimum of the cluster described from scratch, and no task sits on a machine
    that left or that its selectors exclude."""
    from svc_util import RpcModel, model_cost, random_rpcs, rpc_shapes
    rng = np.random.default_rng(seed)
    shapes, st = rpc_shapes(), RpcModel()
    where = {}
    with pa.SchedulerService(device=0, max_machines=8, max_tasks=64, max_jobs=4, reschedule_running=True) as s:
        warm = 0
        for burst in range(14):
            random_rpcs(s, rng, st, shapes, int(rng.integers(10, 80)))
            deltas = s.Schedule()
            stt = s.stats()
            warm += stt.last_round_warm + stt.solve_warm_declined  # rounds applied in place
            assert stt.total_cost == model_cost(st.machines, st.tasks, shapes), (seed, burst)
            for d in deltas:
                assert d.task_id in st.tasks
                where[d.task_id] = d.resource_id if d.type != CT.PREEMPT else ""
            for u, (e, _) in st.tasks.items():
                res = s.task_placement(u)
                assert res == "" or res in st.machines
        assert warm >= 5 and s.stats().rebuilds >= 2


# ---- keep-running mode (the default): placed tasks hold their machine, a round only answers PLACE
def _rounds_until_quiet(s, max_rounds=4):
    """Schedule until a round reports nothing; every round's optimum is checked against the oracle's
    on the network that round solved.  Returns all deltas."""
    out = []
    for _ in range(max_rounds):
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        deltas = s.Schedule()
        st = s.stats()
        if st.solve_ms > 0:
            assert rc == 0 and tc == st.total_cost
        if not deltas:
            break
        out += deltas
    return out


def test_running_tasks_keep_their_machines_through_churn():
    """Default mode: a task that has been placed leaves the network and holds its machine.  After
    5 % churn the rounds report only tasks that arrived, were pending, or whose machine went away, all
    as PLACE; nobody else moves, and every round's optimum equals the oracle's on the network the
    round solved (running tasks priced in as usage of their machines)."""
    cfg, seed = 0, 4
    cl = synth.make_cluster(cfg, seed)
    with pa.SchedulerService(device=0) as s:
        uids = feed_cluster(s, cl)
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        deltas = s.Schedule()
        st = s.stats()
        assert rc == 0 and tc == st.total_cost == oracle_cost(cl)  # nothing runs yet: the synthetic optimum
        assert all(d.type == CT.PLACE for d in deltas)
        assert st.tasks_running == len(deltas) == st.tasks_placed
        deltas += _rounds_until_quiet(s)
        nxt = int(uids.max()) + 1
        restat = set()
        for rnd in (1, 2):
            before = {int(u): s.task_placement(int(u)) for u in uids}
            cl, ch = synth.churn(cl, cfg, seed, rnd)
            new_uids = apply_churn(s, cl, ch, uids, nxt)
            nxt += ch.pods_replaced.shape[0]
            restat |= set(int(m) for m in ch.machines_restat)
            live = set(int(u) for u in new_uids)
            arrived = live - set(int(u) for u in uids)
            gone_machines = set(machine_uuid(int(m)) for m in np.nonzero(cl.machine_out)[0])
            displaced = {u for u, res in before.items() if res in gone_machines and u in live}
            pending_before = {u for u, res in before.items() if res == "" and u in live}
            uids = new_uids
            deltas = _rounds_until_quiet(s)
            assert all(d.type == CT.PLACE for d in deltas)
            moved = set(d.task_id for d in deltas)
            assert len(moved) == len(deltas)
            assert moved <= arrived | displaced | pending_before
            assert len(moved & (arrived | displaced)) >= (len(arrived) + len(displaced)) // 2
            for d in deltas:
                assert d.resource_id not in gone_machines
            for u in live - moved - arrived - displaced:  # everybody else is exactly where they were
                assert s.task_placement(u) == before[u]
            # no machine holds more than its slots, nor (where the statistics did not change under it) its capacity
            load, cpu = collections.Counter(), collections.Counter()
            shape_of = {int(u): int(cl.pod_shape[t]) for t, u in enumerate(uids)}
            for u in live:
                res = s.task_placement(u)
                if res:
                    assert res not in gone_machines
                    m = machine_index(res)
                    load[m] += 1
                    cpu[m] += int(cl.req_cpu[shape_of[u]])
            assert max(load.values()) <= cl.slots
            for m, c in cpu.items():
                if m not in restat:
                    assert c + int(cl.used_cpu[m]) <= int(cl.cap_cpu[m]), m


def test_filled_nodes_leave_the_extra_pod_pending():
    """/root/reference/test/e2e/poseidon_integration.go:310-408: one filler pod per node asking for
    70 % of that node's CPU, then a pod asking for 50 % of the largest node: it must stay pending."""
    caps = [8000, 16000, 4000]
    with pa.SchedulerService(device=0) as s:
        for m, c in enumerate(caps):
            assert s.NodeAdded(machine_uuid(m), c, 64 << 20) == NR.NODE_ADDED_OK
        from poseidon_amd.service import IN_SET
        for m in range(len(caps)):
            assert s.NodeLabels(machine_uuid(m), {"kubernetes.io/hostname": f"node-{m}"}) == NR.NODE_UPDATED_OK
        for i, c in enumerate(caps):  # "one pod with affinity to each node that uses 70% of the node CPU" (:301)
            assert s.TaskSubmitted(1 + i, job_uuid(i), c * 7 // 10, 1 << 20,
                                   selectors=[(IN_SET, "kubernetes.io/hostname", [f"node-{i}"])]) == TR.TASK_SUBMITTED_OK
        placed = {}
        for _ in range(3):
            for d in s.Schedule():
                assert d.type == CT.PLACE
                placed[d.task_id] = machine_index(d.resource_id)
        assert placed == {1: 0, 2: 1, 3: 2}
        used = collections.Counter()
        for u, m in placed.items():
            used[m] += caps[u - 1] * 7 // 10
        assert all(used[m] <= caps[m] for m in used)
        assert s.TaskSubmitted(10, job_uuid(9), max(caps) // 2, 1 << 20) == TR.TASK_SUBMITTED_OK
        for _ in range(3):
            assert s.Schedule() == []
        assert s.task_placement(10) == ""
        # a filler goes away: the extra pod takes the node that has room for it now
        big = [u for u, m in placed.items() if m == 1]
        for u in big:
            assert s.TaskRemoved(u) == TR.TASK_REMOVED_OK
        deltas = s.Schedule()
        assert [(d.task_id, d.type, machine_index(d.resource_id)) for d in deltas] == [(10, CT.PLACE, 1)]


def test_two_classes_cannot_overcommit_a_machine_in_one_round():
    with pa.SchedulerService(device=0) as s:
        assert s.NodeAdded(machine_uuid(0), 10000, 64 << 20) == NR.NODE_ADDED_OK
        assert s.TaskSubmitted(1, job_uuid(0), 6000, 1 << 20) == TR.TASK_SUBMITTED_OK
        assert s.TaskSubmitted(2, job_uuid(0), 6000, 2 << 20) == TR.TASK_SUBMITTED_OK  # another class: same CPU, other memory
        d1 = s.Schedule()
        assert len(d1) == 1 and d1[0].type == CT.PLACE and s.stats().tasks_deferred == 1
        assert s.Schedule() == [] and s.stats().tasks_running == 1
        other = 2 if d1[0].task_id == 1 else 1
        assert s.task_placement(other) == ""
        assert s.TaskCompleted(d1[0].task_id) == TR.TASK_COMPLETED_OK
        d2 = s.Schedule()
        assert [(d.task_id, d.type) for d in d2] == [(other, CT.PLACE)]


def test_measured_machine_usage_excludes_the_running_tasks_own_share():
    """A measured whole-machine figure (ResourceStats) includes what the service's own running tasks use;
    their share (observed usage where TaskStats reported it, else the request) is taken out before the
    requests of the running tasks are added back as what is reserved."""
    with pa.SchedulerService(device=0) as s:
        assert s.NodeAdded(machine_uuid(0), 10000, 1000000) == NR.NODE_ADDED_OK
        assert s.TaskSubmitted(1, job_uuid(0), 100, 1000) == TR.TASK_SUBMITTED_OK
        assert s.TaskSubmitted(2, job_uuid(0), 400, 4000) == TR.TASK_SUBMITTED_OK
        assert len(s.Schedule()) == 2 and s.stats().tasks_running == 2
        assert s.AddTaskStats(1, cpu_usage=30, mem_usage=500) == TR.TASK_UPDATED_OK   # task 2 reports nothing
        assert s.AddNodeStatsRaw(machine_uuid(0), [(10000, 2000.0)], 1000000, 100000.0) == NR.NODE_UPDATED_OK
        assert s.TaskSubmitted(3, job_uuid(0), 100, 1000) == TR.TASK_SUBMITTED_OK     # something pending: a network exists
        ex = s.export_network()
        # external = 2000 - (30 + 400) ; reserved = 100 + 400
        assert int(ex["used_cpu"][0]) == (2000 - 430) + 500
        assert int(ex["used_mem"][0]) == (100000 - (500 + 4000)) + 5000
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        d = s.Schedule()
        assert rc == 0 and tc == s.stats().total_cost and [x.task_id for x in d] == [3]
