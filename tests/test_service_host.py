"""Scheduler service core without a GPU: reply-code state machine
(/root/reference/pkg/firmament/firmament_client.go:43-179 lists the replies the client
accepts) and in-place maintenance of the flow network, checked by solving the exported
network with the CPU oracle and comparing with the from-scratch network of the same
cluster."""
import numpy as np
import pytest

import poseidon_amd as pa
from poseidon_amd import _lib
from poseidon_amd.firmament import NodeReplyType as NR, TaskReplyType as TR
from oracle import capi, synth

from svc_util import apply_churn, exported_graph, feed_cluster, job_uuid, machine_uuid, oracle_cost


def test_reply_codes_follow_the_rpc_contract():
    with pa.SchedulerService(device=-1) as s:
        assert not s.Check()  # no device bound: NOT_SERVING
        assert s.NodeAdded(machine_uuid(0), 8000, 16 << 20) == NR.NODE_ADDED_OK
        assert s.NodeAdded(machine_uuid(0), 8000, 16 << 20) == NR.NODE_ALREADY_EXISTS
        assert s.NodeUpdated(machine_uuid(1), 8000, 16 << 20) == NR.NODE_NOT_FOUND
        assert s.NodeUpdated(machine_uuid(0), 16000, 16 << 20) == NR.NODE_UPDATED_OK
        assert s.AddNodeStats(machine_uuid(0), 100, 100) == NR.NODE_UPDATED_OK
        assert s.AddNodeStats(machine_uuid(7), 100, 100) == NR.NODE_NOT_FOUND
        assert s.NodeFailed(machine_uuid(1)) == NR.NODE_NOT_FOUND
        assert s.NodeFailed(machine_uuid(0)) == NR.NODE_FAILED_OK
        assert s.NodeRemoved(machine_uuid(0)) == NR.NODE_NOT_FOUND
        # a failed node comes back under the same uuid (nodewatcher.go:254-266 forgets it on failure)
        assert s.NodeAdded(machine_uuid(0), 8000, 16 << 20) == NR.NODE_ADDED_OK
        assert s.NodeRemoved(machine_uuid(0)) == NR.NODE_REMOVED_OK

        assert s.TaskSubmitted(11, job_uuid(1), 100, 1024) == TR.TASK_SUBMITTED_OK
        assert s.TaskSubmitted(11, job_uuid(1), 100, 1024) == TR.TASK_ALREADY_SUBMITTED
        assert s.TaskSubmitted(12, job_uuid(1), 100, 1024, state=2) == TR.TASK_STATE_NOT_CREATED
        assert s.TaskUpdated(13, job_uuid(1), 100, 1024) == TR.TASK_NOT_FOUND
        assert s.TaskUpdated(11, job_uuid(9), 100, 1024) == TR.TASK_JOB_NOT_FOUND
        assert s.TaskUpdated(11, job_uuid(1), 200, 2048) == TR.TASK_UPDATED_OK
        assert s.AddTaskStats(11) == TR.TASK_UPDATED_OK
        assert s.TaskCompleted(99) == TR.TASK_NOT_FOUND
        assert s.TaskCompleted(11) == TR.TASK_COMPLETED_OK
        # a completed pod is still deleted later (podwatcher.go:279-297): the task must still be known
        assert s.TaskSubmitted(11, job_uuid(1), 100, 1024) == TR.TASK_ALREADY_SUBMITTED
        assert s.TaskRemoved(11) == TR.TASK_REMOVED_OK
        assert s.TaskRemoved(11) == TR.TASK_NOT_FOUND
        assert s.TaskSubmitted(12, job_uuid(1), 100, 1024) == TR.TASK_SUBMITTED_OK
        assert s.TaskFailed(12) == TR.TASK_FAILED_OK
        assert s.TaskFailed(77) == TR.TASK_NOT_FOUND
        with pytest.raises(_lib.PsnError) as e:
            s.Schedule()
        assert e.value.code == _lib.PSN_ESTATE


def test_bad_arguments_are_errors_not_replies():
    with pa.SchedulerService(device=-1) as s:
        for call in (lambda: s.NodeAdded("", 1000, 1000), lambda: s.NodeAdded("x" * 40, 1000, 1000),
                     lambda: s.NodeAdded(machine_uuid(0), 0, 1000), lambda: s.TaskSubmitted(0, job_uuid(0), 1, 1),
                     lambda: s.TaskSubmitted(5, "", 1, 1), lambda: s.TaskSubmitted(5, job_uuid(0), -1, 1)):
            with pytest.raises(_lib.PsnError) as e:
                call()
            assert e.value.code == _lib.PSN_EINVAL


def _check_structure(g):
    a = np.arange(g.rev.shape[0])
    assert (g.rev[g.rev] == a).all()
    tail = np.repeat(np.arange(g.n), np.diff(g.row_off))
    assert (g.head[g.rev] == tail).all()
    assert (g.cost[g.rev] == -g.cost).all()
    selfp = g.rev == a
    assert (g.cap[selfp] == 0).all() and (g.head[selfp] == tail[selfp]).all()
    assert g.supply.sum() == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_network_kept_in_place_matches_from_scratch(seed):
    """Feed C0, then two rounds of churn; after each step the oracle's optimum on the service's
    network equals the optimum of the network built from scratch for the same cluster."""
    cl = synth.make_cluster(0, seed)
    with pa.SchedulerService(device=-1) as s:
        uids = feed_cluster(s, cl)
        g, ex = exported_graph(s)
        _check_structure(g)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0 and tc == oracle_cost(cl)
        m2_first = ex["m2"]
        nxt = int(uids.max()) + 1
        for rnd in (1, 2):
            cl, ch = synth.churn(cl, 0, seed, rnd)
            uids = apply_churn(s, cl, ch, uids, nxt)
            nxt += ch.pods_replaced.shape[0]
            g, ex = exported_graph(s)
            _check_structure(g)
            assert ex["m2"] == m2_first, "churn within head-room must not lay the network out again"
            rc, tc, _, _, _ = capi.cost_scaling(g)
            assert rc == 0 and tc == oracle_cost(cl)
        assert s.stats().rebuilds == 1


def test_running_out_of_room_lays_out_again():
    cl = synth.make_cluster(0, 3, n_machines=20, n_pods=100)
    with pa.SchedulerService(device=-1) as s:
        uids = feed_cluster(s, cl)
        g, ex = exported_graph(s)
        n0 = ex["n"]
        # many more tasks of one shape than the class row has room for, and new machines
        for i in range(400):
            assert s.TaskSubmitted(10_000 + i, job_uuid(0), int(cl.req_cpu[5]), int(cl.req_mem[5])) == TR.TASK_SUBMITTED_OK
        for m in range(20, 60):
            assert s.NodeAdded(machine_uuid(m), 8000, 32 << 20) == NR.NODE_ADDED_OK
        g, ex = exported_graph(s)
        _check_structure(g)
        assert ex["n"] > n0 and s.stats().rebuilds == 2
        assert int(g.supply[0]) == -500
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0
        # the same cluster described from scratch
        cl2 = synth.Cluster(60, 500, cl.n_jobs,
                            np.concatenate([cl.cap_cpu, np.full(40, 8000, np.int64)]),
                            np.concatenate([cl.cap_mem, np.full(40, 32 << 20, np.int64)]),
                            np.concatenate([cl.used_cpu, np.zeros(40, np.int64)]),
                            np.concatenate([cl.used_mem, np.zeros(40, np.int64)]),
                            cl.req_cpu, cl.req_mem,
                            np.concatenate([cl.pod_shape, np.full(400, 5, np.int32)]),
                            np.concatenate([cl.pod_job, np.zeros(400, np.int32)]))
        assert tc == oracle_cost(cl2)


def test_selectors_shape_the_admission_mask():
    """nodeSelector semantics (/root/reference/test/e2e/poseidon_integration.go:417-472 expects a
    pod with a non-matching selector to stay pending and a matching one to land on that node):
    selectors become part of the class key and a per-(class, machine) mask."""
    from poseidon_amd.service import EXISTS_KEY, IN_SET, NOT_EXISTS_KEY, NOT_IN_SET
    with pa.SchedulerService(device=-1) as s:
        for m in range(4):
            assert s.NodeAdded(machine_uuid(m), 8000, 32 << 20) == NR.NODE_ADDED_OK
        assert s.NodeLabels(machine_uuid(0), {"zone": "a", "gpu": "yes"}) == NR.NODE_UPDATED_OK
        assert s.NodeLabels(machine_uuid(1), {"zone": "b"}) == NR.NODE_UPDATED_OK
        assert s.NodeLabels(machine_uuid(9), {"zone": "b"}) == NR.NODE_NOT_FOUND
        sub = lambda uid, sel: s.TaskSubmitted(uid, job_uuid(0), 100, 1 << 20, selectors=sel)
        assert sub(1, None) == TR.TASK_SUBMITTED_OK
        assert sub(2, [(IN_SET, "zone", ["a"])]) == TR.TASK_SUBMITTED_OK
        assert sub(3, [(IN_SET, "zone", ["b", "a"])]) == TR.TASK_SUBMITTED_OK
        assert sub(4, [(NOT_IN_SET, "zone", ["a"])]) == TR.TASK_SUBMITTED_OK
        assert sub(5, [(EXISTS_KEY, "gpu", [])]) == TR.TASK_SUBMITTED_OK
        assert sub(6, [(NOT_EXISTS_KEY, "zone", []), (NOT_EXISTS_KEY, "gpu", [])]) == TR.TASK_SUBMITTED_OK
        assert sub(7, [(IN_SET, "zone", ["a", "b"])]) == TR.TASK_SUBMITTED_OK  # same class as task 3
        assert sub(8, [(IN_SET, "zone", ["nowhere"])]) == TR.TASK_SUBMITTED_OK
        ex = s.export_network()
        al = ex["allowed"][:, :4]
        want = {0: [1, 1, 1, 1], 1: [1, 0, 0, 0], 2: [1, 1, 0, 0], 3: [0, 1, 1, 1], 4: [1, 0, 0, 0], 5: [0, 0, 1, 1],
                6: [0, 0, 0, 0]}
        for e, row in want.items():
            assert al[e].astype(int).tolist() == row, (e, al[e])
        # labels change in place: machine 2 joins zone a
        assert s.NodeLabels(machine_uuid(2), {"zone": "a"}) == NR.NODE_UPDATED_OK
        al = s.export_network()["allowed"][:, :4]
        assert al[1].astype(int).tolist() == [1, 0, 1, 0] and al[5].astype(int).tolist() == [0, 0, 0, 1]
        g, _ = exported_graph(s)
        rc, tc, _, _, _ = capi.cost_scaling(g)
        assert rc == 0 and tc >= synth.C_UNSCHED  # task 8 can go nowhere: it pays the unscheduled cost
        assert tc < 2 * synth.C_UNSCHED


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_rpc_sequences_keep_the_network_right(seed):
    """Random interleavings of every mutating RPC; after each burst the optimum of the service's
    network (solved by the oracle) equals the optimum of the same cluster described from scratch."""
    from svc_util import RpcModel, model_cost, random_rpcs, rpc_shapes
    rng = np.random.default_rng(seed)
    shapes, st = rpc_shapes(), RpcModel()
    with pa.SchedulerService(device=-1, max_machines=8, max_tasks=64, max_jobs=4) as s:
        for burst in range(12):
            random_rpcs(s, rng, st, shapes, int(rng.integers(20, 120)))
            g, _ = exported_graph(s)
            _check_structure(g)
            rc, tc, _, _, _ = capi.cost_scaling(g)
            assert rc == 0 and tc == model_cost(st.machines, st.tasks, shapes), (seed, burst)
        assert s.stats().rebuilds >= 2  # the small initial capacities were outgrown on the way


def test_resource_stats_fields_map_to_machine_usage():
    """AddNodeStats with the fields of the message: the reference's converter vectors
    (/root/reference/pkg/stats/stats_test.go:100-127: CPU capacity 10000 / utilisation 2000.0, memory
    capacity 1000000 / utilisation 100000.0) give 2000 millicores and 100000 Kb in use; the same
    figures as fractions of the capacity (what the message text describes) give the same."""
    with pa.SchedulerService(device=-1) as s:
        for m in range(3):
            assert s.NodeAdded(machine_uuid(m), 10000, 1000000) == NR.NODE_ADDED_OK
        assert s.AddNodeStatsRaw(machine_uuid(0), [(10000, 2000.0)], 1000000, 100000.0) == NR.NODE_UPDATED_OK
        assert s.AddNodeStatsRaw(machine_uuid(1), [(10000, 0.2)], 1000000, 0.1) == NR.NODE_UPDATED_OK
        # per-CPU entries add up; a figure above the capacity is cut to it
        assert s.AddNodeStatsRaw(machine_uuid(2), [(2500, 0.5), (2500, 2000.0), (2500, 9999.0), (2500, 0.0)], 1000000, 5e9) \
            == NR.NODE_UPDATED_OK
        assert s.AddNodeStatsRaw("no-such-node", [(1, 0.5)], 1, 0.5) == NR.NODE_NOT_FOUND
        assert s.TaskSubmitted(1, "job-0", 100, 1000) == TR.TASK_SUBMITTED_OK
        # the reference's TaskStats vector: cpu_usage 30, mem_usage 500 (stats_test.go:41-46)
        assert s.AddTaskStats(1, cpu_usage=30, mem_usage=500) == TR.TASK_UPDATED_OK
        assert s.AddTaskStats(2, cpu_usage=30, mem_usage=500) == TR.TASK_NOT_FOUND
        assert s.AddTaskStats(1) == TR.TASK_UPDATED_OK
        ex = s.export_network()
        assert ex["used_cpu"][:3].tolist() == [2000, 2000, 1250 + 2000 + 2500]
        assert ex["used_mem"][:3].tolist() == [100000, 100000, 1000000]
        # plain AddNodeStats replaces a measurement
        assert s.AddNodeStats(machine_uuid(0), 700, 7000) == NR.NODE_UPDATED_OK
        ex = s.export_network()
        assert (int(ex["used_cpu"][0]), int(ex["used_mem"][0])) == (700, 7000)


def test_resource_stats_reading_at_the_boundary_is_a_configured_choice():
    """A utilisation of exactly 1.0 is the whole capacity as a fraction and one millicore / one Kb as
    a quantity.  By default values up to 1.0 are fractions (what the message text says); a service
    configured for absolute figures (what the reference's own vectors carry) reads 1.0 and 0.5 as
    one unit and half a unit (rounded), and one configured for fractions reads 2000.0 as over the
    capacity and cuts it to the capacity."""
    want = {0: (10000, 1000000, 5000, 2000), 1: (10000, 1000000, 5000, 10000), 2: (1, 1, 1, 2000)}
    for mode, (cpu1, mem1, cpu_half, cpu_2000) in want.items():
        with pa.SchedulerService(device=-1, stats_utilization=mode) as s:
            for m in range(3):
                assert s.NodeAdded(machine_uuid(m), 10000, 1000000) == NR.NODE_ADDED_OK
            assert s.AddNodeStatsRaw(machine_uuid(0), [(10000, 1.0)], 1000000, 1.0) == NR.NODE_UPDATED_OK
            assert s.AddNodeStatsRaw(machine_uuid(1), [(10000, 0.5)], 1000000, 0.0) == NR.NODE_UPDATED_OK
            assert s.AddNodeStatsRaw(machine_uuid(2), [(10000, 2000.0)], 1000000, 0.0) == NR.NODE_UPDATED_OK
            assert s.TaskSubmitted(1, "job-0", 100, 1000) == TR.TASK_SUBMITTED_OK
            ex = s.export_network()
            assert (int(ex["used_cpu"][0]), int(ex["used_mem"][0])) == (cpu1, mem1), mode
            assert int(ex["used_cpu"][1]) == cpu_half and int(ex["used_cpu"][2]) == cpu_2000, mode
