"""Helpers shared by the service tests: drive a SchedulerService from a synthetic cluster
the way the reference's watchers would (TaskSubmitted per pod, NodeAdded + AddNodeStats per
machine), and solve the service's exported network with the CPU oracle."""
import dataclasses

import numpy as np

from oracle import capi, synth


def machine_uuid(i):
    return f"00000000-0000-4000-8000-{i:012d}"


def job_uuid(j):
    return f"job-{j:06d}"


def feed_cluster(svc, cl, uid_base=1):
    from poseidon_amd.firmament import NodeReplyType, TaskReplyType
    for m in range(cl.n_machines):
        assert svc.NodeAdded(machine_uuid(m), int(cl.cap_cpu[m]), int(cl.cap_mem[m])) == NodeReplyType.NODE_ADDED_OK
        assert svc.AddNodeStats(machine_uuid(m), int(cl.used_cpu[m]), int(cl.used_mem[m])) == NodeReplyType.NODE_UPDATED_OK
    uids = np.arange(uid_base, uid_base + cl.n_pods, dtype=np.int64)
    for t in range(cl.n_pods):
        e = int(cl.pod_shape[t])
        assert svc.TaskSubmitted(int(uids[t]), job_uuid(int(cl.pod_job[t])), int(cl.req_cpu[e]), int(cl.req_mem[e])) \
            == TaskReplyType.TASK_SUBMITTED_OK
    return uids


def apply_churn(svc, cl_new, ch, uids, next_uid):
    """Replay one synth.churn round as RPCs; returns the updated uid array."""
    from poseidon_amd.firmament import NodeReplyType, TaskReplyType
    uids = uids.copy()
    for i, t in enumerate(ch.pods_replaced):
        assert svc.TaskRemoved(int(uids[t])) == TaskReplyType.TASK_REMOVED_OK
        uids[t] = next_uid + i
        e = int(ch.new_shape[i])
        assert svc.TaskSubmitted(int(uids[t]), job_uuid(int(ch.new_job[i])), int(cl_new.req_cpu[e]), int(cl_new.req_mem[e])) \
            == TaskReplyType.TASK_SUBMITTED_OK
    gone = getattr(svc, "_machines_gone", set())  # a later round may draw a machine that already left
    for i, m in enumerate(ch.machines_restat):
        rc = svc.AddNodeStats(machine_uuid(int(m)), int(ch.new_used_cpu[i]), int(ch.new_used_mem[i]))
        assert rc == (NodeReplyType.NODE_NOT_FOUND if int(m) in gone else NodeReplyType.NODE_UPDATED_OK)
    for m in ch.machines_out:
        rc = svc.NodeRemoved(machine_uuid(int(m)))
        assert rc == (NodeReplyType.NODE_NOT_FOUND if int(m) in gone else NodeReplyType.NODE_REMOVED_OK)
        gone.add(int(m))
    svc._machines_gone = gone
    return uids


def exported_graph(svc):
    """The service's network as an oracle Graph, slot arcs priced by the oracle's cost model."""
    ex = svc.export_network()
    lay = ex["layout"]
    N, E, K = lay.n_machines, lay.n_classes, lay.k_slots
    cl = synth.Cluster(N, lay.n_tasks, lay.n_jobs, ex["cap_cpu"], ex["cap_mem"], ex["used_cpu"], ex["used_mem"],
                       ex["req_cpu"], ex["req_mem"], np.zeros(0, np.int32), np.zeros(0, np.int32), k_slots=K)
    cost3, cap3 = capi.eval_costs(cl)
    cap3 = cap3 * ex["allowed"][:, :, None].astype(np.int32)  # label selectors
    idx = (ex["ec_arc_base"][:, None, None] + (np.arange(N)[None, :, None] * K) + np.arange(K)[None, None, :]).ravel()
    cost, cap, rev = ex["cost"].copy(), ex["cap"].copy(), ex["rev"]
    cost[idx] = cost3.ravel()
    cap[idx] = cap3.ravel()
    cost[rev[idx]] = -cost3.ravel()
    g = synth.Graph(ex["n"], ex["m2"] // 2, ex["row_off"], ex["head"], cap, cost, rev, ex["supply"], np.zeros(0, np.int64))
    return g, ex


def oracle_cost(cl):
    rc, tc, _, _, _ = capi.cost_scaling(synth.build_graph(cl))
    assert rc == 0
    return tc
