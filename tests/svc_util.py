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


# ---- random RPC sequences (shared by the host and GPU service tests)
from poseidon_amd.firmament import NodeReplyType as NR, TaskReplyType as TR  # noqa: E402


def rpc_shapes():
    from poseidon_amd.service import EXISTS_KEY, IN_SET, NOT_IN_SET
    return [(int(c), int(m) << 10, sel) for c, m, sel in
            [(100, 64, ()), (500, 256, ()), (2000, 1024, ()), (250, 128, ((IN_SET, "zone", ("a",)),)),
             (250, 128, ((NOT_IN_SET, "zone", ("a",)),)), (1000, 512, ((EXISTS_KEY, "ssd", ()),)), (4000, 2048, ())]]


class RpcModel:
    """Plain-dict picture of what the service has been told."""

    def __init__(self):
        self.machines, self.tasks, self.done = {}, {}, set()
        self.next_uid, self.next_m = 1, 0


def random_rpcs(s, rng, st, shapes, count):
    for _ in range(count):
        op = rng.choice(["submit", "submit", "submit", "remove", "complete", "update", "node+", "node-", "stats", "labels"])
        machines, tasks, done = st.machines, st.tasks, st.done
        if op == "submit":
            e = int(rng.integers(0, len(shapes)))
            job = f"job-{int(rng.integers(0, 6))}"
            assert s.TaskSubmitted(st.next_uid, job, shapes[e][0], shapes[e][1],
                                   selectors=[(t, k, list(v)) for t, k, v in shapes[e][2]]) == TR.TASK_SUBMITTED_OK
            tasks[st.next_uid] = (e, job)
            st.next_uid += 1
        elif op == "remove" and (tasks or done):
            u = int(rng.choice(sorted(tasks) + sorted(done)))
            assert s.TaskRemoved(u) == TR.TASK_REMOVED_OK
            tasks.pop(u, None); done.discard(u)
        elif op == "complete" and tasks:
            u = int(rng.choice(sorted(tasks)))
            assert (s.TaskCompleted(u) if rng.random() < 0.5 else s.TaskFailed(u)) in (TR.TASK_COMPLETED_OK, TR.TASK_FAILED_OK)
            tasks.pop(u); done.add(u)
        elif op == "update" and tasks:
            u = int(rng.choice(sorted(tasks)))
            e_old, job = tasks[u]
            cands = [i for i, sh in enumerate(shapes) if sh[2] == shapes[e_old][2]]  # the selectors stay with the task
            e = int(rng.choice(cands))
            assert s.TaskUpdated(u, job, shapes[e][0], shapes[e][1]) == TR.TASK_UPDATED_OK
            tasks[u] = (e, job)
        elif op == "node+":
            u = machine_uuid(st.next_m); st.next_m += 1
            cpu, mem = int(rng.choice([4000, 16000, 64000])), int(rng.choice([8, 64, 256])) << 20
            assert s.NodeAdded(u, cpu, mem) == NR.NODE_ADDED_OK
            machines[u] = [cpu, mem, 0, 0, {}]
        elif op == "node-" and machines:
            u = str(rng.choice(sorted(machines)))
            assert (s.NodeRemoved(u) if rng.random() < 0.5 else s.NodeFailed(u)) in (NR.NODE_REMOVED_OK, NR.NODE_FAILED_OK)
            machines.pop(u)
        elif op == "stats" and machines:
            u = str(rng.choice(sorted(machines)))
            machines[u][2] = int(rng.integers(0, machines[u][0])); machines[u][3] = int(rng.integers(0, machines[u][1]))
            assert s.AddNodeStats(u, machines[u][2], machines[u][3]) == NR.NODE_UPDATED_OK
        elif op == "labels" and machines:
            u = str(rng.choice(sorted(machines)))
            labels = {}
            if rng.random() < 0.6:
                labels["zone"] = str(rng.choice(["a", "b"]))
            if rng.random() < 0.3:
                labels["ssd"] = "yes"
            assert s.NodeLabels(u, labels) == NR.NODE_UPDATED_OK
            machines[u][4] = labels


def model_cost(machines, tasks, shapes):
    """From-scratch optimum of a cluster given as plain dicts: machines uuid -> (cpu, mem, used_cpu,
    used_mem, labels), tasks uid -> (shape index, job name), shapes list of (cpu, mem, selectors)."""
    from poseidon_amd.service import EXISTS_KEY, IN_SET, NOT_EXISTS_KEY, NOT_IN_SET
    if not tasks:
        return 0
    mlist = sorted(machines)
    N = max(1, len(mlist))
    cc = np.array([machines[u][0] for u in mlist] or [1 << 40], np.int64)
    cm = np.array([machines[u][1] for u in mlist] or [1 << 40], np.int64)
    uc = np.array([min(machines[u][2], machines[u][0]) for u in mlist] or [1 << 40], np.int64)
    um = np.array([min(machines[u][3], machines[u][1]) for u in mlist] or [1 << 40], np.int64)
    rq = np.ones(64, np.int64); rm = np.ones(64, np.int64)
    allowed = np.ones((64, N), bool)

    def admits(sels, labels):
        for typ, key, vals in sels:
            has = key in labels
            inn = has and labels[key] in vals
            if (typ == IN_SET and not inn) or (typ == NOT_IN_SET and inn) or (typ == EXISTS_KEY and not has) or \
                    (typ == NOT_EXISTS_KEY and has):
                return False
        return True
    for e, (c, m, sels) in enumerate(shapes):
        rq[e], rm[e] = c, m
        for i, u in enumerate(mlist):
            allowed[e, i] = admits(sels, machines[u][4])
    if not mlist:
        allowed[:] = False
    jobs = sorted({j for _, j in tasks.values()})
    jidx = {j: i for i, j in enumerate(jobs)}
    uids = sorted(tasks)
    cl = synth.Cluster(N, len(uids), len(jobs), cc, cm, uc, um, rq, rm,
                       np.array([tasks[u][0] for u in uids], np.int32), np.array([jidx[tasks[u][1]] for u in uids], np.int32))
    cl.allowed = allowed
    return oracle_cost(cl)


