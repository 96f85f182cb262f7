"""Python face of the scheduler service core (include/psn_service.h).

Method names and reply codes are those of the Firmament scheduler service the reference
talks to (/root/reference/pkg/firmament/firmament_scheduler.proto:15-45, enums :110-135);
the wrappers return the enum members of ``poseidon_amd.firmament`` so that tests read like
the reference's client code (/root/reference/pkg/firmament/firmament_client.go:29-207).
"""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib
from .firmament import ChangeType, NodeReplyType, TaskReplyType

TASK_STATE_CREATED = 0  # /root/reference/pkg/firmament/task_desc.proto:32-33


class SvcConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_machines", C.c_int32), ("max_tasks", C.c_int32),
                ("max_jobs", C.c_int32), ("max_classes", C.c_int32), ("k_slots", C.c_int32),
                ("slots_per_machine", C.c_int32), ("unsched_cost", C.c_int32), ("emit_noop", C.c_int32),
                ("reschedule_running", C.c_int32), ("stats_utilization", C.c_int32), ("solver", _lib.Config)]


class SvcStats(C.Structure):
    _fields_ = [("rounds", C.c_int64), ("rebuilds", C.c_int64), ("warm_rounds", C.c_int64),
                ("patched_arcs", C.c_int64), ("cap_updates", C.c_int64), ("supply_updates", C.c_int64),
                ("tasks_runnable", C.c_int64), ("tasks_placed", C.c_int64), ("machines_live", C.c_int64),
                ("n", C.c_int64), ("m2", C.c_int64),
                ("build_ms", C.c_double), ("upload_ms", C.c_double), ("update_ms", C.c_double),
                ("solve_ms", C.c_double), ("extract_ms", C.c_double),
                ("total_cost", C.c_int64), ("last_round_warm", C.c_int32),
                ("tasks_running", C.c_int64), ("total_ms", C.c_double), ("tasks_deferred", C.c_int64),
                ("solve_replays", C.c_int64), ("solve_last_phase_replays", C.c_int64), ("solve_refines", C.c_int64),
                ("solve_rounds", C.c_int64), ("solve_global_updates", C.c_int64), ("solve_pushes", C.c_int64),
                ("solve_first_units", C.c_int64), ("solve_first_eps_log2", C.c_int64), ("solve_device_ms", C.c_double),
                ("solve_warm_declined", C.c_int64)]


class Selector(C.Structure):  # include/psn_service.h psn_selector
    _fields_ = [("type", C.c_int32), ("key", C.c_char_p), ("values", C.POINTER(C.c_char_p)), ("n_values", C.c_int32)]


IN_SET, NOT_IN_SET, EXISTS_KEY, NOT_EXISTS_KEY = 0, 1, 2, 3  # /root/reference/pkg/firmament/label_selector.proto:25-30


@dataclasses.dataclass
class SchedulingDelta:  # /root/reference/pkg/firmament/scheduling_delta.proto:25-41
    task_id: int
    resource_id: str
    type: ChangeType


_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp, i32, i64, u64, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_char_p
    lib.psn_svc_create.argtypes = [C.POINTER(SvcConfig), C.POINTER(vp)]
    lib.psn_svc_destroy.argtypes = [vp]
    lib.psn_svc_destroy.restype = None
    lib.psn_svc_last_error.argtypes = [vp]
    lib.psn_svc_last_error.restype = cp
    lib.psn_svc_task_submitted.argtypes = [vp, u64, cp, i32, i64, i64]
    lib.psn_svc_task_submitted_sel.argtypes = [vp, u64, cp, i32, i64, i64, C.POINTER(Selector), i32]
    lib.psn_svc_node_labels.argtypes = [vp, cp, C.POINTER(cp), C.POINTER(cp), i32]
    lib.psn_svc_export_mask.argtypes = [vp, vp]
    lib.psn_svc_task_updated.argtypes = [vp, u64, cp, i64, i64]
    for name in ("completed", "failed", "removed"):
        getattr(lib, "psn_svc_task_" + name).argtypes = [vp, u64]
    lib.psn_svc_add_task_stats.argtypes = [vp, u64]
    lib.psn_svc_node_added.argtypes = [vp, cp, i64, i64]
    lib.psn_svc_node_updated.argtypes = [vp, cp, i64, i64]
    lib.psn_svc_add_node_stats.argtypes = [vp, cp, i64, i64]
    lib.psn_svc_add_node_stats_raw.argtypes = [vp, cp, i32, vp, vp, i64, C.c_double]
    lib.psn_svc_add_task_stats_raw.argtypes = [vp, u64, i64, i64]
    lib.psn_svc_node_failed.argtypes = [vp, cp]
    lib.psn_svc_node_removed.argtypes = [vp, cp]
    lib.psn_svc_check.argtypes = [vp]
    lib.psn_svc_schedule.argtypes = [vp]
    lib.psn_svc_schedule.restype = i64
    lib.psn_svc_get_deltas.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.psn_svc_get_stats.argtypes = [vp, C.POINTER(SvcStats)]
    lib.psn_svc_network_size.argtypes = [vp, C.POINTER(_lib.Layout), C.POINTER(i32), C.POINTER(i64)]
    lib.psn_svc_export_network.argtypes = [vp] + [vp] * 13
    lib.psn_svc_task_placement.argtypes = [vp, u64, vp]
    lib.psn_svc_solver.argtypes = [vp]
    lib.psn_svc_solver.restype = vp
    _bound = True


class SchedulerService:
    """One Firmament-compatible scheduler state machine; ``device=-1`` keeps everything on the
    host (no Schedule), any other value binds the solver to that GPU."""

    def __init__(self, device: int = 0, max_machines: int = 0, max_tasks: int = 0, max_jobs: int = 0,
                 emit_noop: bool = False, slots_per_machine: int = 0, unsched_cost: int = 0,
                 reschedule_running: bool = False, stats_utilization: int = 0):
        self._lib = _lib.load_library()
        _bind(self._lib)
        cfg = SvcConfig(device=device, max_machines=max_machines, max_tasks=max_tasks, max_jobs=max_jobs,
                        emit_noop=int(emit_noop), slots_per_machine=slots_per_machine, unsched_cost=unsched_cost,
                        reschedule_running=int(reschedule_running), stats_utilization=int(stats_utilization))
        self._svc = C.c_void_p(None)
        rc = self._lib.psn_svc_create(C.byref(cfg), C.byref(self._svc))
        if rc != _lib.PSN_OK:
            self._svc = C.c_void_p(None)
            _lib.check(rc)

    def close(self):
        if self._svc:
            self._lib.psn_svc_destroy(self._svc)
            self._svc = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _reply(self, rc, enum_type):
        if rc < 0:
            raise _lib.PsnError(rc, self._lib.psn_svc_last_error(self._svc).decode())
        return enum_type(rc)

    # ---- task RPCs
    def TaskSubmitted(self, uid: int, job_uuid: str, cpu_millicores: int, ram_kb: int, state: int = TASK_STATE_CREATED,
                      selectors=None):
        """selectors: iterable of (type, key, [values]) label selectors, e.g. a pod's nodeSelector as
        [(IN_SET, key, [value]), ...] (/root/reference/pkg/k8sclient/podwatcher.go:455-465)."""
        if selectors:
            sels = list(selectors)
            arr = (Selector * len(sels))()
            keep = []
            for i, (typ, key, values) in enumerate(sels):
                vals = (C.c_char_p * max(1, len(values)))(*[v.encode() for v in values])
                keep.append(vals)
                arr[i] = Selector(int(typ), key.encode(), vals, len(values))
            return self._reply(self._lib.psn_svc_task_submitted_sel(self._svc, uid, job_uuid.encode(), state, cpu_millicores,
                                                                    ram_kb, arr, len(sels)), TaskReplyType)
        return self._reply(self._lib.psn_svc_task_submitted(self._svc, uid, job_uuid.encode(), state, cpu_millicores, ram_kb),
                           TaskReplyType)

    def TaskUpdated(self, uid: int, job_uuid: str, cpu_millicores: int, ram_kb: int):
        return self._reply(self._lib.psn_svc_task_updated(self._svc, uid, job_uuid.encode(), cpu_millicores, ram_kb), TaskReplyType)

    def TaskCompleted(self, uid: int):
        return self._reply(self._lib.psn_svc_task_completed(self._svc, uid), TaskReplyType)

    def TaskFailed(self, uid: int):
        return self._reply(self._lib.psn_svc_task_failed(self._svc, uid), TaskReplyType)

    def TaskRemoved(self, uid: int):
        return self._reply(self._lib.psn_svc_task_removed(self._svc, uid), TaskReplyType)

    def AddTaskStats(self, uid: int, cpu_usage: int | None = None, mem_usage: int | None = None):
        """TaskStats (/root/reference/pkg/firmament/task_stats.proto:22-50): cpu_usage in millicores,
        mem_usage in Kb; without them the call only acknowledges."""
        if cpu_usage is None and mem_usage is None:
            return self._reply(self._lib.psn_svc_add_task_stats(self._svc, uid), TaskReplyType)
        return self._reply(self._lib.psn_svc_add_task_stats_raw(self._svc, uid, int(cpu_usage or 0), int(mem_usage or 0)), TaskReplyType)

    def AddNodeStatsRaw(self, uuid: str, cpus_stats, mem_capacity: int, mem_utilization: float):
        """ResourceStats as the message carries it (/root/reference/pkg/firmament/resource_stats.proto:22-59):
        cpus_stats = [(cpu_capacity millicores, cpu_utilization), ...], memory capacity in Kb and utilisation."""
        n = len(cpus_stats)
        caps = (C.c_int64 * max(1, n))(*[int(c) for c, _ in cpus_stats])
        utils = (C.c_double * max(1, n))(*[float(u) for _, u in cpus_stats])
        return self._reply(self._lib.psn_svc_add_node_stats_raw(self._svc, uuid.encode(), n, caps, utils, int(mem_capacity),
                                                                float(mem_utilization)), NodeReplyType)

    # ---- node RPCs
    def NodeAdded(self, uuid: str, cpu_millicores: int, ram_kb: int):
        return self._reply(self._lib.psn_svc_node_added(self._svc, uuid.encode(), cpu_millicores, ram_kb), NodeReplyType)

    def NodeUpdated(self, uuid: str, cpu_millicores: int, ram_kb: int):
        return self._reply(self._lib.psn_svc_node_updated(self._svc, uuid.encode(), cpu_millicores, ram_kb), NodeReplyType)

    def AddNodeStats(self, uuid: str, used_cpu_millicores: int, used_ram_kb: int):
        return self._reply(self._lib.psn_svc_add_node_stats(self._svc, uuid.encode(), used_cpu_millicores, used_ram_kb), NodeReplyType)

    def NodeLabels(self, uuid: str, labels: dict):
        """Labels carried by NodeAdded / NodeUpdated (/root/reference/pkg/k8sclient/nodewatcher.go:309-315)."""
        n = len(labels)
        keys = (C.c_char_p * max(1, n))(*[k.encode() for k in labels.keys()])
        vals = (C.c_char_p * max(1, n))(*[v.encode() for v in labels.values()])
        return self._reply(self._lib.psn_svc_node_labels(self._svc, uuid.encode(), keys, vals, n), NodeReplyType)

    def NodeFailed(self, uuid: str):
        return self._reply(self._lib.psn_svc_node_failed(self._svc, uuid.encode()), NodeReplyType)

    def NodeRemoved(self, uuid: str):
        return self._reply(self._lib.psn_svc_node_removed(self._svc, uuid.encode()), NodeReplyType)

    def Check(self) -> bool:
        return self._lib.psn_svc_check(self._svc) == 1  # SERVING

    # ---- the scheduling round
    def schedule_arrays(self):
        """One scheduling round with the reply as arrays: (task_uid uint64[d], type int32[d],
        resource_uuid S36[d]) - what a caller that forwards the deltas (or counts them) wants; no
        Python object per delta, so a first round over 100 000 tasks costs what the round costs."""
        cnt = self._lib.psn_svc_schedule(self._svc)
        if cnt < 0:
            raise _lib.PsnError(int(cnt), self._lib.psn_svc_last_error(self._svc).decode())
        uid = np.empty(cnt, np.uint64)
        typ = np.empty(cnt, np.int32)
        res = np.zeros((cnt, 37), np.uint8)
        if cnt:
            _lib.check(self._lib.psn_svc_get_deltas(self._svc, 0, cnt, uid.ctypes.data, typ.ctypes.data, res.ctypes.data))
        return uid, typ, np.ascontiguousarray(res[:, :36]).view("S36").reshape(cnt)

    def Schedule(self) -> list[SchedulingDelta]:
        uid, typ, res = self.schedule_arrays()
        return [SchedulingDelta(int(u), r.decode(), ChangeType(int(t))) for u, t, r in zip(uid.tolist(), typ.tolist(), res.tolist())]

    def stats(self) -> SvcStats:
        st = SvcStats()
        _lib.check(self._lib.psn_svc_get_stats(self._svc, C.byref(st)))
        return st

    def task_placement(self, uid: int) -> str:
        buf = C.create_string_buffer(37)
        rc = self._lib.psn_svc_task_placement(self._svc, uid, buf)
        if rc != _lib.PSN_OK:
            raise KeyError(uid)
        return buf.value.decode()

    def export_network(self):
        """Host mirror of the resident network: dict of numpy arrays (slot arcs unpriced) plus the
        cost-model inputs and the layout."""
        lay = _lib.Layout()
        n, m2 = C.c_int32(), C.c_int64()
        rc = self._lib.psn_svc_network_size(self._svc, C.byref(lay), C.byref(n), C.byref(m2))
        if rc != _lib.PSN_OK:
            raise _lib.PsnError(rc, self._lib.psn_svc_last_error(self._svc).decode())
        n, m2 = n.value, m2.value
        N, E = lay.n_machines, lay.n_classes
        out = {"row_off": np.empty(n + 1, np.int64), "head": np.empty(m2, np.int32), "cap": np.empty(m2, np.int32),
               "cost": np.empty(m2, np.int32), "rev": np.empty(m2, np.int32), "supply": np.empty(n, np.int64),
               "ec_arc_base": np.empty(E, np.int64), "cap_cpu": np.empty(N, np.int64), "cap_mem": np.empty(N, np.int64),
               "used_cpu": np.empty(N, np.int64), "used_mem": np.empty(N, np.int64), "req_cpu": np.empty(E, np.int64),
               "req_mem": np.empty(E, np.int64)}
        order = ["row_off", "head", "cap", "cost", "rev", "supply", "ec_arc_base", "cap_cpu", "cap_mem", "used_cpu", "used_mem",
                 "req_cpu", "req_mem"]
        _lib.check(self._lib.psn_svc_export_network(self._svc, *[out[k].ctypes.data for k in order]))
        words = (N + 31) // 32
        mask = np.empty((E, words), np.uint32)
        _lib.check(self._lib.psn_svc_export_mask(self._svc, mask.ctypes.data))
        bits = np.unpackbits(mask.view(np.uint8).reshape(E, words * 4), axis=1, bitorder="little")
        out["allowed"] = bits[:, :N].astype(bool)
        out["layout"] = lay
        out["n"], out["m2"] = n, m2
        return out
