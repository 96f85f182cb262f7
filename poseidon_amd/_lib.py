"""ctypes binding of libpsn.so (include/psn.h, include/psn_builder.h).

The library is HIP-only: there is no CPU implementation behind these calls.  If
the shared object is missing, or no GPU is present when a context is created, the
call raises instead of falling back to anything else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSN_LIB") or os.path.join(_HERE, "libpsn.so")  # PSN_LIB: A/B builds during tuning

PSN_OK = 0
PSN_EINVAL, PSN_ENOMEM, PSN_EDEVICE, PSN_EINFEASIBLE = -1, -2, -3, -4
PSN_EOVERFLOW, PSN_ESTATE, PSN_ELIMIT = -5, -6, -7
PSN_WARM_START = 1
PSN_CFG_EAGER = 1
PSN_CFG_KTIME = 2
PSN_CFG_EPS_FROM_MAX = 4
KT_NAMES = ["push", "gu_expand", "relabel_compact", "saturate", "other"]

EXPORTS = [
    "psn_version", "psn_kernel_list", "psn_strerror", "psn_last_error", "psn_create", "psn_destroy",
    "psn_upload_csr", "psn_eval_costs", "psn_update_caps", "psn_update_supplies", "psn_solve",
    "psn_get_flow", "psn_get_residual", "psn_get_prices", "psn_get_costs", "psn_get_placements",
    "psn_placements_device_ptr", "psn_get_placements_device", "psn_kernel_times", "psn_debug_trace",
    "psn_synth_cluster", "psn_network_size", "psn_build_network", "psn_fnv1a64",
    "psn_csr_from_arcs", "psn_dimacs_write", "psn_dimacs_size", "psn_dimacs_read",
    "psn_set_admission_mask", "psn_reprice_costs", "psn_patch_arcs", "psn_settle_sources", "psn_solve_batch",
    "psn_svc_create", "psn_svc_destroy", "psn_svc_last_error", "psn_svc_task_submitted", "psn_svc_task_updated",
    "psn_svc_task_completed", "psn_svc_task_failed", "psn_svc_task_removed", "psn_svc_add_task_stats",
    "psn_svc_node_added", "psn_svc_node_updated", "psn_svc_node_failed", "psn_svc_node_removed",
    "psn_svc_add_node_stats", "psn_svc_check", "psn_svc_schedule", "psn_svc_get_deltas", "psn_svc_get_stats",
    "psn_svc_task_submitted_sel", "psn_svc_node_labels", "psn_svc_export_mask", "psn_svc_network_size", "psn_svc_export_network", "psn_svc_task_placement", "psn_svc_solver",
    "psn_gather_placements", "psn_debug_alloc_count", "psn_get_placements_device_noalloc", "psn_price_scale", "psn_update_costs",
    "psn_dimacs_changes_size", "psn_dimacs_read_changes", "psn_dimacs_read_lower", "psn_dimacs_read_changes_lower",
    "psn_svc_add_node_stats_raw", "psn_svc_add_task_stats_raw", "psn_build_network_device", "psn_get_structure", "psn_get_rows", "psn_build_network_rooms_device",
]


class PsnError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"libpsn error {code}: {detail}")


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("alpha_log2", C.c_int32), ("rounds_per_batch", C.c_int32),
                ("flags", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("total_cost", C.c_int64), ("status", C.c_int32), ("refines", C.c_int32),
                ("arc_scans", C.c_uint64), ("pushes", C.c_uint64), ("relabels", C.c_uint64),
                ("vertex_visits", C.c_uint64), ("gu_arc_scans", C.c_uint64),
                ("gu_relaxations", C.c_uint64), ("sat_arc_scans", C.c_uint64), ("rounds", C.c_uint64),
                ("gu_iterations", C.c_uint64), ("global_updates", C.c_uint64), ("launches", C.c_uint64),
                ("solve_ms", C.c_double), ("gu_arc_skips", C.c_uint64), ("arc_skips", C.c_uint64),
                ("warm_started", C.c_int32), ("mirrored_replays", C.c_int32), ("replays", C.c_uint64),
                ("first_units", C.c_uint64), ("first_eps_log2", C.c_int32), ("last_phase_replays", C.c_int32),
                ("warm_declined", C.c_int32), ("reserved_", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("n_machines", C.c_int32), ("n_classes", C.c_int32), ("n_jobs", C.c_int32),
                ("n_tasks", C.c_int32), ("k_slots", C.c_int32)]


_lib = None


def load_library():
    """Load libpsn.so from the package directory (built in-tree by `make`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PsnError(PSN_EDEVICE, f"{LIB_PATH} is missing: run `make` (hipcc, gfx950) first")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    lib.psn_version.restype = C.c_int
    lib.psn_kernel_list.restype = C.c_char_p
    lib.psn_strerror.restype = C.c_char_p
    lib.psn_strerror.argtypes = [C.c_int]
    lib.psn_last_error.restype = C.c_char_p
    lib.psn_last_error.argtypes = [vp]
    lib.psn_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.psn_destroy.argtypes = [vp]
    lib.psn_destroy.restype = None
    lib.psn_upload_csr.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp]
    lib.psn_build_network_device.argtypes = [vp, C.POINTER(Layout), i32, i32, vp, vp, vp]
    lib.psn_eval_costs.argtypes = [vp, C.POINTER(Layout), vp, vp, vp, vp, vp, vp, vp]
    lib.psn_reprice_costs.argtypes = [vp, C.POINTER(Layout), vp, vp, vp, vp, vp, vp, vp]
    lib.psn_set_admission_mask.argtypes = [vp, C.POINTER(Layout), vp]
    lib.psn_patch_arcs.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.psn_settle_sources.argtypes = [vp, i64, vp, C.POINTER(i64)]
    lib.psn_update_caps.argtypes = [vp, i64, vp, vp]
    lib.psn_update_supplies.argtypes = [vp, i64, vp, vp]
    lib.psn_update_costs.argtypes = [vp, i64, vp, vp]
    lib.psn_dimacs_changes_size.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(i64)]
    lib.psn_dimacs_read_changes.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.psn_dimacs_read_changes_lower.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.psn_dimacs_read_lower.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, C.POINTER(i64), vp]
    lib.psn_solve.argtypes = [vp, u32, C.POINTER(Result)]
    lib.psn_solve_batch.argtypes = [vp, i32, u32, i32, vp]
    lib.psn_get_flow.argtypes = [vp, vp]
    lib.psn_get_residual.argtypes = [vp, vp]
    lib.psn_get_prices.argtypes = [vp, vp]
    lib.psn_get_costs.argtypes = [vp, vp, vp]
    lib.psn_get_structure.argtypes = [vp, vp, vp]
    lib.psn_get_rows.argtypes = [vp, vp, vp]
    lib.psn_price_scale.argtypes = [vp, C.POINTER(i64)]
    lib.psn_get_placements.argtypes = [vp, C.POINTER(Layout), vp]
    lib.psn_get_placements_device.argtypes = [vp, C.POINTER(Layout), vp]
    lib.psn_get_placements_device_noalloc.argtypes = [vp, C.POINTER(Layout), vp]
    lib.psn_kernel_times.argtypes = [vp, vp, vp]
    lib.psn_placements_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    lib.psn_synth_cluster.argtypes = [i32, u64, C.POINTER(Layout), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.psn_network_size.argtypes = [C.POINTER(Layout), C.POINTER(i32), C.POINTER(i64)]
    lib.psn_build_network.argtypes = [C.POINTER(Layout), i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.psn_csr_from_arcs.argtypes = [i32, i64] + [vp] * 10
    lib.psn_dimacs_write.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, C.c_char_p]
    lib.psn_dimacs_size.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i64)]
    lib.psn_dimacs_read.argtypes = [C.c_char_p, i32, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    lib.psn_gather_placements.argtypes = [vp, i32, vp, i64, i32, vp]
    lib.psn_debug_alloc_count.restype = u64
    lib.psn_debug_alloc_count.argtypes = []
    lib.psn_fnv1a64.restype = u64
    lib.psn_fnv1a64.argtypes = [vp, i64, u64]
    _lib = lib
    return lib


def check(rc: int, ctx=None):
    if rc != PSN_OK:
        lib = load_library()
        detail = lib.psn_strerror(rc).decode()
        if ctx:
            extra = lib.psn_last_error(ctx).decode()
            if extra:
                detail += " (" + extra + ")"
        raise PsnError(rc, detail)
