"""Host-side mirror of the reference's Firmament boundary for the scheduling path.

Names and reply codes follow the protobuf contract the reference keeps
(/root/reference/pkg/firmament/firmament_scheduler.proto:15-45, enums :110-135); the
unit conversions follow the watchers that feed it
(/root/reference/pkg/k8sclient/podwatcher.go:135-175,
/root/reference/pkg/k8sclient/nodewatcher.go:101-121).
"""
from __future__ import annotations

import enum
import re

BYTES_TO_KB = 1024  # /root/reference/pkg/k8sclient/types.go:28


class TaskReplyType(enum.IntEnum):  # firmament_scheduler.proto:110-120
    TASK_COMPLETED_OK = 0
    TASK_SUBMITTED_OK = 1
    TASK_REMOVED_OK = 2
    TASK_FAILED_OK = 3
    TASK_UPDATED_OK = 4
    TASK_NOT_FOUND = 5
    TASK_JOB_NOT_FOUND = 6
    TASK_ALREADY_SUBMITTED = 7
    TASK_STATE_NOT_CREATED = 8


class NodeReplyType(enum.IntEnum):  # firmament_scheduler.proto:122-129
    NODE_ADDED_OK = 0
    NODE_FAILED_OK = 1
    NODE_REMOVED_OK = 2
    NODE_UPDATED_OK = 3
    NODE_NOT_FOUND = 4
    NODE_ALREADY_EXISTS = 5


class ChangeType(enum.IntEnum):  # scheduling_delta.proto:26-35
    NOOP = 0
    PLACE = 1
    PREEMPT = 2
    MIGRATE = 3


_SUFFIX = {"": 1, "k": 10**3, "M": 10**6, "G": 10**9, "T": 10**12, "P": 10**15, "E": 10**18,
           "Ki": 2**10, "Mi": 2**20, "Gi": 2**30, "Ti": 2**40, "Pi": 2**50, "Ei": 2**60}
_QTY = re.compile(r"^([+-]?\d+(?:\.\d+)?)(m|k|M|G|T|P|E|Ki|Mi|Gi|Ti|Pi|Ei)?$")


def _quantity_milli(q: str) -> int:
    """Kubernetes resource quantity -> integer thousandths (rounded up, as MilliValue does)."""
    m = _QTY.match(q.strip())
    if not m:
        raise ValueError(f"bad quantity {q!r}")
    num, suf = m.group(1), m.group(2) or ""
    whole, _, frac = num.partition(".")
    digits = int(whole + frac)
    scale = 10 ** len(frac)
    if suf == "m":
        return -((-digits) // scale)
    return -((-digits * 1000 * _SUFFIX[suf]) // scale)


def cpu_millicores(q: str) -> int:
    """CPU request/capacity as the reference sends it: Quantity.MilliValue()
    (/root/reference/pkg/k8sclient/podwatcher.go:141)."""
    return _quantity_milli(q)


def memory_kb(q: str) -> int:
    """Memory as the reference sends it: bytes / 1024, truncated
    (/root/reference/pkg/k8sclient/podwatcher.go:143-144,169)."""
    return (_quantity_milli(q) // 1000) // BYTES_TO_KB
