"""Device specification strings: ``/job:<name>/replica:<r>/task:<t>/device:<TYPE>:<i>``.

The reference only ever writes two shapes of device string -- ``/job:worker/task:%d``
(reference ``tf_distributed.py:35``) and the ``/job:ps`` prefix that
``replica_device_setter`` produces -- but device functions return partial specs that are
merged with enclosing scopes, so the full grammar with optional fields and merge semantics is
implemented here.

On MI355X a fully resolved worker device maps to one process / one GPU: ``/job:worker/task:N``
is rank N of the worker group and ``device:GPU:0`` is that rank's (only) visible HIP device.
"""
from __future__ import annotations

from typing import Optional

_DEVICE_TYPES = ("CPU", "GPU")


class DeviceSpec:
    """Parsed, mergeable device specification. Every field is optional (``None`` = unset)."""

    __slots__ = ("job", "replica", "task", "device_type", "device_index")

    def __init__(self, job: Optional[str] = None, replica: Optional[int] = None,
                 task: Optional[int] = None, device_type: Optional[str] = None,
                 device_index: Optional[int] = None):
        self.job = job
        self.replica = replica
        self.task = task
        self.device_type = device_type.upper() if device_type else None
        self.device_index = device_index

    # -- parsing -------------------------------------------------------------------------
    @classmethod
    def from_string(cls, spec: Optional[str]) -> "DeviceSpec":
        out = cls()
        if not spec:
            return out
        if not spec.startswith("/"):
            raise ValueError("device spec must start with '/': %r" % (spec,))
        for part in spec.split("/"):
            if not part:
                continue
            fields = part.split(":")
            key = fields[0].lower()
            if key == "job" and len(fields) == 2 and fields[1]:
                out.job = fields[1]
            elif key == "replica" and len(fields) == 2:
                out.replica = _to_index(fields[1], spec)
            elif key == "task" and len(fields) == 2:
                out.task = _to_index(fields[1], spec)
            elif key == "device" and len(fields) == 3:
                out.device_type = fields[1].upper()
                out.device_index = None if fields[2] == "*" else _to_index(fields[2], spec)
            elif fields[0].upper() in _DEVICE_TYPES and len(fields) == 2:
                # legacy short form "/gpu:0", "/cpu:0"
                out.device_type = fields[0].upper()
                out.device_index = None if fields[1] == "*" else _to_index(fields[1], spec)
            else:
                raise ValueError("unknown field %r in device spec %r" % (part, spec))
        return out

    # -- merging -------------------------------------------------------------------------
    def merge_from(self, other: "DeviceSpec") -> "DeviceSpec":
        """Overlay ``other`` on ``self``: fields set in ``other`` win (inner scope wins)."""
        return DeviceSpec(
            job=other.job if other.job is not None else self.job,
            replica=other.replica if other.replica is not None else self.replica,
            task=other.task if other.task is not None else self.task,
            device_type=other.device_type if other.device_type is not None else self.device_type,
            device_index=(other.device_index if other.device_type is not None
                          or other.device_index is not None else self.device_index),
        )

    def is_subset_of(self, full: "DeviceSpec") -> bool:
        """True when every field set here equals the same field of ``full``."""
        for f in self.__slots__:
            mine = getattr(self, f)
            if mine is not None and mine != getattr(full, f):
                return False
        return True

    # -- formatting ----------------------------------------------------------------------
    def to_string(self) -> str:
        out = ""
        if self.job is not None:
            out += "/job:%s" % self.job
        if self.replica is not None:
            out += "/replica:%d" % self.replica
        if self.task is not None:
            out += "/task:%d" % self.task
        if self.device_type is not None:
            idx = "*" if self.device_index is None else str(self.device_index)
            out += "/device:%s:%s" % (self.device_type, idx)
        return out

    def __str__(self) -> str:
        return self.to_string()

    def __repr__(self) -> str:
        return "DeviceSpec(%r)" % self.to_string()

    def __eq__(self, other) -> bool:
        return isinstance(other, DeviceSpec) and all(
            getattr(self, f) == getattr(other, f) for f in self.__slots__)

    def __hash__(self) -> int:
        return hash(tuple(getattr(self, f) for f in self.__slots__))

    def __bool__(self) -> bool:
        return any(getattr(self, f) is not None for f in self.__slots__)


def _to_index(text: str, spec: str) -> int:
    try:
        value = int(text)
    except ValueError:
        raise ValueError("non-integer index %r in device spec %r" % (text, spec))
    if value < 0:
        raise ValueError("negative index in device spec %r" % (spec,))
    return value


def merge_device(outer: Optional[str], inner: Optional[str]) -> str:
    """Merge two device strings; fields of ``inner`` override ``outer``."""
    return DeviceSpec.from_string(outer).merge_from(DeviceSpec.from_string(inner)).to_string()


def canonical_name(spec: Optional[str]) -> str:
    return DeviceSpec.from_string(spec).to_string()
