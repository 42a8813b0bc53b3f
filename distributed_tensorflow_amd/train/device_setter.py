"""``replica_device_setter``: between-graph-replication placement.

Reference usage (``tf_distributed.py:34-36``)::

    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d" % task_index, cluster=cluster)):

returns a *device function*: variable-holding ops go to ``/job:ps/task:K`` (K chosen by a
strategy, round-robin by default, over the ps tasks of the cluster), every other op goes to
``worker_device``. Ops that mutate a variable follow the variable (``Graph._resolve_device``),
so the SGD update executes on the parameter owner.

On MI355X the ps task index is the variable's *owner label*: the collective strategies turn
it into an owner rank (``K % num_workers``) for reduce-scatter / broadcast, and the CPU
parameter-server mode into the address of a real ps process.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

from ..framework.device import DeviceSpec
from .cluster import ClusterSpec

# op types that hold state and therefore live on parameter servers
STANDARD_PS_OPS = ("Variable", "VariableV2", "AutoReloadVariable", "VarHandleOp",
                   "MutableHashTable", "MutableHashTableV2", "MutableHashTableOfTensors",
                   "MutableHashTableOfTensorsV2", "MutableDenseHashTable",
                   "MutableDenseHashTableV2")


class _RoundRobinStrategy:
    """ps task = creation order modulo number of ps tasks."""

    def __init__(self, num_tasks: int):
        self._num_tasks = num_tasks
        self._next = 0

    def __call__(self, op) -> int:
        task = self._next
        self._next = (self._next + 1) % self._num_tasks
        return task


class GreedyLoadBalancingStrategy:
    """ps task = the one holding the fewest bytes so far (size-balanced owners, which is
    what makes reduce-scatter shards even when variable sizes differ a lot, e.g. the
    reference's 78 400-element ``W1`` next to a 10-element ``b2``)."""

    def __init__(self, num_tasks: int, load_fn: Optional[Callable] = None):
        self._loads = [0] * num_tasks
        self._load_fn = load_fn or byte_size_load_fn

    def __call__(self, op) -> int:
        task = min(range(len(self._loads)), key=lambda i: (self._loads[i], i))
        self._loads[task] += self._load_fn(op)
        return task


def byte_size_load_fn(op) -> int:
    shape = op.attrs.get("shape") or ()
    n = 1
    for d in shape:
        n *= int(d or 1)
    dtype = op.attrs.get("dtype")
    return n * (dtype.size if dtype is not None else 4)


class _ReplicaDeviceChooser:
    def __init__(self, ps_tasks: int, ps_device: str, worker_device: str, merge_devices: bool,
                 ps_ops: Iterable[str], ps_strategy: Callable):
        self._ps_tasks = ps_tasks
        self._ps_device = ps_device
        self._worker_device = worker_device
        self._merge_devices = merge_devices
        self._ps_ops = tuple(ps_ops)
        self._ps_strategy = ps_strategy

    def device_function(self, op) -> str:
        # an explicit, complete outer placement wins unless merging is requested
        current = DeviceSpec.from_string(op.device or "")
        if not self._merge_devices and op.device:
            return op.device
        if self._ps_tasks and self._ps_device and op.type in self._ps_ops:
            ps = DeviceSpec.from_string(self._ps_device)
            if ps.job is None or (current.job is not None and current.job != ps.job):
                target = ps                       # malformed / conflicting: leave as given
            else:
                ps.task = self._ps_strategy(op)
                target = ps
            return target.merge_from(_without_job_task(current)).to_string() \
                if self._merge_devices else target.to_string()
        worker = DeviceSpec.from_string(self._worker_device or "")
        # fields already set by enclosing scopes (e.g. an explicit /device:GPU:0) are kept
        return worker.merge_from(_only_device(current)).to_string()

    __call__ = device_function


def _without_job_task(spec: DeviceSpec) -> DeviceSpec:
    return DeviceSpec(device_type=spec.device_type, device_index=spec.device_index)


def _only_device(spec: DeviceSpec) -> DeviceSpec:
    return DeviceSpec(device_type=spec.device_type, device_index=spec.device_index)


def replica_device_setter(ps_tasks: int = 0, ps_device: str = "/job:ps",
                          worker_device: str = "/job:worker", merge_devices: bool = True,
                          cluster=None, ps_ops: Optional[Iterable[str]] = None,
                          ps_strategy: Optional[Callable] = None):
    """Return a device function for ``tf.device``, or ``None`` when there are no ps tasks."""
    if cluster is not None:
        spec = cluster if isinstance(cluster, ClusterSpec) else ClusterSpec(cluster)
        ps_job = DeviceSpec.from_string(ps_device).job
        ps_tasks = spec.num_tasks(ps_job) if ps_job in spec.jobs else 0
    if ps_tasks == 0:
        return None
    if ps_ops is None:
        ps_ops = STANDARD_PS_OPS
    if ps_strategy is None:
        ps_strategy = _RoundRobinStrategy(ps_tasks)
    if not callable(ps_strategy):
        raise TypeError("ps_strategy must be callable")
    return _ReplicaDeviceChooser(ps_tasks, ps_device, worker_device, merge_devices, ps_ops,
                                 ps_strategy).device_function
