"""ClusterSpec: the job -> task -> address table that every process of a job shares.

Mirrors the behaviour the reference relies on at ``tf_distributed.py:11``,
``tf_distributed_worker.py:10`` and ``tf_distributed_1000Matrix.py:10``
(``ClusterSpec({"ps": [...], "worker": [...]})``), plus the constructors the reference's dead
argparse tail intended (comma-separated ``--ps`` / ``--worker`` host lists,
``tf_distributed.py:137-148``) and launcher-driven synthesis from ``RANK``/``WORLD_SIZE``.

MI355X mapping: worker task N is rank N of the ``torch.distributed`` process group (one
process per GPU); ps tasks are either real processes (CPU parameter-server mode) or *owner
labels* that the collective strategies fold onto worker ranks (``owner = ps_task % workers``).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Mapping, Optional, Union

from ..utils import protowire as pw

_TaskTable = Dict[int, str]


class ClusterDef:
    """Serializable form of a cluster: ordered jobs, each an index->address table.

    ``SerializeToString`` emits the protobuf wire layout ``ClusterDef{repeated JobDef job=1}``,
    ``JobDef{string name=1; map<int32,string> tasks=2}`` so that event files / server
    definitions written by this framework carry a standard cluster description.
    """

    def __init__(self, jobs: Optional[Mapping[str, Mapping[int, str]]] = None):
        self.jobs: Dict[str, _TaskTable] = {
            name: dict(tasks) for name, tasks in (jobs or {}).items()}

    def SerializeToString(self) -> bytes:
        out = b""
        for name in sorted(self.jobs):
            job = pw.f_bytes(1, name)
            for idx in sorted(self.jobs[name]):
                entry = pw.f_varint(1, idx) + pw.f_bytes(2, self.jobs[name][idx])
                job += pw.f_message(2, entry)
            out += pw.f_message(1, job)
        return out

    @classmethod
    def FromString(cls, data: bytes) -> "ClusterDef":
        jobs: Dict[str, _TaskTable] = {}
        for field, _w, value in pw.iter_fields(data):
            if field != 1:
                continue
            name = ""
            tasks: _TaskTable = {}
            for jf, _jw, jv in pw.iter_fields(value):
                if jf == 1:
                    name = jv.decode("utf-8")
                elif jf == 2:
                    entry = pw.parse(jv)
                    tasks[int(entry.get(1, [0])[0])] = entry[2][0].decode("utf-8")
            jobs[name] = tasks
        return cls(jobs)

    def __eq__(self, other) -> bool:
        return isinstance(other, ClusterDef) and self.jobs == other.jobs


class ClusterSpec:
    """Immutable mapping of job name -> {task index -> ``host:port``}."""

    def __init__(self, cluster: Union[None, Mapping, "ClusterSpec", ClusterDef] = None):
        self._spec: Dict[str, _TaskTable] = {}
        if cluster is None:
            cluster = {}
        if isinstance(cluster, ClusterSpec):
            self._spec = {j: dict(t) for j, t in cluster._spec.items()}
        elif isinstance(cluster, ClusterDef):
            self._spec = {j: dict(t) for j, t in cluster.jobs.items()}
        elif isinstance(cluster, Mapping):
            for job, tasks in cluster.items():
                if not isinstance(job, str) or not job:
                    raise TypeError("job names must be non-empty strings, got %r" % (job,))
                if isinstance(tasks, Mapping):
                    table = {}
                    for idx, addr in tasks.items():
                        table[_check_index(idx, job)] = _check_addr(addr, job)
                elif isinstance(tasks, (list, tuple)):
                    table = {i: _check_addr(a, job) for i, a in enumerate(tasks)}
                else:
                    raise TypeError(
                        "tasks of job %r must be a list or a dict of index->address, got %r"
                        % (job, type(tasks).__name__))
                self._spec[job] = table
        else:
            raise TypeError("cluster must be a dict, ClusterSpec or ClusterDef, got %r"
                            % type(cluster).__name__)

    # -- queries -------------------------------------------------------------------------
    @property
    def jobs(self) -> List[str]:
        return sorted(self._spec)

    def _tasks(self, job_name: str) -> _TaskTable:
        try:
            return self._spec[job_name]
        except KeyError:
            raise ValueError("No such job in cluster: %r" % (job_name,))

    def num_tasks(self, job_name: str) -> int:
        return len(self._tasks(job_name))

    def task_indices(self, job_name: str) -> List[int]:
        return sorted(self._tasks(job_name))

    def task_address(self, job_name: str, task_index: int) -> str:
        tasks = self._tasks(job_name)
        try:
            return tasks[task_index]
        except KeyError:
            raise ValueError("No task with index %r in job %r" % (task_index, job_name))

    def job_tasks(self, job_name: str) -> List[Optional[str]]:
        tasks = self._tasks(job_name)
        if not tasks:
            return []
        return [tasks.get(i) for i in range(max(tasks) + 1)]

    def as_dict(self) -> Dict[str, Union[List[str], Dict[int, str]]]:
        out: Dict[str, Union[List[str], Dict[int, str]]] = {}
        for job, tasks in self._spec.items():
            dense = sorted(tasks) == list(range(len(tasks)))
            out[job] = [tasks[i] for i in range(len(tasks))] if dense else dict(tasks)
        return out

    def as_cluster_def(self) -> ClusterDef:
        return ClusterDef(self._spec)

    # -- rank arithmetic used by the collective runtime ----------------------------------
    def flat_tasks(self, job_order: Iterable[str] = ("worker", "ps")) -> List[tuple]:
        """Every (job, task) pair in a fixed global order: workers first, then ps, then the
        remaining jobs alphabetically. The position in this list is the task's global rank in
        process groups that span the whole cluster."""
        order = [j for j in job_order if j in self._spec]
        order += [j for j in self.jobs if j not in order]
        return [(j, i) for j in order for i in self.task_indices(j)]

    def global_rank(self, job_name: str, task_index: int) -> int:
        self.task_address(job_name, task_index)
        return self.flat_tasks().index((job_name, task_index))

    # -- dunder --------------------------------------------------------------------------
    def __bool__(self) -> bool:
        return bool(self._spec)

    __nonzero__ = __bool__

    def __eq__(self, other) -> bool:
        return isinstance(other, ClusterSpec) and self._spec == other._spec

    def __ne__(self, other) -> bool:
        return not self == other

    def __hash__(self) -> int:
        return hash(json.dumps(self._spec, sort_keys=True))

    def __repr__(self) -> str:
        items = ", ".join("%r: %r" % (j, self.as_dict()[j]) for j in self.jobs)
        return "ClusterSpec({%s})" % items

    # -- alternative constructors --------------------------------------------------------
    @classmethod
    def from_host_lists(cls, ps_hosts: Union[str, Iterable[str], None],
                        worker_hosts: Union[str, Iterable[str], None]) -> "ClusterSpec":
        """Build from comma-separated ``host:port`` lists (the CLI form the reference's
        argparse block declares at ``tf_distributed.py:137-148`` but never wires up)."""
        spec = {}
        ps = _split_hosts(ps_hosts)
        workers = _split_hosts(worker_hosts)
        if ps:
            spec["ps"] = ps
        if workers:
            spec["worker"] = workers
        return cls(spec)

    @classmethod
    def from_tf_config(cls, env: Optional[Mapping[str, str]] = None) -> Optional["ClusterSpec"]:
        """Parse a ``TF_CONFIG``-style JSON env var: ``{"cluster": {...}, "task": {...}}``."""
        env = os.environ if env is None else env
        raw = env.get("TF_CONFIG")
        if not raw:
            return None
        cfg = json.loads(raw)
        return cls(cfg.get("cluster", {}))

    @classmethod
    def from_launcher_env(cls, env: Optional[Mapping[str, str]] = None, num_ps: int = 0,
                          base_port: int = 0) -> "ClusterSpec":
        """Synthesise a cluster from ``torchrun``-style variables: ``WORLD_SIZE`` workers on
        ``MASTER_ADDR``; worker N <-> rank N <-> GPU N. ``num_ps`` adds ps *owner labels* that
        the collective strategies fold onto worker ranks (no extra processes are started)."""
        env = os.environ if env is None else env
        world = int(env.get("WORLD_SIZE", "1"))
        addr = env.get("MASTER_ADDR", "127.0.0.1")
        port = base_port or int(env.get("MASTER_PORT", "29500")) + 1
        spec = {"worker": ["%s:%d" % (addr, port + i) for i in range(world)]}
        if num_ps:
            spec["ps"] = ["%s:%d" % (addr, port + world + i) for i in range(num_ps)]
        return cls(spec)


def task_from_env(env: Optional[Mapping[str, str]] = None) -> tuple:
    """``(job_name, task_index)`` from ``TF_CONFIG`` or, failing that, ``RANK``."""
    env = os.environ if env is None else env
    raw = env.get("TF_CONFIG")
    if raw:
        task = json.loads(raw).get("task", {})
        return str(task.get("type", "worker")), int(task.get("index", 0))
    return "worker", int(env.get("RANK", "0"))


def _split_hosts(hosts: Union[str, Iterable[str], None]) -> List[str]:
    if hosts is None:
        return []
    if isinstance(hosts, str):
        return [h.strip() for h in hosts.split(",") if h.strip()]
    return [str(h) for h in hosts]


def _check_index(idx, job: str) -> int:
    try:
        value = int(idx)
    except (TypeError, ValueError):
        raise TypeError("task index %r of job %r is not an integer" % (idx, job))
    if value < 0:
        raise ValueError("task index %r of job %r is negative" % (idx, job))
    return value


def _check_addr(addr, job: str) -> str:
    if isinstance(addr, bytes):
        addr = addr.decode("utf-8")
    if not isinstance(addr, str) or not addr:
        raise TypeError("task address %r of job %r is not a string" % (addr, job))
    return addr
