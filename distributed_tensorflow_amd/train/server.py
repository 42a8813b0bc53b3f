"""``Server``: the in-process endpoint every task of a cluster starts.

Reference usage (``tf_distributed.py:18,31,96``)::

    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index)
    if FLAGS.job_name == "ps":
        server.join()
    ...
    with sv.prepare_or_wait_for_session(server.target) as sess:

Every process -- ps and worker alike -- creates a ``Server`` before branching on its role.
Here a server owns the task's *runtime* (see ``parallel.runtime``):

=============  ===================================================================
protocol       meaning
=============  ===================================================================
``grpc``       (default, the reference's implicit choice) asynchronous parameter
               server over TCP: a ps task hosts variables and applies updates;
               a worker task connects to every ps task. Needs no GPU.
``rccl``       collective strategies: one worker process per GPU joins a
               ``torch.distributed`` group (backend ``nccl`` = RCCL over xGMI on
               GPUs, ``gloo`` on CPUs); ps tasks are owner labels, not processes.
               ``config.strategy`` picks ``allreduce_sync`` (default) or
               ``reduce_scatter_sync``.
``local``      a cluster of one (``Server.create_local_server``).
=============  ===================================================================

``server.target`` is ``"<protocol>://host:port"``; ``Session(target)`` finds the server (and
through it the runtime) in this process's registry. ``join()`` blocks a ps task until
``stop()`` is called or a worker sends the shutdown request (the reference's ps can only be
killed).
"""
from __future__ import annotations

import os
import socket
import threading
from typing import Dict, Optional, Union

import torch

from ..framework.device import DeviceSpec
from ..utils import protowire as pw
from .cluster import ClusterDef, ClusterSpec

_REGISTRY: Dict[str, "Server"] = {}
_REGISTRY_LOCK = threading.Lock()


def lookup_server(target: str) -> "Server":
    with _REGISTRY_LOCK:
        server = _REGISTRY.get(target)
    if server is None:
        raise ValueError("no server for target %r was created in this process (known: %s)"
                         % (target, sorted(_REGISTRY)))
    return server


class ServerDef:
    """``ServerDef{cluster=1, job_name=2, task_index=3, protocol=5}``."""

    def __init__(self, cluster: ClusterDef, job_name: str, task_index: int, protocol: str):
        self.cluster = cluster
        self.job_name = job_name
        self.task_index = task_index
        self.protocol = protocol

    def SerializeToString(self) -> bytes:
        return (pw.f_message(1, self.cluster.SerializeToString()) + pw.f_bytes(2, self.job_name)
                + pw.f_varint(3, self.task_index) + pw.f_bytes(5, self.protocol))


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _default_device(job_name: str) -> torch.device:
    if job_name != "ps" and torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))
                            % max(1, torch.cuda.device_count()))
    return torch.device("cpu")


class Server:
    def __init__(self, server_or_cluster_def: Union[ClusterSpec, ClusterDef, dict, ServerDef],
                 job_name: Optional[str] = None, task_index: Optional[int] = None,
                 protocol: Optional[str] = None, config=None, start: bool = True,
                 comm=None):
        if isinstance(server_or_cluster_def, ServerDef):
            sd = server_or_cluster_def
            cluster = ClusterSpec(sd.cluster)
            job_name = sd.job_name if job_name is None else job_name
            task_index = sd.task_index if task_index is None else task_index
            protocol = sd.protocol if protocol is None else protocol
        else:
            cluster = server_or_cluster_def if isinstance(server_or_cluster_def, ClusterSpec) \
                else ClusterSpec(server_or_cluster_def)
        if job_name is None:
            if len(cluster.jobs) != 1:
                raise ValueError("job_name is required when the cluster has several jobs")
            job_name = cluster.jobs[0]
        if task_index is None:
            if cluster.num_tasks(job_name) != 1:
                raise ValueError("task_index is required when the job has several tasks")
            task_index = cluster.task_indices(job_name)[0]
        if job_name not in cluster.jobs:
            raise ValueError("job %r is not in the cluster %s" % (job_name, cluster.jobs))
        address = cluster.task_address(job_name, int(task_index))   # raises on a bad index
        self._cluster = cluster
        self._job_name = job_name
        self._task_index = int(task_index)
        self._protocol = protocol or os.environ.get("DTF_PROTOCOL") or "grpc"
        if self._protocol not in ("grpc", "rccl", "local"):
            raise ValueError("unknown protocol %r" % (self._protocol,))
        self._config = config
        self._address = address
        self._target = "%s://%s" % (self._protocol, address)
        self._started = False
        self._stopped = threading.Event()
        self._ps_service = None
        self._comm = comm
        self.runtime = None
        self.device = torch.device(config.device) if config is not None and \
            getattr(config, "device", None) else _default_device(job_name)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))
                                       % max(1, torch.cuda.device_count()))
        if start:
            self.start()

    # -- properties ----------------------------------------------------------------------
    @property
    def target(self) -> str:
        return self._target

    @property
    def server_def(self) -> ServerDef:
        return ServerDef(self._cluster.as_cluster_def(), self._job_name, self._task_index,
                         self._protocol)

    @property
    def cluster_spec(self) -> ClusterSpec:
        return self._cluster

    @property
    def job_name(self) -> str:
        return self._job_name

    @property
    def task_index(self) -> int:
        return self._task_index

    @property
    def device_name(self) -> str:
        return DeviceSpec(job=self._job_name, task=self._task_index).to_string()

    def _share_cpu_threads(self) -> None:
        """Tasks of one cluster that share a host share its cores: give this process
        ``cores / co-located tasks`` intra-op threads unless the user already chose
        (``OMP_NUM_THREADS``). Three tasks each spinning a full-machine thread pool on small
        ops is 10-20x slower than the same job with the pools sized to fit (measured on the
        1 ps + 2 workers MNIST job: 50 ms vs 2.6 ms per step on 8 cores)."""
        if self.device.type != "cpu" or os.environ.get("OMP_NUM_THREADS"):
            return
        host = self._address.rpartition(":")[0]
        local = {"localhost", "127.0.0.1", "::1", host}
        same_host = 0
        for job in self._cluster.jobs:
            for idx in self._cluster.task_indices(job):
                if self._cluster.task_address(job, idx).rpartition(":")[0] in local:
                    same_host += 1
        if same_host > 1:
            torch.set_num_threads(max(1, (os.cpu_count() or 1) // same_host))

    # -- lifecycle -----------------------------------------------------------------------
    def start(self) -> None:
        if self._started:
            return
        self._share_cpu_threads()
        from ..parallel import runtime as rt
        host, _, port = self._address.rpartition(":")
        if self._protocol == "local":
            self.runtime = rt.LocalRuntime(self.device)
        elif self._protocol == "grpc":
            if self._job_name == "ps":
                from ..parallel.ps_transport import make_ps_service
                workers = self._cluster.num_tasks("worker") \
                    if "worker" in self._cluster.jobs else 0
                self._ps_service = make_ps_service(host or "127.0.0.1", int(port),
                                                   expected_workers=workers)
                self._ps_service.start()
                self.runtime = rt.LocalRuntime(torch.device("cpu"))
                self.runtime.local_store = self._ps_service.store
            elif "ps" in self._cluster.jobs:
                self.runtime = rt.PsRuntime(self._cluster, self._job_name, self._task_index,
                                            self.device)
            else:
                self.runtime = rt.LocalRuntime(self.device)
        else:   # rccl: collective strategies, one process per worker
            strategy = getattr(self._config, "strategy", None) or \
                os.environ.get("DTF_STRATEGY") or "allreduce_sync"
            if self._job_name == "ps":
                # ps tasks are owner labels under collectives; a process started as one has
                # nothing to host -- it only needs to be joinable for script compatibility
                self.runtime = rt.LocalRuntime(torch.device("cpu"))
            else:
                comm = self._comm
                self._owns_comm = comm is None
                if comm is None:
                    from ..parallel.comm import make_communicator
                    # Rank, world size and rendezvous address are the ClusterSpec's: task N of
                    # the job is rank N, the chief's host:port is where the world meets. A
                    # launcher's MASTER_* (it exports the same values) may override the address
                    # only. DTF_COMM_BACKEND=gloo: several ranks rehearsed on ONE GPU (RCCL
                    # refuses two ranks per device); =rccl_native: own rendezvous + RCCL calls;
                    # default: torch.distributed with RCCL on GPUs, gloo on CPUs.
                    world = self._cluster.num_tasks(self._job_name)
                    ranks = self._cluster.task_indices(self._job_name)
                    chief = self._cluster.task_address(self._job_name, ranks[0])
                    chost, _, cport = chief.rpartition(":")
                    timeout = getattr(self._config, "rendezvous_timeout_s", None) or \
                        float(os.environ.get("DTF_RENDEZVOUS_TIMEOUT", "600"))
                    comm = make_communicator(
                        device=self.device, rank=ranks.index(self._task_index),
                        world_size=world,
                        master_addr=os.environ.get("MASTER_ADDR") or chost or "127.0.0.1",
                        master_port=int(os.environ.get("MASTER_PORT") or cport),
                        timeout_s=timeout)
                    if comm.world_size != world:
                        raise RuntimeError("the cluster declares %d %s tasks, the communicator "
                                           "has %d ranks" % (world, self._job_name,
                                                             comm.world_size))
                    self._comm = comm
                self.runtime = rt.CollectiveRuntime(
                    comm, self.device, strategy,
                    getattr(self._config, "grad_aggregation", None) or "sum")
        with _REGISTRY_LOCK:
            _REGISTRY[self._target] = self
        self._started = True

    def join(self, timeout: Optional[float] = None) -> None:
        """Block until the server is stopped. A ps task sits here for the whole job
        (reference ``tf_distributed.py:31``)."""
        if not self._started:
            self.start()
        restore = self._stop_on_signals()
        try:
            if self._ps_service is not None:
                if timeout is None:
                    while not self._ps_service.wait(0.5):
                        if self._stopped.is_set():
                            break
                else:
                    self._ps_service.wait(timeout)
                if self._ps_service.stopped or self._stopped.is_set():
                    import time
                    time.sleep(0.1)               # let the reply to the last request drain
                    self._ps_service.stop()       # join the service threads before returning
                return
            if timeout is None:
                while not self._stopped.wait(0.5):    # short waits: signal handlers get to run
                    pass
            else:
                self._stopped.wait(timeout)
        finally:
            restore()

    def _stop_on_signals(self):
        """While a task is parked in ``join()``, SIGTERM / SIGINT end the wait instead of
        killing the process mid-request: the service threads are joined and ``join`` returns
        (the reference's ps has no shutdown path at all and has to be killed,
        ``tf_distributed.py:31``). Only possible from the main thread; returns the undo."""
        import signal
        import threading
        if threading.current_thread() is not threading.main_thread():
            return lambda: None
        previous = {}

        def handler(signum, frame):
            self._stopped.set()

        for sig in (signal.SIGTERM, signal.SIGINT):
            try:
                previous[sig] = signal.signal(sig, handler)
            except (ValueError, OSError):
                pass

        def restore():
            for sig, old in previous.items():
                try:
                    signal.signal(sig, old)
                except (ValueError, OSError):
                    pass
        return restore

    def stop(self) -> None:
        self._stopped.set()
        if self._ps_service is not None:
            self._ps_service.stop()
        if self.runtime is not None:
            self.runtime.shutdown()
        if getattr(self, "_owns_comm", False) and self._comm is not None:
            # the process group this server created: tear it down here, not at interpreter
            # exit (a backend's worker threads must not outlive their group)
            try:
                self._comm.barrier()
            except Exception:  # noqa: BLE001 -- peers may already be gone
                pass
            self._comm.close()
            self._comm = None
        with _REGISTRY_LOCK:
            if _REGISTRY.get(self._target) is self:
                del _REGISTRY[self._target]

    @staticmethod
    def create_local_server(config=None, start: bool = True) -> "Server":
        """A single-task cluster on an ephemeral localhost port (the cheapest way to test
        code written against ``server.target``)."""
        cluster = ClusterSpec({"local": ["127.0.0.1:%d" % _free_port()]})
        return Server(cluster, job_name="local", task_index=0, protocol="local", config=config,
                      start=start)
