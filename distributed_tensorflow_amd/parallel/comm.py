"""Collective communicators.

The reference moves every tensor between tasks through TensorFlow's gRPC ``RecvTensor``
(implicit ``_Send/_Recv`` pairs created by ``replica_device_setter`` placement,
``tf_distributed.py:34-36``): variable pull ps->worker, gradient push worker->ps, the
``global_step`` round trip, the chief's init values and the non-chief readiness poll. On an
MI355X node those become collectives over xGMI: pull = ``broadcast`` / ``all_gather``, push =
``reduce_scatter`` / ``all_reduce``, init = ``broadcast``, readiness = ``barrier``.

One interface, three implementations *inside this runtime* (this is test scaffolding plus the
CPU plumbing configuration, not multi-vendor GPU dispatch):

* ``TorchDistComm``   -- ``torch.distributed``: backend ``nccl`` (= RCCL over xGMI) for GPU
  tensors, ``gloo`` for the CPU configuration. One process per GPU.
* ``LoopbackComm``    -- N ranks as threads of one process exchanging tensors through shared
  slots; deterministic rank-ordered reductions; optional fault injection (delay / drop a
  rank) for failure-detection tests.
* ``SingleComm``      -- world size 1 (every collective is the identity).

All collectives are synchronous from the caller's point of view unless ``async_op`` is
requested, in which case a handle with ``wait()`` is returned so callers can overlap the
transfer with GEMM tiles on the compute stream.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (required by the driver)


class CommError(RuntimeError):
    """A collective failed or timed out (peer died, watchdog fired)."""


class _Done:
    def wait(self, timeout: Optional[float] = None) -> bool:
        return True

    def is_completed(self) -> bool:
        return True


class _Work:
    """Handle of an asynchronous collective. ``wait()`` turns whatever the backend raises when
    the collective fails after it was enqueued -- a watchdog timeout, a peer that died, an RCCL
    asynchronous error -- into :class:`CommError`, the one exception the recovery paths
    (``Supervisor.run_with_recovery``) look for."""

    __slots__ = ("_work", "_what")

    def __init__(self, work, what: str):
        self._work, self._what = work, what

    def wait(self, timeout: Optional[float] = None) -> bool:
        try:
            if timeout is None:
                return bool(self._work.wait())
            import datetime
            return bool(self._work.wait(datetime.timedelta(seconds=timeout)))
        except CommError:
            raise
        except Exception as e:  # noqa: BLE001 -- RuntimeError, DistBackendError, timeouts
            raise CommError("%s failed while in flight: %s" % (self._what, e)) from e

    def is_completed(self) -> bool:
        try:
            return bool(self._work.is_completed())
        except Exception as e:  # noqa: BLE001
            raise CommError("%s failed while in flight: %s" % (self._what, e)) from e


class Communicator:
    rank: int = 0
    world_size: int = 1
    device: torch.device = torch.device("cpu")
    backend: str = "single"
    # all_gather(output, input) accepts ``input`` = this rank's slot of ``output`` (no copy of
    # the local shard): true for RCCL, the loopback world and a world of one
    inplace_all_gather = True

    def broadcast(self, tensor: torch.Tensor, src: int = 0, async_op: bool = False):
        raise NotImplementedError

    def all_reduce(self, tensor: torch.Tensor, op: str = "sum", async_op: bool = False):
        raise NotImplementedError

    def reduce_scatter(self, output: torch.Tensor, input: torch.Tensor, op: str = "sum",
                       async_op: bool = False):
        raise NotImplementedError

    def all_gather(self, output: torch.Tensor, input: torch.Tensor, async_op: bool = False):
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def broadcast_object(self, obj: Any, src: int = 0) -> Any:
        raise NotImplementedError

    def all_gather_object(self, obj: Any) -> List[Any]:
        raise NotImplementedError

    def split(self, color: int, key: Optional[int] = None) -> "Communicator":
        """Sub-communicator of the ranks that pass the same ``color``, ordered by ``key``
        (default: current rank). Collective: every rank of this communicator must call it."""
        raise NotImplementedError

    def close(self) -> None:
        pass


class SingleComm(Communicator):
    def __init__(self, device: torch.device = torch.device("cpu")):
        self.rank, self.world_size, self.device = 0, 1, torch.device(device)

    def broadcast(self, tensor, src=0, async_op=False):
        return _Done() if async_op else None

    def all_reduce(self, tensor, op="sum", async_op=False):
        return _Done() if async_op else None

    def reduce_scatter(self, output, input, op="sum", async_op=False):
        output.copy_(input.reshape(output.shape))
        return _Done() if async_op else None

    def all_gather(self, output, input, async_op=False):
        output.reshape(input.shape).copy_(input)
        return _Done() if async_op else None

    def barrier(self):
        pass

    def broadcast_object(self, obj, src=0):
        return obj

    def all_gather_object(self, obj):
        return [obj]

    def split(self, color, key=None):
        return SingleComm(self.device)


# ---------------------------------------------------------------------------------------
# torch.distributed (RCCL on GPU, gloo on CPU)
# ---------------------------------------------------------------------------------------
class TorchDistComm(Communicator):
    """``torch.distributed`` process group. ``backend="nccl"`` is RCCL on ROCm."""

    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None,
                 device: Optional[torch.device] = None, backend: Optional[str] = None,
                 master_addr: Optional[str] = None, master_port: Optional[int] = None,
                 timeout_s: float = 600.0, group=None):
        import datetime
        import torch.distributed as dist
        self._dist = dist
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) \
                if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.backend = backend or ("nccl" if self.device.type == "cuda" else "gloo")
        self._owns_pg = False
        if not dist.is_initialized():
            rank = int(os.environ.get("RANK", "0")) if rank is None else rank
            world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None \
                else world_size
            addr = master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
            port = master_port or int(os.environ.get("MASTER_PORT", "29500"))
            kwargs = {}
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
                if self.backend == "nccl":
                    kwargs["device_id"] = self.device
            dist.init_process_group(self.backend, init_method="tcp://%s:%d" % (addr, port),
                                    rank=rank, world_size=world_size,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
            self._owns_pg = True
        self._group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def _op(self, op: str):
        ops = self._dist.ReduceOp
        return {"sum": ops.SUM, "max": ops.MAX, "min": ops.MIN, "avg": ops.AVG}[op]

    def _wrap(self, fn: Callable, async_op: bool, what: str = "collective"):
        try:
            work = fn()
        except Exception as e:  # noqa: BLE001 -- RCCL async error / watchdog timeout at enqueue
            raise CommError("%s failed: %s" % (what, e)) from e
        if not async_op:
            return None
        return _Work(work, what) if work is not None else _Done()

    @property
    def inplace_all_gather(self) -> bool:      # gloo gathers into a list of chunks
        return self.backend == "nccl"

    def _global(self, group_rank: int) -> int:
        """torch.distributed addresses broadcast sources by global rank."""
        if self._group is None:
            return group_rank
        return self._dist.get_global_rank(self._group, group_rank)

    @staticmethod
    def _bytes_view(t: torch.Tensor) -> torch.Tensor:
        """Pure data movement of 1-byte float types goes over the wire as uint8 (not every
        backend build knows the fp8 dtypes)."""
        return t.view(torch.uint8) if t.element_size() == 1 and t.is_floating_point() else t

    def _staged(self, tensor: torch.Tensor) -> bool:
        """gloo with GPU tensors (several ranks rehearsed on one GPU): collectives go through
        host copies, synchronously -- gloo's own device path stalls with several asynchronous
        collectives in flight from processes that share a device."""
        return self.backend == "gloo" and tensor.is_cuda

    def broadcast(self, tensor, src=0, async_op=False):
        src = self._global(src)
        tensor = self._bytes_view(tensor)
        if self._staged(tensor):
            host = tensor.cpu()
            self._wrap(lambda: self._dist.broadcast(host, src=src, group=self._group), False)
            tensor.copy_(host)
            return _Done() if async_op else None
        return self._wrap(lambda: self._dist.broadcast(tensor, src=src, group=self._group,
                                                       async_op=async_op), async_op, "broadcast")

    def all_reduce(self, tensor, op="sum", async_op=False):
        if self._staged(tensor):
            host = tensor.float().cpu() if tensor.dtype == torch.bfloat16 else tensor.cpu()
            self.all_reduce(host, op)
            tensor.copy_(host)
            return _Done() if async_op else None
        if op == "avg" and self.backend == "gloo":
            work = self._wrap(lambda: self._dist.all_reduce(tensor, op=self._op("sum"),
                                                            group=self._group), False)
            tensor.div_(self.world_size)
            return _Done() if async_op else work
        return self._wrap(lambda: self._dist.all_reduce(tensor, op=self._op(op),
                                                        group=self._group, async_op=async_op),
                          async_op, "all_reduce")

    def reduce_scatter(self, output, input, op="sum", async_op=False):
        if self._staged(input):
            host = input.float().cpu() if input.dtype == torch.bfloat16 else input.cpu()
            self.all_reduce(host, op)
            n = output.numel()
            output.copy_(host.reshape(-1)[self.rank * n:(self.rank + 1) * n].reshape(output.shape))
            return _Done() if async_op else None
        if self.backend == "gloo":          # gloo has no reduce_scatter: all_reduce + slice
            tmp = input.clone()
            self._wrap(lambda: self._dist.all_reduce(tmp, op=self._op(op), group=self._group),
                       False)
            n = output.numel()
            output.copy_(tmp.reshape(-1)[self.rank * n:(self.rank + 1) * n].reshape(output.shape))
            return _Done() if async_op else None
        return self._wrap(lambda: self._dist.reduce_scatter_tensor(
            output, input, op=self._op(op), group=self._group, async_op=async_op), async_op,
            "reduce_scatter")

    def all_gather(self, output, input, async_op=False):
        if self._staged(input):
            host_in = self._bytes_view(input).cpu().reshape(-1)
            host_out = torch.empty((self.world_size,) + tuple(host_in.shape), dtype=host_in.dtype)
            self._wrap(lambda: self._dist.all_gather(list(host_out.unbind(0)), host_in,
                                                     group=self._group), False)
            self._bytes_view(output).reshape(self.world_size, -1).copy_(host_out)
            return _Done() if async_op else None
        if self.backend == "gloo":
            chunks = list(output.reshape(self.world_size, -1).unbind(0))
            return self._wrap(lambda: self._dist.all_gather(
                chunks, input.reshape(-1), group=self._group, async_op=async_op), async_op,
                "all_gather")
        return self._wrap(lambda: self._dist.all_gather_into_tensor(
            output, input, group=self._group, async_op=async_op), async_op, "all_gather")

    def barrier(self):
        try:
            if self.device.type == "cuda" and self.backend == "nccl":
                self._dist.barrier(group=self._group, device_ids=[self.device.index or 0])
            else:
                self._dist.barrier(group=self._group)
        except Exception as e:  # noqa: BLE001
            raise CommError("barrier failed: %s" % e) from e

    def broadcast_object(self, obj, src=0):
        box = [obj]
        try:
            self._dist.broadcast_object_list(box, src=self._global(src), group=self._group)
        except Exception as e:  # noqa: BLE001
            raise CommError("broadcast_object failed: %s" % e) from e
        return box[0]

    def all_gather_object(self, obj):
        out = [None] * self.world_size
        try:
            self._dist.all_gather_object(out, obj, group=self._group)
        except Exception as e:  # noqa: BLE001
            raise CommError("all_gather_object failed: %s" % e) from e
        return out

    def split(self, color, key=None):
        key = self.rank if key is None else key
        table = self.all_gather_object((int(color), int(key), self._dist.get_rank()))
        mine = None
        for c in sorted({t[0] for t in table}):
            members = [g for _, _, g in sorted((k, i, g) for i, (cc, k, g) in enumerate(table)
                                               if cc == c)]
            group = self._dist.new_group(ranks=members)   # every rank creates every group
            if c == color:
                mine = group
        return TorchDistComm(device=self.device, backend=self.backend, group=mine)

    def close(self):
        if self._owns_pg and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._owns_pg = False


def make_communicator(device: Optional[torch.device] = None, backend: Optional[str] = None,
                      rank: Optional[int] = None, world_size: Optional[int] = None,
                      master_addr: Optional[str] = None, master_port: Optional[int] = None,
                      timeout_s: float = 600.0) -> Communicator:
    """The job's communicator for this process.

    ``backend`` (default: ``DTF_COMM_BACKEND``, then ``nccl`` on a GPU / ``gloo`` on a CPU) is
    ``nccl`` or ``gloo`` (``torch.distributed``) or ``rccl_native`` (``parallel/rccl_native.py``:
    own rendezvous and RCCL calls). Rank, world size and rendezvous address are taken from the
    arguments first -- ``Server`` passes what the ``ClusterSpec`` says, so a task started by
    hand as ``python script --job_name=worker --task_index=N`` (the reference's launch style,
    ``tf_distributed.py:9-18``) joins the same world as one started by a launcher -- and from
    the launcher's ``RANK`` / ``WORLD_SIZE`` / ``MASTER_*`` environment otherwise.

    A world of one gets ``SingleComm`` (no process group to set up) unless ``rccl_native`` is
    asked for by name, which then runs as a real RCCL communicator of one rank. A world of
    more than one NEVER yields a communicator of another size: the constructors below either
    rendezvous with the declared peers or raise ``CommError``."""
    backend = backend or os.environ.get("DTF_COMM_BACKEND") or None
    if backend not in (None, "nccl", "gloo", "rccl_native"):
        raise ValueError("unknown communicator backend %r" % (backend,))
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None \
        else int(world_size)
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("rank %d is not in a world of %d" % (rank, world_size))
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) \
            if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    if backend == "rccl_native":
        from .rccl_native import RcclComm
        comm: Communicator = RcclComm(rank=rank, world_size=world_size, device=device,
                                      master_addr=master_addr, master_port=master_port,
                                      timeout_s=timeout_s)
    elif world_size == 1:
        return SingleComm(device)
    else:
        try:
            comm = TorchDistComm(rank=rank, world_size=world_size, device=device,
                                 backend=backend, master_addr=master_addr,
                                 master_port=master_port, timeout_s=timeout_s)
        except CommError:
            raise
        except Exception as e:  # noqa: BLE001 -- store time-out, refused connection, ...
            raise CommError("rank %d could not join the world of %d at %s:%s: %s"
                            % (rank, world_size, master_addr, master_port, e)) from e
    if comm.world_size != world_size or comm.rank != rank:
        found = (comm.rank, comm.world_size)
        comm.close()
        raise CommError("asked for rank %d of %d, the backend formed rank %d of %d"
                        % ((rank, world_size) + found))
    return comm


# ---------------------------------------------------------------------------------------
# in-process loopback (threads)
# ---------------------------------------------------------------------------------------
class _LoopbackWorld:
    def __init__(self, world_size: int, timeout_s: float):
        self.world_size = world_size
        self.timeout_s = timeout_s
        self.cond = threading.Condition()
        self.slots: Dict[int, Dict[int, Any]] = {}
        self.generation = [0] * world_size
        self.done: Dict[int, int] = {}
        self.dead: set = set()
        self.delay_s: Dict[int, float] = {}


class LoopbackComm(Communicator):
    """Rank ``r`` of an in-process world; create all ranks with ``LoopbackComm.create_world``.

    Every collective is an exchange: each rank deposits its contribution under the sequence
    number of the call, waits until all live ranks have deposited, then computes its own
    result from the full set in rank order (so reductions are bitwise reproducible and
    identical on every rank). A rank marked dead (``kill``) or delayed (``set_delay``) makes
    its peers raise ``CommError`` after ``timeout_s`` -- the hook failure-detection tests use.
    """

    backend = "loopback"

    def __init__(self, world: _LoopbackWorld, rank: int):
        self._w = world
        self.rank = rank
        self.world_size = world.world_size
        self.device = torch.device("cpu")
        self._seq = 0

    @classmethod
    def create_world(cls, world_size: int, timeout_s: float = 30.0) -> List["LoopbackComm"]:
        world = _LoopbackWorld(world_size, timeout_s)
        return [cls(world, r) for r in range(world_size)]

    # -- fault injection -----------------------------------------------------------------
    def kill(self) -> None:
        with self._w.cond:
            self._w.dead.add(self.rank)
            self._w.cond.notify_all()

    def set_delay(self, seconds: float) -> None:
        self._w.delay_s[self.rank] = seconds

    # -- exchange primitive --------------------------------------------------------------
    def _exchange(self, value: Any) -> List[Any]:
        w = self._w
        delay = w.delay_s.get(self.rank, 0.0)
        if delay:
            time.sleep(delay)
        seq = self._seq
        self._seq += 1
        deadline = time.monotonic() + w.timeout_s
        with w.cond:
            if self.rank in w.dead:
                raise CommError("rank %d is dead" % self.rank)
            w.slots.setdefault(seq, {})[self.rank] = value
            w.cond.notify_all()
            while len(w.slots[seq]) < w.world_size:
                if w.dead:
                    raise CommError("peer rank(s) %s died during collective #%d"
                                    % (sorted(w.dead), seq))
                remaining = deadline - time.monotonic()
                if remaining <= 0:
                    missing = sorted(set(range(w.world_size)) - set(w.slots[seq]))
                    raise CommError("collective #%d timed out waiting for rank(s) %s"
                                    % (seq, missing))
                w.cond.wait(min(remaining, 0.5))
            values = [w.slots[seq][r] for r in range(w.world_size)]
            w.done[seq] = w.done.get(seq, 0) + 1
            if w.done[seq] == w.world_size:
                del w.slots[seq]
                del w.done[seq]
            return values

    @staticmethod
    def _reduce(values: Sequence[torch.Tensor], op: str) -> torch.Tensor:
        acc = values[0].clone()
        for v in values[1:]:
            if op in ("sum", "avg"):
                acc += v
            elif op == "max":
                acc = torch.maximum(acc, v)
            elif op == "min":
                acc = torch.minimum(acc, v)
            else:
                raise ValueError("unknown reduce op %r" % op)
        if op == "avg":
            acc /= len(values)
        return acc

    # -- collectives ---------------------------------------------------------------------
    def broadcast(self, tensor, src=0, async_op=False):
        values = self._exchange(tensor.detach().clone() if self.rank == src else None)
        if self.rank != src:
            tensor.copy_(values[src])
        return _Done() if async_op else None

    def all_reduce(self, tensor, op="sum", async_op=False):
        values = self._exchange(tensor.detach().clone())
        tensor.copy_(self._reduce(values, op))
        return _Done() if async_op else None

    def reduce_scatter(self, output, input, op="sum", async_op=False):
        values = self._exchange(input.detach().clone())
        total = self._reduce(values, op).reshape(-1)
        n = output.numel()
        output.copy_(total[self.rank * n:(self.rank + 1) * n].reshape(output.shape))
        return _Done() if async_op else None

    def all_gather(self, output, input, async_op=False):
        values = self._exchange(input.detach().clone())
        output.reshape(self.world_size, -1).copy_(
            torch.stack([v.reshape(-1) for v in values], dim=0))
        return _Done() if async_op else None

    def barrier(self):
        self._exchange(None)

    def broadcast_object(self, obj, src=0):
        return self._exchange(obj if self.rank == src else None)[src]

    def all_gather_object(self, obj):
        return self._exchange(obj)

    def split(self, color, key=None):
        key = self.rank if key is None else key
        table = self._exchange((int(color), int(key)))
        members = [r for _, r in sorted((k, r) for r, (c, k) in enumerate(table) if c == color)]
        # the lowest member creates the sub-world and publishes it to the others
        world = _LoopbackWorld(len(members), self._w.timeout_s) \
            if self.rank == members[0] else None
        worlds = self._exchange(world)
        return LoopbackComm(worlds[members[0]], members.index(self.rank))
