"""RCCL communicator without ``torch.distributed``: own rendezvous, own ``ncclUniqueId``
exchange, collectives issued on raw device pointers and HIP streams.

``RcclComm`` implements the ``Communicator`` interface on two native pieces of
``libdtf_host.so``:

* ``csrc/runtime/rendezvous.cc`` -- the control plane. Rank 0 listens on
  ``MASTER_ADDR:MASTER_PORT``, the others connect; all-gather / broadcast / barrier of byte
  strings with bounded waits. Python objects travel pickled (between the ranks of one job only).
* ``csrc/runtime/rccl_comm.cc`` -- the data plane. ``librccl`` is opened at run time; rank 0
  creates the ``ncclUniqueId``, the rendezvous broadcasts it, every rank calls
  ``ncclCommInitRank`` on its own GPU.

Collectives run on a communication stream of the communicator's own, the way the process-group
backend does it: the stream first waits for everything enqueued so far on the caller's stream,
the collective is enqueued there, and either the caller's stream waits for it right away
(``async_op=False``) or the returned handle's ``wait()`` makes it do so -- compute enqueued in
between overlaps the transfer. All of this is stream-ordered (events, no host synchronisation),
so a step with its collectives can be captured into a HIP graph.

Every collective in flight is bounded by a watchdog (``_Watchdog``): a host thread that follows
the completion events of the communication stream and the communicator's asynchronous error
state. A collective that has not completed ``collective_timeout_s`` after it was enqueued (a peer
that died or never arrived), or an error RCCL reports on its own, makes the watchdog abort the
communicator (``ncclCommAbort``: kernels of it that spin on a peer are released, so a host
thread blocked in a stream synchronisation behind them returns) and park a :class:`CommError`
that every later call on the communicator -- collectives, ``wait()``, ``barrier()``,
``check_async_error()`` -- raises. ``Supervisor.run_with_recovery`` catches that exception; the
process is never replaced.

Selected with ``DTF_COMM_BACKEND=rccl_native`` (``make_communicator``, and through it
``tf.train.Server(protocol="rccl")``); the default backend stays ``torch.distributed``'s ``nccl``
(the same RCCL underneath).
"""
from __future__ import annotations

import collections
import ctypes
import os
import pickle
import threading
import time
from typing import Any, List, Optional

import torch

from ..utils import native
from .comm import CommError, Communicator, _Done

_DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6,
           torch.float32: 7, torch.float64: 8, torch.bfloat16: 9}
_OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3, "avg": 4}
_RDZV_CODES = {-1: "bad argument", -2: "connection lost", -3: "timed out",
               -4: "refused: another process already holds this rank, or the rank is not in "
                   "the world rank 0 was started with"}


def available() -> bool:
    lib = native.load()
    return bool(lib is not None and hasattr(lib, "dtf_rccl_available")
                and lib.dtf_rccl_available())


class Rendezvous:
    """Byte-string collectives of a job's ranks over TCP (star through rank 0)."""

    def __init__(self, rank: int, world_size: int, addr: str = "127.0.0.1", port: int = 29500,
                 timeout_s: float = 600.0):
        lib = native.load()
        if lib is None or not hasattr(lib, "dtf_rdzv_create"):
            raise CommError("the native host library (libdtf_host.so) is not built")
        self._lib = lib
        self.rank, self.world_size = int(rank), int(world_size)
        err = ctypes.c_int(0)
        self._h = lib.dtf_rdzv_create(addr.encode(), int(port), self.rank, self.world_size,
                                      float(timeout_s), ctypes.byref(err))
        if not self._h:
            raise CommError("rendezvous at %s:%d failed for rank %d of %d (code %d: %s)"
                            % (addr, port, rank, world_size, err.value,
                               _RDZV_CODES.get(err.value, "error")))

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise CommError("rendezvous %s failed (code %d: %s)"
                            % (what, rc, _RDZV_CODES.get(rc, "error")))

    def all_gather_bytes(self, mine: bytes) -> List[bytes]:
        """Every rank's byte string, in rank order (lengths may differ)."""
        w = self.world_size
        sizes = ctypes.create_string_buffer(8 * w)
        self._check(self._lib.dtf_rdzv_all_gather(self._h, len(mine).to_bytes(8, "little"), 8,
                                                  sizes), "all_gather")
        lens = [int.from_bytes(sizes.raw[8 * r:8 * r + 8], "little") for r in range(w)]
        n = max(lens)
        table = ctypes.create_string_buffer(max(1, n * w))
        self._check(self._lib.dtf_rdzv_all_gather(self._h, mine.ljust(n, b"\0"), n, table),
                    "all_gather")
        return [table.raw[r * n:r * n + lens[r]] for r in range(w)]

    def broadcast_bytes(self, data: Optional[bytes], src: int = 0) -> bytes:
        size = ctypes.create_string_buffer(
            len(data).to_bytes(8, "little") if self.rank == src else b"\0" * 8, 8)
        self._check(self._lib.dtf_rdzv_broadcast(self._h, size, 8, int(src)), "broadcast")
        n = int.from_bytes(size.raw, "little")
        buf = ctypes.create_string_buffer(data if self.rank == src else b"", max(1, n))
        self._check(self._lib.dtf_rdzv_broadcast(self._h, buf, n, int(src)), "broadcast")
        return buf.raw[:n]

    def barrier(self) -> None:
        self._check(self._lib.dtf_rdzv_barrier(self._h), "barrier")

    def close(self) -> None:
        if self._h:
            self._lib.dtf_rdzv_destroy(self._h)
            self._h = None


class _Watchdog(threading.Thread):
    """Bounds the collectives in flight of one communicator.

    ``watch(event, what)`` registers the completion event of a collective (anything with a
    ``query()`` that turns true when the work is done). The thread drops completed entries,
    asks RCCL for an asynchronous error while anything is in flight, and fails the communicator
    (``RcclComm._fail``) when the oldest entry is older than ``timeout_s`` or RCCL reports an
    error. It sleeps on a condition while nothing is in flight."""

    def __init__(self, comm: "RcclComm", timeout_s: float, poll_s: float = 0.02):
        super().__init__(name="dtf-rccl-watchdog", daemon=True)
        self._comm_ref = comm
        self.timeout_s = float(timeout_s)
        self._poll_s = float(poll_s)
        self._cond = threading.Condition()
        self._inflight: collections.deque = collections.deque()
        self._stop_flag = False

    def watch(self, event, what: str) -> None:
        with self._cond:
            self._inflight.append((event, what, time.monotonic()))
            self._cond.notify()

    def in_flight(self) -> int:
        with self._cond:
            return len(self._inflight)

    def stop(self) -> None:
        with self._cond:
            self._stop_flag = True
            self._inflight.clear()
            self._cond.notify()

    def run(self) -> None:
        while True:
            with self._cond:
                while not self._inflight and not self._stop_flag:
                    self._cond.wait()
                if self._stop_flag:
                    return
                try:
                    while self._inflight and self._inflight[0][0].query():
                        self._inflight.popleft()
                except Exception as e:  # noqa: BLE001 -- a faulted device fails the query itself
                    self._inflight.clear()
                    failure = "completion query failed: %s" % (str(e)[:200],)
                else:
                    failure = None
                    if self._inflight:
                        _, what, t0 = self._inflight[0]
                        age = time.monotonic() - t0
                        if age > self.timeout_s:
                            failure = "%s did not complete within %.1f s (%d collective(s) in " \
                                "flight): a peer is gone or never arrived" \
                                % (what, self.timeout_s, len(self._inflight))
                            self._inflight.clear()
                busy = bool(self._inflight)
            comm = self._comm_ref
            if failure is None and busy:
                failure = comm._async_error_text()
                if failure is not None:
                    with self._cond:
                        self._inflight.clear()
            if failure is not None:
                comm._fail(failure)
            time.sleep(self._poll_s)


class _StreamWork:
    """Handle of a collective in flight on the communication stream."""

    def __init__(self, comm: "RcclComm", event, what: str):
        self._comm, self._event, self._what = comm, event, what

    def wait(self, timeout: Optional[float] = None) -> bool:
        """Without ``timeout``: the caller's stream waits for the collective (stream-ordered, the
        host does not block; the watchdog bounds it). With ``timeout`` (seconds): the HOST waits
        until the collective has completed, and a collective still running at the deadline fails
        the communicator and raises :class:`CommError`."""
        self._comm.check_async_error(self._what)
        if timeout is None:
            torch.cuda.current_stream(self._comm.device).wait_event(self._event)
            return True
        deadline = time.monotonic() + float(timeout)
        while not self._event.query():
            self._comm.check_async_error(self._what)
            if time.monotonic() >= deadline:
                self._comm._fail("%s did not complete within %.1f s" % (self._what, timeout))
                self._comm.check_async_error(self._what)
            time.sleep(0.0005)
        torch.cuda.current_stream(self._comm.device).wait_event(self._event)
        return True

    def is_completed(self) -> bool:
        self._comm.check_async_error(self._what)
        return bool(self._event.query())


class RcclComm(Communicator):
    backend = "rccl_native"
    inplace_all_gather = True

    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None,
                 device: Optional[torch.device] = None, master_addr: Optional[str] = None,
                 master_port: Optional[int] = None, timeout_s: float = 600.0,
                 collective_timeout_s: Optional[float] = None,
                 _parent: Optional["RcclComm"] = None, _handle=None, _members=None):
        lib = native.load()
        if not available():
            raise CommError("native RCCL is not available (libdtf_host.so not built, or librccl "
                            "cannot be opened)")
        self._lib = lib
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise CommError("RcclComm needs a GPU device")
        torch.cuda.set_device(self.device)
        self._stream = torch.cuda.Stream(device=self.device)
        self._lock = threading.RLock()               # enqueue vs abort
        self._failed: Optional[CommError] = None
        if collective_timeout_s is None:
            collective_timeout_s = _parent._watchdog.timeout_s if _parent is not None else \
                float(os.environ.get("DTF_COLLECTIVE_TIMEOUT", "300"))
        self._watchdog = _Watchdog(self, collective_timeout_s)
        self._watchdog.start()
        if _parent is not None:                      # a sub-communicator made by split()
            self._rdzv, self._owns_rdzv = _parent._rdzv, False
            self._members = list(_members)           # parent-world ranks, in group order
            self.rank, self.world_size = self._members.index(_parent._world_rank), len(_members)
            self._world_rank = _parent._world_rank
            self._comm = _handle
            return
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None \
            else int(world_size)
        self._world_rank = self.rank
        self._members = list(range(self.world_size))
        addr = master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = master_port or int(os.environ.get("MASTER_PORT", "29500"))
        self._rdzv = Rendezvous(self.rank, self.world_size, addr, port, timeout_s)
        self._owns_rdzv = True
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            self._rc(lib.dtf_rccl_unique_id(uid), "ncclGetUniqueId")
        uid = ctypes.create_string_buffer(self._rdzv.broadcast_bytes(uid.raw, 0), 128)
        handle = ctypes.c_void_p()
        self._rc(lib.dtf_rccl_init(ctypes.byref(handle), self.world_size, uid, self.rank),
                 "ncclCommInitRank")
        self._comm = handle

    # -- plumbing ------------------------------------------------------------------------
    def _rc(self, code: int, what: str) -> None:
        if code != 0:
            raise CommError("%s failed: %s (code %d)"
                            % (what, self._lib.dtf_rccl_error_string(code).decode(), code))

    def _async_error_text(self) -> Optional[str]:
        """RCCL's asynchronous error of this communicator as text, ``None`` when there is none."""
        with self._lock:
            if not self._comm:
                return None
            err = ctypes.c_int(0)
            self._lib.dtf_rccl_async_error(self._comm, ctypes.byref(err))
            if err.value == 0:
                return None
            return "RCCL reported an asynchronous error: %s" \
                % self._lib.dtf_rccl_error_string(err.value).decode()

    def _fail(self, why: str) -> None:
        """Abort the communicator and park the error for every later call (idempotent; called
        by the watchdog thread or by a ``wait(timeout)`` that ran out)."""
        with self._lock:
            if self._failed is not None:
                return
            self._failed = CommError("rank %d of %d: %s; the communicator has been aborted"
                                     % (self.rank, self.world_size, why))
            handle, self._comm = self._comm, None
            if handle:
                self._lib.dtf_rccl_destroy(handle, 1)        # ncclCommAbort
        self._watchdog.stop()

    def check_async_error(self, what: str = "collective") -> None:
        """Raise the parked failure, or an asynchronous error RCCL reports right now."""
        if self._failed is not None:
            raise self._failed
        text = self._async_error_text()
        if text is not None:
            self._fail("%s: %s" % (what, text))
            raise self._failed

    def watch(self, event, what: str = "captured step") -> None:
        """Put work that contains collectives of this communicator under the watchdog: ``event``
        completes (``query()``) when that work has. The session records one behind every replay
        of a HIP graph that holds captured collectives -- events recorded DURING capture are
        graph nodes, not queryable handles, so ``_run`` cannot register those itself."""
        if self._failed is not None:
            raise self._failed
        self._watchdog.watch(event, what)

    @staticmethod
    def _wire(t: torch.Tensor) -> torch.Tensor:
        """1-byte float types travel as uint8 (pure data movement)."""
        return t.view(torch.uint8) if t.element_size() == 1 and t.is_floating_point() else t

    def _dtype(self, t: torch.Tensor, reducing: bool) -> int:
        code = _DTYPES.get(t.dtype)
        if code is None or (reducing and t.dtype == torch.uint8 and False):
            raise CommError("RcclComm: dtype %s is not supported" % t.dtype)
        return code

    def _need(self, *tensors: torch.Tensor) -> None:
        for t in tensors:
            if not t.is_cuda or not t.is_contiguous():
                raise CommError("RcclComm: tensors must be contiguous GPU tensors")

    def _run(self, what: str, async_op: bool, tensors, launch):
        """Fork the communication stream off the caller's, enqueue, join (now or on wait())."""
        if self._failed is not None:
            raise self._failed
        cur = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        self._stream.wait_event(ready)
        for t in tensors:
            t.record_stream(self._stream)
        with self._lock:
            if self._failed is not None:
                raise self._failed
            self._rc(launch(self._stream.cuda_stream), what)
        done = torch.cuda.Event()
        done.record(self._stream)
        if not torch.cuda.is_current_stream_capturing():
            self._watchdog.watch(done, what)
        if async_op:
            return _StreamWork(self, done, what)
        cur.wait_event(done)
        return None

    # -- collectives -----------------------------------------------------------------------
    def broadcast(self, tensor, src=0, async_op=False):
        tensor = self._wire(tensor)
        self._need(tensor)
        return self._run("broadcast", async_op, (tensor,), lambda s: self._lib.dtf_rccl_broadcast(
            self._comm, tensor.data_ptr(), tensor.numel(), self._dtype(tensor, False), int(src), s))

    def all_reduce(self, tensor, op="sum", async_op=False):
        self._need(tensor)
        return self._run("all_reduce", async_op, (tensor,), lambda s: self._lib.dtf_rccl_all_reduce(
            self._comm, tensor.data_ptr(), tensor.data_ptr(), tensor.numel(),
            self._dtype(tensor, True), _OPS[op], s))

    def reduce_scatter(self, output, input, op="sum", async_op=False):
        self._need(output, input)
        if input.numel() != output.numel() * self.world_size or input.dtype != output.dtype:
            raise CommError("reduce_scatter: input must hold world_size outputs of the same dtype")
        return self._run("reduce_scatter", async_op, (output, input),
                         lambda s: self._lib.dtf_rccl_reduce_scatter(
                             self._comm, input.data_ptr(), output.data_ptr(), output.numel(),
                             self._dtype(input, True), _OPS[op], s))

    def all_gather(self, output, input, async_op=False):
        output, input = self._wire(output), self._wire(input)
        self._need(output, input)
        if output.numel() != input.numel() * self.world_size or input.dtype != output.dtype:
            raise CommError("all_gather: output must hold world_size inputs of the same dtype")
        return self._run("all_gather", async_op, (output, input),
                         lambda s: self._lib.dtf_rccl_all_gather(
                             self._comm, input.data_ptr(), output.data_ptr(), input.numel(),
                             self._dtype(input, False), s))

    def barrier(self, timeout: Optional[float] = None) -> None:
        """All ranks have reached this point. The host waits (bounded by ``timeout``, default
        the watchdog's limit) for a one-word all-reduce."""
        flag = torch.zeros(1, dtype=torch.float32, device=self.device)
        work = self.all_reduce(flag, async_op=True)
        work.wait(self._watchdog.timeout_s if timeout is None else timeout)
        self.check_async_error("barrier")

    # -- host objects (control plane; whole-world communicators only) --------------------------
    def _whole_world(self, what: str) -> None:
        if not self._owns_rdzv:
            raise CommError("%s is only available on the job's whole communicator" % what)

    def broadcast_object(self, obj: Any, src: int = 0) -> Any:
        self._whole_world("broadcast_object")
        data = self._rdzv.broadcast_bytes(pickle.dumps(obj) if self.rank == src else None, src)
        return pickle.loads(data)

    def all_gather_object(self, obj: Any) -> List[Any]:
        self._whole_world("all_gather_object")
        return [pickle.loads(b) for b in self._rdzv.all_gather_bytes(pickle.dumps(obj))]

    def split(self, color: int, key: Optional[int] = None) -> "RcclComm":
        """Collective over the whole communicator: ranks of the same ``color`` form a new RCCL
        communicator, ordered by ``key``."""
        self._whole_world("split")
        key = self.rank if key is None else int(key)
        table = self.all_gather_object((int(color), key, self.rank))
        members = [r for _, _, r in sorted((k, i, r) for i, (c, k, r) in enumerate(table)
                                           if c == color)]
        handle = ctypes.c_void_p()
        if self._lib.dtf_rccl_has_split():
            self._rc(self._lib.dtf_rccl_split(self._comm, int(color), key, ctypes.byref(handle)),
                     "ncclCommSplit")
        else:
            # no ncclCommSplit in this RCCL: every group's first member makes a unique id, the
            # whole world exchanges them, each rank initialises into its group's
            uid = ctypes.create_string_buffer(128)
            if members[0] == self.rank:
                self._rc(self._lib.dtf_rccl_unique_id(uid), "ncclGetUniqueId")
            ids = self._rdzv.all_gather_bytes(uid.raw)
            self._rc(self._lib.dtf_rccl_init(ctypes.byref(handle), len(members),
                                             ctypes.create_string_buffer(ids[members[0]], 128),
                                             members.index(self.rank)), "ncclCommInitRank")
        return RcclComm(device=self.device, _parent=self, _handle=handle, _members=members)

    def close(self) -> None:
        if getattr(self, "_watchdog", None) is not None:
            self._watchdog.stop()
        if getattr(self, "_comm", None):
            try:
                torch.cuda.synchronize(self.device)
            except Exception:  # noqa: BLE001 -- a faulted device must not keep the process alive
                pass
            with self._lock:
                handle, self._comm = self._comm, None
                if handle:
                    self._lib.dtf_rccl_destroy(handle, 0)
        if getattr(self, "_owns_rdzv", False) and self._rdzv is not None:
            self._rdzv.close()
            self._rdzv = None
