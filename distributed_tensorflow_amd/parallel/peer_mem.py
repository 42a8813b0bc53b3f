"""Peer-memory collectives for the GPUs of one node: symmetric buffers mapped through HIP IPC.

The reference moves every tensor through one parameter server over gRPC/TCP
(``tf_distributed.py:9-10,18``). On an MI355X node the eight GPUs form a fully connected xGMI
mesh -- seven point-to-point links per GPU -- and a ring collective drives one of them at a
time. The collectives here are *direct*: every rank talks to all of its peers at once.

* :class:`SymmetricBuffer` -- one device allocation per rank, the same shape everywhere, mapped
  into every process of the job (``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle`` through
  torch's CUDA-tensor sharing; one process per GPU, all GPUs visible to every process). The
  handles travel over the job's ordinary communicator (RCCL or gloo) once, at allocation.
* ``all_gather_rows`` (push): a rank copies its block of rows into the same rows of every peer's
  buffer, one copy per peer on that peer's own stream -- N-1 concurrent transfers on N-1 links
  (or, with ``transport="kernel"``, one ``peer_push`` kernel whose workgroups store the rows
  into all peers themselves; or, with ``transport="epilogue"``, no copy at all: the kernel that
  PRODUCES the rows stores every result vector into the peers' buffers as well as its own --
  ``mirror_table`` gives it the addresses, ``begin_epilogue_push`` / ``publish`` / ``finish``
  order the exchange on the producing stream).
* ``reduce_scatter_rows`` (pull): a rank sums its block of rows over all ranks' buffers with the
  ``reduce_partials`` kernel reading the peers' rows IN PLACE through the mapped views, in rank
  order (deterministic) -- the rows cross xGMI once, straight into the reduction.
* ``all_reduce_rows``: the two above back to back (two-shot).

Completion (``sync``):

* ``"flags"`` (default when the shared flag page can be set up): device-side. A pusher writes a
  sequence number for each peer behind its copies (:class:`PeerFlags`: one page of fine-grained
  device memory per rank, mapped into every peer through HIP IPC; system-scope release store
  from a one-lane kernel on the pushing stream);
  the consumer's stream waits for all peers' numbers with a one-wave kernel. No host
  synchronisation and no control-plane barrier per step, so pipelined steps never drain the
  compute stream. Overwriting is ordered the same way: a rank acknowledges what it has finished
  reading, and a pusher's streams wait for the acknowledgements of the previous exchange before
  they touch a peer's rows again. Every wait is bounded (``DTF_PEER_TIMEOUT_S``, default 10 s):
  a peer that never signals leaves a mark in a status word and :meth:`SymmetricBuffer.check`
  (called by ``finish`` and by ``ShardedMatmul.wait``) raises :class:`CommError`.
* ``"host"``: a transfer is complete when the stream that carried it has been synchronised, and
  a rank learns that its peers' transfers are complete from a barrier on the control
  communicator. One host synchronisation and one barrier per collective call.
"""
from __future__ import annotations

import os
import threading
from typing import List, Optional, Sequence

import torch

from .comm import CommError, Communicator


class PeerMemError(RuntimeError):
    pass


KIND_PUSHED, KIND_READY, KIND_ACK, KIND_READ_ACK = 0, 1, 2, 3
_FLAG_BYTES = 64


def _agree(comm: Communicator, error: str) -> None:
    """Collective: raise on every rank if any rank reports an error (also a barrier)."""
    verdicts = comm.all_gather_object(error)
    bad = [(r, e) for r, e in enumerate(verdicts) if e]
    if bad:
        raise PeerMemError("; ".join("rank %d: %s" % (r, e) for r, e in bad))


class PeerFlags:
    """Sequence-number flags for the ranks of ``comm`` (one node). Every rank owns one page of
    fine-grained device memory, mapped into all peers through HIP IPC; flag ``(kind, src)`` of
    rank ``dst``'s page is written by ``src`` and polled by ``dst``. Collective constructor;
    raises :class:`PeerMemError` on every rank if any rank cannot set its part up."""

    def __init__(self, comm: Communicator, device: torch.device, kinds: int = 4,
                 timeout_s: Optional[float] = None):
        from ..ops import hip
        self._hip = hip
        self.comm, self.rank, self.world = comm, comm.rank, comm.world_size
        self.kinds = kinds
        self.timeout_s = float(os.environ.get("DTF_PEER_TIMEOUT_S", "10")) \
            if timeout_s is None else float(timeout_s)
        self.nbytes = max(4096, kinds * self.world * _FLAG_BYTES)
        self._pages: List[Optional[int]] = [None] * self.world     # address of rank r's page here
        error, handle = "", None
        try:
            self._pages[self.rank], handle = hip.flags_alloc(self.nbytes)
        except Exception as e:  # noqa: BLE001
            error = "allocating the flag page failed: %s" % e
        table = comm.all_gather_object(handle)
        if not error and all(h is not None for h in table):
            try:
                for r, h in enumerate(table):
                    if r != self.rank:
                        self._pages[r] = hip.flags_open(h)
            except Exception as e:  # noqa: BLE001
                error = "mapping a peer's flag page failed: %s" % e
        elif not error:
            error = "a peer could not allocate its flag page"
        try:
            _agree(comm, error)
        except PeerMemError:
            self.close()
            raise
        self.status = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_np = self.status.numpy()

    def _flag(self, page_rank: int, kind: int, src: int) -> int:
        return self._pages[page_rank] + (kind * self.world + src) * _FLAG_BYTES

    def signal(self, kind: int, peers: Sequence[int], value: int) -> None:
        """Current stream: tell ``peers`` that this rank's exchange ``value`` of ``kind`` is done."""
        if peers:
            self._hip.peer_signal([self._flag(p, kind, self.rank) for p in peers], value)

    def wait(self, kind: int, peers: Sequence[int], value: int) -> None:
        """Current stream: wait until every rank in ``peers`` has signalled ``value``."""
        if peers:
            self._hip.peer_wait([self._flag(self.rank, kind, p) for p in peers], value,
                                self.timeout_s, self.status)

    def check(self, what: str = "peer-memory exchange") -> None:
        """Raise :class:`CommError` if a wait kernel that has run so far gave up."""
        code = int(self._status_np[0])
        if code:
            raise CommError("%s: a peer did not signal within %.1f s (wait slot %d on rank %d)"
                            % (what, self.timeout_s, code - 1, self.rank))

    def close(self) -> None:
        for r, page in enumerate(self._pages):
            if page is not None:
                self._hip.flags_close(page, r == self.rank)
        self._pages = [None] * self.world


class SymmetricBuffer:
    """``local`` is this rank's allocation; ``views[r]`` aliases rank r's (``views[rank] is
    local``). Collective constructor: every rank of ``comm`` must call it with the same shape."""

    def __init__(self, comm: Communicator, shape: Sequence[int], dtype: torch.dtype,
                 device: torch.device, transport: str = "copy", blocks_per_peer: int = 8,
                 sync: Optional[str] = None):
        if transport not in ("copy", "kernel", "epilogue"):
            raise ValueError("transport must be 'copy', 'kernel' or 'epilogue'")
        sync = sync or os.environ.get("DTF_PEER_SYNC") or "flags"
        if sync not in ("flags", "host"):
            raise ValueError("sync must be 'flags' or 'host'")
        self.transport = transport
        self.blocks_per_peer = int(blocks_per_peer)
        self._push_stream: Optional[torch.cuda.Stream] = None
        self._tables: dict = {}       # (row0, rows) -> device table of the peers' row addresses
        self.tracer = None            # utils.tracing.StepTracer while a traced step runs
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PeerMemError("symmetric buffers live in GPU memory")
        self.local = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self.views: List[torch.Tensor] = [self.local] * self.world
        self._streams: List[Optional[torch.cuda.Stream]] = [None] * self.world
        self.flags: Optional[PeerFlags] = None
        self.sync = "host"
        self._seq = 0                 # push exchanges published by this rank
        self._acked = 0               # push exchanges of the peers this rank has acknowledged
        self._ready_seq = 0           # ready marks set by this rank (partials readable in place)
        self._read_acked = 0          # ready marks of the peers this rank has finished reading
        self._pushing = False         # pushes queued since the last publish
        self._peers = [r for r in range(self.world) if r != self.rank]
        self.host_syncs = 0           # host-side stream synchronisations made by finish/quiesce
        if self.world == 1:
            return
        # Every step below is collective and ends in an exchange of verdicts, so that a failure
        # on one rank (no IPC support, no peer access) makes ALL ranks raise instead of leaving
        # the others waiting in a barrier.
        from torch.multiprocessing.reductions import reduce_tensor
        error = ""
        try:
            packed = reduce_tensor(self.local)
        except Exception as e:  # noqa: BLE001
            packed, error = None, "exporting the IPC handle failed: %s" % e
        table = comm.all_gather_object(packed)
        views: List[torch.Tensor] = []
        if not error and all(t is not None for t in table):
            try:
                for r, (rebuild, args) in enumerate(table):
                    v = self.local if r == self.rank else rebuild(*args)
                    if tuple(v.shape) != tuple(self.local.shape) or v.dtype != dtype:
                        raise PeerMemError("rank %d allocated %s %s" % (r, tuple(v.shape), v.dtype))
                    views.append(v)
            except Exception as e:  # noqa: BLE001
                error = "mapping peer memory failed: %s" % e
        elif not error:
            error = "a peer could not export its IPC handle"
        _agree(comm, error)
        self.views = views
        # one copy stream per peer so the N-1 transfers of a push run concurrently
        self._streams = [None if r == self.rank else torch.cuda.Stream(device=self.device)
                         for r in range(self.world)]
        if transport == "kernel":
            self._push_stream = torch.cuda.Stream(device=self.device)
        # prove the copy path before anything depends on it: every rank writes its rank number
        # into byte `rank` of every peer's buffer (the buffer holds no data yet)
        nbytes = self.local.numel() * self.local.element_size()
        if nbytes >= self.world:
            try:
                stamp = torch.full((1,), self.rank, dtype=torch.uint8, device=self.device)
                for r, v in enumerate(self.views):
                    v.view(-1).view(torch.uint8)[self.rank:self.rank + 1].copy_(stamp)
                torch.cuda.synchronize(self.device)
            except Exception as e:  # noqa: BLE001
                error = "peer copy failed: %s" % e
            _agree(comm, error)
            got = self.local.view(-1).view(torch.uint8)[:self.world].cpu().tolist()
            if got != list(range(self.world)):
                error = "peer copies did not land: expected %s, found %s" % (
                    list(range(self.world)), got)
            _agree(comm, error)
        if sync == "flags":
            # the flag page is optional equipment: if any rank cannot map it, all fall back to
            # host synchronisation + barrier
            try:
                self.flags = PeerFlags(comm, self.device)
                self.sync = "flags"
            except PeerMemError:
                self.flags = None

    # -- data movement -------------------------------------------------------------------
    def _ordered_streams(self):
        if self.transport == "epilogue":       # the producing (current) stream carries the rows
            return [(None, torch.cuda.current_stream(self.device))]
        if self.transport == "kernel":
            return [(None, self._push_stream)]
        return [(r, self._streams[r]) for r in self._peers]

    def push_rows(self, row0: int, rows: int, ready: Optional[torch.cuda.Event] = None) -> None:
        """Queue the copy of ``local[row0:row0+rows]`` into the same rows of every peer. The
        copies start once ``ready`` (default: everything queued so far on the current stream)
        has happened; :meth:`finish` completes them.

        ``transport = "copy"`` (default): one runtime copy per peer, each on its own stream (the
        copy engines move the data). ``transport = "kernel"``: one ``peer_push`` kernel on a side
        stream stores the rows into all peers itself. Under ``sync="flags"`` the first push of an
        exchange makes each pushing stream wait until its peer has acknowledged the previous
        exchange (the rows it is about to overwrite are no longer being read)."""
        if self.world == 1:
            return
        if self.transport == "epilogue":
            raise PeerMemError("transport='epilogue': the producing kernel stores the rows "
                               "(mirror_table / begin_epilogue_push), nothing is left to push")
        if ready is None:
            ready = torch.cuda.Event()
            ready.record()
        src = self.local[row0:row0 + rows]
        first = not self._pushing
        self._pushing = True
        if self.transport == "kernel":
            from ..ops import hip
            st = self._push_stream
            st.wait_event(ready)
            with torch.cuda.stream(st):
                if first and self.flags is not None and self._seq > 0:
                    self.flags.wait(KIND_ACK, self._peers, self._seq)
                hip.peer_push(src, [self.views[r][row0:row0 + rows] for r in range(self.world)
                                    if r != self.rank], self.blocks_per_peer)
            return
        tracer = self.tracer
        for r in self._peers:
            st = self._streams[r]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                if first and self.flags is not None and self._seq > 0:
                    self.flags.wait(KIND_ACK, [r], self._seq)
                # (a traced step: the copy's bracket is recorded on ITS stream, so the trace
                # shows it next to the products running on the compute stream meanwhile)
                rec = tracer.begin("peer_push[rows %d+%d -> rank %d]" % (row0, rows, r),
                                   "peer_push", "collective",
                                   bytes=int(src.numel() * src.element_size())) \
                    if tracer is not None else None
                try:
                    self.views[r][row0:row0 + rows].copy_(src, non_blocking=True)
                except RuntimeError as e:      # the peer's allocation is gone (it has exited)
                    raise CommError("push to rank %d failed: %s" % (r, str(e)[:200])) from e
                finally:
                    if rec is not None:
                        tracer.end(rec)

    def mirror_table(self, row0: int, rows: int) -> torch.Tensor:
        """``transport="epilogue"``: int64 device tensor with the addresses of rows
        ``row0 .. row0+rows`` in every PEER's buffer (rank order, this rank left out) -- what a
        producing kernel takes as its list of further destinations. Cached per row range."""
        key = (int(row0), int(rows))
        table = self._tables.get(key)
        if table is None:
            addrs = [self.views[r][row0:row0 + rows].data_ptr() for r in self._peers]
            table = self._tables[key] = torch.tensor(addrs, dtype=torch.int64, device=self.device)
        return table

    def begin_epilogue_push(self) -> None:
        """``transport="epilogue"``, current stream, before a kernel that stores into the peers:
        the first one of an exchange waits until every peer has acknowledged the previous
        exchange (the rows about to be overwritten are no longer being read)."""
        if self.world == 1:
            return
        first = not self._pushing
        self._pushing = True
        if first and self.flags is not None and self._seq > 0:
            self.flags.wait(KIND_ACK, self._peers, self._seq)

    def publish(self) -> int:
        """Close the current exchange of pushes: behind its copies every pushing stream tells its
        peer the new sequence number. Returns that number (``sync="flags"`` only)."""
        self._seq += 1
        self._pushing = False
        for r, st in self._ordered_streams():
            with torch.cuda.stream(st):
                self.flags.signal(KIND_PUSHED, self._peers if r is None else [r], self._seq)
        return self._seq

    def finish(self, release: bool = True) -> None:
        """Complete the current exchange: afterwards (in stream order under ``sync="flags"``, on
        the host otherwise) ``local`` holds what the peers pushed into it and this rank's own
        pushes have left its rows. ``release``: also acknowledge the received rows right away
        (peers may overwrite them in their next exchange); pass False and call :meth:`release`
        later when kernels queued after ``finish`` still read them."""
        if self.world == 1:
            return
        if self.flags is None:
            for st in self._streams:
                if st is not None:
                    st.synchronize()
            if self._push_stream is not None:
                self._push_stream.synchronize()
            if self.transport == "epilogue":
                torch.cuda.current_stream(self.device).synchronize()
            self.host_syncs += 1
            self._pushing = False
            self.comm.barrier()
            return
        self.flags.check()
        seq = self.publish() if self._pushing else self._seq   # (already published: pipelined)
        cur = torch.cuda.current_stream(self.device)
        for _, st in self._ordered_streams():       # own rows may be rewritten after the pushes
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
        self.flags.wait(KIND_PUSHED, self._peers, seq)
        if release:
            self.release()

    def release(self) -> None:
        """Current stream: everything queued so far has finished reading the rows the peers
        pushed (or, for partials read in place, the peers' rows): they may be overwritten."""
        if self.flags is None:
            return
        if self._acked < self._seq:
            self._acked = self._seq
            self.flags.signal(KIND_ACK, self._peers, self._acked)
        if self._read_acked < self._ready_seq:
            self._read_acked = self._ready_seq
            self.flags.signal(KIND_READ_ACK, self._peers, self._read_acked)

    def mark_ready(self) -> int:
        """Pull mode: everything queued so far on the current stream has produced ``local``;
        peers may read it in place once they have waited for the returned number."""
        self._ready_seq += 1
        self.flags.signal(KIND_READY, self._peers, self._ready_seq)
        return self._ready_seq

    def quiesce(self) -> None:
        """Everything this rank has queued on its current stream is complete and every rank has
        reached this point: peers may now read ``local`` in place. Device-side under
        ``sync="flags"`` (ready marks + a wait kernel), host synchronisation + barrier
        otherwise."""
        if self.world == 1:
            return
        if self.flags is None:
            torch.cuda.current_stream(self.device).synchronize()
            self.host_syncs += 1
            self.comm.barrier()
            return
        self.flags.check()
        seq = self.mark_ready()
        self.flags.wait(KIND_READY, self._peers, seq)

    def before_overwrite(self) -> None:
        """Pull mode, current stream: wait until every peer has acknowledged reading the last
        ready-marked contents of ``local`` (call before the kernel that rewrites it)."""
        if self.flags is not None and self._ready_seq > 0:
            self.flags.wait(KIND_READ_ACK, self._peers, self._ready_seq)

    def check(self) -> None:
        if self.flags is not None:
            self.flags.check()


def all_gather_rows(buf: SymmetricBuffer, rows_per_rank: int) -> None:
    """Rank r owns rows [r*rows_per_rank, (r+1)*rows_per_rank) of ``buf``; afterwards every
    rank's buffer holds all of them."""
    buf.push_rows(buf.rank * rows_per_rank, rows_per_rank)
    buf.finish()


def reduce_scatter_rows(partial: SymmetricBuffer, out: torch.Tensor, row0: int,
                        rows: int) -> None:
    """``out = sum over ranks of partial[row0:row0+rows]`` in rank order (deterministic): the
    ``reduce_partials`` kernel reads the peers' rows in place through the mapped views, so they
    cross xGMI once and are never staged in local memory. Call
    :meth:`SymmetricBuffer.quiesce` first so that every rank's partials are complete; under
    ``sync="flags"`` the kernel's completion is acknowledged to the peers (they may rewrite
    their partials)."""
    from ..ops import hip
    w = partial.world
    if w > 8:
        raise PeerMemError("at most 8 ranks per node")
    if w == 1:
        out.copy_(partial.local[row0:row0 + rows])
        return
    parts = [partial.views[peer][row0:row0 + rows] for peer in range(w)]
    hip.reduce_partials(parts, out=out)
    partial.release()


def all_reduce_rows(partial: SymmetricBuffer, result: SymmetricBuffer) -> None:
    """Two-shot all-reduce of ``partial`` (rows split evenly over ranks) into ``result`` on
    every rank. Both buffers have the same shape; rows must divide by the world size."""
    w, r = partial.world, partial.rank
    total = partial.local.shape[0]
    if total % w:
        raise PeerMemError("rows (%d) must divide by the world size (%d)" % (total, w))
    per = total // w
    partial.quiesce()
    reduce_scatter_rows(partial, result.local[r * per:(r + 1) * per], r * per, per)
    result.push_rows(r * per, per)
    result.finish()
