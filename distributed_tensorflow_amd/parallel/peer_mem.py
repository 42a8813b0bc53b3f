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
  into all peers themselves).
* ``reduce_scatter_rows`` (pull): a rank fetches its block of rows from every peer's buffer --
  again N-1 concurrent copies -- and sums the N blocks with the ``reduce_partials`` kernel in rank
  order (deterministic).
* ``all_reduce_rows``: the two above back to back (two-shot).

Synchronisation is deliberately coarse and uses only runtime guarantees: a transfer is complete
when the stream that carried it has been synchronised, and a rank learns that its *peers'*
transfers are complete from a barrier on the control communicator. There are no device-side
flags and no spin loops, so nothing here can hang a GPU; the price is one host synchronisation
and one barrier per collective call (per step of the sharded GEMM, not per panel: the copies of
panel p are queued behind the event of panel p's product and overlap the product of panel p+1).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .comm import Communicator


class PeerMemError(RuntimeError):
    pass


class SymmetricBuffer:
    """``local`` is this rank's allocation; ``views[r]`` aliases rank r's (``views[rank] is
    local``). Collective constructor: every rank of ``comm`` must call it with the same shape."""

    def __init__(self, comm: Communicator, shape: Sequence[int], dtype: torch.dtype,
                 device: torch.device, transport: str = "copy", blocks_per_peer: int = 8):
        if transport not in ("copy", "kernel"):
            raise ValueError("transport must be 'copy' or 'kernel'")
        self.transport = transport
        self.blocks_per_peer = int(blocks_per_peer)
        self._push_stream: Optional[torch.cuda.Stream] = None
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PeerMemError("symmetric buffers live in GPU memory")
        self.local = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self.views: List[torch.Tensor] = [self.local] * self.world
        self._streams: List[Optional[torch.cuda.Stream]] = [None] * self.world
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
        self._agree(error)
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
            self._agree(error)
            got = self.local.view(-1).view(torch.uint8)[:self.world].cpu().tolist()
            if got != list(range(self.world)):
                error = "peer copies did not land: expected %s, found %s" % (
                    list(range(self.world)), got)
            self._agree(error)

    def _agree(self, error: str) -> None:
        """Collective: raise on every rank if any rank reports an error (also a barrier)."""
        verdicts = self.comm.all_gather_object(error)
        bad = [(r, e) for r, e in enumerate(verdicts) if e]
        if bad:
            raise PeerMemError("; ".join("rank %d: %s" % (r, e) for r, e in bad))

    # -- data movement -------------------------------------------------------------------
    def push_rows(self, row0: int, rows: int, ready: Optional[torch.cuda.Event] = None) -> None:
        """Queue the copy of ``local[row0:row0+rows]`` into the same rows of every peer. The
        copies start once ``ready`` (default: everything queued so far on the current stream)
        has happened; :meth:`finish` completes them.

        ``transport = "copy"`` (default): one runtime copy per peer, each on its own stream (the
        copy engines move the data). ``transport = "kernel"``: one ``peer_push`` kernel on a side
        stream stores the rows into all peers itself."""
        if self.world == 1:
            return
        if ready is None:
            ready = torch.cuda.Event()
            ready.record()
        src = self.local[row0:row0 + rows]
        if self.transport == "kernel":
            from ..ops import hip
            st = self._push_stream
            st.wait_event(ready)
            with torch.cuda.stream(st):
                hip.peer_push(src, [self.views[r][row0:row0 + rows] for r in range(self.world)
                                    if r != self.rank], self.blocks_per_peer)
            return
        for r in range(self.world):
            if r == self.rank:
                continue
            st = self._streams[r]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                self.views[r][row0:row0 + rows].copy_(src, non_blocking=True)

    def finish(self) -> None:
        """Complete every queued push of this rank and wait until all ranks have done the same:
        afterwards ``local`` holds what the peers pushed into it."""
        if self.world == 1:
            return
        for st in self._streams:
            if st is not None:
                st.synchronize()
        if self._push_stream is not None:
            self._push_stream.synchronize()
        self.comm.barrier()

    def quiesce(self) -> None:
        """Everything this rank has queued on its current stream is complete and every rank has
        reached this point: peers may now read ``local`` in place."""
        torch.cuda.current_stream(self.device).synchronize()
        if self.world > 1:
            self.comm.barrier()


def all_gather_rows(buf: SymmetricBuffer, rows_per_rank: int) -> None:
    """Rank r owns rows [r*rows_per_rank, (r+1)*rows_per_rank) of ``buf``; afterwards every
    rank's buffer holds all of them."""
    buf.push_rows(buf.rank * rows_per_rank, rows_per_rank)
    buf.finish()


def reduce_scatter_rows(partial: SymmetricBuffer, out: torch.Tensor, row0: int, rows: int,
                        staging: Optional[torch.Tensor] = None) -> None:
    """``out = sum over ranks of partial[row0:row0+rows]`` in rank order (deterministic). The
    peers' blocks are pulled into ``staging`` ([world-1, rows, ...], allocated when not given)
    by N-1 concurrent copies and summed locally by the ``reduce_partials`` kernel. Call
    :meth:`SymmetricBuffer.quiesce` first so that every rank's partials are complete."""
    from ..ops import hip
    w, r = partial.world, partial.rank
    if w > 8:
        raise PeerMemError("at most 8 ranks per node")
    mine = partial.local[row0:row0 + rows]
    if w == 1:
        out.copy_(mine)
        return
    if staging is None:
        staging = torch.empty((w - 1,) + tuple(mine.shape), dtype=mine.dtype,
                              device=partial.device)
    parts, slot = [], 0
    done = []
    for peer in range(w):
        if peer == r:
            parts.append(mine)
            continue
        st = partial._streams[peer]
        with torch.cuda.stream(st):
            staging[slot].copy_(partial.views[peer][row0:row0 + rows], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        done.append(ev)
        parts.append(staging[slot])
        slot += 1
    for ev in done:
        torch.cuda.current_stream(partial.device).wait_event(ev)
    hip.reduce_partials(parts, out=out)


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
