"""Step tracing: per-op device timings and collective durations without serialising the step.

``Session.run(..., options=RunOptions(trace_level=RunOptions.FULL_TRACE), run_metadata=md)``
fills ``md.step_stats``; ``md.to_chrome_trace()`` / ``write_chrome_trace`` turn it into the JSON
that ``chrome://tracing`` / Perfetto load. The reference has no tracing at all -- its only timing
is wall-clock deltas around 100-step windows (``tf_distributed.py:94,116-122``).

How a traced run is measured (:class:`StepTracer`):

* every plan step is bracketed by two HIP events recorded on the stream it launches on
  (``hipEventRecord`` is a marker in the stream, not a synchronisation): the kernels of
  consecutive steps still run back to back, and the events give each step's time ON THE DEVICE;
* the host side of the same step (kernel selection, argument checks, the launch calls) is timed
  with ``perf_counter`` and shown on a second track, so launch-bound steps are visible as such;
* collectives issued through the runtime's communicator (RCCL all-reduce / reduce-scatter /
  all-gather / broadcast) get the same two brackets plus their payload size: the device bracket
  spans from the moment the stream reaches the call to the moment the stream may continue,
  i.e. the time the step actually waited for the transfer;
* one stream synchronisation at the END of the run resolves all events at once
  (``hipEventElapsedTime`` against a base event recorded, synchronised and time-stamped at the
  start), so device times land on the host's clock axis.

A traced run executes the plan's kernels eagerly (a captured HIP graph has no per-node events),
so the gaps between kernels are those of eager launches; the per-kernel device times are the
real ones. For kernel-level statistics over many steps use ``rocprofv3 --kernel-trace --stats``.
"""
from __future__ import annotations

import contextlib
import json
import os
import time
from typing import Any, Dict, List, Optional

import torch

_COLLECTIVES = ("broadcast", "all_reduce", "reduce_scatter", "all_gather")


class _TracedWork:
    """Handle of an asynchronous collective whose ``wait`` is recorded as its own bracket."""

    def __init__(self, work, tracer: "StepTracer", name: str, nbytes: int):
        self._work, self._tracer, self._name, self._nbytes = work, tracer, name, nbytes

    def wait(self, *args, **kwargs):
        with self._tracer.span("wait:%s" % self._name, self._name, "collective",
                               bytes=self._nbytes):
            return self._work.wait(*args, **kwargs)

    def is_completed(self):
        return self._work.is_completed()


class StepTracer:
    def __init__(self, device: torch.device):
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.records: List[Dict[str, Any]] = []
        self._open: List[Dict[str, Any]] = []
        self._base_event = None
        self._base_host = time.perf_counter()
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
            self._base_event = torch.cuda.Event(enable_timing=True)
            self._base_event.record(torch.cuda.current_stream(self.device))
            self._base_event.synchronize()
            self._base_host = time.perf_counter()

    # -- brackets ------------------------------------------------------------------------
    def begin(self, name: str, op: str, category: str = "op", **args) -> Dict[str, Any]:
        rec = {"name": name, "op": op, "cat": category, "args": args,
               "host_start": time.perf_counter()}
        if self.on_gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            rec["_ev0"] = ev
        self._open.append(rec)
        return rec

    def end(self, rec: Dict[str, Any]) -> None:
        if self.on_gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            rec["_ev1"] = ev
        rec["host_end"] = time.perf_counter()
        self._open.remove(rec)
        self.records.append(rec)

    @contextlib.contextmanager
    def span(self, name: str, op: str, category: str = "op", **args):
        rec = self.begin(name, op, category, **args)
        try:
            yield rec
        finally:
            self.end(rec)

    # -- collectives ---------------------------------------------------------------------
    @contextlib.contextmanager
    def collectives(self, comm):
        """While active, every collective issued through ``comm`` is recorded (name, bytes,
        host and device brackets). Works on any communicator: the four methods are wrapped on
        the instance and restored afterwards."""
        if comm is None or getattr(comm, "world_size", 1) < 1:
            yield
            return
        saved = {}
        tracer = self

        def wrap(method_name, fn):
            def traced(*a, **kw):
                tensors = [t for t in a if isinstance(t, torch.Tensor)]
                nbytes = max((t.numel() * t.element_size() for t in tensors), default=0)
                rec = tracer.begin("%s[%s]" % (method_name, getattr(comm, "backend", "?")),
                                   method_name, "collective", bytes=int(nbytes),
                                   world=int(getattr(comm, "world_size", 1)))
                try:
                    work = fn(*a, **kw)
                finally:
                    tracer.end(rec)
                if work is not None and hasattr(work, "wait") and kw.get("async_op"):
                    return _TracedWork(work, tracer, method_name, int(nbytes))
                return work
            return traced

        for m in _COLLECTIVES:
            fn = getattr(comm, m, None)
            if fn is not None:
                saved[m] = comm.__dict__.get(m)
                setattr(comm, m, wrap(m, fn))
        try:
            yield
        finally:
            for m, old in saved.items():
                if old is None:
                    comm.__dict__.pop(m, None)
                else:
                    setattr(comm, m, old)

    # -- results -------------------------------------------------------------------------
    def finish(self) -> List[Dict[str, Any]]:
        """Resolve all events (ONE stream synchronisation) and return the step statistics: one
        entry per bracket with ``start`` / ``end`` on the device (seconds on the host clock) and
        ``host_start`` / ``host_end`` of the enqueueing code."""
        out = []
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
        for rec in self.records:
            entry = {"name": rec["name"], "op": rec["op"], "cat": rec["cat"],
                     "args": dict(rec["args"]), "device": str(self.device),
                     "host_start": rec["host_start"], "host_end": rec["host_end"]}
            if self.on_gpu:
                entry["start"] = self._base_host + self._base_event.elapsed_time(rec["_ev0"]) * 1e-3
                entry["end"] = self._base_host + self._base_event.elapsed_time(rec["_ev1"]) * 1e-3
                entry["clock"] = "hip_event"
            else:
                entry["start"], entry["end"] = rec["host_start"], rec["host_end"]
                entry["clock"] = "host"
            out.append(entry)
        self.records = []
        return out


def chrome_trace(step_stats: List[Dict[str, Any]]) -> Dict[str, Any]:
    """Chrome trace events: per process (rank) one "device" track (HIP-event times) and one
    "host" track (enqueue times); collectives carry their payload size."""
    events: List[Dict[str, Any]] = []
    if not step_stats:
        return {"traceEvents": events}
    t0 = min(min(s["start"], s.get("host_start", s["start"])) for s in step_stats)
    pid = int(os.environ.get("RANK", os.getpid()))
    for s in step_stats:
        args = dict(s.get("args", {}), op=s["op"])
        dev_tid = "%s %s" % (s.get("device", "device"),
                             "collectives" if s.get("cat") == "collective" else "kernels")
        events.append({"name": s["name"],
                       "cat": "collective" if s.get("cat") == "collective" else s["op"],
                       "ph": "X", "pid": pid,
                       "tid": dev_tid, "ts": (s["start"] - t0) * 1e6,
                       "dur": max(0.0, (s["end"] - s["start"]) * 1e6), "args": args})
        if "host_start" in s and s.get("clock") == "hip_event":
            events.append({"name": s["name"], "cat": "enqueue", "ph": "X", "pid": pid,
                           "tid": "host enqueue", "ts": (s["host_start"] - t0) * 1e6,
                           "dur": max(0.0, (s["host_end"] - s["host_start"]) * 1e6),
                           "args": args})
    return {"traceEvents": events, "displayTimeUnit": "ms"}


def write_chrome_trace(path: str, step_stats: List[Dict[str, Any]]) -> None:
    with open(path, "w") as fh:
        json.dump(chrome_trace(step_stats), fh)


def summarize(step_stats: List[Dict[str, Any]]) -> Dict[str, Dict[str, float]]:
    """Per op type: count, total device seconds, total host-enqueue seconds, bytes moved."""
    out: Dict[str, Dict[str, float]] = {}
    for s in step_stats:
        row = out.setdefault(s["op"], {"count": 0, "device_s": 0.0, "host_s": 0.0, "bytes": 0})
        row["count"] += 1
        row["device_s"] += max(0.0, s["end"] - s["start"])
        row["host_s"] += max(0.0, s.get("host_end", s["end"]) - s.get("host_start", s["start"]))
        row["bytes"] += int(s.get("args", {}).get("bytes", 0))
    return out
