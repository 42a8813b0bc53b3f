"""Chrome-trace export of per-op timings collected by
``Session.run(..., options=RunOptions(trace_level=RunOptions.FULL_TRACE), run_metadata=md)``.

Load the JSON written by ``write_chrome_trace`` in ``chrome://tracing`` / Perfetto. Device
kernels are timed by bracketing each plan step with a device synchronisation, so a traced run
is slower than a normal one; it shows *where* a step's time goes, not its steady-state rate
(use ``rocprofv3 --kernel-trace --stats`` for that).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List


def chrome_trace(step_stats: List[Dict[str, Any]]) -> Dict[str, Any]:
    events = []
    if not step_stats:
        return {"traceEvents": events}
    t0 = min(s["start"] for s in step_stats)
    for s in step_stats:
        events.append({"name": s["name"], "cat": s["op"], "ph": "X", "pid": os.getpid(),
                       "tid": s.get("device", "device"),
                       "ts": (s["start"] - t0) * 1e6, "dur": (s["end"] - s["start"]) * 1e6,
                       "args": {"op": s["op"]}})
    return {"traceEvents": events, "displayTimeUnit": "ms"}


def write_chrome_trace(path: str, step_stats: List[Dict[str, Any]]) -> None:
    with open(path, "w") as fh:
        json.dump(chrome_trace(step_stats), fh)
