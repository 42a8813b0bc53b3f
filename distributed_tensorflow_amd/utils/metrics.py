"""JSON-lines metrics sink: one self-describing JSON object per line, appended and flushed as
it is produced, so a run that is killed still leaves every record written so far.

The reference reports only through ``print`` (``tf_distributed.py:118-128``); its stdout format
is kept by ``utils.step_timer.StepTimeReporter``, which can mirror every report into a sink of
this kind (``--metrics_path`` of the examples), and the ``bench/`` scripts write their results
the same way.
"""
from __future__ import annotations

import json
import os
import threading
import time
from typing import Any, Dict, Iterator, Optional


class JsonLinesSink:
    def __init__(self, path: str, common: Optional[Dict[str, Any]] = None,
                 clock=time.time):
        """``common`` fields (job name, task index, ...) are merged into every record."""
        directory = os.path.dirname(os.path.abspath(path))
        os.makedirs(directory, exist_ok=True)
        self.path = path
        self._common = dict(common or {})
        self._clock = clock
        self._lock = threading.Lock()
        self._fh = open(path, "a", buffering=1)       # line buffered

    def write(self, kind: str, **fields: Any) -> Dict[str, Any]:
        record = {"kind": kind, "wall_time": self._clock()}
        record.update(self._common)
        record.update(fields)
        line = json.dumps(record, sort_keys=True, default=_plain)
        with self._lock:
            if self._fh is None:
                raise ValueError("write to a closed metrics sink")
            self._fh.write(line + "\n")
        return record

    def close(self) -> None:
        with self._lock:
            if self._fh is not None:
                self._fh.close()
                self._fh = None

    def __enter__(self) -> "JsonLinesSink":
        return self

    def __exit__(self, *exc) -> None:
        self.close()


def _plain(value: Any) -> Any:
    """numpy / torch scalars and arrays as plain JSON values."""
    if hasattr(value, "item") and getattr(value, "ndim", 1) == 0:
        return value.item()
    if hasattr(value, "tolist"):
        return value.tolist()
    return str(value)


def read_jsonl(path: str) -> Iterator[Dict[str, Any]]:
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line:
                yield json.loads(line)
