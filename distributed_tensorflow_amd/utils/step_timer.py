"""Step-time reporting in the reference's stdout format.

The reference prints, every ``frequency`` = 100 local steps or at the last batch of an epoch
(``tf_distributed.py:114-123``)::

    Step: 1101,  Epoch:  2,  Batch: 100 of 550,  Cost: 3.1211,  AvgTime: 4.21ms

but always divides the elapsed time by ``frequency``, so the 50-step window at the end of
each 550-batch epoch (550 = 5*100 + 50) reports half its true per-step time. The reporter
below keeps the format and divides by the number of steps actually in the window. With a
``sink`` (``utils.metrics.JsonLinesSink``) every report is also appended as a JSON line.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional


class StepTimeReporter:
    def __init__(self, frequency: int = 100, printer: Callable[[str], None] = print,
                 clock: Callable[[], float] = time.time, rank_prefix: str = "", sink=None):
        self.frequency = frequency
        self._print = printer
        self._clock = clock
        self._prefix = rank_prefix
        self._sink = sink
        self._count = 0
        self._window_start = clock()
        self.history: List[float] = []      # ms per step of every reported window

    def start_epoch(self) -> None:
        self._count = 0

    def restart_window(self) -> None:
        self._window_start = self._clock()
        self._count = 0

    def step(self, global_step: float, epoch: int, batch_index: int, batch_count: int,
             cost: float) -> Optional[float]:
        """Record one finished step; prints and returns the window's ms/step when a report is
        due (every ``frequency`` steps, or at the epoch's last batch), else ``None``."""
        self._count += 1
        if self._count % self.frequency != 0 and batch_index + 1 != batch_count:
            return None
        now = self._clock()
        avg_ms = (now - self._window_start) * 1000.0 / self._count
        self._window_start = now
        self.history.append(avg_ms)
        self._print("%sStep: %d,  Epoch: %2d,  Batch: %3d of %3d,  Cost: %.4f,  AvgTime: %3.2fms"
                    % (self._prefix, int(global_step) + 1, epoch + 1, batch_index + 1,
                       batch_count, cost, avg_ms))
        if self._sink is not None:
            self._sink.write("step_window", global_step=int(global_step) + 1, epoch=epoch + 1,
                             batch=batch_index + 1, batch_count=batch_count, cost=float(cost),
                             steps_in_window=self._count, avg_ms_per_step=avg_ms)
        self._count = 0
        return avg_ms
