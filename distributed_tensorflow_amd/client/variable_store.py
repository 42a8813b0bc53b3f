"""Where variable values live.

In the reference every variable lives in the resource manager of the single ``/job:ps`` task
and is read / updated over gRPC each step (``tf_distributed.py:34-36,111``). Here a *store*
abstracts the owner of variable state:

* ``LocalVariableStore`` -- values are torch tensors in this process (HBM when the session
  device is a GPU). Used for single-process sessions, for the replicated / owner-sharded
  collective strategies, and *inside* a parameter-server task.
* ``parallel.ps_transport.RemoteVariableStore`` -- values live in ps tasks reached over TCP
  (the CPU plumbing configuration with asynchronous updates).

Stores are keyed by the variable op's name (``weights/Variable``), which is also the
checkpoint key.
"""
from __future__ import annotations

import threading
from typing import Dict, Iterable, List, Optional

import torch


class FailedPreconditionError(RuntimeError):
    """Raised when an uninitialized variable is read."""


class LocalVariableStore:
    def __init__(self, device: torch.device = torch.device("cpu")):
        self.device = torch.device(device)
        self._values: Dict[str, torch.Tensor] = {}
        self._lock = threading.Lock()
        self.update_count = 0     # number of apply_sgd calls served (async-PS accounting)
        # bumped whenever a variable gets NEW storage (first value, reshaped value, adopted
        # arena view, clear): HIP-graph captures hold raw addresses and are dropped on a change
        self.generation = 0
        # per variable: how often its VALUE was written through this interface (kernels that
        # update storage in place through ``local_tensor`` do not count: callers that cache
        # something derived from a variable only do so for variables no such kernel touches)
        self._versions: Dict[str, int] = {}
        self._imaged: set = set()     # variables something captured holds a derived image of

    # -- reads ---------------------------------------------------------------------------
    def is_initialized(self, name: str) -> bool:
        return name in self._values

    def read(self, name: str) -> torch.Tensor:
        try:
            return self._values[name]
        except KeyError:
            raise FailedPreconditionError(
                "Attempting to use uninitialized value %s" % name)

    def local_tensor(self, name: str) -> Optional[torch.Tensor]:
        """The live storage of ``name`` (for kernels that update it in place), or ``None``."""
        return self._values.get(name)

    def names(self) -> List[str]:
        return sorted(self._values)

    def version(self, name: str) -> int:
        return self._versions.get(name, 0)

    def _touch(self, name: str) -> None:
        self._versions[name] = self._versions.get(name, 0) + 1
        if name in self._imaged:
            # a captured step may hold an image derived from this variable (the K-contiguous
            # copy of a right-hand matmul operand): captures are dropped like after a move
            self.generation += 1

    def watch_writes(self, name: str) -> None:
        """From now on a write to ``name`` also invalidates captured steps (``generation``)."""
        self._imaged.add(name)

    def uninitialized(self, names: Iterable[str]) -> List[str]:
        return [n for n in names if n not in self._values]

    # -- writes --------------------------------------------------------------------------
    def assign(self, name: str, value: torch.Tensor) -> torch.Tensor:
        value = value.detach()
        with self._lock:
            self._touch(name)
            cur = self._values.get(name)
            if cur is not None and cur.shape == value.shape and cur.dtype == value.dtype:
                cur.copy_(value)             # keep the address stable (captured graphs, views)
            else:
                self._values[name] = value.to(self.device, copy=True).contiguous()
                self.generation += 1         # new storage: captured steps point at the old one
            return self._values[name]

    def adopt(self, name: str, storage: torch.Tensor) -> None:
        """Install ``storage`` (e.g. a view into a flat parameter arena) as the variable."""
        with self._lock:
            self._touch(name)
            self._values[name] = storage
            self.generation += 1

    def assign_add(self, name: str, delta: torch.Tensor) -> torch.Tensor:
        var = self.read(name)
        self._touch(name)
        var.add_(delta.to(var.device, var.dtype) if isinstance(delta, torch.Tensor) else delta)
        return var

    def apply_sgd(self, name: str, lr: float, grad: torch.Tensor) -> torch.Tensor:
        """``var -= lr * grad`` in place, without a lock across variables: concurrent callers
        interleave freely, which is the asynchronous update rule of the reference's plain
        ``GradientDescentOptimizer`` (``use_locking=False``)."""
        var = self.read(name)
        self._touch(name)
        var.add_(grad.to(var.device, var.dtype), alpha=-float(lr))
        self.update_count += 1
        return var

    def clear(self) -> None:
        with self._lock:
            for name in self._values:
                self._touch(name)
            self._values.clear()
            self.generation += 1

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._values)
