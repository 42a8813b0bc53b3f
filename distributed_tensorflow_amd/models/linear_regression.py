"""Linear regression trained with plain SGD -- the smallest between-graph-replicated model.

``y = x W + b`` with a mean-squared-error loss. Under ``replica_device_setter`` the two
parameters are placed on the ps job; with the ``reduce_scatter_sync`` runtime that ownership
becomes shards of one flat arena: gradients are reduce-scattered to the owners, each owner
applies ``w -= lr * g`` to its shard and the shards are all-gathered (the broadcast of the
updated parameters), i.e. the parameter-server step of ``tf_distributed.py:33-36,75-76``
expressed as collectives.

Synthetic data: ``x ~ N(0, 1)`` and ``y = x W* + b* + noise`` with a fixed ground truth drawn
from a counter-based stream, so every worker can generate its own batches and convergence to
``W*`` is checkable.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Tuple

import numpy as np

from ..framework import dtypes
from ..framework import ops as F
from ..framework.graph import get_default_graph, name_scope, set_random_seed
from ..framework.variables import (Variable, constant_initializer, get_variable,
                                   global_variables_initializer)
from ..ops import philox
from ..train.optimizer import GradientDescentOptimizer


def build_model(features: int = 1024, outputs: int = 1, learning_rate: float = 0.05,
                seed: Optional[int] = 1, fused_loss: bool = True) -> SimpleNamespace:
    global_step = get_variable("global_step", [], initializer=constant_initializer(0),
                               trainable=False)
    with name_scope("input"):
        x = F.placeholder(dtypes.float32, shape=[None, features], name="x-input")
        y_ = F.placeholder(dtypes.float32, shape=[None, outputs], name="y-input")
    set_random_seed(seed)
    with name_scope("weights"):
        w = Variable(F.zeros([features, outputs]))
    with name_scope("biases"):
        b = Variable(F.zeros([outputs]))
    with name_scope("model"):
        y = F.add(F.matmul(x, w), b)
    with name_scope("loss"):
        # (fused: three launches forward + backward; the composition takes eight)
        loss = F.mean_squared_error(y_, y) if fused_loss else \
            F.reduce_mean(F.square(y - y_))
    with name_scope("train"):
        optimizer = GradientDescentOptimizer(learning_rate)
        train_op = optimizer.minimize(loss, global_step=global_step)
    init_op = global_variables_initializer()
    return SimpleNamespace(graph=get_default_graph(), global_step=global_step, x=x, y_=y_, w=w,
                           b=b, y=y, loss=loss, train_op=train_op, init_op=init_op,
                           optimizer=optimizer, features=features, outputs=outputs)


def ground_truth(features: int, outputs: int, seed: int = 7) -> Tuple[np.ndarray, np.ndarray]:
    """The fixed ``(W*, b*)`` every worker regresses towards."""
    w = philox.window(features, outputs, seed, 11, 0, 0, outputs, True).astype(np.float32)
    b = philox.window(1, outputs, seed, 12, 0, 0, outputs, True).astype(np.float32)[0]
    return w / np.sqrt(features), b


class SyntheticRegressionData:
    """Per-worker batch stream: worker ``task`` draws from its own counter range.

    ``to_device(device)`` moves the generator into GPU memory: a batch is then one window of a
    counter-based Philox stream generated in place (rows advance by ``batch_size`` per call),
    the targets one small fp32 product -- no host random numbers, no host-to-device copy (on the
    host, drawing 256 x 1024 normals costs ~2.5 ms per step, 25 times the training step)."""

    def __init__(self, features: int, outputs: int, task: int = 0, noise: float = 0.01,
                 seed: int = 7):
        self.features, self.outputs, self.noise = features, outputs, noise
        self.w, self.b = ground_truth(features, outputs, seed)
        self._rng = np.random.RandomState(seed * 1000 + task)
        self._seed, self._task = seed, task
        self._device = None
        self._row = 0
        self._chunk, self._chunk_pos, self._chunk_batch = None, 0, 0

    def to_device(self, device) -> "SyntheticRegressionData":
        import torch
        device = torch.device(device)
        if device.type == "cuda":
            self._device = device
            self._w_dev = torch.from_numpy(np.ascontiguousarray(self.w, dtype=np.float32)).to(device)
            self._b_dev = torch.from_numpy(np.ascontiguousarray(self.b, dtype=np.float32)).to(device)
        return self

    def next_batch(self, batch_size: int):
        if self._device is not None:
            # generated 64 batches at a time (three launches per chunk instead of per step);
            # a batch is a view of the chunk
            if self._chunk is None or self._chunk_batch != batch_size \
                    or self._chunk_pos + batch_size > self._chunk[0].shape[0]:
                from ..ops import hip
                rows = 64 * batch_size
                seed = self._seed * 1000 + self._task
                x = hip.philox_window(rows, self.features, seed, 1, self._row, 0,
                                      self.features, device=self._device)
                y = hip.gemm_f32(x, self._w_dev, bias=self._b_dev)
                if self.noise:
                    hip.axpby(1.0, y, self.noise,
                              hip.philox_window(rows, self.outputs, seed, 2, self._row, 0,
                                                self.outputs, device=self._device), out=y)
                self._row += rows
                self._chunk, self._chunk_pos, self._chunk_batch = (x, y), 0, batch_size
            lo = self._chunk_pos
            self._chunk_pos += batch_size
            return self._chunk[0][lo:lo + batch_size], self._chunk[1][lo:lo + batch_size]
        x = self._rng.standard_normal((batch_size, self.features)).astype(np.float32)
        y = x @ self.w + self.b
        if self.noise:
            y = y + self.noise * self._rng.standard_normal(y.shape).astype(np.float32)
        return x, y.astype(np.float32)
