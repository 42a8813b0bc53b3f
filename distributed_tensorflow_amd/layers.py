"""``layers`` namespace: ``tf.layers.dense`` -- the reference's hand-written
``tf.add(tf.matmul(x, W), b)`` followed by an activation (``tf_distributed.py:62-65``) as one
call that creates its ``kernel`` / ``bias`` variables. The session's fusion pass sees the same
``MatMul`` / ``Add`` / activation ops and turns them into the fused dense kernel."""
from __future__ import annotations

from typing import Callable, Optional

from .framework import ops as F
from .framework.graph import Tensor, get_default_graph
from .framework.variables import get_variable, zeros_initializer


def dense(inputs, units: int, activation: Optional[Callable[[Tensor], Tensor]] = None,
          use_bias: bool = True, kernel_initializer=None, bias_initializer=None,
          trainable: bool = True, name: Optional[str] = None) -> Tensor:
    """``activation(inputs @ kernel + bias)`` for ``inputs`` of shape ``[batch, features]``;
    variables ``<name>/kernel`` (glorot-uniform unless given) and ``<name>/bias`` (zeros)."""
    inputs = F.convert_to_tensor(inputs)
    if inputs.shape is None or len(inputs.shape) != 2 or inputs.shape[-1] is None:
        raise ValueError("dense expects [batch, features] inputs with a static feature count, "
                         "got shape %s" % (inputs.shape,))
    g = get_default_graph()
    with g.name_scope(name or "dense"):
        kernel = get_variable("kernel", [int(inputs.shape[-1]), int(units)], dtype=inputs.dtype,
                              initializer=kernel_initializer, trainable=trainable)
        out = F.matmul(inputs, kernel)
        if use_bias:
            bias = get_variable("bias", [int(units)], dtype=inputs.dtype,
                                initializer=bias_initializer or zeros_initializer(),
                                trainable=trainable)
            out = F.add(out, bias)
        if activation is not None:
            out = activation(out)
    return out
