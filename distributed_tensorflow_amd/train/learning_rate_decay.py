"""Learning-rate schedules as graph tensors: ``tf.train.exponential_decay`` and friends.

The schedule is a handful of scalar ops on ``global_step`` inside the training step's own graph;
the optimizers hand the resulting device scalar to the update kernel (``csrc/kernels/optim.hip``
reads it when it runs), so a captured step replays with a changing rate and the host never
computes or uploads one. ``global_step`` is read at its value BEFORE the step's increment, as in
TensorFlow.
"""
from __future__ import annotations

import math
from typing import Optional

from ..framework import dtypes
from ..framework import ops as F
from ..framework.graph import Tensor, get_default_graph


def _step_as_float(global_step) -> Tensor:
    if global_step is None:
        raise ValueError("global_step is required for a learning-rate schedule")
    step = F.convert_to_tensor(global_step)
    return step if step.dtype == dtypes.float32 else F.cast(step, dtypes.float32)


def exponential_decay(learning_rate: float, global_step, decay_steps: float, decay_rate: float,
                      staircase: bool = False, name: Optional[str] = None) -> Tensor:
    """``learning_rate * decay_rate ** (global_step / decay_steps)`` (``staircase``: the exponent
    is floored, so the rate drops once every ``decay_steps`` steps)."""
    with get_default_graph().name_scope(name or "ExponentialDecay"):
        p = F.divide(_step_as_float(global_step), float(decay_steps))
        if staircase:
            p = F.floor(p)
        return F.multiply(float(learning_rate), F.pow(float(decay_rate), p))


def inverse_time_decay(learning_rate: float, global_step, decay_steps: float, decay_rate: float,
                       staircase: bool = False, name: Optional[str] = None) -> Tensor:
    """``learning_rate / (1 + decay_rate * global_step / decay_steps)``."""
    with get_default_graph().name_scope(name or "InverseTimeDecay"):
        p = F.divide(_step_as_float(global_step), float(decay_steps))
        if staircase:
            p = F.floor(p)
        return F.divide(float(learning_rate), F.add(1.0, F.multiply(float(decay_rate), p)))


def natural_exp_decay(learning_rate: float, global_step, decay_steps: float, decay_rate: float,
                      staircase: bool = False, name: Optional[str] = None) -> Tensor:
    """``learning_rate * exp(-decay_rate * global_step / decay_steps)``."""
    with get_default_graph().name_scope(name or "NaturalExpDecay"):
        p = F.divide(_step_as_float(global_step), float(decay_steps))
        if staircase:
            p = F.floor(p)
        return F.multiply(float(learning_rate), F.exp(F.multiply(-float(decay_rate), p)))


def cosine_decay(learning_rate: float, global_step, decay_steps: float, alpha: float = 0.0,
                 name: Optional[str] = None) -> Tensor:
    """``learning_rate * ((1 - alpha) * 0.5 * (1 + cos(pi * min(global_step, decay_steps) /
    decay_steps)) + alpha)``: half a cosine period from the full rate down to ``alpha`` of it."""
    with get_default_graph().name_scope(name or "CosineDecay"):
        frac = F.divide(F.minimum(_step_as_float(global_step), float(decay_steps)),
                        float(decay_steps))
        cosine = F.multiply(0.5, F.add(1.0, F.cos(F.multiply(math.pi, frac))))
        return F.multiply(float(learning_rate),
                          F.add(F.multiply(1.0 - float(alpha), cosine), float(alpha)))


def polynomial_decay(learning_rate: float, global_step, decay_steps: float,
                     end_learning_rate: float = 0.0001, power: float = 1.0,
                     name: Optional[str] = None) -> Tensor:
    """``(learning_rate - end) * (1 - min(global_step, decay_steps) / decay_steps) ** power +
    end``."""
    with get_default_graph().name_scope(name or "PolynomialDecay"):
        step = F.minimum(_step_as_float(global_step), float(decay_steps))
        frac = F.subtract(1.0, F.divide(step, float(decay_steps)))
        return F.add(F.multiply(float(learning_rate) - float(end_learning_rate),
                                F.pow(frac, float(power))), float(end_learning_rate))


def piecewise_constant(x, boundaries, values, name: Optional[str] = None) -> Tensor:
    """``values[0]`` while ``x <= boundaries[0]``, ``values[i]`` while ``boundaries[i-1] < x <=
    boundaries[i]``, ``values[-1]`` beyond the last boundary. Built from comparisons and a
    weighted sum, all scalar device ops."""
    boundaries, values = [float(b) for b in boundaries], [float(v) for v in values]
    if len(values) != len(boundaries) + 1:
        raise ValueError("piecewise_constant needs one more value than boundaries")
    if any(a >= b for a, b in zip(boundaries, boundaries[1:])):
        raise ValueError("boundaries must increase strictly")
    with get_default_graph().name_scope(name or "PiecewiseConstant"):
        step = _step_as_float(x)
        # below[i] = 1 while x <= boundaries[i]; it only ever switches from 1 to 0 as x grows
        below = [F.subtract(1.0, F.cast(F.greater(step, b), dtypes.float32)) for b in boundaries]
        terms = [F.multiply(values[0], below[0])]
        for i in range(1, len(boundaries)):
            terms.append(F.multiply(values[i], F.subtract(below[i], below[i - 1])))
        terms.append(F.multiply(values[-1], F.subtract(1.0, below[-1])))
        return terms[0] if len(terms) == 1 else F.add_n(terms)
