"""``losses`` namespace: ``tf.losses.mean_squared_error`` (the regression loss of
``examples/linear_regression.py``) and ``tf.losses.softmax_cross_entropy`` /
``sparse_softmax_cross_entropy`` as batch means of the fused cross-entropy ops,
``sigmoid_cross_entropy`` and ``absolute_difference`` as means over all elements."""
from __future__ import annotations

from .framework import ops as F
from .framework.graph import Tensor
from .framework.ops import mean_squared_error  # noqa: F401


def softmax_cross_entropy(onehot_labels, logits, name=None) -> Tensor:
    return F.reduce_mean(F.softmax_cross_entropy_with_logits(labels=onehot_labels, logits=logits),
                         name=name)


def sparse_softmax_cross_entropy(labels, logits, name=None) -> Tensor:
    return F.reduce_mean(F.sparse_softmax_cross_entropy_with_logits(labels=labels, logits=logits),
                         name=name)


def sigmoid_cross_entropy(multi_class_labels, logits, name=None) -> Tensor:
    return F.reduce_mean(F.sigmoid_cross_entropy_with_logits(labels=multi_class_labels,
                                                             logits=logits), name=name)


def absolute_difference(labels, predictions, name=None) -> Tensor:
    return F.reduce_mean(F.abs(F.subtract(predictions, labels)), name=name)
