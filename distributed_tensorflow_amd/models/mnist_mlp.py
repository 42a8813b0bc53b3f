"""The reference's MNIST model: 784 -> 100 (sigmoid) -> 10 (softmax) MLP trained with plain SGD.

Graph structure, names, initialisers and hyper-parameters follow ``tf_distributed.py:39-89``
(and the identical ``tf_distributed_worker.py:38-88``): a float32 scalar ``global_step``
variable, placeholders ``input/x-input`` [None,784] and ``input/y-input`` [None,10], graph seed
1, ``weights/Variable`` [784,100] and ``weights/Variable_1`` [100,10] ~ N(0,1),
``biases/Variable`` [100] and ``biases/Variable_1`` [10] zeros, the hand-written
``reduce_mean(-reduce_sum(y_ * log(y), axis=[1]))`` loss, ``GradientDescentOptimizer(0.0005)``,
the argmax accuracy, and ``cost`` / ``accuracy`` scalar summaries.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

from .. import summary
from ..framework import dtypes
from ..framework import ops as F
from ..framework.graph import get_default_graph, name_scope, set_random_seed
from ..framework.variables import (Variable, constant_initializer, get_variable,
                                   global_variables_initializer)
from ..train.optimizer import GradientDescentOptimizer
from ..train.training_util import get_or_create_global_step

INPUT_DIM, HIDDEN_DIM, NUM_CLASSES = 784, 100, 10
BATCH_SIZE = 100
LEARNING_RATE = 0.0005
NUM_PARAMS = INPUT_DIM * HIDDEN_DIM + HIDDEN_DIM + HIDDEN_DIM * NUM_CLASSES + NUM_CLASSES  # 79 510


def build_model(learning_rate: float = LEARNING_RATE, seed: Optional[int] = 1,
                hidden: int = HIDDEN_DIM, with_summaries: bool = True,
                optimizer=None) -> SimpleNamespace:
    """Build the model into the default graph (inside whatever device scope is active) and
    return its handles. ``optimizer``: a ``tf.train.Optimizer`` to train with instead of the
    reference's ``GradientDescentOptimizer(learning_rate)``."""
    # (a caller that built a learning-rate schedule has created the counter already)
    global_step = get_or_create_global_step()
    with name_scope("input"):
        x = F.placeholder(dtypes.float32, shape=[None, INPUT_DIM], name="x-input")
        y_ = F.placeholder(dtypes.float32, shape=[None, NUM_CLASSES], name="y-input")
    set_random_seed(seed)
    with name_scope("weights"):
        w1 = Variable(F.random_normal([INPUT_DIM, hidden]))
        w2 = Variable(F.random_normal([hidden, NUM_CLASSES]))
    with name_scope("biases"):
        b1 = Variable(F.zeros([hidden]))
        b2 = Variable(F.zeros([NUM_CLASSES]))
    with name_scope("softmax"):
        z2 = F.add(F.matmul(x, w1), b1)
        a2 = F.sigmoid(z2)
        z3 = F.add(F.matmul(a2, w2), b2)
        y = F.softmax(z3)
    with name_scope("cross_entropy"):
        loss = F.reduce_mean(-F.reduce_sum(y_ * F.log(y), axis=[1]))
    with name_scope("train"):
        if optimizer is None:
            optimizer = GradientDescentOptimizer(learning_rate)
        train_op = optimizer.minimize(loss, global_step=global_step)
    with name_scope("Accuracy"):
        correct = F.equal(F.argmax(y, 1), F.argmax(y_, 1))
        accuracy = F.reduce_mean(F.cast(correct, dtypes.float32))
    summary_op = None
    if with_summaries:
        summary.scalar("cost", loss)
        summary.scalar("accuracy", accuracy)
        summary_op = summary.merge_all()
    init_op = global_variables_initializer()
    return SimpleNamespace(graph=get_default_graph(), global_step=global_step, x=x, y_=y_,
                           w1=w1, w2=w2, b1=b1, b2=b2, logits=z3, y=y, loss=loss,
                           train_op=train_op, accuracy=accuracy, summary_op=summary_op,
                           init_op=init_op, optimizer=optimizer)
