"""Optimizers: ``GradientDescentOptimizer(lr).minimize(loss, global_step=...)``.

Reference usage: ``tf_distributed.py:75-76``. ``minimize`` = ``compute_gradients``
(reverse-mode autodiff) + ``apply_gradients`` (one ``ApplyGradientDescent`` per variable,
colocated with the variable so the update runs on the parameter owner, then
``global_step += 1`` under a control dependency on every update).

How the gradient reaches the variable depends on the session's strategy: applied as-is by
the owner as it arrives (asynchronous parameter server -- the reference's semantics), or
summed across workers by an RCCL all-reduce / reduce-scatter first (synchronous). The graph
built here is the same in every case.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from ..framework import ops as F
from ..framework.gradients import gradients
from ..framework.graph import GraphKeys, Operation, Tensor, get_default_graph
from ..framework.variables import Variable, trainable_variables


class Optimizer:
    GATE_NONE, GATE_OP, GATE_GRAPH = 0, 1, 2

    def __init__(self, use_locking: bool = False, name: str = "Optimizer"):
        self._use_locking = use_locking
        self._name = name

    def compute_gradients(self, loss: Tensor, var_list: Optional[Sequence[Variable]] = None
                          ) -> List[Tuple[Optional[Tensor], Variable]]:
        if var_list is None:
            var_list = trainable_variables()
        var_list = list(var_list)
        if not var_list:
            raise ValueError("No variables to optimize.")
        grads = gradients(loss, [v._variable for v in var_list])
        return list(zip(grads, var_list))

    def apply_gradients(self, grads_and_vars: Sequence[Tuple[Optional[Tensor], Variable]],
                        global_step: Optional[Variable] = None,
                        name: Optional[str] = None) -> Operation:
        grads_and_vars = [(g, v) for g, v in grads_and_vars if g is not None]
        if not grads_and_vars:
            raise ValueError("No gradients provided for any variable.")
        g = get_default_graph()
        with g.name_scope(name or self._name) as scope:
            updates = [self._apply_dense(grad, var) for grad, var in grads_and_vars]
            if global_step is None:
                train_op = F.group(*updates, name="update")
            else:
                with g.control_dependencies(updates):
                    one = F.fill(global_step.shape or (), 1, dtype=global_step.dtype,
                                 name="step_increment")
                    train_op = g.create_op(
                        "AssignAdd", [global_step._as_graph_element(), one],
                        [(global_step.dtype, global_step.shape)], name="global_step_add")
        g.add_to_collection(GraphKeys.TRAIN_OP, train_op)
        return train_op

    def minimize(self, loss: Tensor, global_step: Optional[Variable] = None,
                 var_list: Optional[Sequence[Variable]] = None,
                 name: Optional[str] = None) -> Operation:
        grads_and_vars = self.compute_gradients(loss, var_list)
        missing = [v.name for g, v in grads_and_vars if g is None]
        if len(missing) == len(grads_and_vars):
            raise ValueError("No gradients provided for any variable: %s" % missing)
        return self.apply_gradients(grads_and_vars, global_step=global_step, name=name)

    def _apply_dense(self, grad: Tensor, var: Variable) -> Operation:
        raise NotImplementedError


class GradientDescentOptimizer(Optimizer):
    def __init__(self, learning_rate: float, use_locking: bool = False,
                 name: str = "GradientDescent"):
        super().__init__(use_locking, name)
        self._learning_rate = float(learning_rate)

    def _apply_dense(self, grad: Tensor, var: Variable) -> Operation:
        g = var.graph
        return g.create_op("ApplyGradientDescent", [var._variable, grad],
                           [(var.dtype, var.shape)], name="update_" + var.op.name.split("/")[-1],
                           attrs={"learning_rate": self._learning_rate,
                                  "use_locking": self._use_locking})
