"""``tf.train.ExponentialMovingAverage``: shadow copies of variables that follow them with
``shadow -= (1 - decay) * (shadow - variable)``.

Each shadow is a non-trainable variable named ``<variable>/<name>``, colocated with its variable
(a parameter-server task that owns a variable owns its average; ``Saver`` checkpoints both) and
initialised from the same initial-value tensor, so it starts equal to the variable. The update is
three element-wise device ops per variable inside whatever step fetches it -- typically under
``tf.control_dependencies([train_op])``, so the average sees the freshly updated value and the
whole thing is part of the captured training step. With ``num_updates`` the effective decay
``min(decay, (1 + num_updates) / (10 + num_updates))`` is computed on the device from that
counter, like a learning-rate schedule.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

from ..framework import dtypes
from ..framework import ops as F
from ..framework.graph import Operation, get_default_graph
from ..framework.variables import Variable, global_variables, trainable_variables


class ExponentialMovingAverage:
    def __init__(self, decay: float, num_updates=None, zero_debias: bool = False,
                 name: str = "ExponentialMovingAverage"):
        if zero_debias:
            raise NotImplementedError("zero_debias is not provided")
        if not 0.0 <= float(decay) <= 1.0:
            raise ValueError("decay must lie in [0, 1]: %s" % decay)
        self._decay = float(decay)
        self._num_updates = num_updates
        self._name = name
        self._averages: Dict[Variable, Variable] = {}

    @property
    def name(self) -> str:
        return self._name

    def _one_minus_decay(self):
        if self._num_updates is None:
            return 1.0 - self._decay
        n = F.convert_to_tensor(self._num_updates)
        if n.dtype != dtypes.float32:
            n = F.cast(n, dtypes.float32)
        warm = F.divide(F.add(1.0, n), F.add(10.0, n))
        return F.subtract(1.0, F.minimum(self._decay, warm))

    def apply(self, var_list: Optional[Sequence[Variable]] = None) -> Operation:
        """Create the shadow variables of ``var_list`` (default: the trainable variables) and
        return the op that moves every average one step towards its variable."""
        var_list = list(trainable_variables() if var_list is None else var_list)
        g = get_default_graph()
        for var in var_list:
            if not isinstance(var, Variable):
                raise TypeError("ExponentialMovingAverage.apply takes variables, got %r" % (var,))
            if var.dtype not in (dtypes.float32, dtypes.bfloat16):
                raise TypeError("the variables must be float32 or bfloat16: %s" % var.name)
            if var in self._averages:
                raise ValueError("moving average already computed for %s" % var.name)
        for var in var_list:
            # (named after the variable from the root scope, wherever apply() is called; created
            # outside any control-dependency scope: the initialiser must not wait for a train op)
            with g.control_dependencies(None), g.colocate_with(var), g.name_scope(None):
                self._averages[var] = Variable(var.initial_value, trainable=False,
                                               name=self.average_name(var))
        with g.name_scope(self._name):
            rate = self._one_minus_decay()
            updates = []
            for var in var_list:
                shadow = self._averages[var]
                with g.colocate_with(var):
                    delta = F.multiply(F.subtract(shadow._variable, var._variable), rate)
                    updates.append(shadow.assign_sub(delta, name="update_" +
                                                     var.op.name.split("/")[-1]).op)
            return F.group(*updates, name="apply")

    def average(self, var: Variable) -> Optional[Variable]:
        return self._averages.get(var)

    def average_name(self, var: Variable) -> str:
        return var.op.name + "/" + self._name

    def variables_to_restore(self, moving_avg_variables: Optional[Sequence[Variable]] = None
                             ) -> Dict[str, Variable]:
        """``{checkpoint name: variable}`` that loads the AVERAGED values into the variables
        themselves (evaluation with the smoothed weights): averaged variables are looked up
        under their shadow's name, every other global variable under its own."""
        averaged = set(trainable_variables() if moving_avg_variables is None
                       else moving_avg_variables) | set(self._averages)
        shadows = set(self._averages.values())
        out: Dict[str, Variable] = {}
        for v in global_variables():
            if v in shadows:
                continue
            out[self.average_name(v) if v in averaged else v.op.name] = v
        return out

