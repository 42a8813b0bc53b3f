"""Optimizers: ``GradientDescentOptimizer(lr).minimize(loss, global_step=...)`` and the
slot-carrying rules of the same family (``MomentumOptimizer``, ``AdagradOptimizer``,
``RMSPropOptimizer``, ``AdamOptimizer``).

Reference usage: ``tf_distributed.py:75-76``. ``minimize`` = ``compute_gradients``
(reverse-mode autodiff) + ``apply_gradients`` (one ``ApplyGradientDescent`` per variable,
colocated with the variable so the update runs on the parameter owner, then
``global_step += 1`` under a control dependency on every update).

How the gradient reaches the variable depends on the session's strategy: applied as-is by
the owner as it arrives (asynchronous parameter server -- the reference's semantics), or
summed across workers by an RCCL all-reduce / reduce-scatter first (synchronous). The graph
built here is the same in every case.

Rules with per-variable state keep it in *slot variables* (non-trainable, colocated with their
variable, named ``<variable>/<optimizer name>`` -- so they are placed on the parameter owner,
initialised by ``global_variables_initializer`` and saved by ``Saver`` like any variable). One
``ApplyOptimizer`` op per variable takes ``[variable, slots..., (running powers), gradient]``;
the session fuses the ops of a step into one multi-tensor launch
(``csrc/kernels/optim.hip``).
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
        self._create_slots([v for _, v in grads_and_vars])
        with g.name_scope(name or self._name) as scope:
            updates = [self._apply_dense(grad, var) for grad, var in grads_and_vars]
            updates = updates + self._finish(updates)
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

    # -- per-variable state ------------------------------------------------------------------
    def _create_slots(self, var_list: Sequence[Variable]) -> None:
        """Create the rule's state variables (called once per ``apply_gradients``, before the
        update ops are built)."""

    def _finish(self, updates: List[Operation]) -> List[Operation]:
        """Ops that run after every variable's update (e.g. advancing Adam's running powers)."""
        return []

    def _slot_dict(self):
        if not hasattr(self, "_slots"):
            self._slots = {}
        return self._slots

    def _zeros_slot(self, var: Variable, slot_name: str, op_name: str,
                    value: float = 0.0) -> Variable:
        slots = self._slot_dict().setdefault(slot_name, {})
        if var not in slots:
            g = var.graph
            # (named after the variable from the root scope, wherever minimize() is called)
            with g.colocate_with(var), g.name_scope(None):
                slots[var] = Variable(
                    F.fill(tuple(var.shape), value, dtype=var.dtype), trainable=False,
                    name=var.op.name + "/" + op_name)
            # per-element state of `var`: a runtime that gives every parameter shard one owner
            # (reduce_scatter_sync) keeps only the owner's part of it
            slots[var].op.attrs["slot_of"] = var.op.name
            slots[var].op.attrs["slot_key"] = op_name
        return slots[var]

    def get_slot(self, var: Variable, name: str) -> Optional[Variable]:
        return self._slot_dict().get(name, {}).get(var)

    def get_slot_names(self) -> List[str]:
        return sorted(self._slot_dict())

    @staticmethod
    def _split_lr(learning_rate):
        """``(constant, tensor)``: a Python number is a constant baked into the update; a
        tensor or variable (a schedule such as ``exponential_decay``) is read on the device by
        the update kernel every step."""
        if isinstance(learning_rate, (int, float)):
            return float(learning_rate), None
        return 0.0, F.convert_to_tensor(learning_rate, dtype=None)

    def _apply_rule(self, rule: str, hyper: Sequence[float], var: Variable,
                    refs: Sequence[Variable], grad: Tensor,
                    lr_tensor: Optional[Tensor] = None) -> Operation:
        g = var.graph
        ref_tensors = [var._variable] + [r._variable for r in refs]
        values = ([lr_tensor] if lr_tensor is not None else []) + [grad]
        return g.create_op("ApplyOptimizer", ref_tensors + values, [(var.dtype, var.shape)],
                           name="update_" + var.op.name.split("/")[-1],
                           attrs={"rule": rule, "hyper": tuple(float(h) for h in hyper),
                                  "num_refs": len(ref_tensors),
                                  "lr_input": lr_tensor is not None,
                                  "use_locking": self._use_locking})


class GradientDescentOptimizer(Optimizer):
    def __init__(self, learning_rate: float, use_locking: bool = False,
                 name: str = "GradientDescent"):
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)

    def _apply_dense(self, grad: Tensor, var: Variable) -> Operation:
        g = var.graph
        if self._lr_tensor is not None:
            return self._apply_rule("sgd", (0.0,), var, [], grad, self._lr_tensor)
        return g.create_op("ApplyGradientDescent", [var._variable, grad],
                           [(var.dtype, var.shape)], name="update_" + var.op.name.split("/")[-1],
                           attrs={"learning_rate": self._learning_rate,
                                  "use_locking": self._use_locking})


class MomentumOptimizer(Optimizer):
    """``accum = momentum * accum + grad; var -= lr * accum`` (``use_nesterov``:
    ``var -= lr * (grad + momentum * accum)``)."""

    def __init__(self, learning_rate: float, momentum: float, use_locking: bool = False,
                 name: str = "Momentum", use_nesterov: bool = False):
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)
        self._momentum = float(momentum)
        self._use_nesterov = bool(use_nesterov)

    def _create_slots(self, var_list):
        for v in var_list:
            self._zeros_slot(v, "momentum", self._name)

    def _apply_dense(self, grad, var):
        return self._apply_rule("nesterov" if self._use_nesterov else "momentum",
                                (self._learning_rate, self._momentum), var,
                                [self.get_slot(var, "momentum")], grad, self._lr_tensor)


class AdagradOptimizer(Optimizer):
    """``accum += grad**2; var -= lr * grad / sqrt(accum)``."""

    def __init__(self, learning_rate: float, initial_accumulator_value: float = 0.1,
                 use_locking: bool = False, name: str = "Adagrad"):
        if initial_accumulator_value <= 0.0:
            raise ValueError("initial_accumulator_value must be positive: %s"
                             % initial_accumulator_value)
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)
        self._initial_accumulator_value = float(initial_accumulator_value)

    def _create_slots(self, var_list):
        for v in var_list:
            self._zeros_slot(v, "accumulator", self._name, self._initial_accumulator_value)

    def _apply_dense(self, grad, var):
        return self._apply_rule("adagrad", (self._learning_rate,), var,
                                [self.get_slot(var, "accumulator")], grad, self._lr_tensor)


class RMSPropOptimizer(Optimizer):
    """``ms = decay * ms + (1 - decay) * grad**2; mom = momentum * mom + lr * grad /
    sqrt(ms + epsilon); var -= mom`` (``ms`` starts at one, as in TensorFlow)."""

    def __init__(self, learning_rate: float, decay: float = 0.9, momentum: float = 0.0,
                 epsilon: float = 1e-10, use_locking: bool = False, centered: bool = False,
                 name: str = "RMSProp"):
        if centered:
            raise NotImplementedError("centered RMSProp is not provided")
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)
        self._decay = float(decay)
        self._momentum, self._epsilon = float(momentum), float(epsilon)

    def _create_slots(self, var_list):
        for v in var_list:
            self._zeros_slot(v, "rms", self._name, 1.0)
            self._zeros_slot(v, "momentum", self._name + "_1")

    def _apply_dense(self, grad, var):
        return self._apply_rule(
            "rmsprop", (self._learning_rate, self._decay, self._momentum, self._epsilon), var,
            [self.get_slot(var, "rms"), self.get_slot(var, "momentum")], grad, self._lr_tensor)


class AdadeltaOptimizer(Optimizer):
    """``accum = rho * accum + (1 - rho) * grad**2; update = sqrt(accum_update + epsilon) /
    sqrt(accum + epsilon) * grad; accum_update = rho * accum_update + (1 - rho) * update**2;
    var -= lr * update``."""

    def __init__(self, learning_rate: float = 0.001, rho: float = 0.95, epsilon: float = 1e-8,
                 use_locking: bool = False, name: str = "Adadelta"):
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)
        self._rho, self._epsilon = float(rho), float(epsilon)

    def _create_slots(self, var_list):
        for v in var_list:
            self._zeros_slot(v, "accum", self._name)
            self._zeros_slot(v, "accum_update", self._name + "_1")

    def _apply_dense(self, grad, var):
        return self._apply_rule(
            "adadelta", (self._learning_rate, self._rho, self._epsilon), var,
            [self.get_slot(var, "accum"), self.get_slot(var, "accum_update")], grad,
            self._lr_tensor)


class AdamOptimizer(Optimizer):
    """``m += (1 - beta1) * (grad - m); v += (1 - beta2) * (grad**2 - v);
    var -= lr * sqrt(1 - beta2**t) / (1 - beta1**t) * m / (sqrt(v) + epsilon)``. ``beta1**t``
    and ``beta2**t`` are two scalar variables (``beta1_power``, ``beta2_power``) colocated with
    the first variable and advanced once per step after every update."""

    def __init__(self, learning_rate: float = 0.001, beta1: float = 0.9, beta2: float = 0.999,
                 epsilon: float = 1e-8, use_locking: bool = False, name: str = "Adam"):
        super().__init__(use_locking, name)
        self._learning_rate, self._lr_tensor = self._split_lr(learning_rate)
        self._epsilon = float(epsilon)
        self._beta1, self._beta2 = float(beta1), float(beta2)
        self._powers: Optional[Tuple[Variable, Variable]] = None

    def _create_slots(self, var_list):
        if self._powers is None:
            first = min(var_list, key=lambda v: v.name)
            with first.graph.colocate_with(first), first.graph.name_scope(None):
                self._powers = (
                    Variable(F.fill((), self._beta1, dtype=first.dtype), trainable=False,
                             name="beta1_power"),
                    Variable(F.fill((), self._beta2, dtype=first.dtype), trainable=False,
                             name="beta2_power"))
        for v in var_list:
            self._zeros_slot(v, "m", self._name)
            self._zeros_slot(v, "v", self._name + "_1")

    def _apply_dense(self, grad, var):
        return self._apply_rule(
            "adam", (self._learning_rate, self._beta1, self._beta2, self._epsilon), var,
            [self.get_slot(var, "m"), self.get_slot(var, "v"), *self._powers], grad,
            self._lr_tensor)

    def _finish(self, updates):
        g = get_default_graph()
        out = []
        with g.control_dependencies(updates):
            for power, beta in zip(self._powers, (self._beta1, self._beta2)):
                out.append(g.create_op(
                    "AffineUpdateVariable", [power._variable], [(power.dtype, power.shape)],
                    name="advance_" + power.op.name.split("/")[-1],
                    attrs={"factor": beta, "increment": 0.0}))
        return out


class SyncReplicasOptimizer(Optimizer):
    """Synchronous between-graph training: every step applies the MEAN of the replicas'
    gradients once, and ``global_step`` advances by one per step.

    TensorFlow implements this on top of the asynchronous parameter server with gradient
    accumulators and a token queue. Here the synchronous strategies (``allreduce_sync``,
    ``reduce_scatter_sync``: servers started with ``protocol="rccl"``) already aggregate in
    lock-step with one collective per step, so the wrapper only has to select averaging and to
    refuse configurations it cannot honour: the hook from ``make_session_run_hook`` (pass it to
    ``MonitoredTrainingSession(hooks=...)``, or call ``configure_session(sess)`` under a
    ``Supervisor``) checks that the session runs a collective strategy over exactly
    ``total_num_replicas`` ranks. Backup replicas (``replicas_to_aggregate <
    total_num_replicas``) would need the stragglers' gradients dropped; a collective waits for
    every rank, so that is rejected."""

    def __init__(self, opt: Optimizer, replicas_to_aggregate: int,
                 total_num_replicas: Optional[int] = None, use_locking: bool = False,
                 name: str = "sync_replicas"):
        super().__init__(use_locking, name)
        if total_num_replicas is None:
            total_num_replicas = replicas_to_aggregate
        if replicas_to_aggregate != total_num_replicas:
            raise ValueError("replicas_to_aggregate=%d != total_num_replicas=%d: backup replicas "
                             "are not available with collective aggregation"
                             % (replicas_to_aggregate, total_num_replicas))
        self._opt = opt
        self._total_num_replicas = int(total_num_replicas)

    def compute_gradients(self, *args, **kwargs):
        return self._opt.compute_gradients(*args, **kwargs)

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        if global_step is None:
            raise ValueError("global_step is required for SyncReplicasOptimizer")
        return self._opt.apply_gradients(grads_and_vars, global_step=global_step, name=name)

    def get_slot(self, *args, **kwargs):
        return self._opt.get_slot(*args, **kwargs)

    def get_slot_names(self, *args, **kwargs):
        return self._opt.get_slot_names(*args, **kwargs)

    def configure_session(self, session) -> None:
        """Check that ``session`` can aggregate synchronously over ``total_num_replicas`` ranks
        and select mean aggregation in its runtime."""
        rt = getattr(session, "_runtime", None)
        comm = getattr(rt, "comm", None)
        world = comm.world_size if comm is not None else 1
        strategy = getattr(rt, "strategy", None)
        synchronous = strategy in ("allreduce_sync", "reduce_scatter_sync")
        if world != self._total_num_replicas:
            raise ValueError("SyncReplicasOptimizer(total_num_replicas=%d) in a job of %d "
                             "replica(s)" % (self._total_num_replicas, world))
        if self._total_num_replicas > 1 and not synchronous:
            raise ValueError("SyncReplicasOptimizer needs a synchronous strategy (servers with "
                             "protocol='rccl'); this session runs %r" % (strategy,))
        if synchronous:
            rt.grad_aggregation = "mean"

    def make_session_run_hook(self, is_chief: bool = True, num_tokens: int = -1):
        from .monitored_session import SessionRunHook
        outer = self

        class _SyncReplicasHook(SessionRunHook):
            def after_create_session(self, session, coord=None):
                outer.configure_session(session)

        return _SyncReplicasHook()
