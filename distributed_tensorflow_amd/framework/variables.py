"""Variables: persistent tensors owned by a parameter store and shared between steps.

Covers ``tf.Variable(initial_value)`` (reference ``tf_distributed.py:51-57``),
``tf.get_variable(name, shape, initializer=, trainable=)`` with ``constant_initializer``
(``tf_distributed.py:39``), ``tf.global_variables_initializer()`` (``:89``) and the readiness
probe the Supervisor polls (``report_uninitialized_variables``).

A variable is one ``VariableV2`` op -- the op type ``replica_device_setter`` keys on to place
it on a ``/job:ps`` task -- plus an ``Assign`` initializer that is colocated with it. The
*value* lives outside the graph, in the session's ``VariableStore`` (local memory, a remote
parameter-server task, or a rank-sharded flat HBM buffer, depending on the strategy).
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence

from . import dtypes
from . import ops as F
from .graph import GraphKeys, Operation, Tensor, as_shape, get_default_graph


class Variable:
    def __init__(self, initial_value: Any = None, trainable: bool = True,
                 collections: Optional[Sequence[str]] = None, name: Optional[str] = None,
                 dtype=None, validate_shape: bool = True):
        if initial_value is None:
            raise ValueError("initial_value must be specified")
        g = get_default_graph()
        with g.name_scope(name or "Variable") as scope:
            scope_name = scope[:-1] if scope else "Variable"
            if callable(initial_value):
                initial_value = initial_value()
            init = F.convert_to_tensor(initial_value, dtype=dtype, name="initial_value")
            if init.shape is None or None in init.shape:
                raise ValueError("initial_value of a Variable must have a fully defined shape")
            op = g.create_op("VariableV2", [], [(init.dtype, init.shape)], name=scope_name + "/",
                             attrs={"shape": init.shape, "dtype": init.dtype,
                                    "shared_name": scope_name})
            self._variable: Tensor = op.outputs[0]
            self._initial_value = init
            self._initializer_op: Operation = g.create_op(
                "Assign", [self._variable, init], [(init.dtype, init.shape)], name="Assign",
                attrs={"validate_shape": validate_shape})
        self._trainable = trainable
        if collections is None:
            collections = [GraphKeys.GLOBAL_VARIABLES]
        collections = list(collections)
        if trainable and GraphKeys.TRAINABLE_VARIABLES not in collections:
            collections.append(GraphKeys.TRAINABLE_VARIABLES)
        g.add_to_collections(collections, self)

    # -- graph-element protocol ----------------------------------------------------------
    def _as_graph_element(self) -> Tensor:
        return self._variable

    @property
    def name(self) -> str:
        return self._variable.name

    @property
    def op(self) -> Operation:
        return self._variable.op

    @property
    def graph(self):
        return self._variable.graph

    @property
    def dtype(self) -> dtypes.DType:
        return self._variable.dtype

    @property
    def shape(self):
        return self._variable.shape

    def get_shape(self):
        return self._variable.shape

    @property
    def device(self) -> str:
        return self._variable.device

    @property
    def trainable(self) -> bool:
        return self._trainable

    @property
    def initializer(self) -> Operation:
        return self._initializer_op

    @property
    def initial_value(self) -> Tensor:
        return self._initial_value

    def value(self) -> Tensor:
        return self._variable

    def read_value(self) -> Tensor:
        return F.identity(self._variable, name="read")

    # -- mutation ops --------------------------------------------------------------------
    def assign(self, value, name: Optional[str] = None) -> Tensor:
        value = F.convert_to_tensor(value, dtype=self.dtype)
        return self.graph.create_op("Assign", [self._variable, value],
                                    [(self.dtype, self.shape)], name=name or "Assign").outputs[0]

    def assign_add(self, delta, name: Optional[str] = None) -> Tensor:
        delta = F.convert_to_tensor(delta, dtype=self.dtype)
        return self.graph.create_op("AssignAdd", [self._variable, delta],
                                    [(self.dtype, self.shape)],
                                    name=name or "AssignAdd").outputs[0]

    def assign_sub(self, delta, name: Optional[str] = None) -> Tensor:
        delta = F.convert_to_tensor(delta, dtype=self.dtype)
        return self.graph.create_op("AssignSub", [self._variable, delta],
                                    [(self.dtype, self.shape)],
                                    name=name or "AssignSub").outputs[0]

    def eval(self, session=None):
        return self._variable.eval(session=session)

    def __repr__(self) -> str:
        return "<dtf.Variable %r shape=%s dtype=%s device=%r>" % (
            self.name, self.shape, self.dtype.name, self.device)

    # arithmetic on a Variable acts on its value
    def __add__(self, o): return F.add(self, o)
    def __radd__(self, o): return F.add(o, self)
    def __sub__(self, o): return F.subtract(self, o)
    def __rsub__(self, o): return F.subtract(o, self)
    def __mul__(self, o): return F.multiply(self, o)
    def __rmul__(self, o): return F.multiply(o, self)
    def __truediv__(self, o): return F.divide(self, o)
    def __neg__(self): return F.negative(self)
    def __matmul__(self, o): return F.matmul(self, o)


# ---------------------------------------------------------------------------------------
# initializers
# ---------------------------------------------------------------------------------------
def constant_initializer(value=0, dtype=dtypes.float32) -> Callable:
    def init(shape, dtype=dtype, partition_info=None):
        return F.fill(shape, value, dtype=dtype, name="Const")
    return init


def zeros_initializer(dtype=dtypes.float32) -> Callable:
    return constant_initializer(0, dtype)


def ones_initializer(dtype=dtypes.float32) -> Callable:
    return constant_initializer(1, dtype)


def random_normal_initializer(mean=0.0, stddev=1.0, seed=None, dtype=dtypes.float32) -> Callable:
    def init(shape, dtype=dtype, partition_info=None):
        return F.random_normal(shape, mean, stddev, dtype=dtype, seed=seed)
    return init


def truncated_normal_initializer(mean=0.0, stddev=1.0, seed=None,
                                 dtype=dtypes.float32) -> Callable:
    def init(shape, dtype=dtype, partition_info=None):
        return F.truncated_normal(shape, mean, stddev, dtype=dtype, seed=seed)
    return init


def random_uniform_initializer(minval=0.0, maxval=1.0, seed=None,
                               dtype=dtypes.float32) -> Callable:
    def init(shape, dtype=dtype, partition_info=None):
        return F.random_uniform(shape, minval, maxval, dtype=dtype, seed=seed)
    return init


def glorot_uniform_initializer(seed=None, dtype=dtypes.float32) -> Callable:
    def init(shape, dtype=dtype, partition_info=None):
        shape = as_shape(shape)
        fan_in = shape[0] if len(shape) >= 1 else 1
        fan_out = shape[-1] if len(shape) >= 2 else fan_in
        limit = (6.0 / max(1, fan_in + fan_out)) ** 0.5
        return F.random_uniform(shape, -limit, limit, dtype=dtype, seed=seed)
    return init


def get_variable(name: str, shape=None, dtype=None, initializer=None, trainable: bool = True,
                 collections: Optional[Sequence[str]] = None) -> Variable:
    """Named variable; a second call with the same (scoped) name raises, as without
    ``reuse``. Default dtype float32 and default initializer glorot-uniform."""
    g = get_default_graph()
    full = (g._name_stack + "/" + name) if g._name_stack else name
    if full in g._var_store:
        raise ValueError("Variable %s already exists" % full)
    dt = dtypes.as_dtype(dtype) if dtype is not None else dtypes.float32
    if initializer is None:
        initializer = glorot_uniform_initializer()
    if callable(initializer):
        if shape is None:
            raise ValueError("shape of a new variable (%s) must be fully defined" % name)
        shape = as_shape(shape)
        init_fn = lambda: initializer(shape, dtype=dt)  # noqa: E731
    else:
        init_fn = lambda: F.convert_to_tensor(initializer, dtype=dt)  # noqa: E731
    var = Variable(init_fn, trainable=trainable, collections=collections, name=name)
    g._var_store[full] = var
    return var


# ---------------------------------------------------------------------------------------
# collections helpers
# ---------------------------------------------------------------------------------------
def global_variables() -> List[Variable]:
    return get_default_graph().get_collection(GraphKeys.GLOBAL_VARIABLES)


def trainable_variables() -> List[Variable]:
    return get_default_graph().get_collection(GraphKeys.TRAINABLE_VARIABLES)


def local_variables() -> List[Variable]:
    return get_default_graph().get_collection(GraphKeys.LOCAL_VARIABLES)


def variables_initializer(var_list: Sequence[Variable], name: str = "init") -> Operation:
    if not var_list:
        return F.no_op(name=name)
    return F.group(*[v.initializer for v in var_list], name=name)


def global_variables_initializer() -> Operation:
    return variables_initializer(global_variables())


def local_variables_initializer() -> Operation:
    return variables_initializer(local_variables(), name="init_local")


initialize_all_variables = global_variables_initializer


def report_uninitialized_variables(var_list: Optional[Sequence[Variable]] = None,
                                   name: str = "report_uninitialized_variables") -> Tensor:
    """String vector of the names of variables that hold no value yet (empty = ready)."""
    g = get_default_graph()
    if var_list is None:
        var_list = global_variables() + local_variables()
    with g.device(None):
        op = g.create_op("ReportUninitializedVariables", [v._variable for v in var_list],
                         [(dtypes.string, (None,))], name=name,
                         attrs={"names": [v.op.name for v in var_list]})
    return op.outputs[0]


def is_variable_initialized(variable: Variable) -> Tensor:
    return variable.graph.create_op("IsVariableInitialized", [variable._variable],
                                    [(dtypes.bool_, ())],
                                    name="IsVariableInitialized").outputs[0]
