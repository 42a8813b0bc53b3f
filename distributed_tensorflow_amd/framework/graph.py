"""Deferred dataflow graph: ``Graph`` / ``Operation`` / ``Tensor`` plus name and device scopes.

This is the graph-construction layer the reference scripts program against
(``tf_distributed.py:34-89``: ``tf.device`` scopes, ``tf.name_scope``, placeholders,
variables, math ops, ``tf.get_default_graph``). It is deliberately small: a graph is an
append-only list of operations in creation order (which is a valid topological order), each
with typed output tensors, a device string resolved at creation time from the enclosing
device scopes / device functions, and op-type specific attributes. Execution, pruning and
kernel selection live in ``client.session``.
"""
from __future__ import annotations

import contextlib
import re
import threading
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

from . import dtypes
from .device import DeviceSpec
from ..utils import protowire as pw


class GraphKeys:
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"
    LOCAL_VARIABLES = "local_variables"
    SUMMARIES = "summaries"
    GLOBAL_STEP = "global_step"
    INIT_OP = "init_op"
    READY_OP = "ready_op"
    SAVERS = "savers"
    TRAIN_OP = "train_op"
    UPDATE_OPS = "update_ops"


Shape = Optional[Tuple[Optional[int], ...]]

_VALID_OP_NAME = re.compile(r"^[A-Za-z0-9.][A-Za-z0-9_.\-/]*$")
_VALID_SCOPE_NAME = re.compile(r"^[A-Za-z0-9_.\-/]*$")


def as_shape(shape) -> Shape:
    if shape is None:
        return None
    if isinstance(shape, int):
        return (shape,)
    return tuple(None if d is None or (isinstance(d, int) and d < 0) else int(d) for d in shape)


class Tensor:
    """Symbolic handle to one output of an ``Operation``."""

    __slots__ = ("op", "value_index", "dtype", "shape", "__weakref__")
    # make numpy defer to our reflected operators (np_array * tensor)
    __array_priority__ = 100

    def __init__(self, op: "Operation", value_index: int, dtype: dtypes.DType, shape: Shape):
        self.op = op
        self.value_index = value_index
        self.dtype = dtype
        self.shape = shape

    @property
    def name(self) -> str:
        return "%s:%d" % (self.op.name, self.value_index)

    @property
    def graph(self) -> "Graph":
        return self.op.graph

    @property
    def device(self) -> str:
        return self.op.device

    def get_shape(self) -> Shape:
        return self.shape

    def consumers(self) -> List["Operation"]:
        return [o for o in self.op.graph.get_operations() if self in o.inputs]

    def eval(self, feed_dict=None, session=None):
        from ..client.session import get_default_session
        session = session or get_default_session()
        if session is None:
            raise ValueError("no default session; pass session=")
        return session.run(self, feed_dict=feed_dict)

    def __repr__(self) -> str:
        return "<dtf.Tensor %r shape=%s dtype=%s>" % (self.name, self.shape, self.dtype.name)

    def __hash__(self) -> int:
        return id(self)

    def __eq__(self, other) -> bool:  # identity: tensors are dict keys (feed_dict)
        return self is other

    def __bool__(self):
        raise TypeError("a symbolic Tensor has no truth value; fetch it with Session.run")

    def __iter__(self):
        raise TypeError("a symbolic Tensor is not iterable")

    # arithmetic operator overloads are attached by framework.ops at import time


class Operation:
    """One node: type, inputs, attributes, outputs and its resolved device."""

    __slots__ = ("graph", "type", "name", "inputs", "control_inputs", "attrs", "outputs",
                 "device", "id", "__weakref__")

    def __init__(self, graph: "Graph", op_type: str, name: str, inputs: Sequence[Tensor],
                 control_inputs: Sequence["Operation"], attrs: Dict[str, Any], op_id: int):
        self.graph = graph
        self.type = op_type
        self.name = name
        self.inputs = list(inputs)
        self.control_inputs = list(control_inputs)
        self.attrs = dict(attrs)
        self.outputs: List[Tensor] = []
        self.device = ""
        self.id = op_id

    def get_attr(self, key: str) -> Any:
        return self.attrs[key]

    @property
    def node_def(self) -> "Operation":  # device functions written for node_def.op still work
        return self

    @property
    def op(self) -> str:  # ``node_def.op`` spelling
        return self.type

    def run(self, feed_dict=None, session=None) -> None:
        from ..client.session import get_default_session
        session = session or get_default_session()
        if session is None:
            raise ValueError("no default session; pass session=")
        session.run(self, feed_dict=feed_dict)

    def __repr__(self) -> str:
        return "<dtf.Operation %r type=%s device=%r>" % (self.name, self.type, self.device)


DeviceLike = Union[None, str, Callable[[Operation], Optional[str]]]

# op types that take a variable's storage as their first input and therefore run wherever
# the variable lives (so that the SGD update executes on the parameter owner)
REF_CONSUMER_TYPES = frozenset(
    ["Assign", "AssignAdd", "AssignSub", "ApplyGradientDescent", "IsVariableInitialized",
     "ApplyOptimizer", "AffineUpdateVariable"])
# op types all of whose inputs are variable references (never evaluated as values)
ALL_REF_INPUT_TYPES = frozenset(["ReportUninitializedVariables"])


def value_inputs(node):
    """The inputs of ``node`` that are consumed as *values*. A variable handed to a mutating
    op is a reference to storage, not a read: it creates no data dependency (so an ``Assign``
    can run before the variable holds anything)."""
    if node.type in ALL_REF_INPUT_TYPES:
        return []
    if node.type == "ApplyOptimizer":
        # [variable, its slots, (running powers), gradient]: only the gradient is a value
        return list(enumerate(node.inputs))[node.attrs["num_refs"]:]
    if node.type in REF_CONSUMER_TYPES:
        return list(enumerate(node.inputs))[1:]
    return list(enumerate(node.inputs))


class Graph:
    def __init__(self):
        self._ops: List[Operation] = []
        self._by_name: Dict[str, Operation] = {}
        self._names_in_use: Dict[str, int] = {}
        self._collections: Dict[str, List[Any]] = {}
        self._name_stack = ""
        self._device_stack: List[DeviceLike] = []
        self._control_stack: List[List[Operation]] = []
        self._colocate_stack: List[Operation] = []
        self.seed: Optional[int] = None
        self._finalized = False
        self._version = 0
        # tensor -> tensor holding d(loss)/d(tensor) for the most recent gradients() call that
        # touched it; the session's fusion pass uses it to replace backward chains wholesale
        self.grad_map: Dict[Tensor, Tensor] = {}
        self._var_store: Dict[str, Any] = {}   # get_variable() name -> Variable
        self._lock = threading.RLock()

    # -- default-graph stack -------------------------------------------------------------
    @contextlib.contextmanager
    def as_default(self):
        _default_stack.stack.append(self)
        try:
            yield self
        finally:
            _default_stack.stack.pop()

    # -- bookkeeping ---------------------------------------------------------------------
    @property
    def version(self) -> int:
        return self._version

    @property
    def finalized(self) -> bool:
        return self._finalized

    def finalize(self) -> None:
        self._finalized = True

    def _unfinalize(self) -> None:
        self._finalized = False

    def get_operations(self) -> List[Operation]:
        return list(self._ops)

    def get_operation_by_name(self, name: str) -> Operation:
        try:
            return self._by_name[name]
        except KeyError:
            raise KeyError("no operation named %r in the graph" % (name,))

    def get_tensor_by_name(self, name: str) -> Tensor:
        op_name, _, idx = name.partition(":")
        op = self.get_operation_by_name(op_name)
        return op.outputs[int(idx or 0)]

    def as_graph_element(self, obj, allow_tensor=True, allow_operation=True):
        if hasattr(obj, "_as_graph_element"):
            obj = obj._as_graph_element()
        if isinstance(obj, str):
            if ":" in obj and allow_tensor:
                return self.get_tensor_by_name(obj)
            if allow_operation:
                return self.get_operation_by_name(obj)
        if isinstance(obj, Tensor) and allow_tensor:
            if obj.graph is not self:
                raise ValueError("%r is not an element of this graph" % (obj,))
            return obj
        if isinstance(obj, Operation) and allow_operation:
            if obj.graph is not self:
                raise ValueError("%r is not an element of this graph" % (obj,))
            return obj
        raise TypeError("cannot interpret %r as a graph element" % (obj,))

    # -- collections ---------------------------------------------------------------------
    def add_to_collection(self, name: str, value: Any) -> None:
        self._collections.setdefault(name, []).append(value)

    def add_to_collections(self, names: Iterable[str], value: Any) -> None:
        for n in ([names] if isinstance(names, str) else names):
            self.add_to_collection(n, value)

    def get_collection(self, name: str, scope: Optional[str] = None) -> List[Any]:
        items = list(self._collections.get(name, []))
        if scope is None:
            return items
        return [i for i in items if getattr(i, "name", "").startswith(scope)]

    def get_collection_ref(self, name: str) -> List[Any]:
        return self._collections.setdefault(name, [])

    def clear_collection(self, name: str) -> None:
        self._collections.pop(name, None)

    # -- names ---------------------------------------------------------------------------
    def unique_name(self, name: str, mark_as_used: bool = True) -> str:
        if self._name_stack:
            name = self._name_stack + "/" + name
        count = self._names_in_use.get(name, 0)
        if mark_as_used:
            self._names_in_use[name] = count + 1
        if count > 0:
            base = name
            while True:
                name = "%s_%d" % (base, count)
                if name not in self._names_in_use:
                    break
                count += 1
            if mark_as_used:
                self._names_in_use[name] = 1
        return name

    @contextlib.contextmanager
    def name_scope(self, name: Optional[str]):
        old = self._name_stack
        if not name:
            new = ""
        elif name.endswith("/"):
            new = name[:-1]          # absolute re-entry of an existing scope
        else:
            if not _VALID_SCOPE_NAME.match(name):
                raise ValueError("%r is not a valid scope name" % (name,))
            new = self.unique_name(name)
        self._name_stack = new
        try:
            yield new + "/" if new else ""
        finally:
            self._name_stack = old

    # -- device / control / colocation scopes --------------------------------------------
    @contextlib.contextmanager
    def device(self, device_name_or_function: DeviceLike):
        self._device_stack.append(device_name_or_function)
        try:
            yield
        finally:
            self._device_stack.pop()

    @contextlib.contextmanager
    def control_dependencies(self, control_inputs: Optional[Sequence[Any]]):
        if control_inputs is None:
            saved, self._control_stack = self._control_stack, []
            try:
                yield
            finally:
                self._control_stack = saved
            return
        ops = []
        for c in control_inputs:
            c = getattr(c, "_as_graph_element", lambda: c)()
            ops.append(c.op if isinstance(c, Tensor) else c)
        self._control_stack.append(ops)
        try:
            yield
        finally:
            self._control_stack.pop()

    @contextlib.contextmanager
    def colocate_with(self, op: Union[Operation, Tensor, Any]):
        op = getattr(op, "op", op)
        self._colocate_stack.append(op)
        try:
            yield
        finally:
            self._colocate_stack.pop()

    def _resolve_device(self, op: Operation) -> str:
        if self._colocate_stack:
            return self._colocate_stack[-1].device
        if op.type in REF_CONSUMER_TYPES and op.inputs:
            return op.inputs[0].op.device
        spec = DeviceSpec()
        for entry in self._device_stack:           # outer -> inner; inner fields override
            if entry is None:
                spec = DeviceSpec()                 # device(None) clears enclosing scopes
                continue
            if callable(entry):
                op.device = spec.to_string()
                result = entry(op)
                if result:
                    spec = spec.merge_from(DeviceSpec.from_string(result))
            else:
                spec = spec.merge_from(DeviceSpec.from_string(entry))
        return spec.to_string()

    # -- op creation ---------------------------------------------------------------------
    def create_op(self, op_type: str, inputs: Sequence[Tensor], output_specs: Sequence[Tuple],
                  name: Optional[str] = None, attrs: Optional[Dict[str, Any]] = None,
                  control_inputs: Sequence[Operation] = ()) -> Operation:
        """Append an operation. ``output_specs`` is a list of ``(dtype, shape)``."""
        with self._lock:
            if self._finalized:
                raise RuntimeError("Graph is finalized and cannot be modified.")
            for t in inputs:
                if not isinstance(t, Tensor):
                    raise TypeError("op input %r is not a Tensor" % (t,))
                if t.graph is not self:
                    raise ValueError("input %r belongs to a different graph" % (t,))
            base = name or op_type
            if name and name.endswith("/"):
                full = name[:-1]          # absolute name chosen by an enclosing name_scope
                self._names_in_use.setdefault(full, 1)
            else:
                if not _VALID_OP_NAME.match(base):
                    raise ValueError("%r is not a valid op name" % (base,))
                full = self.unique_name(base)
            ctrl = list(control_inputs)
            for frame in self._control_stack:
                for c in frame:
                    if c not in ctrl:
                        ctrl.append(c)
            op = Operation(self, op_type, full, inputs, ctrl, attrs or {}, len(self._ops))
            op.outputs = [Tensor(op, i, dtypes.as_dtype(dt), as_shape(shape))
                          for i, (dt, shape) in enumerate(output_specs)]
            op.device = self._resolve_device(op)
            self._ops.append(op)
            self._by_name[full] = op
            self._version += 1
            return op

    # -- serialisation (for FileWriter(graph=...)) ---------------------------------------
    def as_graph_def(self) -> bytes:
        """Serialized ``GraphDef``: one ``NodeDef{name=1, op=2, input=3, device=4}`` per op,
        plus ``versions{producer}``. Attributes are omitted (tensor contents never travel in
        event files written by this framework)."""
        out = b""
        for op in self._ops:
            node = pw.f_bytes(1, op.name) + pw.f_bytes(2, op.type)
            for t in op.inputs:
                node += pw.f_bytes(3, t.name if t.value_index else t.op.name)
            for c in op.control_inputs:
                node += pw.f_bytes(3, "^" + c.name)
            if op.device:
                node += pw.f_bytes(4, op.device)
            out += pw.f_message(1, node)
        out += pw.f_message(4, pw.f_varint(1, 27))
        return out


class _DefaultStack(threading.local):
    def __init__(self):
        self.stack: List[Graph] = []


_default_stack = _DefaultStack()
_global_default = Graph()


def get_default_graph() -> Graph:
    return _default_stack.stack[-1] if _default_stack.stack else _global_default


def reset_default_graph() -> None:
    global _global_default
    if _default_stack.stack:
        raise AssertionError("reset_default_graph inside a `with graph.as_default()` block")
    _global_default = Graph()


def name_scope(name: Optional[str]):
    return get_default_graph().name_scope(name)


def device(device_name_or_function: DeviceLike):
    return get_default_graph().device(device_name_or_function)


def control_dependencies(control_inputs):
    return get_default_graph().control_dependencies(control_inputs)


def colocate_with(op):
    return get_default_graph().colocate_with(op)


def add_to_collection(name: str, value: Any) -> None:
    get_default_graph().add_to_collection(name, value)


def get_collection(name: str, scope: Optional[str] = None) -> List[Any]:
    return get_default_graph().get_collection(name, scope)


def set_random_seed(seed: Optional[int]) -> None:
    """Graph-level seed (reference ``tf_distributed.py:49``): together with each random op's
    creation index it fixes the counter-based generator stream of that op."""
    get_default_graph().seed = seed
