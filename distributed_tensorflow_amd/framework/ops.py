"""Graph op constructors (the ``tf.*`` math / array / random surface) with static shape
inference, and the gradient builders for each differentiable op.

Covers every op the reference graphs contain -- ``matmul``, ``add``, ``nn.sigmoid``,
``nn.softmax``, ``log``, ``*``, unary ``-``, ``reduce_sum``, ``reduce_mean``, ``argmax``,
``equal``, ``cast`` (``tf_distributed.py:62-81``), ``random_normal`` / ``zeros``
(``tf_distributed.py:51-57``, ``tf_distributed_1000Matrix.py:43-44``) -- and the handful more
that their gradients and the linear-regression / sharded-GEMM workloads need.
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import dtypes
from .graph import Graph, Operation, Shape, Tensor, as_shape, get_default_graph
from .op_registry import gradient

TensorLike = Union[Tensor, int, float, bool, np.ndarray, list, tuple]


# ---------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------
def convert_to_tensor(value: Any, dtype=None, name: Optional[str] = None) -> Tensor:
    if hasattr(value, "_as_graph_element"):
        value = value._as_graph_element()
    if isinstance(value, Tensor):
        if dtype is not None and dtypes.as_dtype(dtype) != value.dtype:
            raise TypeError("tensor %r has dtype %s, expected %s"
                            % (value.name, value.dtype.name, dtypes.as_dtype(dtype).name))
        return value
    return constant(value, dtype=dtype, name=name or "Const")


def _binary_inputs(x: TensorLike, y: TensorLike) -> Tuple[Tensor, Tensor]:
    if isinstance(x, Tensor) or hasattr(x, "_as_graph_element"):
        x = convert_to_tensor(x)
        y = convert_to_tensor(y, dtype=None if isinstance(y, Tensor)
                              or hasattr(y, "_as_graph_element") else x.dtype)
    else:
        y = convert_to_tensor(y)
        x = convert_to_tensor(x, dtype=y.dtype)
    if x.dtype != y.dtype:
        raise TypeError("dtype mismatch: %s vs %s" % (x.dtype.name, y.dtype.name))
    return x, y


def broadcast_static_shape(a: Shape, b: Shape) -> Shape:
    if a is None or b is None:
        return None
    out: List[Optional[int]] = []
    for i in range(1, max(len(a), len(b)) + 1):
        da = a[-i] if i <= len(a) else 1
        db = b[-i] if i <= len(b) else 1
        if da == 1:
            out.append(db)
        elif db == 1:
            out.append(da)
        elif da is None or db is None:
            out.append(da if db is None else db)
        elif da == db:
            out.append(da)
        else:
            raise ValueError("shapes %s and %s are not broadcast-compatible" % (a, b))
    return tuple(reversed(out))


def _norm_axes(axis, rank: Optional[int]) -> Optional[Tuple[int, ...]]:
    if axis is None:
        return None
    if isinstance(axis, (int, np.integer)):
        axis = [int(axis)]
    axes = []
    for a in axis:
        a = int(a)
        if a < 0:
            if rank is None:
                raise ValueError("negative axis needs a known rank")
            a += rank
        axes.append(a)
    return tuple(sorted(set(axes)))


def _reduced_shape(shape: Shape, axes: Optional[Tuple[int, ...]], keepdims: bool) -> Shape:
    if shape is None:
        return None
    if axes is None:
        return tuple(1 for _ in shape) if keepdims else ()
    out = []
    for i, d in enumerate(shape):
        if i in axes:
            if keepdims:
                out.append(1)
        else:
            out.append(d)
    return tuple(out)


def _g(inputs: Sequence[Tensor]) -> Graph:
    return inputs[0].graph if inputs else get_default_graph()


def _op1(op_type: str, inputs: Sequence[Tensor], out_dtype, out_shape: Shape, name=None,
         **attrs) -> Tensor:
    return _g(inputs).create_op(op_type, inputs, [(out_dtype, out_shape)], name=name,
                                attrs=attrs).outputs[0]


# ---------------------------------------------------------------------------------------
# sources
# ---------------------------------------------------------------------------------------
def constant(value: Any, dtype=None, shape=None, name: str = "Const") -> Tensor:
    if dtype is not None:
        dt = dtypes.as_dtype(dtype)
    elif isinstance(value, np.ndarray):
        dt = dtypes.as_dtype(value.dtype)
    elif isinstance(value, bool):
        dt = dtypes.bool_
    else:
        probe = np.asarray(value)
        dt = (dtypes.float32 if probe.dtype.kind == "f" else
              dtypes.int32 if probe.dtype.kind in "iu" else
              dtypes.bool_ if probe.dtype.kind == "b" else dtypes.as_dtype(probe.dtype))
    np_dtype = dt.numpy if dt.numpy is not None else np.float32
    arr = np.asarray(value, dtype=np_dtype)
    if shape is not None:
        shape = as_shape(shape)
        arr = np.broadcast_to(arr, shape).copy() if arr.shape != shape else arr
    return get_default_graph().create_op(
        "Const", [], [(dt, arr.shape)], name=name, attrs={"value": arr, "dtype": dt}).outputs[0]


def placeholder(dtype, shape=None, name: Optional[str] = None) -> Tensor:
    """Value supplied through ``feed_dict`` (``tf_distributed.py:44,46``)."""
    dt = dtypes.as_dtype(dtype)
    return get_default_graph().create_op(
        "Placeholder", [], [(dt, as_shape(shape))], name=name or "Placeholder",
        attrs={"dtype": dt, "shape": as_shape(shape)}).outputs[0]


def zeros(shape, dtype=dtypes.float32, name: Optional[str] = None) -> Tensor:
    return fill(shape, 0, dtype=dtype, name=name or "zeros")


def ones(shape, dtype=dtypes.float32, name: Optional[str] = None) -> Tensor:
    return fill(shape, 1, dtype=dtype, name=name or "ones")


def fill(shape, value, dtype=dtypes.float32, name: Optional[str] = None) -> Tensor:
    shape = as_shape(shape)
    dt = dtypes.as_dtype(dtype)
    return get_default_graph().create_op(
        "Fill", [], [(dt, shape)], name=name or "Fill",
        attrs={"shape": shape, "value": value, "dtype": dt}).outputs[0]


def zeros_like(x: TensorLike, name: Optional[str] = None) -> Tensor:
    x = convert_to_tensor(x)
    return _op1("ZerosLike", [x], x.dtype, x.shape, name=name or "zeros_like")


def ones_like(x: TensorLike, name: Optional[str] = None) -> Tensor:
    x = convert_to_tensor(x)
    return _op1("OnesLike", [x], x.dtype, x.shape, name=name or "ones_like")


def _random(op_type: str, shape, dtype, seed, name, **attrs) -> Tensor:
    g = get_default_graph()
    shape = as_shape(shape)
    dt = dtypes.as_dtype(dtype)
    # stream key = (graph seed, op seed); the op seed defaults to the op's creation index so
    # two random ops in one graph draw different, reproducible streams
    op_seed = len(g.get_operations()) if seed is None else int(seed)
    graph_seed = 0 if g.seed is None else int(g.seed)
    attrs.update(shape=shape, dtype=dt, seed=graph_seed, seed2=op_seed,
                 deterministic=(g.seed is not None or seed is not None))
    return g.create_op(op_type, [], [(dt, shape)], name=name or op_type.lower(),
                       attrs=attrs).outputs[0]


def random_normal(shape, mean=0.0, stddev=1.0, dtype=dtypes.float32, seed=None,
                  name: Optional[str] = None) -> Tensor:
    """N(mean, stddev) samples from a counter-based Philox stream
    (``tf_distributed.py:51-52``)."""
    return _random("RandomStandardNormal", shape, dtype, seed, name or "random_normal",
                   mean=float(mean), stddev=float(stddev))


def truncated_normal(shape, mean=0.0, stddev=1.0, dtype=dtypes.float32, seed=None,
                     name: Optional[str] = None) -> Tensor:
    return _random("TruncatedNormal", shape, dtype, seed, name or "truncated_normal",
                   mean=float(mean), stddev=float(stddev))


def random_uniform(shape, minval=0.0, maxval=1.0, dtype=dtypes.float32, seed=None,
                   name: Optional[str] = None) -> Tensor:
    return _random("RandomUniform", shape, dtype, seed, name or "random_uniform",
                   minval=float(minval), maxval=float(maxval))


# ---------------------------------------------------------------------------------------
# elementwise
# ---------------------------------------------------------------------------------------
def _binary(op_type: str, x, y, name=None, out_dtype=None) -> Tensor:
    x, y = _binary_inputs(x, y)
    return _op1(op_type, [x, y], out_dtype or x.dtype, broadcast_static_shape(x.shape, y.shape),
                name=name)


def add(x, y, name=None) -> Tensor:
    return _binary("Add", x, y, name)


def subtract(x, y, name=None) -> Tensor:
    return _binary("Sub", x, y, name)


def multiply(x, y, name=None) -> Tensor:
    return _binary("Mul", x, y, name)


def divide(x, y, name=None) -> Tensor:
    return _binary("RealDiv", x, y, name)


def maximum(x, y, name=None) -> Tensor:
    return _binary("Maximum", x, y, name)


def minimum(x, y, name=None) -> Tensor:
    return _binary("Minimum", x, y, name)


def pow(x, y, name=None) -> Tensor:  # noqa: A001
    return _binary("Pow", x, y, name)


def equal(x, y, name=None) -> Tensor:
    return _binary("Equal", x, y, name, out_dtype=dtypes.bool_)


def not_equal(x, y, name=None) -> Tensor:
    return _binary("NotEqual", x, y, name, out_dtype=dtypes.bool_)


def less(x, y, name=None) -> Tensor:
    return _binary("Less", x, y, name, out_dtype=dtypes.bool_)


def greater(x, y, name=None) -> Tensor:
    return _binary("Greater", x, y, name, out_dtype=dtypes.bool_)


def less_equal(x, y, name=None) -> Tensor:
    return _binary("LessEqual", x, y, name, out_dtype=dtypes.bool_)


def greater_equal(x, y, name=None) -> Tensor:
    return _binary("GreaterEqual", x, y, name, out_dtype=dtypes.bool_)


def where(condition, x=None, y=None, name=None) -> Tensor:
    """``condition ? x : y`` element-wise (``tf.where`` with both branches; the one-argument form
    that returns coordinates is not provided). ``x`` and ``y`` may be scalars."""
    if x is None or y is None:
        raise ValueError("where needs both x and y")
    condition = convert_to_tensor(condition)
    if condition.dtype != dtypes.bool_:
        raise TypeError("where: condition must be bool, got %s" % condition.dtype)
    x, y = _binary_inputs(x, y)
    shape = broadcast_static_shape(broadcast_static_shape(condition.shape, x.shape), y.shape)
    return _op1("Select", [condition, x, y], x.dtype, shape, name=name or "Select")


def expand_dims(x, axis: int, name=None) -> Tensor:
    x = convert_to_tensor(x)
    if x.shape is None:
        raise ValueError("expand_dims needs a known rank")
    rank = len(x.shape)
    axis = axis + rank + 1 if axis < 0 else axis
    if not 0 <= axis <= rank:
        raise ValueError("expand_dims: axis %d out of range for rank %d" % (axis, rank))
    dims = [-1 if d is None else d for d in x.shape]
    if dims.count(-1) > 1:
        raise ValueError("expand_dims supports at most one unknown dimension")
    return reshape(x, dims[:axis] + [1] + dims[axis:], name=name or "ExpandDims")


def squeeze(x, axis=None, name=None) -> Tensor:
    x = convert_to_tensor(x)
    if x.shape is None:
        raise ValueError("squeeze needs a known rank")
    rank = len(x.shape)
    axes = None if axis is None else {a + rank if a < 0 else a
                                      for a in ([axis] if isinstance(axis, int) else axis)}
    dims = []
    for i, d in enumerate(x.shape):
        drop = d == 1 and (axes is None or i in axes)
        if axes is not None and i in axes and d != 1:
            raise ValueError("squeeze: dimension %d has extent %s, not 1" % (i, d))
        if not drop:
            dims.append(-1 if d is None else d)
    if dims.count(-1) > 1:
        raise ValueError("squeeze supports at most one unknown dimension")
    return reshape(x, dims, name=name or "Squeeze")


def _unary(op_type: str, x, name=None) -> Tensor:
    x = convert_to_tensor(x)
    return _op1(op_type, [x], x.dtype, x.shape, name=name)


def negative(x, name=None) -> Tensor:
    return _unary("Neg", x, name)


def log(x, name=None) -> Tensor:
    return _unary("Log", x, name)


def exp(x, name=None) -> Tensor:
    return _unary("Exp", x, name)


def sqrt(x, name=None) -> Tensor:
    return _unary("Sqrt", x, name)


def square(x, name=None) -> Tensor:
    return _unary("Square", x, name)


def reciprocal(x, name=None) -> Tensor:
    return _unary("Reciprocal", x, name)


def abs(x, name=None) -> Tensor:  # noqa: A001
    return _unary("Abs", x, name)


def floor(x, name=None) -> Tensor:
    """Largest integer not above ``x``, element-wise (no gradient flows through it)."""
    return _unary("Floor", x, name)


def cos(x, name=None) -> Tensor:
    return _unary("Cos", x, name)


def sin(x, name=None) -> Tensor:
    return _unary("Sin", x, name)


def sign(x, name=None) -> Tensor:
    return _unary("Sign", x, name)


def sigmoid(x, name=None) -> Tensor:
    return _unary("Sigmoid", x, name)


def tanh(x, name=None) -> Tensor:
    return _unary("Tanh", x, name)


def relu(x, name=None) -> Tensor:
    return _unary("Relu", x, name)


def identity(x, name=None) -> Tensor:
    return _unary("Identity", x, name)


def stop_gradient(x, name=None) -> Tensor:
    return _unary("StopGradient", x, name)


def cast(x, dtype, name=None) -> Tensor:
    x = convert_to_tensor(x)
    dt = dtypes.as_dtype(dtype)
    if dt == x.dtype:
        return x
    return _op1("Cast", [x], dt, x.shape, name=name or "Cast", dtype=dt)


def add_n(inputs: Sequence[TensorLike], name=None) -> Tensor:
    inputs = [convert_to_tensor(i) for i in inputs]
    if len(inputs) == 1:
        return inputs[0]
    shape = inputs[0].shape
    for t in inputs[1:]:
        shape = broadcast_static_shape(shape, t.shape)
    return _op1("AddN", inputs, inputs[0].dtype, shape, name=name or "AddN")


def scale(x, alpha: float, name=None) -> Tensor:
    """``alpha * x`` with a compile-time scalar (one fused multiply, no constant tensor)."""
    x = convert_to_tensor(x)
    return _op1("Scale", [x], x.dtype, x.shape, name=name or "Scale", alpha=float(alpha))


def axpby(alpha: float, x, beta: float, y, name=None) -> Tensor:
    """``alpha*x + beta*y`` -- the elementwise add/scale primitive of the sharded-GEMM
    partial-sum path."""
    x, y = _binary_inputs(x, y)
    return _op1("Axpby", [x, y], x.dtype, broadcast_static_shape(x.shape, y.shape),
                name=name or "Axpby", alpha=float(alpha), beta=float(beta))


# ---------------------------------------------------------------------------------------
# matmul
# ---------------------------------------------------------------------------------------
def matmul(a, b, transpose_a: bool = False, transpose_b: bool = False, name=None) -> Tensor:
    """2-D GEMM (``tf_distributed.py:62,64``; ``tf_distributed_1000Matrix.py:48``)."""
    a, b = _binary_inputs(a, b)
    sa, sb = a.shape, b.shape
    if sa is not None and len(sa) != 2 or sb is not None and len(sb) != 2:
        raise ValueError("matmul needs rank-2 operands, got %s and %s" % (sa, sb))
    m = k1 = k2 = n = None
    if sa is not None:
        m, k1 = (sa[1], sa[0]) if transpose_a else sa
    if sb is not None:
        k2, n = (sb[1], sb[0]) if transpose_b else sb
    if k1 is not None and k2 is not None and k1 != k2:
        raise ValueError("matmul inner dimensions differ: %s vs %s" % (sa, sb))
    # fp8 (OCP e4m3) operands accumulate in fp32 on the matrix cores and produce bf16
    out_dtype = dtypes.bfloat16 if a.dtype == dtypes.float8_e4m3fn else a.dtype
    return _op1("MatMul", [a, b], out_dtype, (m, n), name=name or "MatMul",
                transpose_a=bool(transpose_a), transpose_b=bool(transpose_b))


# ---------------------------------------------------------------------------------------
# reductions / shape ops
# ---------------------------------------------------------------------------------------
def _reduce(op_type: str, x, axis, keepdims, name) -> Tensor:
    x = convert_to_tensor(x)
    rank = None if x.shape is None else len(x.shape)
    axes = _norm_axes(axis, rank)
    return _op1(op_type, [x], x.dtype, _reduced_shape(x.shape, axes, keepdims), name=name,
                axis=axes, keepdims=bool(keepdims))


def reduce_sum(x, axis=None, keepdims=False, name=None, reduction_indices=None,
               keep_dims=None) -> Tensor:
    axis = reduction_indices if axis is None else axis
    keepdims = keepdims if keep_dims is None else keep_dims
    return _reduce("Sum", x, axis, keepdims, name or "Sum")


def reduce_mean(x, axis=None, keepdims=False, name=None, reduction_indices=None,
                keep_dims=None) -> Tensor:
    axis = reduction_indices if axis is None else axis
    keepdims = keepdims if keep_dims is None else keep_dims
    return _reduce("Mean", x, axis, keepdims, name or "Mean")


def reduce_max(x, axis=None, keepdims=False, name=None) -> Tensor:
    return _reduce("Max", x, axis, keepdims, name or "Max")


def reduce_min(x, axis=None, keepdims=False, name=None) -> Tensor:
    """``-reduce_max(-x)``: the row-maximum kernel serves both."""
    return negative(reduce_max(negative(x), axis=axis, keepdims=keepdims), name=name or "Min")


def argmin(x, axis=None, name=None, dimension=None, output_type=dtypes.int64) -> Tensor:
    return argmax(negative(x), axis=axis, name=name or "ArgMin", dimension=dimension,
                  output_type=output_type)


def argmax(x, axis=None, name=None, dimension=None, output_type=dtypes.int64) -> Tensor:
    x = convert_to_tensor(x)
    axis = dimension if axis is None else axis
    axis = 0 if axis is None else int(axis)
    rank = None if x.shape is None else len(x.shape)
    if axis < 0:
        axis += rank
    shape = None if x.shape is None else tuple(d for i, d in enumerate(x.shape) if i != axis)
    return _op1("ArgMax", [x], dtypes.as_dtype(output_type), shape, name=name or "ArgMax",
                axis=axis)


def reshape(x, shape, name=None) -> Tensor:
    x = convert_to_tensor(x)
    shape = tuple(int(d) for d in shape)
    static = tuple(None if d == -1 else d for d in shape)
    if -1 in shape and x.shape is not None and None not in x.shape:
        known = int(np.prod([d for d in shape if d != -1])) or 1
        static = tuple(int(np.prod(x.shape)) // known if d == -1 else d for d in shape)
    return _op1("Reshape", [x], x.dtype, static, name=name or "Reshape", shape=shape)


def transpose(x, perm=None, name=None) -> Tensor:
    x = convert_to_tensor(x)
    rank = len(x.shape)
    perm = tuple(reversed(range(rank))) if perm is None else tuple(int(p) for p in perm)
    return _op1("Transpose", [x], x.dtype, tuple(x.shape[p] for p in perm),
                name=name or "transpose", perm=perm)


def sum_to_shape_of(x: Tensor, like: Tensor, name=None) -> Tensor:
    """Sum ``x`` over its broadcast dimensions so it has the runtime shape of ``like``."""
    return _op1("SumToShapeOf", [x, like], x.dtype, like.shape, name=name or "SumToShapeOf")


def _reduce_grad(g: Tensor, x: Tensor, axes, keepdims: bool, mean: bool) -> Tensor:
    return _op1("ReduceGrad", [g, x], x.dtype, x.shape, name="ReduceGrad", axis=axes,
                keepdims=keepdims, mean=mean)


# ---------------------------------------------------------------------------------------
# nn
# ---------------------------------------------------------------------------------------
def softmax(logits, axis: int = -1, name=None) -> Tensor:
    logits = convert_to_tensor(logits)
    if axis not in (-1, len(logits.shape) - 1):
        raise ValueError("softmax is implemented over the last axis only")
    return _op1("Softmax", [logits], logits.dtype, logits.shape, name=name or "Softmax")


def log_softmax(logits, axis: int = -1, name=None) -> Tensor:
    logits = convert_to_tensor(logits)
    return _op1("LogSoftmax", [logits], logits.dtype, logits.shape, name=name or "LogSoftmax")


def softmax_cross_entropy_with_logits(labels=None, logits=None, name=None,
                                      _sentinel=None) -> Tensor:
    """Per-example ``-sum(labels * log_softmax(logits))`` -- the numerically safe spelling of
    the loss the reference writes out by hand at ``tf_distributed.py:70``."""
    logits, labels = _binary_inputs(logits, labels)
    shape = None if logits.shape is None else logits.shape[:-1]
    op = logits.graph.create_op(
        "SoftmaxCrossEntropyWithLogits", [logits, labels],
        [(logits.dtype, shape), (logits.dtype, logits.shape)],
        name=name or "softmax_cross_entropy_with_logits")
    return op.outputs[0]


def bias_add(value, bias, name=None) -> Tensor:
    return add(value, bias, name=name or "BiasAdd")


def xw_plus_b(x, weights, biases, name=None) -> Tensor:
    """``matmul(x, weights) + biases`` (one GEMM with the bias in its epilogue once the fusion
    pass has seen it)."""
    with get_default_graph().name_scope(name or "xw_plus_b"):
        return bias_add(matmul(x, weights), biases)


def relu6(x, name=None) -> Tensor:
    with get_default_graph().name_scope(name or "Relu6"):
        return minimum(relu(x), 6.0)


def softplus(x, name=None) -> Tensor:
    """``log(1 + exp(x))``, one kernel, evaluated as ``max(x, 0) + log1p(exp(-|x|))`` (no
    overflow); its gradient is ``sigmoid(x)``."""
    return _unary("Softplus", x, name)


def sigmoid_cross_entropy_with_logits(labels=None, logits=None, name=None,
                                      _sentinel=None) -> Tensor:
    """Element-wise ``-z * log(sigmoid(x)) - (1 - z) * log(1 - sigmoid(x))`` for logits ``x`` and
    targets ``z``, computed as ``softplus(x) - x * z``."""
    if _sentinel is not None or labels is None or logits is None:
        raise ValueError("sigmoid_cross_entropy_with_logits takes labels= and logits= by name")
    with get_default_graph().name_scope(name or "logistic_loss"):
        logits = convert_to_tensor(logits)
        labels = convert_to_tensor(labels, dtype=logits.dtype)
        return subtract(softplus(logits), multiply(logits, labels))


def dropout(x, keep_prob, noise_shape=None, seed=None, name=None) -> Tensor:
    """``x * mask / keep_prob`` with an independent Bernoulli(keep_prob) mask at every run (and
    at every replay of a captured step: the mask's position in its Philox stream is a counter
    on the device). ``keep_prob`` may be a scalar tensor, e.g. a placeholder fed 1.0 at
    evaluation time."""
    if noise_shape is not None:
        raise NotImplementedError("dropout: noise_shape is not provided")
    x = convert_to_tensor(x)
    g = x.graph
    op_seed = len(g.get_operations()) if seed is None else int(seed)
    graph_seed = 0 if g.seed is None else int(g.seed)
    inputs, constant_keep = [x], 1.0
    if isinstance(keep_prob, (int, float)):
        if not 0.0 < keep_prob <= 1.0:
            raise ValueError("keep_prob must be in (0, 1], got %r" % (keep_prob,))
        constant_keep = float(keep_prob)
    else:
        inputs.append(convert_to_tensor(keep_prob))
    op = g.create_op("Dropout", inputs, [(x.dtype, x.shape), (x.dtype, x.shape)],
                     name=name or "dropout",
                     attrs={"keep_prob": constant_keep, "seed": graph_seed, "seed2": op_seed,
                            "deterministic": g.seed is not None or seed is not None})
    return op.outputs[0]


def one_hot(indices, depth: int, on_value=1.0, off_value=0.0, axis: int = -1,
            dtype=dtypes.float32, name=None) -> Tensor:
    """``[..., depth]`` tensor with ``on_value`` at the index of the last axis (an index outside
    ``[0, depth)`` gives a row of ``off_value``)."""
    indices = convert_to_tensor(indices)
    if axis != -1:
        raise ValueError("one_hot is implemented for the last axis only")
    if indices.dtype not in (dtypes.int32, dtypes.int64):
        raise TypeError("one_hot: indices must be int32 or int64, got %s" % indices.dtype)
    shape = None if indices.shape is None else tuple(indices.shape) + (int(depth),)
    return _op1("OneHot", [indices], dtypes.as_dtype(dtype), shape, name=name or "one_hot",
                depth=int(depth), on_value=float(on_value), off_value=float(off_value),
                dtype=dtypes.as_dtype(dtype))


def sparse_softmax_cross_entropy_with_logits(labels=None, logits=None, name=None,
                                             _sentinel=None) -> Tensor:
    """Per-example cross-entropy against integer class labels: the labels are expanded on the
    device (``one_hot``) and the fused softmax-cross-entropy kernels take it from there."""
    logits = convert_to_tensor(logits)
    if logits.shape is None or logits.shape[-1] is None:
        raise ValueError("sparse_softmax_cross_entropy_with_logits needs a static class count")
    dense = one_hot(labels, logits.shape[-1], dtype=logits.dtype)
    return softmax_cross_entropy_with_logits(
        labels=dense, logits=logits, name=name or "sparse_softmax_cross_entropy_with_logits")


def mean_squared_error(labels, predictions, name=None) -> Tensor:
    """``mean((predictions - labels) ** 2)`` over all elements as one fused op (forward: partial
    sums + fold; backward: one element-wise launch) when both have the same static shape; the
    composition ``reduce_mean(square(predictions - labels))`` otherwise."""
    predictions, labels = _binary_inputs(predictions, labels)
    ps, ls = predictions.shape, labels.shape
    if ps is None or ls is None or len(ps) != len(ls) or any(
            a != b for a, b in zip(ps, ls)):
        return reduce_mean(square(subtract(predictions, labels)), name=name)
    return _op1("MeanSquaredError", [predictions, labels], predictions.dtype, (),
                name=name or "mean_squared_error")


def l2_loss(t, name=None) -> Tensor:
    """``sum(t ** 2) / 2`` (weight decay term)."""
    return scale(reduce_sum(square(t)), 0.5, name=name or "L2Loss")


def global_norm(t_list, name=None) -> Tensor:
    """``sqrt(sum_i ||t_i||^2)`` over the tensors of ``t_list`` (``None`` entries are skipped)."""
    parts = [reduce_sum(square(convert_to_tensor(t))) for t in t_list if t is not None]
    if not parts:
        raise ValueError("global_norm of an empty list")
    total = parts[0] if len(parts) == 1 else add_n(parts)
    return sqrt(total, name=name or "global_norm")


def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
    """``t_i * clip_norm / max(global_norm, clip_norm)`` for every tensor; returns
    ``(clipped list, global_norm)`` -- gradient clipping between ``compute_gradients`` and
    ``apply_gradients``."""
    norm = use_norm if use_norm is not None else global_norm(t_list)
    factor = divide(clip_norm, maximum(norm, clip_norm))
    return [None if t is None else multiply(convert_to_tensor(t), factor) for t in t_list], norm


def clip_by_norm(t, clip_norm, name=None) -> Tensor:
    t = convert_to_tensor(t)
    norm = sqrt(reduce_sum(square(t)))
    return multiply(t, divide(clip_norm, maximum(norm, clip_norm)), name=name or "clip_by_norm")


def clip_by_value(t, clip_value_min, clip_value_max, name=None) -> Tensor:
    return minimum(maximum(t, clip_value_min), clip_value_max, name=name or "clip_by_value")


# ---------------------------------------------------------------------------------------
# control / grouping
# ---------------------------------------------------------------------------------------
def group(*inputs, name: Optional[str] = None) -> Operation:
    deps = []
    for i in inputs:
        i = getattr(i, "_as_graph_element", lambda: i)()
        deps.append(i.op if isinstance(i, Tensor) else i)
    g = deps[0].graph if deps else get_default_graph()
    with g.control_dependencies(None):
        return g.create_op("NoOp", [], [], name=name or "group_deps", control_inputs=deps)


def no_op(name: Optional[str] = None) -> Operation:
    return get_default_graph().create_op("NoOp", [], [], name=name or "NoOp")


# ---------------------------------------------------------------------------------------
# gradients
# ---------------------------------------------------------------------------------------
def _unbroadcast(g: Tensor, like: Tensor, other: Optional[Tensor] = None) -> Tensor:
    """Reduce the gradient ``g`` of a broadcasting op's output to the shape of input ``like``.

    No reduction is emitted when it is statically certain that ``like`` was not expanded: equal
    rank and, per dimension, either equal known sizes or an ``other`` operand that is absent /
    size 1 there (then the output dimension *is* ``like``'s, whatever its runtime size).
    Otherwise a ``SumToShapeOf`` resolves it at run time."""
    ls, gs = like.shape, g.shape
    if ls is not None and gs is not None and len(ls) == len(gs):
        os_ = None if other is None else other.shape
        certain = True
        for i in range(1, len(ls) + 1):
            if ls[-i] is not None and ls[-i] == gs[-i]:
                continue
            if other is not None and os_ is not None and (i > len(os_) or os_[-i] == 1):
                continue
            certain = False
            break
        if certain:
            return g
    return sum_to_shape_of(g, like)


@gradient("Add")
def _add_grad(op, g):
    x, y = op.inputs
    return _unbroadcast(g, x, y), _unbroadcast(g, y, x)


@gradient("Sub")
def _sub_grad(op, g):
    x, y = op.inputs
    return _unbroadcast(g, x, y), _unbroadcast(negative(g), y, x)


@gradient("Mul")
def _mul_grad(op, g):
    x, y = op.inputs
    return _unbroadcast(multiply(g, y), x, y), _unbroadcast(multiply(g, x), y, x)


@gradient("Pow")
def _pow_grad(op, g):
    """d/dx x^y = y x^(y-1); d/dy x^y = x^y ln x, taken as 0 where x <= 0 (TensorFlow's
    convention: the logarithm is not evaluated into the result there)."""
    x, y = op.inputs
    z = op.outputs[0]
    gx = multiply(g, multiply(y, pow(x, subtract(y, 1.0))))
    safe_log = where(greater(x, 0.0), log(x), zeros_like(x))
    gy = multiply(g, multiply(z, safe_log))
    return _unbroadcast(gx, x, y), _unbroadcast(gy, y, x)


@gradient("RealDiv")
def _div_grad(op, g):
    x, y = op.inputs
    gx = divide(g, y)
    gy = negative(multiply(gx, op.outputs[0]))
    return _unbroadcast(gx, x, y), _unbroadcast(gy, y, x)


@gradient("Axpby")
def _axpby_grad(op, g):
    x, y = op.inputs
    return (_unbroadcast(scale(g, op.attrs["alpha"]), x),
            _unbroadcast(scale(g, op.attrs["beta"]), y))


@gradient("Scale")
def _scale_grad(op, g):
    return (scale(g, op.attrs["alpha"]),)


@gradient("Neg")
def _neg_grad(op, g):
    return (negative(g),)


@gradient("Log")
def _log_grad(op, g):
    return (divide(g, op.inputs[0]),)


@gradient("Softplus")
def _softplus_grad(op, g):
    return (multiply(g, sigmoid(op.inputs[0])),)


@gradient("Cos")
def _cos_grad(op, g):
    return (negative(multiply(g, sin(op.inputs[0]))),)


@gradient("Sin")
def _sin_grad(op, g):
    return (multiply(g, cos(op.inputs[0])),)


@gradient("Exp")
def _exp_grad(op, g):
    return (multiply(g, op.outputs[0]),)


@gradient("Sqrt")
def _sqrt_grad(op, g):
    return (divide(scale(g, 0.5), op.outputs[0]),)


@gradient("Square")
def _square_grad(op, g):
    return (multiply(scale(g, 2.0), op.inputs[0]),)


@gradient("Abs")
def _abs_grad(op, g):
    return (multiply(g, sign(op.inputs[0])),)


@gradient("MeanSquaredError")
def _mse_grad(op, g):
    pred, labels = op.inputs
    dpred = _op1("MeanSquaredErrorGrad", [pred, labels, g], pred.dtype, pred.shape,
                 name="MeanSquaredErrorGrad")
    return dpred, negative(dpred)


@gradient("Select")
def _select_grad(op, g):
    cond, x, y = op.inputs
    zero = constant(0.0, dtype=g.dtype)
    gx = _unbroadcast(where(cond, g, zero), x)
    gy = _unbroadcast(where(cond, zero, g), y)
    return None, gx, gy


@gradient("Maximum")
def _maximum_grad(op, g):
    x, y = op.inputs
    x_wins = subtract(ones_like(g), cast(less(x, y), g.dtype))     # 1 where x >= y
    return (_unbroadcast(multiply(g, x_wins), x, y),
            _unbroadcast(multiply(g, subtract(ones_like(g), x_wins)), y, x))


@gradient("Minimum")
def _minimum_grad(op, g):
    x, y = op.inputs
    x_wins = subtract(ones_like(g), cast(greater(x, y), g.dtype))  # 1 where x <= y
    return (_unbroadcast(multiply(g, x_wins), x, y),
            _unbroadcast(multiply(g, subtract(ones_like(g), x_wins)), y, x))


@gradient("Reciprocal")
def _recip_grad(op, g):
    y = op.outputs[0]
    return (negative(multiply(g, multiply(y, y))),)


@gradient("Sigmoid")
def _sigmoid_grad(op, g):
    y = op.outputs[0]
    return (_op1("SigmoidGrad", [y, g], y.dtype, y.shape, name="SigmoidGrad"),)


@gradient("Tanh")
def _tanh_grad(op, g):
    y = op.outputs[0]
    return (_op1("TanhGrad", [y, g], y.dtype, y.shape, name="TanhGrad"),)


@gradient("Relu")
def _relu_grad(op, g):
    y = op.outputs[0]
    return (_op1("ReluGrad", [y, g], y.dtype, y.shape, name="ReluGrad"),)


@gradient("Identity")
def _identity_grad(op, g):
    return (g,)


@gradient("Cast")
def _cast_grad(op, g):
    src = op.inputs[0].dtype
    if src.is_floating and g.dtype.is_floating:
        return (cast(g, src),)
    return (None,)


@gradient("AddN")
def _addn_grad(op, g):
    return tuple(_unbroadcast(g, x) for x in op.inputs)


@gradient("MatMul")
def _matmul_grad(op, g):
    a, b = op.inputs
    ta, tb = op.attrs["transpose_a"], op.attrs["transpose_b"]
    if not ta and not tb:
        return matmul(g, b, transpose_b=True), matmul(a, g, transpose_a=True)
    if not ta and tb:
        return matmul(g, b), matmul(g, a, transpose_a=True)
    if ta and not tb:
        return matmul(b, g, transpose_b=True), matmul(a, g)
    return (matmul(b, g, transpose_a=True, transpose_b=True),
            matmul(g, a, transpose_a=True, transpose_b=True))


@gradient("Sum")
def _sum_grad(op, g):
    return (_reduce_grad(g, op.inputs[0], op.attrs["axis"], op.attrs["keepdims"], False),)


@gradient("Mean")
def _mean_grad(op, g):
    return (_reduce_grad(g, op.inputs[0], op.attrs["axis"], op.attrs["keepdims"], True),)


@gradient("Dropout")
def _dropout_grad(op, g, g_mask=None):
    gx = multiply(g, op.outputs[1])
    return (gx,) if len(op.inputs) == 1 else (gx, None)


@gradient("Max")
def _max_grad(op, g):
    """The gradient goes to the positions that hold the maximum, shared equally among ties."""
    x, axes, keep = op.inputs[0], op.attrs["axis"], op.attrs["keepdims"]
    y_b = _reduce_grad(op.outputs[0], x, axes, keep, False)          # max, spread back over x
    hit = cast(equal(x, y_b), x.dtype)
    ties_b = _reduce_grad(_reduce("Sum", hit, axes, keep, None), x, axes, keep, False)
    return (multiply(divide(hit, ties_b), _reduce_grad(g, x, axes, keep, False)),)


@gradient("Reshape")
def _reshape_grad(op, g):
    x = op.inputs[0]
    return (_op1("ReshapeLike", [g, x], x.dtype, x.shape, name="ReshapeLike"),)


@gradient("Transpose")
def _transpose_grad(op, g):
    perm = op.attrs["perm"]
    inv = tuple(int(i) for i in np.argsort(perm))
    return (transpose(g, inv),)


@gradient("Softmax")
def _softmax_grad(op, g):
    y = op.outputs[0]
    return (_op1("SoftmaxGrad", [y, g], y.dtype, y.shape, name="SoftmaxGrad"),)


@gradient("LogSoftmax")
def _log_softmax_grad(op, g):
    y = op.outputs[0]
    return (_op1("LogSoftmaxGrad", [y, g], y.dtype, y.shape, name="LogSoftmaxGrad"),)


@gradient("SoftmaxCrossEntropyWithLogits")
def _sxent_grad(op, g_loss, g_backprop=None):
    backprop = op.outputs[1]
    scaled = _op1("RowScale", [backprop, g_loss], backprop.dtype, backprop.shape,
                  name="RowScale")
    return scaled, None


@gradient("SumToShapeOf")
def _sum_to_shape_grad(op, g):
    x = op.inputs[0]
    return (_op1("BroadcastLike", [g, x], x.dtype, x.shape, name="BroadcastLike"), None)


# ---------------------------------------------------------------------------------------
# python operator overloads
# ---------------------------------------------------------------------------------------
def _install_overloads() -> None:
    # default op names of the overloads are lower-case ("mul", "add"), as in TF1 graphs
    Tensor.__add__ = lambda s, o: add(s, o, name="add")
    Tensor.__radd__ = lambda s, o: add(o, s, name="add")
    Tensor.__sub__ = lambda s, o: subtract(s, o, name="sub")
    Tensor.__rsub__ = lambda s, o: subtract(o, s, name="sub")
    Tensor.__mul__ = lambda s, o: multiply(s, o, name="mul")
    Tensor.__rmul__ = lambda s, o: multiply(o, s, name="mul")
    Tensor.__truediv__ = lambda s, o: divide(s, o, name="truediv")
    Tensor.__rtruediv__ = lambda s, o: divide(o, s, name="truediv")
    Tensor.__neg__ = lambda s: negative(s)
    Tensor.__matmul__ = lambda s, o: matmul(s, o)
    Tensor.__pow__ = lambda s, o: pow(s, o)
    Tensor.__lt__ = lambda s, o: less(s, o)
    Tensor.__gt__ = lambda s, o: greater(s, o)
    Tensor.__abs__ = lambda s: abs(s)


_install_overloads()
