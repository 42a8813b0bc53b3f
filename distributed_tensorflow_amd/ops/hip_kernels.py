"""GPU kernels of the graph ops: each op type bound to the hand-written gfx950 kernels.

Imported by ``Session`` when its device is a GPU. Importing it loads ``libdtf_hip.so`` and
raises if the library has not been built -- a GPU session never silently runs torch kernels
for an op that has a native one.

Op types without an entry here (shape plumbing such as ``Reshape`` / ``Transpose``, integer
casts, host-side string ops) keep their reference kernel, which on a GPU tensor is a view or a
trivial torch op; none of them sits on the training hot path of the reference workloads.
"""
from __future__ import annotations

import torch

from ..framework.op_registry import hip_kernel
from . import hip

hip.lib()   # fail loudly now if the kernel library is missing

_f32 = torch.float32


def _is_f32(*tensors) -> bool:
    return all(isinstance(t, torch.Tensor) and t.dtype == _f32 for t in tensors)


def _ref(op_type):
    from ..framework.op_registry import get_opdef
    return get_opdef(op_type).ref_kernel


# -- sources ----------------------------------------------------------------------------
@hip_kernel("RandomStandardNormal")
def _random_normal(op, inputs, ctx):
    dt = op.attrs["dtype"].torch
    if dt not in (torch.float32, torch.bfloat16):
        return _ref("RandomStandardNormal")(op, inputs, ctx)
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    return [hip.philox_normal(op.attrs["shape"], seed, seed2, op.attrs["mean"],
                              op.attrs["stddev"], dt, ctx.device)]


@hip_kernel("RandomUniform")
def _random_uniform(op, inputs, ctx):
    dt = op.attrs["dtype"].torch
    if dt not in (torch.float32, torch.bfloat16):
        return _ref("RandomUniform")(op, inputs, ctx)
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    return [hip.philox_uniform(op.attrs["shape"], seed, seed2, op.attrs["minval"],
                               op.attrs["maxval"], dt, ctx.device)]


@hip_kernel("Fill")
def _fill(op, inputs, ctx):
    if op.attrs["dtype"].torch != _f32:
        return _ref("Fill")(op, inputs, ctx)
    # a small constant fill (the 1.0 seeding backprop, zero biases) is materialised once per
    # session; consumers never write into their inputs (Assign copies into the store)
    cached = ctx.const_cache.get(op)
    if cached is not None:
        return [cached]
    out = hip.fill(op.attrs["shape"], op.attrs["value"], ctx.device)
    if out.numel() <= 65536 and not torch.cuda.is_current_stream_capturing():
        ctx.const_cache[op] = out
    return [out]


@hip_kernel("OnesLike")
def _ones_like(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _ref("OnesLike")(op, inputs, ctx)
    return [hip.fill(inputs[0].shape, 1.0, ctx.device)]


@hip_kernel("ZerosLike")
def _zeros_like(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _ref("ZerosLike")(op, inputs, ctx)
    return [hip.fill(inputs[0].shape, 0.0, ctx.device)]


# -- elementwise ------------------------------------------------------------------------
def _register_binary(op_type: str, code: str) -> None:
    @hip_kernel(op_type)
    def kernel(op, inputs, ctx, _code=code, _type=op_type):
        a, b = inputs
        if not _is_f32(a, b):
            return _ref(_type)(op, inputs, ctx)
        try:
            return [hip.binary(_code, a, b)]
        except hip.HipKernelError as e:
            if "trailing-suffix" in str(e):      # general broadcast: rare, keep correctness
                return _ref(_type)(op, inputs, ctx)
            raise


for _t, _c in (("Add", "add"), ("Sub", "sub"), ("Mul", "mul"), ("RealDiv", "div"),
               ("Maximum", "max"), ("Minimum", "min"), ("Pow", "pow"),
               ("SigmoidGrad", "sigmoid_grad"), ("TanhGrad", "tanh_grad"),
               ("ReluGrad", "relu_grad")):
    _register_binary(_t, _c)


def _register_compare(op_type: str, code: str) -> None:
    @hip_kernel(op_type)
    def kernel(op, inputs, ctx, _code=code, _type=op_type):
        a, b = inputs
        if _is_f32(a, b) or (a.dtype == torch.int64 and b.dtype == torch.int64
                             and _code == "eq" and a.shape == b.shape):
            try:
                return [hip.compare(_code, a, b)]
            except hip.HipKernelError as e:
                if "trailing-suffix" not in str(e):
                    raise
        return _ref(_type)(op, inputs, ctx)


for _t, _c in (("Equal", "eq"), ("NotEqual", "ne"), ("Less", "lt"), ("Greater", "gt")):
    _register_compare(_t, _c)


def _register_unary(op_type: str, code: str) -> None:
    @hip_kernel(op_type)
    def kernel(op, inputs, ctx, _code=code, _type=op_type):
        if not _is_f32(inputs[0]):
            return _ref(_type)(op, inputs, ctx)
        return [hip.unary(_code, inputs[0])]


for _t, _c in (("Neg", "neg"), ("Log", "log"), ("Exp", "exp"), ("Sqrt", "sqrt"),
               ("Square", "square"), ("Reciprocal", "recip"), ("Abs", "abs"),
               ("Sigmoid", "sigmoid"), ("Tanh", "tanh"), ("Relu", "relu")):
    _register_unary(_t, _c)


@hip_kernel("Scale")
def _scale(op, inputs, ctx):
    x = inputs[0]
    if x.dtype not in (torch.float32, torch.bfloat16):
        return _ref("Scale")(op, inputs, ctx)
    return [hip.axpby(op.attrs["alpha"], x)]


@hip_kernel("Axpby")
def _axpby(op, inputs, ctx):
    x, y = inputs
    if x.dtype not in (torch.float32, torch.bfloat16) or x.shape != y.shape:
        return _ref("Axpby")(op, inputs, ctx)
    return [hip.axpby(op.attrs["alpha"], x, op.attrs["beta"], y)]


@hip_kernel("AddN")
def _add_n(op, inputs, ctx):
    if not _is_f32(*inputs) or any(t.shape != inputs[0].shape for t in inputs) \
            or len(inputs) > 8:
        return _ref("AddN")(op, inputs, ctx)
    return [hip.reduce_partials(list(inputs))]


@hip_kernel("Cast")
def _cast(op, inputs, ctx):
    x = inputs[0]
    dst = op.attrs["dtype"].torch
    pairs = {(torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32),
             (torch.bool, torch.float32)}
    if (x.dtype, dst) in pairs:
        return [hip.cast(x, dst)]
    return _ref("Cast")(op, inputs, ctx)


@hip_kernel("RowScale")
def _row_scale(op, inputs, ctx):
    x, s = inputs
    if not _is_f32(x, s) or x.dim() != 2:
        return _ref("RowScale")(op, inputs, ctx)
    return [hip.row_scale(x, s)]


# -- matmul -----------------------------------------------------------------------------
@hip_kernel("MatMul")
def _matmul(op, inputs, ctx):
    a, b = inputs
    ta, tb = op.attrs["transpose_a"], op.attrs["transpose_b"]
    if _is_f32(a, b):
        return [hip.gemm_f32(a, b, ta, tb)]
    if a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16:
        return [hip.gemm_bf16(a, b, ta, tb)]
    return _ref("MatMul")(op, inputs, ctx)


@hip_kernel("FusedDense")
def _fused_dense(op, inputs, ctx):
    x, w, b = inputs
    if not _is_f32(x, w, b):
        return _ref("FusedDense")(op, inputs, ctx)
    return [hip.gemm_f32(x, w, bias=b, activation=hip.ACTIVATIONS[op.attrs["activation"]])]


# -- reductions -------------------------------------------------------------------------
def _reduce(op, inputs, ctx, mean: bool, type_name: str):
    x = inputs[0]
    axes = op.attrs["axis"]
    if not _is_f32(x) or op.attrs["keepdims"]:
        return _ref(type_name)(op, inputs, ctx)
    if axes is None or tuple(axes) == tuple(range(x.dim())):
        scale = 1.0 / max(1, x.numel()) if mean else 1.0
        return [hip.reduce_all(x, scale)]
    if x.dim() == 2 and tuple(axes) == (1,):
        return [hip.reduce_rows(x, hip.ROW_MEAN if mean else hip.ROW_SUM)]
    if x.dim() == 2 and tuple(axes) == (0,):
        out = hip.reduce_cols(x)
        return [hip.axpby(1.0 / x.shape[0], out) if mean else out]
    return _ref(type_name)(op, inputs, ctx)


@hip_kernel("Sum")
def _sum(op, inputs, ctx):
    return _reduce(op, inputs, ctx, False, "Sum")


@hip_kernel("Mean")
def _mean(op, inputs, ctx):
    return _reduce(op, inputs, ctx, True, "Mean")


@hip_kernel("Max")
def _max(op, inputs, ctx):
    x = inputs[0]
    if _is_f32(x) and x.dim() == 2 and op.attrs["axis"] == (1,) and not op.attrs["keepdims"]:
        return [hip.reduce_rows(x, hip.ROW_MAX)]
    return _ref("Max")(op, inputs, ctx)


@hip_kernel("ArgMax")
def _argmax(op, inputs, ctx):
    x = inputs[0]
    if _is_f32(x) and x.dim() == 2 and op.attrs["axis"] == 1 \
            and op.outputs[0].dtype.torch == torch.int64:
        return [hip.argmax_rows(x)]
    return _ref("ArgMax")(op, inputs, ctx)


@hip_kernel("SumToShapeOf")
def _sum_to_shape_of(op, inputs, ctx):
    x, like = inputs
    if x.shape == like.shape:
        return [x]
    if _is_f32(x) and x.dim() == 2 and like.dim() == 1 and like.shape[0] == x.shape[1]:
        return [hip.reduce_cols(x)]
    return _ref("SumToShapeOf")(op, inputs, ctx)


@hip_kernel("BiasGrad")
def _bias_grad(op, inputs, ctx):
    if not _is_f32(inputs[0]) or inputs[0].dim() != 2:
        return _ref("BiasGrad")(op, inputs, ctx)
    return [hip.reduce_cols(inputs[0])]


# -- nn ---------------------------------------------------------------------------------
@hip_kernel("Softmax")
def _softmax(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _ref("Softmax")(op, inputs, ctx)
    return [hip.softmax_rows(inputs[0])]


@hip_kernel("LogSoftmax")
def _log_softmax(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _ref("LogSoftmax")(op, inputs, ctx)
    return [hip.softmax_rows(inputs[0], log=True)]


@hip_kernel("SoftmaxGrad")
def _softmax_grad(op, inputs, ctx):
    if not _is_f32(*inputs):
        return _ref("SoftmaxGrad")(op, inputs, ctx)
    return [hip.softmax_grad(inputs[0], inputs[1])]


@hip_kernel("SoftmaxCrossEntropyWithLogits")
def _sxent(op, inputs, ctx):
    logits, labels = inputs
    if not _is_f32(logits, labels) or logits.dim() != 2:
        return _ref("SoftmaxCrossEntropyWithLogits")(op, inputs, ctx)
    _, y, rows = hip.softmax_xent(logits, labels, want_softmax=True, want_rows=True,
                                  want_mean=False)
    # backprop = softmax * rowsum(labels) - labels
    lsum = hip.reduce_rows(labels, hip.ROW_SUM)
    backprop = hip.binary("sub", hip.row_scale(y, lsum), labels)
    return [rows, backprop]


@hip_kernel("SoftmaxXentMean")
def _softmax_xent_mean(op, inputs, ctx):
    logits, labels = inputs
    if not _is_f32(logits, labels) or logits.dim() != 2:
        return _ref("SoftmaxXentMean")(op, inputs, ctx)
    mean, y, _ = hip.softmax_xent(logits, labels)
    return [mean, y]


@hip_kernel("SoftmaxXentMeanGrad")
def _softmax_xent_mean_grad(op, inputs, ctx):
    y, labels, dloss = inputs
    if not _is_f32(y, labels, dloss) or y.dim() != 2:
        return _ref("SoftmaxXentMeanGrad")(op, inputs, ctx)
    want_bias = bool(op.attrs.get("emit_bias_grad"))
    dz, db = hip.softmax_xent_grad(y, labels, dloss, want_bias)
    return [dz, db] if want_bias else [dz]


@hip_kernel("DenseGradSigmoid")
def _dense_grad_sigmoid(op, inputs, ctx):
    da, a = inputs
    if not _is_f32(da, a) or da.dim() != 2:
        return _ref("DenseGradSigmoid")(op, inputs, ctx)
    dz, db = hip.sigmoid_grad_bias(da, a)
    return [dz, db]


@hip_kernel("ClassifierHeadTrain")
def _classifier_head_train(op, inputs, ctx):
    a, w, b, labels = inputs[:4]
    if len(inputs) > 4:
        dloss = inputs[4]
    else:
        key = ("scalar", float(op.attrs["dloss_const"]))
        dloss = ctx.const_cache.get(key)
        if dloss is None:
            dloss = hip.fill((), float(op.attrs["dloss_const"]), ctx.device)
            if not torch.cuda.is_current_stream_capturing():
                ctx.const_cache[key] = dloss
    if not _is_f32(a, w, b, labels, dloss) or a.dim() != 2:
        return _ref("ClassifierHeadTrain")(op, inputs, ctx)
    sig = bool(op.attrs.get("sigmoid_input"))
    want_acc = bool(op.attrs.get("emit_accuracy"))
    if hip.mlp_head_supported(a.shape[0], a.shape[1], w.shape[1]):
        loss, y, dw, db, da, dbp, acc = hip.mlp_head_train(a, w, b, labels, dloss, sig, want_acc)
    else:
        # batch or layer too large for the one-workgroup kernel: the same math as a chain of
        # the general kernels
        z = hip.gemm_f32(a, w, bias=b)
        loss, y, _ = hip.softmax_xent(z, labels)
        dz, db = hip.softmax_xent_grad(y, labels, dloss, True)
        dw = hip.gemm_f32(a, dz, trans_a=True)
        da = hip.gemm_f32(dz, w, trans_b=True)
        dbp = None
        if sig:
            da, dbp = hip.sigmoid_grad_bias(da, a)
        acc = hip.argmax_accuracy(y, labels) if want_acc else None
    outs = [loss, y, dw, db, da]
    if sig:
        outs.append(dbp)
    if want_acc:
        outs.append(acc)
    return outs


@hip_kernel("ArgmaxAccuracy")
def _argmax_accuracy(op, inputs, ctx):
    y, labels = inputs
    if not _is_f32(y, labels) or y.dim() != 2:
        return _ref("ArgmaxAccuracy")(op, inputs, ctx)
    return [hip.argmax_accuracy(y, labels)]


# -- optimizer --------------------------------------------------------------------------
@hip_kernel("MultiApplySGD")
def _multi_apply_sgd(op, inputs, ctx):
    """One launch for every local fp32 variable; anything else (remote parameter-server
    variables, other dtypes) goes through the store interface."""
    var_ops = op.attrs["var_ops"]
    lrs = op.attrs["learning_rates"]
    if ctx.sharded_sgd(var_ops, list(inputs), lrs):      # reduce-scatter / owner update / gather
        inputs, var_ops, lrs = [], [], []
    else:
        inputs = ctx.reduce_gradients(list(inputs))      # no-op without a synchronous runtime
    local_p, local_g, local_lr, rest = [], [], [], []
    for var_op, lr, g in zip(var_ops, lrs, inputs):
        store = ctx.session._store_for(var_op)
        fast = getattr(store, "local_tensor", None)
        p = fast(var_op.name) if fast is not None else None
        if p is not None and p.dtype == _f32 and g.dtype == _f32 and p.is_cuda:
            local_p.append(p)
            local_g.append(g)
            local_lr.append(lr)
            store.update_count += 1
        else:
            rest.append((var_op, lr, g))
    counters, deltas, rest_counters = [], [], []
    for var_op, delta in op.attrs.get("counter_increments", []):
        store = ctx.session._store_for(var_op)
        fast = getattr(store, "local_tensor", None)
        c = fast(var_op.name) if fast is not None else None
        if c is not None and c.dtype == _f32 and c.is_cuda and c.numel() == 1 \
                and len(counters) < 4:
            counters.append(c)
            deltas.append(delta)
        else:
            rest_counters.append((var_op, delta))
    if local_p or counters:
        hip.multi_sgd(local_p, local_g, local_lr, counters, deltas)
    for var_op, lr, g in rest:
        ctx.var_apply_sgd(var_op, lr, g)
    for var_op, delta in rest_counters:
        ctx.var_assign_add(var_op, torch.tensor(delta, dtype=var_op.outputs[0].dtype.torch,
                                                device=ctx.device))
    return []


@hip_kernel("ApplyGradientDescent")
def _apply_sgd(op, inputs, ctx):
    var_op = op.inputs[0].op
    store = ctx.session._store_for(var_op)
    fast = getattr(store, "local_tensor", None)
    p = fast(var_op.name) if fast is not None else None
    g = ctx.reduce_gradients([inputs[1]])[0]
    if p is not None and p.dtype == _f32 and g.dtype == _f32 and p.is_cuda:
        hip.multi_sgd([p], [g], [op.attrs["learning_rate"]])
        store.update_count += 1
        return [p]
    return [ctx.var_apply_sgd(var_op, op.attrs["learning_rate"], g)]
