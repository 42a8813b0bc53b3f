"""Reference kernels: every op type written with plain torch ops.

These run the CPU plumbing configuration (parameter-server training on localhost without a
GPU) and are the numerical oracle for the HIP kernels in tests. They never run on the GPU hot
path of an op that has a HIP kernel: the executor picks ``OpDef.hip_kernel`` there.

A kernel has the signature ``kernel(op, inputs, ctx) -> list of outputs`` where ``inputs`` are
torch tensors (or host objects for string-typed edges) and ``ctx`` is the session's execution
context (device, variable access).
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch

from ..framework.op_registry import ref_kernel
from . import philox


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


# -- sources ----------------------------------------------------------------------------
@ref_kernel("Const")
def _const(op, inputs, ctx):
    cached = ctx.const_cache.get(op)
    if cached is None:
        dt = op.attrs["dtype"]
        value = op.attrs["value"]
        value = np.asarray(value)
        cached = torch.as_tensor(value if value.flags.c_contiguous else value.copy()).to(
            device=ctx.device, dtype=dt.torch)
        ctx.const_cache[op] = cached
    return [cached]


@ref_kernel("Placeholder", stateful=True)
def _placeholder(op, inputs, ctx):
    raise RuntimeError("You must feed a value for placeholder tensor %r with dtype %s"
                       % (op.name, op.outputs[0].dtype.name))


@ref_kernel("Fill")
def _fill(op, inputs, ctx):
    dt = op.attrs["dtype"].torch
    return [torch.full(tuple(op.attrs["shape"]), op.attrs["value"], dtype=dt, device=ctx.device)]


@ref_kernel("ZerosLike")
def _zeros_like(op, inputs, ctx):
    return [torch.zeros_like(inputs[0])]


@ref_kernel("OnesLike")
def _ones_like(op, inputs, ctx):
    return [torch.ones_like(inputs[0])]


def _stream_keys(op, ctx):
    if op.attrs.get("deterministic", True):
        return op.attrs["seed"], op.attrs["seed2"]
    return ctx.nondeterministic_seed(), op.attrs["seed2"]


@ref_kernel("RandomStandardNormal", stateful=True)
def _random_normal(op, inputs, ctx):
    shape = tuple(op.attrs["shape"])
    seed, seed2 = _stream_keys(op, ctx)
    vals = philox.normal(_numel(shape), seed, seed2)
    t = torch.from_numpy(vals.reshape(shape).copy())
    t = t * op.attrs["stddev"] + op.attrs["mean"]
    return [t.to(device=ctx.device, dtype=op.attrs["dtype"].torch)]


@ref_kernel("RandomUniform", stateful=True)
def _random_uniform(op, inputs, ctx):
    shape = tuple(op.attrs["shape"])
    seed, seed2 = _stream_keys(op, ctx)
    vals = philox.uniform(_numel(shape), seed, seed2)
    lo, hi = op.attrs["minval"], op.attrs["maxval"]
    t = torch.from_numpy(vals.reshape(shape).copy()) * (hi - lo) + lo
    return [t.to(device=ctx.device, dtype=op.attrs["dtype"].torch)]


@ref_kernel("TruncatedNormal", stateful=True)
def _truncated_normal(op, inputs, ctx):
    shape = tuple(op.attrs["shape"])
    seed, seed2 = _stream_keys(op, ctx)
    u = torch.from_numpy(philox.uniform(_numel(shape), seed, seed2).reshape(shape).copy())
    lo = 0.5 * (1.0 + math.erf(-2.0 / math.sqrt(2.0)))
    hi = 1.0 - lo
    z = math.sqrt(2.0) * torch.erfinv(2.0 * (lo + u.double() * (hi - lo)) - 1.0)
    t = z.float() * op.attrs["stddev"] + op.attrs["mean"]
    return [t.to(device=ctx.device, dtype=op.attrs["dtype"].torch)]


# -- variables --------------------------------------------------------------------------
@ref_kernel("VariableV2", stateful=True)
def _variable(op, inputs, ctx):
    return [ctx.var_read(op)]


@ref_kernel("Assign", stateful=True)
def _assign(op, inputs, ctx):
    return [ctx.var_assign(op.inputs[0].op, inputs[1])]


@ref_kernel("AssignAdd", stateful=True)
def _assign_add(op, inputs, ctx):
    return [ctx.var_assign_add(op.inputs[0].op, inputs[1])]


@ref_kernel("AssignSub", stateful=True)
def _assign_sub(op, inputs, ctx):
    return [ctx.var_assign_add(op.inputs[0].op, -inputs[1])]


@ref_kernel("ApplyGradientDescent", stateful=True)
def _apply_sgd(op, inputs, ctx):
    grad = ctx.reduce_gradients([inputs[1]], [op.inputs[0].op])[0]
    return [ctx.var_apply_sgd(op.inputs[0].op, op.attrs["learning_rate"], grad)]


def optimizer_rule_step(rule: str, hyper, w, g, s0, s1=None, powers=None):
    """One step of a slot-carrying update rule on plain tensors, in fp32 (or the variable's
    dtype): returns ``(new w, new s0, new s1)``. The formulas are TensorFlow's
    (``ApplyMomentum``, ``ApplyAdagrad``, ``ApplyRMSProp``, ``ApplyAdadelta``, ``ApplyAdam``); the gfx950 kernel
    (``csrc/kernels/optim.hip``) computes the same expressions with fused multiply-adds."""
    g = g.to(w.device, w.dtype)
    if rule == "sgd":
        (lr,) = hyper
        w = w - lr * g
    elif rule in ("momentum", "nesterov"):
        lr, mu = hyper
        s0 = mu * s0 + g
        w = w - lr * (g + mu * s0) if rule == "nesterov" else w - lr * s0
    elif rule == "adagrad":
        (lr,) = hyper
        s0 = s0 + g * g
        w = w - lr * g / torch.sqrt(s0)
    elif rule == "rmsprop":
        lr, rho, mu, eps = hyper
        s0 = s0 + (1.0 - rho) * (g * g - s0)
        s1 = mu * s1 + lr * g / torch.sqrt(s0 + eps)
        w = w - s1
    elif rule == "adadelta":
        lr, rho, eps = hyper
        s0 = s0 + (1.0 - rho) * (g * g - s0)
        update = torch.sqrt(s1 + eps) / torch.sqrt(s0 + eps) * g
        s1 = s1 + (1.0 - rho) * (update * update - s1)
        w = w - lr * update
    elif rule == "adam":
        lr, b1, b2, eps = hyper
        p1, p2 = (float(p) for p in powers)
        lr_t = lr * math.sqrt(1.0 - p2) / (1.0 - p1)
        s0 = s0 + (1.0 - b1) * (g - s0)
        s1 = s1 + (1.0 - b2) * (g * g - s1)
        w = w - lr_t * s0 / (torch.sqrt(s1) + eps)
    else:
        raise ValueError("unknown optimizer rule %r" % (rule,))
    return w, s0, s1


def apply_rule_through_stores(ctx, rule, hyper, var_op, slot_ops, power_ops, grad, lr=None):
    """The update of one variable through the variable-store interface (any store: local CPU
    storage, a parameter-server task). State that lives in this process is assigned; state on a
    parameter server receives the *difference* (queued with the step's other updates and added
    on arrival), so concurrent workers' updates compose the way their plain-SGD updates do."""
    if lr is not None:                 # a learning rate computed in the graph (a schedule)
        hyper = (float(lr),) + tuple(hyper[1:])
    w = ctx.var_read(var_op)
    slots = [ctx.var_read(s) for s in slot_ops]
    powers = [ctx.var_read(p) for p in power_ops] if power_ops else None
    new_w, new_s0, new_s1 = optimizer_rule_step(
        rule, hyper, w, grad, slots[0] if slots else None,
        slots[1] if len(slots) > 1 else None, powers)
    for s_op, old, new in zip(slot_ops, slots, (new_s0, new_s1)):
        store = ctx.session._store_for(s_op)
        if getattr(store, "local_tensor", lambda n: None)(s_op.name) is not None:
            ctx.var_assign(s_op, new)
        else:
            ctx.var_assign_add(s_op, new - old)
    return ctx.var_apply_sgd(var_op, 1.0, w - new_w)


def _optimizer_refs(op):
    """(variable op, slot ops, power ops) of an ``ApplyOptimizer`` op."""
    refs = [t.op for t in op.inputs[:op.attrs["num_refs"]]]
    n_slots = {"rmsprop": 2, "adam": 2, "adadelta": 2, "sgd": 0}.get(op.attrs["rule"], 1)
    return refs[0], refs[1:1 + n_slots], refs[1 + n_slots:]


@ref_kernel("ApplyOptimizer", stateful=True)
def _apply_optimizer(op, inputs, ctx):
    var_op, slot_ops, power_ops = _optimizer_refs(op)
    lr = inputs[op.attrs["num_refs"]] if op.attrs.get("lr_input") else None
    if ctx.sharded_apply(op.attrs["rule"], op.attrs["hyper"], [var_op], [inputs[-1]], [slot_ops],
                         [ctx.var_read(p) for p in power_ops], lr):
        return [ctx.var_read(var_op)]
    grad = ctx.reduce_gradients([inputs[-1]], [var_op])[0]
    return [apply_rule_through_stores(ctx, op.attrs["rule"], op.attrs["hyper"], var_op,
                                      slot_ops, power_ops, grad, lr)]


def affine_through_store(ctx, var_op, factor: float, increment: float):
    """``var = var * factor + increment`` through the store interface (assigned locally, shipped
    as a difference to a parameter server)."""
    cur = ctx.var_read(var_op)
    new = cur * factor + increment
    store = ctx.session._store_for(var_op)
    if getattr(store, "local_tensor", lambda n: None)(var_op.name) is not None:
        return ctx.var_assign(var_op, new)
    return ctx.var_assign_add(var_op, new - cur)


@ref_kernel("AffineUpdateVariable", stateful=True)
def _affine_update_variable(op, inputs, ctx):
    return [affine_through_store(ctx, op.inputs[0].op, op.attrs["factor"],
                                 op.attrs["increment"])]


@ref_kernel("MultiApplyOptimizer", stateful=True)
def _multi_apply_optimizer(op, inputs, ctx):
    var_ops = list(op.attrs["var_ops"])
    slot_ops_list = list(op.attrs["slot_ops"])
    lr = inputs[len(var_ops)] if op.attrs.get("lr_input") else None
    grads = list(inputs[:len(var_ops)])
    # reduce_scatter_sync: the rule runs on the owner's shard, where its state lives
    done = ctx.sharded_apply(op.attrs["rule"], op.attrs["hyper"], var_ops, grads, slot_ops_list,
                             [ctx.var_read(p) for p in op.attrs["power_ops"]], lr)
    if done:
        keep = [i for i in range(len(var_ops)) if i not in set(done)]
        var_ops = [var_ops[i] for i in keep]
        slot_ops_list = [slot_ops_list[i] for i in keep]
        grads = [grads[i] for i in keep]
    if var_ops:
        grads = ctx.reduce_gradients(grads, var_ops)
    for var_op, slot_ops, g in zip(var_ops, slot_ops_list, grads):
        apply_rule_through_stores(ctx, op.attrs["rule"], op.attrs["hyper"], var_op, slot_ops,
                                  op.attrs["power_ops"], g, lr)
    for var_op, factor, increment in op.attrs.get("scalar_updates", []):
        affine_through_store(ctx, var_op, factor, increment)
    return []


@ref_kernel("MultiMovingAverage", stateful=True)
def _multi_moving_average(op, inputs, ctx):
    """``shadow -= rate * (shadow - variable)`` for every (shadow, variable) pair of the node
    (``tf.train.ExponentialMovingAverage``): difference, product and subtraction each rounded on
    their own, like the three element-wise ops the node stands for."""
    rate = inputs[0] if inputs else op.attrs["rate"]
    for shadow_op, var_op in zip(op.attrs["shadow_ops"], op.attrs["var_ops"]):
        shadow, var = ctx.var_read(shadow_op), ctx.var_read(var_op)
        ctx.var_assign_add(shadow_op, -((shadow - var) * rate))
    return []


@ref_kernel("IsVariableInitialized", stateful=True)
def _is_var_init(op, inputs, ctx):
    return [torch.tensor(ctx.var_is_initialized(op.inputs[0].op))]


@ref_kernel("ReportUninitializedVariables", stateful=True, host_only=True)
def _report_uninit(op, inputs, ctx):
    missing = [n.encode() for n in ctx.uninitialized([t.op for t in op.inputs])]
    return [np.array(missing, dtype=object)]


@ref_kernel("NoOp", stateful=True)
def _noop(op, inputs, ctx):
    return []


# -- elementwise ------------------------------------------------------------------------
def _bin(fn):
    def kernel(op, inputs, ctx):
        return [fn(inputs[0], inputs[1])]
    return kernel


def _un(fn):
    def kernel(op, inputs, ctx):
        return [fn(inputs[0])]
    return kernel


ref_kernel("Add")(_bin(torch.add))
ref_kernel("Sub")(_bin(torch.sub))
ref_kernel("Mul")(_bin(torch.mul))
ref_kernel("RealDiv")(_bin(torch.div))
ref_kernel("Maximum")(_bin(torch.maximum))
ref_kernel("Minimum")(_bin(torch.minimum))
ref_kernel("Pow")(_bin(torch.pow))
ref_kernel("Equal")(_bin(torch.eq))
ref_kernel("NotEqual")(_bin(torch.ne))
ref_kernel("Less")(_bin(torch.lt))
ref_kernel("Greater")(_bin(torch.gt))
ref_kernel("LessEqual")(_bin(torch.le))
ref_kernel("GreaterEqual")(_bin(torch.ge))


@ref_kernel("MeanSquaredError")
def _mean_squared_error(op, inputs, ctx):
    d = inputs[0] - inputs[1]
    return [(d * d).mean()]


@ref_kernel("MeanSquaredErrorGrad")
def _mean_squared_error_grad(op, inputs, ctx):
    a, b, g = inputs
    return [(2.0 * g / a.numel()) * (a - b)]


@ref_kernel("MSELossStep")
def _mse_loss_step(op, inputs, ctx):
    """(pred, labels[, g]) -> (loss, dpred[, dbias]): mean squared error, its gradient with
    respect to ``pred`` for d(loss) = g (constant ``g_const`` without the third input) and the
    column sums of that gradient."""
    pred, labels = inputs[0], inputs[1]
    g = inputs[2] if len(inputs) > 2 else op.attrs["g_const"]
    d = pred - labels
    dpred = (2.0 * g / pred.numel()) * d
    outs = [(d * d).mean(), dpred]
    if op.attrs.get("emit_bias_grad"):
        outs.append(dpred.sum(dim=0))
    return outs


@ref_kernel("Select")
def _select(op, inputs, ctx):
    return [torch.where(inputs[0], inputs[1], inputs[2])]
ref_kernel("Neg")(_un(torch.neg))
ref_kernel("Log")(_un(torch.log))
ref_kernel("Exp")(_un(torch.exp))
ref_kernel("Sqrt")(_un(torch.sqrt))
ref_kernel("Floor")(_un(torch.floor))
ref_kernel("Cos")(_un(torch.cos))
ref_kernel("Sin")(_un(torch.sin))
ref_kernel("Softplus")(_un(torch.nn.functional.softplus))
ref_kernel("Square")(_un(torch.square))
ref_kernel("Reciprocal")(_un(torch.reciprocal))
ref_kernel("Abs")(_un(torch.abs))
ref_kernel("Sign")(_un(torch.sign))


@ref_kernel("Dropout", stateful=True)
def _dropout(op, inputs, ctx):
    """Host oracle of the device kernel: round r of the op's stream covers elements
    ``r * 4 * ceil(n / 4) ...``; the round number lives with the session."""
    x = inputs[0]
    keep = float(inputs[1]) if len(inputs) > 1 else op.attrs["keep_prob"]
    seed, seed2 = _stream_keys(op, ctx)
    n = x.numel()
    key = (op, "round")
    rnd = ctx.buffers.get(key, 0)
    ctx.buffers[key] = rnd + 1
    u = philox.uniform(n, seed, seed2, offset=rnd * 4 * ((n + 3) // 4))
    mask = torch.from_numpy(np.where(u < np.float32(keep), np.float32(1.0) / np.float32(keep),
                                     np.float32(0.0)).astype(np.float32).reshape(tuple(x.shape)))
    mask = mask.to(x.device)
    return [x * mask, mask]


@ref_kernel("OneHot")
def _one_hot(op, inputs, ctx):
    idx = inputs[0].long()
    depth = op.attrs["depth"]
    hit = idx.unsqueeze(-1) == torch.arange(depth, device=idx.device)
    on = torch.tensor(op.attrs["on_value"], device=idx.device)
    off = torch.tensor(op.attrs["off_value"], device=idx.device)
    return [torch.where(hit, on, off).to(op.attrs["dtype"].torch)]
ref_kernel("Sigmoid")(_un(torch.sigmoid))
ref_kernel("Tanh")(_un(torch.tanh))
ref_kernel("Relu")(_un(torch.relu))
ref_kernel("Identity")(_un(lambda x: x))
ref_kernel("StopGradient")(_un(lambda x: x))


@ref_kernel("Cast")
def _cast(op, inputs, ctx):
    return [inputs[0].to(op.attrs["dtype"].torch)]


@ref_kernel("AddN")
def _add_n(op, inputs, ctx):
    out = inputs[0]
    for t in inputs[1:]:
        out = out + t
    return [out]


@ref_kernel("Scale")
def _scale(op, inputs, ctx):
    return [inputs[0] * op.attrs["alpha"]]


@ref_kernel("Axpby")
def _axpby(op, inputs, ctx):
    return [inputs[0] * op.attrs["alpha"] + inputs[1] * op.attrs["beta"]]


@ref_kernel("SigmoidGrad")
def _sigmoid_grad(op, inputs, ctx):
    y, dy = inputs
    return [dy * y * (1 - y)]


@ref_kernel("TanhGrad")
def _tanh_grad(op, inputs, ctx):
    y, dy = inputs
    return [dy * (1 - y * y)]


@ref_kernel("ReluGrad")
def _relu_grad(op, inputs, ctx):
    y, dy = inputs
    return [dy * (y > 0).to(dy.dtype)]


# -- matmul -----------------------------------------------------------------------------
@ref_kernel("MatMul")
def _matmul(op, inputs, ctx):
    a, b = inputs
    if op.attrs["transpose_a"]:
        a = a.t()
    if op.attrs["transpose_b"]:
        b = b.t()
    if a.dtype == torch.float8_e4m3fn:
        return [(a.float() @ b.float()).to(torch.bfloat16)]
    if a.dtype in (torch.bfloat16, torch.float16) and a.device.type == "cpu":
        return [(a.float() @ b.float()).to(a.dtype)]
    return [a @ b]


# -- reductions / shapes ----------------------------------------------------------------
def _axes(op, x):
    axes = op.attrs["axis"]
    return tuple(range(x.dim())) if axes is None else tuple(axes)


@ref_kernel("Sum")
def _sum(op, inputs, ctx):
    x = inputs[0]
    axes = _axes(op, x)
    return [x.sum(dim=axes, keepdim=op.attrs["keepdims"]) if axes else x.clone()]


@ref_kernel("Mean")
def _mean(op, inputs, ctx):
    x = inputs[0]
    axes = _axes(op, x)
    return [x.mean(dim=axes, keepdim=op.attrs["keepdims"]) if axes else x.clone()]


@ref_kernel("Max")
def _max(op, inputs, ctx):
    x = inputs[0]
    return [x.amax(dim=_axes(op, x), keepdim=op.attrs["keepdims"])]


@ref_kernel("ArgMax")
def _argmax(op, inputs, ctx):
    return [inputs[0].argmax(dim=op.attrs["axis"]).to(op.outputs[0].dtype.torch)]


@ref_kernel("Reshape")
def _reshape(op, inputs, ctx):
    return [inputs[0].reshape(op.attrs["shape"])]


@ref_kernel("ReshapeLike")
def _reshape_like(op, inputs, ctx):
    return [inputs[0].reshape(inputs[1].shape)]


@ref_kernel("Transpose")
def _transpose(op, inputs, ctx):
    return [inputs[0].permute(op.attrs["perm"]).contiguous()]


@ref_kernel("SumToShapeOf")
def _sum_to_shape_of(op, inputs, ctx):
    x, like = inputs
    if x.shape == like.shape:
        return [x]
    lead = x.dim() - like.dim()
    if lead > 0:
        x = x.sum(dim=tuple(range(lead)))
    dims = tuple(i for i, (a, b) in enumerate(zip(x.shape, like.shape)) if b == 1 and a != 1)
    if dims:
        x = x.sum(dim=dims, keepdim=True)
    return [x]


@ref_kernel("BroadcastLike")
def _broadcast_like(op, inputs, ctx):
    return [inputs[0].expand_as(inputs[1]).contiguous()]


@ref_kernel("ReduceGrad")
def _reduce_grad(op, inputs, ctx):
    g, x = inputs
    axes = op.attrs["axis"]
    axes = tuple(range(x.dim())) if axes is None else tuple(axes)
    if not op.attrs["keepdims"]:
        for a in sorted(axes):
            g = g.unsqueeze(a)
    out = g.expand_as(x)
    if op.attrs["mean"]:
        count = 1
        for a in axes:
            count *= x.shape[a]
        out = out / count
    return [out.contiguous()]


# -- nn ---------------------------------------------------------------------------------
@ref_kernel("Softmax")
def _softmax(op, inputs, ctx):
    return [torch.softmax(inputs[0], dim=-1)]


@ref_kernel("LogSoftmax")
def _log_softmax(op, inputs, ctx):
    return [torch.log_softmax(inputs[0], dim=-1)]


@ref_kernel("SoftmaxGrad")
def _softmax_grad(op, inputs, ctx):
    y, dy = inputs
    return [(dy - (dy * y).sum(dim=-1, keepdim=True)) * y]


@ref_kernel("LogSoftmaxGrad")
def _log_softmax_grad(op, inputs, ctx):
    y, dy = inputs
    return [dy - torch.exp(y) * dy.sum(dim=-1, keepdim=True)]


@ref_kernel("SoftmaxCrossEntropyWithLogits")
def _sxent(op, inputs, ctx):
    logits, labels = inputs
    logp = torch.log_softmax(logits, dim=-1)
    loss = -(labels * logp).sum(dim=-1)
    backprop = torch.exp(logp) * labels.sum(dim=-1, keepdim=True) - labels
    return [loss, backprop]


@ref_kernel("RowScale")
def _row_scale(op, inputs, ctx):
    x, s = inputs
    return [x * s.unsqueeze(-1)]


# -- fused ops produced by the session's planner ----------------------------------------
_ACTS = {"none": lambda x: x, "sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}


@ref_kernel("FusedDense")
def _fused_dense(op, inputs, ctx):
    x, w, b = inputs
    return [_ACTS[op.attrs["activation"]](x @ w + b)]


@ref_kernel("SoftmaxXentMean")
def _softmax_xent_mean(op, inputs, ctx):
    """(logits, labels) -> (mean loss, softmax). Loss uses log-softmax, which agrees with
    ``mean(-sum(labels*log(softmax)))`` wherever the naive form is finite."""
    logits, labels = inputs
    logp = torch.log_softmax(logits, dim=-1)
    return [-(labels * logp).sum(dim=-1).mean(), torch.exp(logp)]


@ref_kernel("SoftmaxXentMeanGrad")
def _softmax_xent_mean_grad(op, inputs, ctx):
    """(softmax, labels, dloss) -> dlogits = dloss/B * (y*sum(labels) - labels), and, when
    ``emit_bias_grad`` is set, the column sum of dlogits (gradient of the logits' bias)."""
    y, labels, dloss = inputs
    b = y.shape[0]
    dz = (y * labels.sum(dim=-1, keepdim=True) - labels) * (dloss / b)
    return [dz, dz.sum(dim=0)] if op.attrs.get("emit_bias_grad") else [dz]


@ref_kernel("DenseGradSigmoid")
def _dense_grad_sigmoid(op, inputs, ctx):
    """(da, a) -> (dz, dbias): dz = da*a*(1-a), dbias = sum over batch."""
    da, a = inputs
    dz = da * a * (1 - a)
    return [dz, dz.sum(dim=0)]


@ref_kernel("ArgmaxAccuracy")
def _argmax_accuracy(op, inputs, ctx):
    """(y, labels) -> mean(argmax(y, 1) == argmax(labels, 1)) as fp32."""
    y, labels = inputs
    return [(y.argmax(dim=1) == labels.argmax(dim=1)).to(torch.float32).mean()]


@ref_kernel("ClassifierHeadTrain")
def _classifier_head_train(op, inputs, ctx):
    """(a, W, b, labels[, dloss]) -> (loss, softmax, dW, db, da | dz_prev[, db_prev][, accuracy]);
    without the fifth input d(loss) is the constant ``dloss_const``.

    Forward and backward of ``softmax(a W + b)`` under the mean cross-entropy in one node; with
    ``sigmoid_input`` the gradient is carried through the sigmoid that produced ``a``
    (``dz_prev = da * a * (1 - a)`` and its column sum ``db_prev``)."""
    a, w, b, labels = inputs[:4]
    dloss = inputs[4] if len(inputs) > 4 else op.attrs["dloss_const"]
    logp = torch.log_softmax(a @ w + b, dim=-1)
    y = torch.exp(logp)
    loss = -(labels * logp).sum(dim=-1).mean()
    dz = (y * labels.sum(dim=-1, keepdim=True) - labels) * (dloss / a.shape[0])
    outs = [loss, y, a.t() @ dz, dz.sum(dim=0)]
    da = dz @ w.t()
    if op.attrs.get("sigmoid_input"):
        dzp = da * a * (1 - a)
        outs += [dzp, dzp.sum(dim=0)]
    else:
        outs.append(da)
    if op.attrs.get("emit_accuracy"):
        outs.append((y.argmax(dim=1) == labels.argmax(dim=1)).to(torch.float32).mean())
    return outs


@ref_kernel("MLPTrainStep", stateful=True)
def _mlp_train_step(op, inputs, ctx):
    """(x, labels) -> (loss, softmax[, accuracy]) and, in place, one SGD step on the four
    parameters ``attrs["var_ops"] = (W1, b1, W2, b2)`` plus the step-counter increments: the
    nodes it replaces (``attrs["parts"]``: first layer, head, dW1 product, optimizer), run one
    after the other."""
    x, labels = inputs
    dense, head, _dw1_op, sgd = op.attrs["parts"]
    w1_op, b1_op, w2_op, b2_op = op.attrs["var_ops"]
    w1, b1, w2, b2 = (ctx.var_read(v) for v in (w1_op, b1_op, w2_op, b2_op))
    a = _fused_dense(dense, [x, w1, b1], ctx)[0]
    h = _classifier_head_train(head, [a, w2, b2, labels], ctx)
    grads = {w1_op: x.t() @ h[4], b1_op: h[5], w2_op: h[2], b2_op: h[3]}
    _multi_apply_sgd(sgd, [grads[v] for v in sgd.attrs["var_ops"]], ctx)
    outs = [h[0], h[1]]
    if op.attrs.get("emit_accuracy"):
        outs.append(h[6])
    return outs


@ref_kernel("MLPStepGrads")
def _mlp_step_grads(op, inputs, ctx):
    """(x, labels) -> (loss, softmax[, accuracy], dW1, db1, dW2, db2): forward and backward of
    the two-layer classifier, the nodes it replaces (``attrs["parts"]``) run one after the
    other; nothing is updated."""
    x, labels = inputs
    dense, head, _dw1_op = op.attrs["parts"]
    w1, b1, w2, b2 = (ctx.var_read(v) for v in op.attrs["var_ops"])
    a = _fused_dense(dense, [x, w1, b1], ctx)[0]
    h = _classifier_head_train(head, [a, w2, b2, labels], ctx)
    outs = [h[0], h[1]]
    if op.attrs.get("emit_accuracy"):
        outs.append(h[6])
    return outs + [x.t() @ h[4], h[5], h[2], h[3]]


@ref_kernel("BiasGrad")
def _bias_grad(op, inputs, ctx):
    return [inputs[0].sum(dim=0)]


@ref_kernel("MultiApplySGD", stateful=True)
def _multi_apply_sgd(op, inputs, ctx):
    var_ops = op.attrs["var_ops"]
    lrs = op.attrs["learning_rates"]
    if not ctx.sharded_sgd(var_ops, list(inputs), lrs):
        for var_op, lr, g in zip(var_ops, lrs, ctx.reduce_gradients(list(inputs), var_ops)):
            ctx.var_apply_sgd(var_op, lr, g)
    for var_op, delta in op.attrs.get("counter_increments", []):
        ctx.var_assign_add(var_op, torch.tensor(delta, dtype=var_op.outputs[0].dtype.torch,
                                                device=ctx.device))
    return []
