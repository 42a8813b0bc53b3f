"""GPU kernels of the graph ops: each op type bound to the hand-written gfx950 kernels.

Imported by ``Session`` when its device is a GPU. Importing it loads ``libdtf_hip.so`` and
raises if the library has not been built -- a GPU session never silently runs torch kernels
for an op that has a native one.

Op types without an entry here are pure views or host bookkeeping (``Reshape``, ``Identity``,
``Const`` upload, variable reads, summaries): ``client.session`` lets exactly that list through
and refuses to plan anything else on a GPU. A kernel here that meets an element type or layout
it has no native code for does not quietly run torch either: ``_fallback`` raises unless the
session was created with ``ConfigProto(allow_torch_fallback=True)`` (then the fallback is
counted in ``Session.stats["torch_fallbacks"]`` and listed in ``Session.fallback_log``).
"""
from __future__ import annotations

import torch

from ..framework.op_registry import hip_kernel
from . import hip

hip.lib()   # fail loudly now if the kernel library is missing

_f32 = torch.float32


def _is_f32(*tensors) -> bool:
    return all(isinstance(t, torch.Tensor) and t.dtype == _f32 for t in tensors)


def _hint(ctx, index: int, shape, numel=None):
    """The caller-placed destination for output ``index`` of the step being run (its slot in the
    runtime's flat gradient arena), shaped like the output -- or ``None``. A kernel that writes
    there saves the copy the executor would otherwise make."""
    hints = ctx.out_hint
    if not hints:
        return None
    t = hints.get(index)
    if t is None or not t.is_cuda or t.dtype != _f32:
        return None
    n = 1
    for d in shape:
        n *= int(d)
    return t.view(tuple(shape)) if t.numel() == n and t.is_contiguous() else None


def _fallback(op_type, op, inputs, ctx, why: str = ""):
    """No native kernel for this case: refuse (default) or, when the session allows it, run
    the torch reference kernel and record that it happened."""
    from ..framework.op_registry import get_opdef
    ctx.note_fallback(op, why or ", ".join(
        str(getattr(t, "dtype", type(t).__name__)) for t in inputs))
    return get_opdef(op_type).ref_kernel(op, inputs, ctx)


# -- sources ----------------------------------------------------------------------------
@hip_kernel("RandomStandardNormal")
def _random_normal(op, inputs, ctx):
    dt = op.attrs["dtype"].torch
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    shape = tuple(op.attrs["shape"])
    if dt == hip.FP8 and len(shape) == 2:
        # fp8 e4m3 matrices (operands of the fp8 MFMA products): the 2-D window fill of the same
        # counter-based stream, element (i, j) = stream element i * cols + j
        return [hip.philox_window(shape[0], shape[1], seed, seed2, 0, 0, shape[1], dt,
                                  ctx.device, True, op.attrs["mean"], op.attrs["stddev"])]
    if dt not in (torch.float32, torch.bfloat16):
        return _fallback("RandomStandardNormal", op, inputs, ctx)
    return [hip.philox_normal(op.attrs["shape"], seed, seed2, op.attrs["mean"],
                              op.attrs["stddev"], dt, ctx.device)]


@hip_kernel("RandomUniform")
def _random_uniform(op, inputs, ctx):
    dt = op.attrs["dtype"].torch
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    shape = tuple(op.attrs["shape"])
    if dt == hip.FP8 and len(shape) == 2:
        lo, hi = float(op.attrs["minval"]), float(op.attrs["maxval"])
        return [hip.philox_window(shape[0], shape[1], seed, seed2, 0, 0, shape[1], dt,
                                  ctx.device, False, lo, hi - lo)]
    if dt not in (torch.float32, torch.bfloat16):
        return _fallback("RandomUniform", op, inputs, ctx)
    return [hip.philox_uniform(op.attrs["shape"], seed, seed2, op.attrs["minval"],
                               op.attrs["maxval"], dt, ctx.device)]


@hip_kernel("Fill")
def _fill(op, inputs, ctx):
    if op.attrs["dtype"].torch != _f32:
        return _fallback("Fill", op, inputs, ctx)
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
        return _fallback("OnesLike", op, inputs, ctx)
    return [hip.fill(inputs[0].shape, 1.0, ctx.device)]


@hip_kernel("ZerosLike")
def _zeros_like(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _fallback("ZerosLike", op, inputs, ctx)
    return [hip.fill(inputs[0].shape, 0.0, ctx.device)]


# -- elementwise ------------------------------------------------------------------------
def _register_binary(op_type: str, code: str) -> None:
    @hip_kernel(op_type)
    def kernel(op, inputs, ctx, _code=code, _type=op_type):
        a, b = inputs
        if not _is_f32(a, b):
            return _fallback(_type, op, inputs, ctx)
        try:
            return [hip.binary(_code, a, b)]
        except hip.HipKernelError as e:
            if "trailing-suffix" in str(e):      # general broadcast: rare, keep correctness
                return _fallback(_type, op, inputs, ctx)
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
        return _fallback(_type, op, inputs, ctx)


for _t, _c in (("Equal", "eq"), ("NotEqual", "ne"), ("Less", "lt"), ("Greater", "gt"),
               ("LessEqual", "le"), ("GreaterEqual", "ge")):
    _register_compare(_t, _c)


def _register_unary(op_type: str, code: str) -> None:
    @hip_kernel(op_type)
    def kernel(op, inputs, ctx, _code=code, _type=op_type):
        if not _is_f32(inputs[0]):
            return _fallback(_type, op, inputs, ctx)
        return [hip.unary(_code, inputs[0])]


for _t, _c in (("Neg", "neg"), ("Log", "log"), ("Exp", "exp"), ("Sqrt", "sqrt"),
               ("Square", "square"), ("Reciprocal", "recip"), ("Abs", "abs"),
               ("Sigmoid", "sigmoid"), ("Tanh", "tanh"), ("Relu", "relu"), ("Floor", "floor"),
               ("Cos", "cos"), ("Sin", "sin"), ("Softplus", "softplus")):
    _register_unary(_t, _c)


@hip_kernel("Scale")
def _scale(op, inputs, ctx):
    x = inputs[0]
    if x.dtype not in (torch.float32, torch.bfloat16):
        return _fallback("Scale", op, inputs, ctx)
    return [hip.axpby(op.attrs["alpha"], x)]


@hip_kernel("Axpby")
def _axpby(op, inputs, ctx):
    x, y = inputs
    if x.dtype not in (torch.float32, torch.bfloat16) or x.shape != y.shape:
        return _fallback("Axpby", op, inputs, ctx)
    return [hip.axpby(op.attrs["alpha"], x, op.attrs["beta"], y)]


@hip_kernel("AddN")
def _add_n(op, inputs, ctx):
    if not _is_f32(*inputs) or any(t.shape != inputs[0].shape for t in inputs) \
            or len(inputs) > 8:
        return _fallback("AddN", op, inputs, ctx)
    return [hip.reduce_partials(list(inputs))]


@hip_kernel("Cast")
def _cast(op, inputs, ctx):
    x = inputs[0]
    dst = op.attrs["dtype"].torch
    if x.dtype in hip._CAST_KINDS and dst in hip._CAST_KINDS:
        return [hip.cast(x, dst)]
    return _fallback("Cast", op, inputs, ctx)


@hip_kernel("RowScale")
def _row_scale(op, inputs, ctx):
    x, s = inputs
    if not _is_f32(x, s) or x.dim() != 2:
        return _fallback("RowScale", op, inputs, ctx)
    return [hip.row_scale(x, s)]


# -- matmul -----------------------------------------------------------------------------
def _ring_index(ctx, t) -> int:
    """Entry of the step ring's address table that holds ``t`` -- a fed tensor the session left
    ungathered because every reader takes it through the ring -- or -1."""
    if ctx.ring is None or not ctx.ring_feeds or not isinstance(t, torch.Tensor):
        return -1
    return ctx.ring_feeds.get(t.data_ptr(), -1)


def _k_contiguous_rhs(op, b: torch.Tensor, ctx) -> torch.Tensor:
    """``b`` ([K, N]) as the bf16 / fp8 kernels read a right-hand operand ([N, K]). When ``b`` is
    a variable that no kernel of the graph updates in place, the transposed image is made once
    and kept until the variable is written through its store (``version``) or gets new storage --
    a matmul benchmark's ``sess.run(C)`` then costs the product alone, not the product plus a
    pass over B. Anything else is transposed every time."""
    src = op.inputs[1].op
    if src.type == "VariableV2" and b.dim() == 2 and b.is_contiguous():
        from ..parallel.sharded_ops import _written_in_place
        store = ctx.session._store_for(src)
        if hasattr(store, "version") and not _written_in_place(ctx.session.graph, src):
            key = (op, "rhs_image")
            stamp = (b.data_ptr(), tuple(b.shape), store.version(src.name))
            hit = ctx.buffers.get(key)
            if hit is None or hit[0] != stamp:
                hit = ctx.buffers[key] = (stamp, hip.transpose2d(b))
                store.watch_writes(src.name)
            return hit[1]
    return hip.transpose2d(b)


@hip_kernel("MatMul")
def _matmul(op, inputs, ctx):
    a, b = inputs
    ta, tb = op.attrs["transpose_a"], op.attrs["transpose_b"]
    if _is_f32(a, b):
        m = a.shape[1] if ta else a.shape[0]
        n = b.shape[0] if tb else b.shape[1]
        idx = _ring_index(ctx, a)
        if idx >= 0:       # a fed tensor inside a captured step: read through the step ring
            return [hip.gemm_f32_ring(ctx.ring, idx, a, b.contiguous(), ta, tb,
                                      out=_hint(ctx, 0, (m, n)))]
        return [hip.gemm_f32(a, b, ta, tb, out=_hint(ctx, 0, (m, n)))]
    if a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16:
        if not ta and not tb and a.dim() == 2 and b.dim() == 2 and b.shape[0] % 64 == 0 \
                and 2.0 * a.shape[0] * b.shape[1] * b.shape[0] >= 2.0 * 512 ** 3:
            # (the sizes at which gemm_bf16 would transpose b itself, every call)
            return [hip.gemm_bf16_nt(a, _k_contiguous_rhs(op, b, ctx), torch.bfloat16)]
        return [hip.gemm_bf16(a, b, ta, tb)]
    if a.dtype == hip.FP8 and b.dtype == hip.FP8 and a.dim() == 2 and b.dim() == 2:
        # block-scaled fp8 MFMA kernel: both operands K-contiguous (one transpose pass for the
        # layouts that are not), fp32 accumulate, bf16 result
        a_kc = hip.transpose2d(a) if ta else a
        bt_kc = b if tb else _k_contiguous_rhs(op, b, ctx)
        return [hip.gemm_fp8_nt(a_kc, bt_kc, torch.bfloat16)]
    return _fallback("MatMul", op, inputs, ctx)


@hip_kernel("FusedDense")
def _fused_dense(op, inputs, ctx):
    x, w, b = inputs
    if not _is_f32(x, w, b):
        return _fallback("FusedDense", op, inputs, ctx)
    idx = _ring_index(ctx, x)
    if idx >= 0:           # a fed tensor inside a captured step: read through the step ring
        return [hip.gemm_f32_ring(ctx.ring, idx, x, w.contiguous(), bias=b,
                                  activation=hip.ACTIVATIONS[op.attrs["activation"]])]
    return [hip.gemm_f32(x, w, bias=b, activation=hip.ACTIVATIONS[op.attrs["activation"]])]


# -- reductions -------------------------------------------------------------------------
def _reduce(op, inputs, ctx, mean: bool, type_name: str):
    x = inputs[0]
    axes = op.attrs["axis"]
    if not _is_f32(x) or op.attrs["keepdims"]:
        return _fallback(type_name, op, inputs, ctx)
    if axes is None or tuple(axes) == tuple(range(x.dim())):
        scale = 1.0 / max(1, x.numel()) if mean else 1.0
        return [hip.reduce_all(x, scale)]
    if x.dim() == 2 and tuple(axes) == (1,):
        return [hip.reduce_rows(x, hip.ROW_MEAN if mean else hip.ROW_SUM)]
    if x.dim() == 2 and tuple(axes) == (0,):
        out = hip.reduce_cols(x)
        return [hip.axpby(1.0 / x.shape[0], out) if mean else out]
    return _fallback(type_name, op, inputs, ctx)


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
    return _fallback("Max", op, inputs, ctx)


@hip_kernel("ArgMax")
def _argmax(op, inputs, ctx):
    x = inputs[0]
    if _is_f32(x) and x.dim() == 2 and op.attrs["axis"] == 1 \
            and op.outputs[0].dtype.torch == torch.int64:
        return [hip.argmax_rows(x)]
    return _fallback("ArgMax", op, inputs, ctx)


@hip_kernel("SumToShapeOf")
def _sum_to_shape_of(op, inputs, ctx):
    x, like = inputs
    if x.shape == like.shape:
        return [x]
    if _is_f32(x) and x.dim() == 2 and like.dim() == 1 and like.shape[0] == x.shape[1]:
        return [hip.reduce_cols(x, out=_hint(ctx, 0, (x.shape[1],)))]
    return _fallback("SumToShapeOf", op, inputs, ctx)


@hip_kernel("BiasGrad")
def _bias_grad(op, inputs, ctx):
    if not _is_f32(inputs[0]) or inputs[0].dim() != 2:
        return _fallback("BiasGrad", op, inputs, ctx)
    return [hip.reduce_cols(inputs[0], out=_hint(ctx, 0, (inputs[0].shape[1],)))]


# -- nn ---------------------------------------------------------------------------------
@hip_kernel("Softmax")
def _softmax(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _fallback("Softmax", op, inputs, ctx)
    return [hip.softmax_rows(inputs[0])]


@hip_kernel("LogSoftmax")
def _log_softmax(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _fallback("LogSoftmax", op, inputs, ctx)
    return [hip.softmax_rows(inputs[0], log=True)]


@hip_kernel("SoftmaxGrad")
def _softmax_grad(op, inputs, ctx):
    if not _is_f32(*inputs):
        return _fallback("SoftmaxGrad", op, inputs, ctx)
    return [hip.softmax_grad(inputs[0], inputs[1])]


@hip_kernel("SoftmaxCrossEntropyWithLogits")
def _sxent(op, inputs, ctx):
    logits, labels = inputs
    if not _is_f32(logits, labels) or logits.dim() != 2:
        return _fallback("SoftmaxCrossEntropyWithLogits", op, inputs, ctx)
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
        return _fallback("SoftmaxXentMean", op, inputs, ctx)
    mean, y, _ = hip.softmax_xent(logits, labels)
    return [mean, y]


@hip_kernel("SoftmaxXentMeanGrad")
def _softmax_xent_mean_grad(op, inputs, ctx):
    y, labels, dloss = inputs
    if not _is_f32(y, labels, dloss) or y.dim() != 2:
        return _fallback("SoftmaxXentMeanGrad", op, inputs, ctx)
    want_bias = bool(op.attrs.get("emit_bias_grad"))
    dz, db = hip.softmax_xent_grad(y, labels, dloss, want_bias,
                                   db_out=_hint(ctx, 1, (y.shape[1],)) if want_bias else None)
    return [dz, db] if want_bias else [dz]


@hip_kernel("DenseGradSigmoid")
def _dense_grad_sigmoid(op, inputs, ctx):
    da, a = inputs
    if not _is_f32(da, a) or da.dim() != 2:
        return _fallback("DenseGradSigmoid", op, inputs, ctx)
    dz, db = hip.sigmoid_grad_bias(da, a, db_out=_hint(ctx, 1, (da.shape[1],)))
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
        return _fallback("ClassifierHeadTrain", op, inputs, ctx)
    sig = bool(op.attrs.get("sigmoid_input"))
    want_acc = bool(op.attrs.get("emit_accuracy"))
    li = _ring_index(ctx, labels)
    if hip.mlp_head_supported(a.shape[0], a.shape[1], w.shape[1]):
        h_, c_ = w.shape
        loss, y, dw, db, da, dbp, acc = hip.mlp_head_train(
            a, w, b, labels, dloss, sig, want_acc, dw_out=_hint(ctx, 2, (h_, c_)),
            db_out=_hint(ctx, 3, (c_,)), dbp_out=_hint(ctx, 5, (h_,)) if sig else None,
            ring=ctx.ring if li >= 0 else None, labels_index=li)
    elif li >= 0:
        raise hip.HipKernelError("ClassifierHeadTrain: the labels were left in the step ring "
                                 "but the head kernel does not take this shape")
    else:
        # batch or layer too large for the one-workgroup kernel: the same math as a chain of
        # the general kernels
        z = hip.gemm_f32(a, w, bias=b)
        loss, y, _ = hip.softmax_xent(z, labels)
        dz, db = hip.softmax_xent_grad(y, labels, dloss, True,
                                       db_out=_hint(ctx, 3, (w.shape[1],)))
        dw = hip.gemm_f32(a, dz, trans_a=True, out=_hint(ctx, 2, tuple(w.shape)))
        da = hip.gemm_f32(dz, w, trans_b=True)
        dbp = None
        if sig:
            da, dbp = hip.sigmoid_grad_bias(da, a, db_out=_hint(ctx, 5, (a.shape[1],)))
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
        return _fallback("ArgmaxAccuracy", op, inputs, ctx)
    return [hip.argmax_accuracy(y, labels)]


# -- shape plumbing / bookkeeping with real data movement ------------------------------------
@hip_kernel("Transpose")
def _transpose(op, inputs, ctx):
    x = inputs[0]
    perm = tuple(op.attrs["perm"])
    if x.dim() == 2 and perm == (1, 0) and x.element_size() in (1, 2, 4):
        return [hip.transpose2d(x)]
    if perm == tuple(range(x.dim())):
        return [x]
    return _fallback("Transpose", op, inputs, ctx, "perm %s of rank %d" % (perm, x.dim()))


@hip_kernel("MeanSquaredError")
def _mean_squared_error(op, inputs, ctx):
    a, b = inputs
    if not _is_f32(a, b) or a.shape != b.shape:
        return _fallback("MeanSquaredError", op, inputs, ctx)
    return [hip.mse(a, b)]


@hip_kernel("MeanSquaredErrorGrad")
def _mean_squared_error_grad(op, inputs, ctx):
    a, b, g = inputs
    if not _is_f32(a, b, g) or a.shape != b.shape:
        return _fallback("MeanSquaredErrorGrad", op, inputs, ctx)
    return [hip.mse_grad(a, b, g)]


@hip_kernel("MSELossStep")
def _mse_loss_step(op, inputs, ctx):
    """Loss, gradient and bias gradient of a mean-squared-error step in one launch
    (``csrc/kernels/mse_step.hip``; larger shapes: the separate kernels). Inside a captured step
    the launch also gathers the step's small results for the host when the session asks it to."""
    pred, labels = inputs[0], inputs[1]
    g = inputs[2] if len(inputs) > 2 else float(op.attrs["g_const"])
    want_db = bool(op.attrs.get("emit_bias_grad"))
    if not _is_f32(pred, labels) or pred.shape != labels.shape or \
            (isinstance(g, torch.Tensor) and not _is_f32(g)):
        return _fallback("MSELossStep", op, inputs, ctx)
    pred = pred if pred.is_contiguous() else pred.contiguous()
    labels = labels if labels.is_contiguous() else labels.contiguous()
    rows, cols = (pred.shape if pred.dim() == 2 else (pred.numel(), 1))
    if hip.mse_step_supported(rows, cols, want_db):
        loss = torch.empty((), dtype=_f32, device=pred.device)
        dpred = _hint(ctx, 1, pred.shape)
        outs = [loss, dpred if dpred is not None else torch.empty_like(pred)]
        if want_db:
            db = _hint(ctx, 2, (cols,))
            outs.append(db if db is not None
                        else torch.empty((cols,), dtype=_f32, device=pred.device))
        ring, req = ctx.ring, ctx.pack_request
        pack = adds = None
        if ring is not None and req is not None:
            parts = req.parts_for(op, outs)
            if parts is not None and not any(_spans_overlap(t, o) for t in parts
                                             for o in outs[1:]):
                adds = req.addends(parts)
                if adds is not None:
                    pack = parts
                    req.claim()
        li = ctx.ring_feeds.get(labels.data_ptr(), -1) if ring is not None else -1
        hip.mse_step(pred, labels, g, want_db, out=outs,
                     ring=ring if (pack is not None or li >= 0) else None,
                     pack=pack, pack_adds=adds, labels_index=li)
        return list(outs)
    if not isinstance(g, torch.Tensor):
        key = ("scalar", g)
        gt = ctx.const_cache.get(key)
        if gt is None:
            gt = hip.fill((), g, ctx.device)
            if not torch.cuda.is_current_stream_capturing():
                ctx.const_cache[key] = gt
        g = gt
    dpred = hip.mse_grad(pred, labels, g)
    outs = [hip.mse(pred, labels), dpred]
    if want_db:
        outs.append(hip.reduce_cols(dpred))
    return outs


@hip_kernel("Select")
def _select(op, inputs, ctx):
    cond, a, b = inputs
    if cond.dtype != torch.bool or not _is_f32(a, b):
        return _fallback("Select", op, inputs, ctx)
    return [hip.select(cond, a, b)]


@hip_kernel("Dropout")
def _dropout(op, inputs, ctx):
    x = inputs[0]
    keep_t = inputs[1] if len(inputs) > 1 else None
    if not _is_f32(x) or (keep_t is not None and not (_is_f32(keep_t) and keep_t.numel() == 1)):
        return _fallback("Dropout", op, inputs, ctx)
    key = (op, "round")
    rnd = ctx.buffers.get(key)
    if rnd is None:
        if torch.cuda.is_current_stream_capturing():
            raise hip.HipKernelError("Dropout: the round counter must exist before the step is "
                                     "captured (the first runs of a plan are eager)")
        # a 64-bit integer: an fp32 counter would stick at 2^24 and repeat one mask for ever
        rnd = ctx.buffers[key] = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    out, mask = hip.dropout(x.contiguous(), op.attrs["keep_prob"], seed, seed2, rnd, keep_t)
    hip.bump_counter(rnd)                            # the next run draws the next round
    return [out, mask]


@hip_kernel("OneHot")
def _one_hot(op, inputs, ctx):
    idx = inputs[0]
    if idx.dtype not in (torch.int32, torch.int64) or op.attrs["dtype"].torch != _f32:
        return _fallback("OneHot", op, inputs, ctx)
    return [hip.one_hot(idx, op.attrs["depth"], op.attrs["on_value"], op.attrs["off_value"])]


@hip_kernel("Sign")
def _sign(op, inputs, ctx):
    if not _is_f32(inputs[0]):
        return _fallback("Sign", op, inputs, ctx)
    return [hip.sign(inputs[0])]


def _mid_extents(shape, axes):
    """(outer, mid, inner) when ``axes`` is one contiguous run of dimensions of ``shape``."""
    axes = sorted(axes)
    if not axes or axes != list(range(axes[0], axes[-1] + 1)):
        return None
    outer = mid = inner = 1
    for i, d in enumerate(shape):
        if i < axes[0]:
            outer *= int(d)
        elif i <= axes[-1]:
            mid *= int(d)
        else:
            inner *= int(d)
    return outer, mid, inner


@hip_kernel("ReduceGrad")
def _reduce_grad(op, inputs, ctx):
    g, x = inputs
    axes = op.attrs["axis"]
    axes = tuple(range(x.dim())) if axes is None else tuple(axes)
    ext = _mid_extents(x.shape, axes) if _is_f32(g) and axes else None
    if ext is None:
        if _is_f32(g) and not axes:
            return [g]
        return _fallback("ReduceGrad", op, inputs, ctx, "axes %s" % (axes,))
    outer, mid, inner = ext
    scale = 1.0 / mid if op.attrs["mean"] else 1.0
    return [hip.broadcast_mid(g, outer, mid, inner, scale, x.shape)]


@hip_kernel("BroadcastLike")
def _broadcast_like(op, inputs, ctx):
    g, x = inputs
    if g.shape == x.shape:
        return [g]
    if _is_f32(g):
        gs = [1] * (x.dim() - g.dim()) + list(g.shape)
        axes = [i for i, (a, b) in enumerate(zip(gs, x.shape)) if a == 1 and b != 1]
        same = all(a == b for i, (a, b) in enumerate(zip(gs, x.shape)) if i not in axes)
        ext = _mid_extents(x.shape, axes) if same and axes else None
        if ext is not None:
            return [hip.broadcast_mid(g, ext[0], ext[1], ext[2], 1.0, x.shape)]
    return _fallback("BroadcastLike", op, inputs, ctx, "%s -> %s" % (tuple(g.shape), tuple(x.shape)))


@hip_kernel("LogSoftmaxGrad")
def _log_softmax_grad(op, inputs, ctx):
    if not _is_f32(*inputs):
        return _fallback("LogSoftmaxGrad", op, inputs, ctx)
    return [hip.log_softmax_grad(inputs[0], inputs[1])]


@hip_kernel("TruncatedNormal")
def _truncated_normal(op, inputs, ctx):
    if op.attrs["dtype"].torch != _f32:
        return _fallback("TruncatedNormal", op, inputs, ctx, str(op.attrs["dtype"]))
    if op.attrs.get("deterministic", True):
        seed, seed2 = op.attrs["seed"], op.attrs["seed2"]
    else:
        seed, seed2 = ctx.nondeterministic_seed(), op.attrs["seed2"]
    return [hip.truncated_normal(op.attrs["shape"], seed, seed2, op.attrs["mean"],
                                 op.attrs["stddev"], ctx.device)]


def _local_f32(ctx, var_op):
    store = ctx.session._store_for(var_op)
    fast = getattr(store, "local_tensor", None)
    p = fast(var_op.name) if fast is not None else None
    return p if (p is not None and p.is_cuda and p.dtype == _f32 and p.is_contiguous()) else None


def _assign_add_kernel(sign: float, type_name: str):
    def kernel(op, inputs, ctx):
        var_op, delta = op.inputs[0].op, inputs[1]
        p = _local_f32(ctx, var_op)
        if p is not None and _is_f32(delta) and delta.is_cuda \
                and (delta.shape == p.shape or delta.numel() == 1):
            return [hip.assign_add(p, delta, sign)]
        if p is None and not getattr(ctx.session._store_for(var_op), "local_tensor", None):
            # the variable lives in another process (parameter server): the store ships it
            return [ctx.var_assign_add(var_op, delta if sign > 0 else -delta)]
        return _fallback(type_name, op, inputs, ctx)
    return kernel


hip_kernel("AssignAdd")(_assign_add_kernel(1.0, "AssignAdd"))
hip_kernel("AssignSub")(_assign_add_kernel(-1.0, "AssignSub"))


@hip_kernel("Assign")
def _assign(op, inputs, ctx):
    """Same-shape re-assignment of a local variable copies into its storage with the block-copy
    kernel (the address stays put for captured steps); a first assignment hands the value to
    the store, which takes ownership of it."""
    var_op, value = op.inputs[0].op, inputs[1]
    store = ctx.session._store_for(var_op)
    fast = getattr(store, "local_tensor", None)
    cur = fast(var_op.name) if fast is not None else None
    if cur is not None and cur.is_cuda and isinstance(value, torch.Tensor) and value.is_cuda \
            and cur.shape == value.shape and cur.dtype == value.dtype and cur.is_contiguous() \
            and value.is_contiguous() and cur.numel() > 0 and cur.data_ptr() != value.data_ptr():
        hip.copy_2d(value.view(1, -1), cur.view(1, -1))
        return [cur]
    return [ctx.var_assign(var_op, value)]


# -- optimizer --------------------------------------------------------------------------
@hip_kernel("MultiApplySGD")
def _multi_apply_sgd(op, inputs, ctx):
    """One launch for every local fp32 variable; anything else (remote parameter-server
    variables, other dtypes) goes through the store interface."""
    var_ops = list(op.attrs["var_ops"])
    lrs = list(op.attrs["learning_rates"])
    inputs = list(inputs)
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
    # reduce_scatter_sync: reduce-scatter / owner update (+ the step counters, same launch) /
    # all-gather
    done = ctx.sharded_apply("sgd", (), var_ops, inputs, lrs=lrs, counters=counters,
                             deltas=deltas)
    if done:
        if ctx.counters_applied():
            counters, deltas = [], []
        keep = [i for i in range(len(var_ops)) if i not in set(done)]
        var_ops, lrs = [var_ops[i] for i in keep], [lrs[i] for i in keep]
        inputs = [inputs[i] for i in keep]
    if var_ops:
        inputs = ctx.reduce_gradients(inputs, var_ops)   # no-op without a sync runtime
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
    if local_p or counters:
        hip.multi_sgd(local_p, local_g, local_lr, counters, deltas)
    for var_op, lr, g in rest:
        ctx.var_apply_sgd(var_op, lr, g)
    for var_op, delta in rest_counters:
        ctx.var_assign_add(var_op, torch.tensor(delta, dtype=var_op.outputs[0].dtype.torch,
                                                device=ctx.device))
    return []


def _optim_launch(ctx, rule, hyper, var_ops, slot_ops_list, power_ops, grads, lr=None) -> list:
    """Update every variable whose storage, slots and powers are local fp32 GPU tensors with the
    multi-tensor kernel; returns the (var_op, slot_ops, grad) triples it could not take."""
    powers = [_local_f32(ctx, p) for p in power_ops]
    rest, ps, gs, s0, s1 = [], [], [], [], []
    if lr is not None and not (_is_f32(lr) and lr.is_cuda and lr.numel() == 1):
        return list(zip(var_ops, slot_ops_list, grads))
    for var_op, slot_ops, g in zip(var_ops, slot_ops_list, grads):
        p = _local_f32(ctx, var_op)
        slots = [_local_f32(ctx, s) for s in slot_ops]
        if p is None or any(t is None for t in slots + powers) or not _is_f32(g) \
                or not g.is_cuda or g.numel() != p.numel():
            rest.append((var_op, slot_ops, g))
            continue
        ps.append(p)
        gs.append(g)
        if slots:
            s0.append(slots[0])
        if len(slots) > 1:
            s1.append(slots[1])
        ctx.session._store_for(var_op).update_count += 1
    if ps:
        hip.multi_optim(rule, ps, gs, s0, s1, hyper, powers, lr=lr)
    return rest


def _affine_launch(ctx, updates) -> None:
    """``var = var * factor + increment`` for scalar variables: local fp32 GPU scalars in one
    launch, anything else through its store."""
    from .ref_kernels import affine_through_store
    local, rest = [], []
    for var_op, factor, increment in updates:
        p = _local_f32(ctx, var_op)
        if p is not None and p.numel() == 1 and len(local) < 8 \
                and all(p.data_ptr() != q.data_ptr() for q, _, _ in local):
            local.append((p, factor, increment))
        else:
            rest.append((var_op, factor, increment))
    if local:
        hip.affine_scalars([p for p, _, _ in local], [f for _, f, _ in local],
                           [d for _, _, d in local])
    for var_op, factor, increment in rest:
        affine_through_store(ctx, var_op, factor, increment)


@hip_kernel("ApplyOptimizer")
def _apply_optimizer(op, inputs, ctx):
    from .ref_kernels import _optimizer_refs, apply_rule_through_stores
    var_op, slot_ops, power_ops = _optimizer_refs(op)
    lr = inputs[op.attrs["num_refs"]] if op.attrs.get("lr_input") else None
    powers = [_local_f32(ctx, p) for p in power_ops]
    if all(t is not None for t in powers) and (
            lr is None or (_is_f32(lr) and lr.is_cuda and lr.numel() == 1)):
        if ctx.sharded_apply(op.attrs["rule"], op.attrs["hyper"], [var_op], [inputs[-1]],
                             [slot_ops], powers, lr):
            return [_local_f32(ctx, var_op)]
    grad = ctx.reduce_gradients([inputs[-1]], [var_op])[0]
    rest = _optim_launch(ctx, op.attrs["rule"], op.attrs["hyper"], [var_op], [slot_ops],
                         power_ops, [grad], lr)
    for v, s_ops, g in rest:       # state in another process (parameter server)
        return [apply_rule_through_stores(ctx, op.attrs["rule"], op.attrs["hyper"], v, s_ops,
                                          power_ops, g, lr)]
    return [_local_f32(ctx, var_op)]


@hip_kernel("MultiApplyOptimizer")
def _multi_apply_optimizer(op, inputs, ctx):
    """Every variable of the step in one launch, then one one-wave launch for the scalars that
    advance with the step (Adam's running powers, step counters)."""
    from .ref_kernels import apply_rule_through_stores
    var_ops = list(op.attrs["var_ops"])
    slot_ops = list(op.attrs["slot_ops"])
    lr = inputs[len(var_ops)] if op.attrs.get("lr_input") else None
    grads = list(inputs[:len(var_ops)])
    # reduce_scatter_sync: reduce-scatter, the rule on the owned shard (state lives there
    # only), all-gather
    powers = [_local_f32(ctx, p) for p in op.attrs["power_ops"]]
    if all(t is not None for t in powers) and (
            lr is None or (_is_f32(lr) and lr.is_cuda and lr.numel() == 1)):
        done = ctx.sharded_apply(op.attrs["rule"], op.attrs["hyper"], var_ops, grads, slot_ops,
                                 powers, lr)
        if done:
            keep = [i for i in range(len(var_ops)) if i not in set(done)]
            var_ops = [var_ops[i] for i in keep]
            slot_ops = [slot_ops[i] for i in keep]
            grads = [grads[i] for i in keep]
    if var_ops:
        grads = ctx.reduce_gradients(grads, var_ops)   # no-op without a sync runtime
    rest = _optim_launch(ctx, op.attrs["rule"], op.attrs["hyper"], var_ops, slot_ops,
                         op.attrs["power_ops"], grads, lr)
    for v, s_ops, g in rest:
        apply_rule_through_stores(ctx, op.attrs["rule"], op.attrs["hyper"], v, s_ops,
                                  op.attrs["power_ops"], g, lr)
    _affine_launch(ctx, op.attrs.get("scalar_updates", []))
    return []


@hip_kernel("MultiMovingAverage")
def _multi_moving_average(op, inputs, ctx):
    """Every moving average of the step in one launch of the multi-tensor optimizer kernel (rule
    ``ema``: the shadow is the updated tensor, the variable takes the gradient's place); shadows
    or variables that are not local fp32 GPU storage go through their stores."""
    rate = inputs[0] if inputs else None
    const = op.attrs["rate"]
    if rate is not None and not (_is_f32(rate) and rate.is_cuda and rate.numel() == 1):
        return _fallback("MultiMovingAverage", op, inputs, ctx)
    shadows, sources, rest = [], [], []
    for shadow_op, var_op in zip(op.attrs["shadow_ops"], op.attrs["var_ops"]):
        s, v = _local_f32(ctx, shadow_op), _local_f32(ctx, var_op)
        if s is None or v is None or s.numel() != v.numel():
            rest.append((shadow_op, var_op))
        else:
            shadows.append(s)
            sources.append(v)
    if shadows:
        hip.multi_optim("ema", shadows, sources, (), hyper=(0.0 if const is None else const,),
                        lr=rate)
    for shadow_op, var_op in rest:
        shadow, var = ctx.var_read(shadow_op), ctx.var_read(var_op)
        r = rate if rate is not None else const
        ctx.var_assign_add(shadow_op, -((shadow - var) * r))
    return []


@hip_kernel("AffineUpdateVariable")
def _affine_update_variable(op, inputs, ctx):
    var_op = op.inputs[0].op
    _affine_launch(ctx, [(var_op, op.attrs["factor"], op.attrs["increment"])])
    p = _local_f32(ctx, var_op)
    return [p if p is not None else ctx.var_read(var_op)]


def _spans_overlap(a: torch.Tensor, b: torch.Tensor) -> bool:
    a0, b0 = a.data_ptr(), b.data_ptr()
    return a0 < b0 + b.numel() * b.element_size() and b0 < a0 + a.numel() * a.element_size()


@hip_kernel("MLPTrainStep")
def _mlp_train_step(op, inputs, ctx):
    """The whole training step of the two-layer classifier. Small shapes (batch and hidden layer
    up to 128, up to 16 classes) run the three-launch step (``csrc/kernels/mlp_step.hip``):
    inside a captured step it reads the batch through the step ring when the session left the
    feeds ungathered, and it takes over the gather of the step's small results when asked to.
    Other shapes run the same math as the chain of general kernels."""
    x, labels = inputs
    var_ops, lrs = op.attrs["var_ops"], op.attrs["learning_rates"]
    params = []
    for v in var_ops:
        store = ctx.session._store_for(v)
        fast = getattr(store, "local_tensor", None)
        p = fast(v.name) if fast is not None else None
        if p is None or p.dtype != _f32 or not p.is_cuda or not p.is_contiguous():
            raise hip.HipKernelError("MLPTrainStep: variable %s is not local fp32 GPU storage"
                                     % v.name)
        params.append(p)
        store.update_count += 1
    counters, deltas = [], []
    for v, delta in op.attrs.get("counter_increments", []):
        store = ctx.session._store_for(v)
        fast = getattr(store, "local_tensor", None)
        t = fast(v.name) if fast is not None else None
        if t is None or t.dtype != _f32 or not t.is_cuda or t.numel() != 1 or len(counters) >= 4:
            raise hip.HipKernelError("MLPTrainStep: counter %s is not a local fp32 GPU scalar"
                                     % v.name)
        counters.append(t)
        deltas.append(delta)
    if not _is_f32(x, labels) or x.dim() != 2 or labels.dim() != 2:
        return _fallback("MLPTrainStep", op, inputs, ctx)
    w1, b1, w2, b2 = params
    b, d = x.shape
    h, c = w2.shape
    want_acc = bool(op.attrs.get("emit_accuracy"))
    dloss = float(op.attrs["dloss_const"])
    if hip.mlp_step_supported(b, d, h, c):
        key = (op, "scratch", b)
        scratch = ctx.buffers.get(key)
        if scratch is None:
            scratch = ctx.buffers[key] = hip.mlp_step_scratch(b, d, h, c, ctx.device)
        outs = [scratch["loss"], scratch["y"]] + ([scratch["accuracy"]] if want_acc else [])
        ring = ctx.ring
        pack, req, claimed = (), ctx.pack_request, False
        if ring is not None and req is not None:
            # the step's small results ride on the last launch unless one of them is a parameter
            # that launch is still updating in other workgroups
            parts = req.parts_for(op, outs)
            if parts is not None and not any(_spans_overlap(t, p) for t in parts
                                             for p in (w2, b2, b1, w1)):
                pack, claimed = parts, True
                req.claim()
        xi = li = -1
        if ring is not None:
            xi = ctx.ring_feeds.get(x.data_ptr(), -1)
            li = ctx.ring_feeds.get(labels.data_ptr(), -1)
        block = hip.mlp_step_train(
            x, labels, w1, b1, w2, b2, lrs, scratch, (b, d, h, c), dloss=dloss,
            want_accuracy=want_acc, want_softmax=True, counters=counters, deltas=deltas,
            pack=pack, ring=ring if (claimed or xi >= 0 or li >= 0) else None,
            x_index=xi, labels_index=li, ring_last=claimed)
        if claimed and xi >= 0 and li >= 0:
            # the whole captured step is this one library call: the session may issue it directly
            import ctypes
            ctx.direct_call = (hip.lib().dtf_mlp_step_train, ctypes.byref(block), block, xi, li)
            ctx.buffers[(op, "call", b)] = block       # the block outlives this call
        return outs
    # larger batch or layers: the chain of general kernels
    key = ("scalar", dloss)
    dl = ctx.const_cache.get(key)
    if dl is None:
        dl = hip.fill((), dloss, ctx.device)
        if not torch.cuda.is_current_stream_capturing():
            ctx.const_cache[key] = dl
    a = hip.gemm_f32(x, w1, bias=b1, activation=hip.ACT_SIGMOID)
    z = hip.gemm_f32(a, w2, bias=b2)
    loss, y, _ = hip.softmax_xent(z, labels)
    dz, db2 = hip.softmax_xent_grad(y, labels, dl, True)
    dw2 = hip.gemm_f32(a, dz, trans_a=True)
    da = hip.gemm_f32(dz, w2, trans_b=True)
    dz1, db1 = hip.sigmoid_grad_bias(da, a)
    dw1 = hip.gemm_f32(x, dz1, trans_a=True)
    acc = hip.argmax_accuracy(y, labels) if want_acc else None
    hip.multi_sgd([w1, b1, w2, b2], [dw1, db1, dw2, db2], lrs, counters, deltas)
    return [loss, y] + ([acc] if want_acc else [])


@hip_kernel("MLPStepGrads")
def _mlp_step_grads(op, inputs, ctx):
    """Forward and backward of the two-layer classifier ending in its four gradients: the three
    launches of ``csrc/kernels/mlp_step.hip`` in gradient mode. The gradients are written where
    the session placed them (the slots of the collective runtime's flat gradient arena) or into
    buffers of the node; inside a captured step the batch is read through the step ring and the
    step's small results ride on the third launch when the session asks for that. Shapes beyond
    the small-step kernels run the chain of general kernels."""
    x, labels = inputs
    params = []
    for v in op.attrs["var_ops"]:
        p = _local_f32(ctx, v)
        if p is None:
            raise hip.HipKernelError("MLPStepGrads: variable %s is not local fp32 GPU storage"
                                     % v.name)
        params.append(p)
    if not _is_f32(x, labels) or x.dim() != 2 or labels.dim() != 2:
        return _fallback("MLPStepGrads", op, inputs, ctx)
    w1, b1, w2, b2 = params
    b, d = x.shape
    h, c = w2.shape
    want_acc = bool(op.attrs.get("emit_accuracy"))
    dloss = float(op.attrs["dloss_const"])
    base = 3 if want_acc else 2
    dests = []
    for i, p in enumerate(params):
        t = _hint(ctx, base + i, p.shape)
        if t is None:
            key = (op, "grad", i, b)
            t = ctx.buffers.get(key)
            if t is None or t.shape != p.shape:
                t = ctx.buffers[key] = torch.empty_like(p)
        dests.append(t)
    if hip.mlp_step_supported(b, d, h, c):
        key = (op, "scratch", b)
        scratch = ctx.buffers.get(key)
        if scratch is None:
            scratch = ctx.buffers[key] = hip.mlp_step_scratch(b, d, h, c, ctx.device)
        outs = [scratch["loss"], scratch["y"]] + ([scratch["accuracy"]] if want_acc else [])
        ring = ctx.ring
        pack, adds, req, claimed = (), None, ctx.pack_request, False
        if ring is not None and req is not None:
            parts = req.parts_for(op, outs + dests)
            # (a gathered value must not be one the launch is still writing in other workgroups)
            if parts is not None and not any(_spans_overlap(t, g) for t in parts for g in dests):
                adds = req.addends(parts)
                if adds is not None:
                    pack, claimed = parts, True
                    req.claim()
        xi = li = -1
        if ring is not None:
            xi = ctx.ring_feeds.get(x.data_ptr(), -1)
            li = ctx.ring_feeds.get(labels.data_ptr(), -1)
        hip.mlp_step_train(
            x, labels, w1, b1, w2, b2, (0.0, 0.0, 0.0, 0.0), scratch, (b, d, h, c), dloss=dloss,
            want_accuracy=want_acc, want_softmax=True, pack=pack,
            ring=ring if (claimed or xi >= 0 or li >= 0) else None,
            x_index=xi, labels_index=li, ring_last=claimed, grads=dests,
            pack_adds=adds if claimed else None)
        return outs + dests
    key = ("scalar", dloss)
    dl = ctx.const_cache.get(key)
    if dl is None:
        dl = hip.fill((), dloss, ctx.device)
        if not torch.cuda.is_current_stream_capturing():
            ctx.const_cache[key] = dl
    a = hip.gemm_f32(x, w1, bias=b1, activation=hip.ACT_SIGMOID)
    z = hip.gemm_f32(a, w2, bias=b2)
    loss, y, _ = hip.softmax_xent(z, labels)
    dz, db2 = hip.softmax_xent_grad(y, labels, dl, True)
    dw2 = hip.gemm_f32(a, dz, trans_a=True, out=dests[2])
    da = hip.gemm_f32(dz, w2, trans_b=True)
    dz1, db1 = hip.sigmoid_grad_bias(da, a)
    dw1 = hip.gemm_f32(x, dz1, trans_a=True, out=dests[0])
    acc = hip.argmax_accuracy(y, labels) if want_acc else None
    return [loss, y] + ([acc] if want_acc else []) + [dw1, db1, dw2, db2]


@hip_kernel("ApplyGradientDescent")
def _apply_sgd(op, inputs, ctx):
    var_op = op.inputs[0].op
    store = ctx.session._store_for(var_op)
    fast = getattr(store, "local_tensor", None)
    p = fast(var_op.name) if fast is not None else None
    g = ctx.reduce_gradients([inputs[1]], [var_op])[0]
    if p is not None and p.dtype == _f32 and g.dtype == _f32 and p.is_cuda:
        hip.multi_sgd([p], [g], [op.attrs["learning_rate"]])
        store.update_count += 1
        return [p]
    return [ctx.var_apply_sgd(var_op, op.attrs["learning_rate"], g)]
