"""Plan-time graph fusion: swap op chains for the fused kernels the GPU actually runs.

A whole MNIST training step (reference ``tf_distributed.py:62-76``) is ~48 MFLOP, so on an
MI355X it is bounded by kernel count, not arithmetic. The planner therefore rewrites the
pruned subgraph before execution:

* ``act(add(matmul(x, W), b))``                      -> ``FusedDense`` (bias + activation in the
  GEMM epilogue);
* ``reduce_mean(-reduce_sum(labels * log(softmax(z)), 1))`` -> ``SoftmaxXentMean`` (log-softmax
  form; also yields the softmax for the accuracy branch);
* the backward chain from d(loss) to d(logits)        -> ``SoftmaxXentMeanGrad`` (emits the
  logits gradient and the bias gradient);
* ``SigmoidGrad`` + the bias-gradient column sum      -> ``DenseGradSigmoid``;
* ``reduce_mean(cast(equal(argmax(y,1), argmax(y_,1))))``  -> ``ArgmaxAccuracy``;
* a linear output layer + softmax cross-entropy + their whole backward (logits gradient, weight
  and bias gradients, the gradient into -- and through -- a preceding sigmoid layer, and the
  accuracy of the same predictions) -> ``ClassifierHeadTrain``, one kernel for small heads;
* every ``ApplyGradientDescent`` (+ the ``global_step`` increment) -> one ``MultiApplySGD``;
* a whole SGD step of a two-layer sigmoid/softmax classifier whose four parameters live on this
  device (first layer, head, weight-gradient product and update -- the reference's model and
  train_op, ``tf_distributed.py:58-76``) -> one ``MLPTrainStep`` node: three launches, the
  first-layer weight gradient applied in the epilogue of its own product.

Rewrites are expressed as *tensor replacements* (``original tensor -> output k of a fused
node``) and *op replacements* (for control edges). The original ops stay in the graph; after
re-pruning from the fetches the replaced chains are simply unreachable. ``graph.grad_map``
(filled by ``gradients``) tells which tensor holds d(loss)/d(t), which is what lets whole
backward chains be replaced without pattern-matching their internals.
"""
from __future__ import annotations

import os

from typing import Any, Dict, List, Optional, Sequence, Set, Tuple

from ..framework import dtypes
from ..framework.graph import Graph, Operation, Tensor


class FusedNode:
    """A synthetic operation that exists only inside an execution plan."""

    __slots__ = ("type", "name", "inputs", "control_inputs", "attrs", "outputs", "device",
                 "graph", "id")

    def __init__(self, graph: Graph, op_type: str, name: str, inputs: Sequence[Tensor],
                 out_specs: Sequence[Tuple], attrs: Optional[dict] = None,
                 control_inputs: Sequence = (), device: str = ""):
        self.graph = graph
        self.type = op_type
        self.name = name
        self.inputs = list(inputs)
        self.control_inputs = list(control_inputs)
        self.attrs = dict(attrs or {})
        self.outputs = [Tensor(self, i, dt, shape) for i, (dt, shape) in enumerate(out_specs)]
        self.device = device
        self.id = -1

    def get_attr(self, key):
        return self.attrs[key]

    def __repr__(self):
        return "<FusedNode %r type=%s>" % (self.name, self.type)


class Rewrite:
    def __init__(self):
        self.tensor_map: Dict[Tensor, Tensor] = {}      # original tensor -> fused output
        self.op_map: Dict[Operation, FusedNode] = {}    # original op -> fused node (control)
        # a whole-step node that MAY replace several fused nodes at once: (node, tensor
        # replacements, op replacements, the nodes it makes redundant). The session applies it
        # and keeps it only if none of the redundant nodes is still reachable from the fetches.
        self.step_candidate = None

    def with_candidate(self) -> "Rewrite":
        node, tmap, omap, _internal = self.step_candidate
        rw = Rewrite()
        rw.tensor_map = dict(self.tensor_map)
        rw.tensor_map.update(tmap)
        rw.op_map = dict(self.op_map)
        rw.op_map.update(omap)
        return rw

    def resolve(self, t: Tensor) -> Tensor:
        seen = 0
        while t in self.tensor_map and seen < 8:
            t = self.tensor_map[t]
            seen += 1
        return t

    def resolve_op(self, op):
        return self.op_map.get(op, op)


def _consumers(ops: Sequence[Operation]) -> Dict[Tensor, List[Operation]]:
    out: Dict[Tensor, List[Operation]] = {}
    for op in ops:
        for t in op.inputs:
            out.setdefault(t, []).append(op)
    return out


_ACT_TYPES = {"Sigmoid": "sigmoid", "Relu": "relu", "Tanh": "tanh"}


def _match_dense(add_op: Operation) -> Optional[Tuple[Tensor, Tensor, Tensor, Operation]]:
    """``Add(MatMul(x, W), b)`` with b rank-1 -> (x, W, b, matmul_op)."""
    if add_op.type != "Add":
        return None
    for mm_t, b_t in (add_op.inputs, reversed(add_op.inputs)):
        mm = mm_t.op
        if mm.type != "MatMul" or mm.attrs["transpose_a"] or mm.attrs["transpose_b"]:
            continue
        if b_t.shape is None or len(b_t.shape) != 1:
            continue
        if mm_t.dtype != dtypes.float32:
            continue
        return mm.inputs[0], mm.inputs[1], b_t, mm
    return None


def _match_xent(mean_op: Operation):
    """``Mean(Neg(Sum(Mul(labels, Log(Softmax(z))), axis=last)))`` -> (z, labels, softmax_t)."""
    if mean_op.type != "Mean" or mean_op.attrs["axis"] is not None:
        return None
    neg = mean_op.inputs[0].op
    if neg.type != "Neg":
        return None
    sm = neg.inputs[0].op
    if sm.type != "Sum" or sm.attrs["keepdims"]:
        return None
    mul = sm.inputs[0].op
    if mul.type != "Mul":
        return None
    rank = None if mul.outputs[0].shape is None else len(mul.outputs[0].shape)
    if rank != 2 or sm.attrs["axis"] != (1,):
        return None
    for labels_t, log_t in (mul.inputs, reversed(mul.inputs)):
        lg = log_t.op
        if lg.type != "Log":
            continue
        soft = lg.inputs[0].op
        if soft.type != "Softmax":
            continue
        if soft.outputs[0].dtype != dtypes.float32:
            continue
        return soft.inputs[0], labels_t, soft.outputs[0]
    return None


def _match_accuracy(mean_op: Operation):
    """``Mean(Cast(Equal(ArgMax(a, 1), ArgMax(b, 1)), float32))`` over rank-2 fp32 -> (a, b)."""
    if mean_op.type != "Mean" or mean_op.attrs["axis"] is not None:
        return None
    cast = mean_op.inputs[0].op
    if cast.type != "Cast" or cast.outputs[0].dtype != dtypes.float32:
        return None
    eq = cast.inputs[0].op
    if eq.type != "Equal":
        return None
    srcs = []
    for t in eq.inputs:
        am = t.op
        if am.type != "ArgMax" or am.attrs["axis"] != 1:
            return None
        x = am.inputs[0]
        if x.dtype != dtypes.float32 or x.shape is None or len(x.shape) != 2:
            return None
        srcs.append(x)
    return srcs[0], srcs[1]


_HEAD_MAX_HIDDEN, _HEAD_MAX_CLASSES = 128, 16


def _match_head(dense, logits, dlogits_t, emit_bias, grad_map, alive, fed, rw, consumers):
    """The pieces around a fused softmax cross-entropy that make it a whole classifier head:
    ``logits = FusedDense(a, W, b)`` without activation, ``dW = MatMul(a, dlogits, transpose_a)``
    and ``da = MatMul(dlogits, W, transpose_b)``; when ``a`` is itself the output of a sigmoid
    ``FusedDense`` whose ``DenseGradSigmoid`` consumes exactly ``da``, that node is absorbed too.
    Returns ``(a, W, dW, da, (dz_prev, db_prev) | None)`` or ``None``."""
    if dense is None or not emit_bias or dense.attrs["activation"] != "none":
        return None
    a_t, w_t, _ = dense.inputs
    if logits in fed:
        return None
    if w_t.shape is None or len(w_t.shape) != 2 or None in w_t.shape:
        return None
    if w_t.shape[0] > _HEAD_MAX_HIDDEN or w_t.shape[1] > _HEAD_MAX_CLASSES:
        return None
    dw_t, da_t = grad_map.get(w_t), grad_map.get(a_t)
    if dw_t is None or da_t is None or dw_t.op not in alive or da_t.op not in alive:
        return None
    dw_op, da_op = dw_t.op, da_t.op
    if dw_op.type != "MatMul" or da_op.type != "MatMul":
        return None
    if not (dw_op.attrs["transpose_a"] and not dw_op.attrs["transpose_b"]
            and dw_op.inputs[0] is a_t and dw_op.inputs[1] is dlogits_t):
        return None
    if not (da_op.attrs["transpose_b"] and not da_op.attrs["transpose_a"]
            and da_op.inputs[0] is dlogits_t and da_op.inputs[1] is w_t):
        return None
    # dlogits itself must have no other reader than dW, da and the bias gradient
    for c in consumers.get(dlogits_t, []):
        if c in alive and c is not dw_op and c is not da_op and c.type != "SumToShapeOf":
            return None
    sig = None
    mapped = rw.tensor_map.get(a_t)
    if mapped is not None and getattr(mapped.op, "type", "") == "FusedDense" \
            and mapped.op.attrs["activation"] == "sigmoid":
        z_prev = mapped.op.attrs["_pre_activation"]
        b_prev = mapped.op.attrs["_bias"]
        dz_prev, db_prev = grad_map.get(z_prev), grad_map.get(b_prev)
        if dz_prev is not None and db_prev is not None:
            g = rw.tensor_map.get(dz_prev)
            if g is not None and getattr(g.op, "type", "") == "DenseGradSigmoid" \
                    and g.op.inputs[0] is da_t and rw.tensor_map.get(db_prev) is g.op.outputs[1] \
                    and all(c is dz_prev.op for c in consumers.get(da_t, []) if c in alive):
                sig = (dz_prev, db_prev)
    return a_t, w_t, dw_t, da_t, sig


def _plan_sharding(graph: Graph, ops: Sequence[Operation], fed: Set[Tensor],
                   fetched: Set[Tensor], consumers, rw: Rewrite, shard_plan, stored_layouts) -> None:
    """Cross-device edges of sharded products: the planned ``MatMul`` ops become
    ``ShardedMatMul`` nodes (local block product + one collective), the initialisers of
    variables stored as shards become ``ShardedVariableInit`` (the rank's own window of the
    random stream), and any other reader of such a variable -- a fetch, an op outside the plan --
    reads it through ``GatherShardedVariable`` (an all-gather of the blocks)."""
    alive = set(ops)
    sharded_ops = set()
    if shard_plan:
        for op in ops:
            spec = shard_plan.matmuls.get(op)
            if spec is not None and op.outputs[0] not in fed:
                out = op.outputs[0]
                node = FusedNode(graph, "ShardedMatMul", op.name + "/sharded", list(op.inputs),
                                 [(out.dtype, out.shape)],
                                 {"shard": spec, "world": shard_plan.world, "_raw_inputs": True},
                                 control_inputs=op.control_inputs, device=op.device)
                rw.tensor_map[out] = node.outputs[0]
                rw.op_map[op] = node          # a fetch of the OP (or a control edge) runs it too
                sharded_ops.add(op)
            lay = shard_plan.init_of.get(op)
            if lay is not None and op.type == "Assign":
                out = op.outputs[0]
                node = FusedNode(graph, "ShardedVariableInit", op.name + "/sharded_init", [],
                                 [(out.dtype, lay.local_shape())],
                                 {"layout": lay, "var_op": op.inputs[0].op},
                                 control_inputs=op.control_inputs, device=op.device)
                rw.op_map[op] = node
                rw.tensor_map[out] = node.outputs[0]
    for op in ops:
        lay = stored_layouts.get(op.name) if (stored_layouts and op.type == "VariableV2") else None
        if lay is None:
            continue
        t = op.outputs[0]
        outside = [c for c in consumers.get(t, []) if c in alive and c not in sharded_ops
                   and c.type not in ("Assign", "AssignAdd", "AssignSub", "ApplyGradientDescent",
                                      "IsVariableInitialized", "ReportUninitializedVariables")]
        if outside or t in fetched:
            node = FusedNode(graph, "GatherShardedVariable", op.name + "/gather", [t],
                             [(t.dtype, t.shape)], {"layout": lay, "_raw_inputs": True},
                             device=op.device)
            rw.tensor_map[t] = node.outputs[0]


def _match_mlp_step(graph, alive, fed, fetched, rw, heads, sgd_node, applies):
    """The whole training step of a two-layer classifier: ``FusedDense(sigmoid)`` feeding a
    ``ClassifierHeadTrain`` that carries the gradient back through the sigmoid, the first layer's
    weight gradient ``MatMul(x, dz1, transpose_a)`` and a ``MultiApplySGD`` over exactly the four
    parameters. Returns ``(node, tensor replacements, op replacements, redundant nodes)``."""
    if sgd_node is None or len(heads) != 1 or len(applies) != 4:
        return None
    hnode, loss_t, soft_t, acc_t = heads[0]
    if not hnode.attrs["sigmoid_input"] or hnode.attrs["dloss_const"] is None \
            or len(hnode.inputs) != 4:
        return None
    a_t, w2_t, b2_t, labels = hnode.inputs
    mapped = rw.tensor_map.get(a_t)
    if mapped is None or getattr(mapped.op, "type", "") != "FusedDense":
        return None
    dense = mapped.op
    x, w1_t, b1_t = dense.inputs
    params = [w1_t, b1_t, w2_t, b2_t]
    if any(t.op.type != "VariableV2" or t.dtype != dtypes.float32 or t in fed for t in params):
        return None
    if len({t.op for t in params}) != 4 or x.dtype != dtypes.float32 \
            or labels.dtype != dtypes.float32:
        return None
    by_var = {a.inputs[0].op: a for a in applies}
    if set(by_var) != {t.op for t in params}:
        return None
    g_w1 = by_var[w1_t.op].inputs[1]
    dw1_op = g_w1.op
    if dw1_op not in alive or dw1_op.type != "MatMul" or g_w1 in fetched or g_w1 in fed \
            or not dw1_op.attrs["transpose_a"] or dw1_op.attrs["transpose_b"] \
            or dw1_op.inputs[0] is not x \
            or rw.tensor_map.get(dw1_op.inputs[1]) is not hnode.outputs[4]:
        return None
    if rw.tensor_map.get(by_var[b1_t.op].inputs[1]) is not hnode.outputs[5] \
            or rw.tensor_map.get(by_var[w2_t.op].inputs[1]) is not hnode.outputs[2] \
            or rw.tensor_map.get(by_var[b2_t.op].inputs[1]) is not hnode.outputs[3]:
        return None
    outs = [(loss_t.dtype, loss_t.shape), (soft_t.dtype, soft_t.shape)]
    if acc_t is not None:
        outs.append((acc_t.dtype, acc_t.shape))
    node = FusedNode(graph, "MLPTrainStep", "fused_train_step", [x, labels], outs,
                     {"var_ops": [t.op for t in params],
                      "learning_rates": [by_var[t.op].attrs["learning_rate"] for t in params],
                      "counter_increments": list(sgd_node.attrs["counter_increments"]),
                      "dloss_const": float(hnode.attrs["dloss_const"]),
                      "emit_accuracy": acc_t is not None,
                      "parts": (dense, hnode, dw1_op, sgd_node)},
                     control_inputs=sgd_node.control_inputs, device=sgd_node.device)
    tmap = {loss_t: node.outputs[0], soft_t: node.outputs[1]}
    if acc_t is not None:
        tmap[acc_t] = node.outputs[2]
    omap = {op: node for op, target in rw.op_map.items() if target is sgd_node}
    return node, tmap, omap, (dense, hnode, dw1_op, sgd_node)


def _grad_input(apply_op) -> Tensor:
    return apply_op.inputs[1] if apply_op.type == "ApplyGradientDescent" else apply_op.inputs[-1]


def _match_mlp_grads(graph, alive, fed, fetched, rw, heads, applies):
    """Forward and backward of the two-layer classifier as ONE node that ends in the four
    gradients -- for steps whose update is not the whole-step node's: several ranks that exchange
    gradients first, or an optimizer rule with state. Replaces ``FusedDense(sigmoid)``, the
    ``ClassifierHeadTrain`` that carries the gradient back through the sigmoid and the first
    layer's weight-gradient ``MatMul``; the optimizer node stays and reads the node's outputs.
    Returns ``(node, tensor replacements, op replacements, redundant nodes)``."""
    if len(heads) != 1 or len(applies) < 4:
        return None
    hnode, loss_t, soft_t, acc_t = heads[0]
    if not hnode.attrs["sigmoid_input"] or hnode.attrs["dloss_const"] is None \
            or len(hnode.inputs) != 4:
        return None
    a_t, w2_t, b2_t, labels = hnode.inputs
    mapped = rw.tensor_map.get(a_t)
    if mapped is None or getattr(mapped.op, "type", "") != "FusedDense":
        return None
    dense = mapped.op
    x, w1_t, b1_t = dense.inputs
    params = [w1_t, b1_t, w2_t, b2_t]
    if any(t.op.type != "VariableV2" or t.dtype != dtypes.float32 or t in fed for t in params):
        return None
    if len({t.op for t in params}) != 4 or x.dtype != dtypes.float32 \
            or labels.dtype != dtypes.float32:
        return None
    by_var = {a.inputs[0].op: a for a in applies}
    if not {t.op for t in params} <= set(by_var):
        return None
    grads = [_grad_input(by_var[t.op]) for t in params]
    g_w1 = grads[0]
    dw1_op = g_w1.op
    if dw1_op not in alive or dw1_op.type != "MatMul" or g_w1 in fed \
            or not dw1_op.attrs["transpose_a"] or dw1_op.attrs["transpose_b"] \
            or dw1_op.inputs[0] is not x \
            or rw.tensor_map.get(dw1_op.inputs[1]) is not hnode.outputs[4]:
        return None
    if rw.tensor_map.get(grads[1]) is not hnode.outputs[5] \
            or rw.tensor_map.get(grads[2]) is not hnode.outputs[2] \
            or rw.tensor_map.get(grads[3]) is not hnode.outputs[3]:
        return None
    if any(g in fed for g in grads):
        return None
    outs = [(loss_t.dtype, loss_t.shape), (soft_t.dtype, soft_t.shape)]
    if acc_t is not None:
        outs.append((acc_t.dtype, acc_t.shape))
    base = len(outs)
    outs += [(g.dtype, t.shape) for g, t in zip(grads, params)]
    node = FusedNode(graph, "MLPStepGrads", "fused_step_grads", [x, labels], outs,
                     {"var_ops": [t.op for t in params],
                      "dloss_const": float(hnode.attrs["dloss_const"]),
                      "emit_accuracy": acc_t is not None,
                      "parts": (dense, hnode, dw1_op)},
                     device=hnode.device)
    tmap = {loss_t: node.outputs[0], soft_t: node.outputs[1]}
    if acc_t is not None:
        tmap[acc_t] = node.outputs[2]
    for i, g in enumerate(grads):
        tmap[g] = node.outputs[base + i]
    return node, tmap, {}, (dense, hnode, dw1_op)


def build_rewrite(graph: Graph, ops: Sequence[Operation], fed: Set[Tensor],
                  fetched: Set[Tensor] = frozenset(), shard_plan=None,
                  stored_layouts=None, fuse: bool = True, step_fusion: bool = False,
                  grads_fusion: bool = False) -> Rewrite:
    """Find fusable patterns among ``ops`` (the pruned, topologically ordered subgraph)."""
    rw = Rewrite()
    alive = set(ops)
    consumers = _consumers(ops)
    if shard_plan or stored_layouts:
        _plan_sharding(graph, ops, fed, fetched, consumers, rw, shard_plan, stored_layouts or {})
    if not fuse:
        return rw
    grad_map = graph.grad_map
    dense_of: Dict[Tensor, FusedNode] = {}     # dense output tensor -> its FusedDense node
    heads: List[Tuple] = []                    # (ClassifierHeadTrain node, loss, softmax, accuracy)

    def only_consumer(t: Tensor, op: Operation) -> bool:
        return all(c is op for c in consumers.get(t, []))

    # -- forward: dense layers ----------------------------------------------------------
    for op in ops:
        hit = _match_dense(op) if op.type == "Add" else None
        if hit is None or op.outputs[0] in fed:
            continue
        x, w, b, mm = hit
        if mm.outputs[0] in fed or not only_consumer(mm.outputs[0], op):
            continue
        if shard_plan and mm in shard_plan.matmuls:
            continue                   # the product runs sharded; bias and activation follow it
        z = op.outputs[0]
        act_name, act_op = "none", None
        users = [c for c in consumers.get(z, []) if c in alive]
        # fold the activation into the GEMM epilogue when nothing else needs the
        # pre-activation (activation gradients are written in terms of the activation output)
        if (len(users) == 1 and users[0].type in _ACT_TYPES and z not in fetched
                and users[0].outputs[0] not in fed):
            act_op = users[0]
            act_name = _ACT_TYPES[act_op.type]
        out_t = act_op.outputs[0] if act_op is not None else z
        node = FusedNode(graph, "FusedDense", out_t.op.name + "/fused_dense", [x, w, b],
                         [(out_t.dtype, out_t.shape)], {"activation": act_name},
                         device=op.device)
        rw.tensor_map[out_t] = node.outputs[0]
        dense_of[out_t] = node
        node.attrs["_pre_activation"] = z
        node.attrs["_bias"] = b

        # -- backward of a sigmoid dense: dz = da*a*(1-a), db = colsum(dz) ---------------
        if act_name == "sigmoid":
            dz_t = grad_map.get(z)
            db_t = grad_map.get(b)
            da_t = grad_map.get(out_t)
            if (dz_t is not None and da_t is not None and dz_t.op in alive
                    and dz_t.op.type == "SigmoidGrad"):
                outs = [(dz_t.dtype, dz_t.shape)]
                fuse_bias = (db_t is not None and db_t.op in alive
                             and db_t.op.type == "SumToShapeOf" and db_t.op.inputs[0] is dz_t)
                if fuse_bias:
                    outs.append((db_t.dtype, db_t.shape))
                    gnode = FusedNode(graph, "DenseGradSigmoid", z.op.name + "/fused_grad",
                                      [da_t, out_t], outs, device=op.device)
                    rw.tensor_map[dz_t] = gnode.outputs[0]
                    rw.tensor_map[db_t] = gnode.outputs[1]

    # -- forward + backward: softmax cross-entropy ----------------------------------------
    for op in ops:
        hit = _match_xent(op) if op.type == "Mean" else None
        if hit is None:
            continue
        logits, labels, soft_t = hit
        loss_t = op.outputs[0]
        node = FusedNode(graph, "SoftmaxXentMean", op.name + "/fused_xent", [logits, labels],
                         [(loss_t.dtype, loss_t.shape), (soft_t.dtype, soft_t.shape)],
                         device=op.device)
        rw.tensor_map[loss_t] = node.outputs[0]
        rw.tensor_map[soft_t] = node.outputs[1]
        dlogits_t = grad_map.get(logits)
        dloss_t = grad_map.get(loss_t)
        if dlogits_t is not None and dloss_t is not None and dlogits_t.op in alive:
            outs = [(dlogits_t.dtype, dlogits_t.shape)]
            bias_t = None
            dense = dense_of.get(logits)
            if dense is not None:
                bias_t = dense.attrs["_bias"]
            db_t = grad_map.get(bias_t) if bias_t is not None else None
            emit_bias = (db_t is not None and db_t.op in alive
                         and db_t.op.type == "SumToShapeOf" and db_t.op.inputs[0] is dlogits_t)
            if emit_bias:
                outs.append((db_t.dtype, db_t.shape))
            gnode = FusedNode(graph, "SoftmaxXentMeanGrad", op.name + "/fused_xent_grad",
                              [soft_t, labels, dloss_t], outs,
                              {"emit_bias_grad": emit_bias}, device=op.device)
            rw.tensor_map[dlogits_t] = gnode.outputs[0]
            if emit_bias:
                rw.tensor_map[db_t] = gnode.outputs[1]

            # -- whole classifier head: dense + xent forward and backward in one node --------
            head = _match_head(dense, logits, dlogits_t, emit_bias, grad_map, alive, fed, rw,
                               consumers)
            if head is not None:
                a_t, w_t, dw_t, da_t, sig = head
                outs = [(loss_t.dtype, loss_t.shape), (soft_t.dtype, soft_t.shape),
                        (dw_t.dtype, dw_t.shape), (db_t.dtype, db_t.shape)]
                prev_dz = prev_db = None
                if sig is not None:
                    prev_dz, prev_db = sig
                    outs += [(prev_dz.dtype, prev_dz.shape), (prev_db.dtype, prev_db.shape)]
                else:
                    outs.append((da_t.dtype, da_t.shape))
                acc_op = None
                for cand in ops:
                    acc_hit = _match_accuracy(cand) if cand.type == "Mean" else None
                    if acc_hit is not None and acc_hit[0] is soft_t and acc_hit[1] is labels:
                        acc_op = cand
                        outs.append((cand.outputs[0].dtype, cand.outputs[0].shape))
                        break
                # d(loss) seeded by gradients() is ones_like(loss): a constant, and one that
                # would make the merged node depend on its own output
                head_inputs = [a_t, w_t, bias_t, labels]
                dloss_const = None
                if dloss_t.op.type == "OnesLike" and dloss_t.op.inputs[0] is loss_t:
                    dloss_const = 1.0
                else:
                    head_inputs.append(dloss_t)
                hnode = FusedNode(graph, "ClassifierHeadTrain", op.name + "/fused_head",
                                  head_inputs, outs,
                                  {"sigmoid_input": sig is not None,
                                   "emit_accuracy": acc_op is not None,
                                   "dloss_const": dloss_const}, device=op.device)
                rw.tensor_map[loss_t] = hnode.outputs[0]
                rw.tensor_map[soft_t] = hnode.outputs[1]
                rw.tensor_map[dw_t] = hnode.outputs[2]
                rw.tensor_map[db_t] = hnode.outputs[3]
                if sig is not None:
                    rw.tensor_map[prev_dz] = hnode.outputs[4]
                    rw.tensor_map[prev_db] = hnode.outputs[5]
                else:
                    rw.tensor_map[da_t] = hnode.outputs[4]
                if acc_op is not None:
                    rw.tensor_map[acc_op.outputs[0]] = hnode.outputs[len(outs) - 1]
                heads.append((hnode, loss_t, soft_t,
                              acc_op.outputs[0] if acc_op is not None else None))

    # -- mean squared error: loss, its gradient and the bias gradient behind it, one node ------
    for op in ops:
        if op.type != "MeanSquaredError" or op.outputs[0] in fed \
                or os.environ.get("DTF_NO_MSE_STEP"):           # (A/B against the separate ops)
            continue
        pred, labels = op.inputs
        loss_t = op.outputs[0]
        gops = [c for c in consumers.get(pred, []) if c in alive
                and c.type == "MeanSquaredErrorGrad" and c.inputs[0] is pred
                and c.inputs[1] is labels]
        if len(gops) != 1 or gops[0].outputs[0] in fed:
            continue
        gop = gops[0]
        dpred_t, g_t = gop.outputs[0], gop.inputs[2]
        g_const = None
        if g_t.op.type == "OnesLike" and g_t.op.inputs[0] is loss_t:
            g_const = 1.0
        elif g_t.op.type == "Fill" and g_t not in fed and not g_t.shape:
            g_const = float(g_t.op.attrs["value"])
        outs = [(loss_t.dtype, loss_t.shape), (dpred_t.dtype, dpred_t.shape)]
        db_t = None
        if pred.shape is not None and len(pred.shape) == 2 and pred.shape[1] is not None:
            for c in consumers.get(dpred_t, []):
                if c in alive and c.type == "SumToShapeOf" and c.inputs[0] is dpred_t \
                        and c.outputs[0].shape is not None \
                        and tuple(c.outputs[0].shape) == (pred.shape[1],) \
                        and c.outputs[0] not in fed:
                    db_t = c.outputs[0]
                    outs.append((db_t.dtype, db_t.shape))
                    break
        node = FusedNode(graph, "MSELossStep", op.name + "/fused_mse_step",
                         [pred, labels] + ([] if g_const is not None else [g_t]), outs,
                         {"g_const": g_const, "emit_bias_grad": db_t is not None},
                         device=op.device)
        rw.tensor_map[loss_t] = node.outputs[0]
        rw.tensor_map[dpred_t] = node.outputs[1]
        if db_t is not None:
            rw.tensor_map[db_t] = node.outputs[2]

    # -- metric: argmax-compare-mean --------------------------------------------------------
    for op in ops:
        hit = _match_accuracy(op) if op.type == "Mean" else None
        if hit is None or op.outputs[0] in rw.tensor_map:
            continue
        node = FusedNode(graph, "ArgmaxAccuracy", op.name + "/fused_accuracy", list(hit),
                         [(op.outputs[0].dtype, op.outputs[0].shape)], device=op.device)
        rw.tensor_map[op.outputs[0]] = node.outputs[0]

    # -- optimizer: one multi-tensor SGD launch --------------------------------------------
    applies = [op for op in ops if op.type == "ApplyGradientDescent"
               and not consumers.get(op.outputs[0]) and op.outputs[0] not in fetched]
    if len(applies) >= 1:
        var_ops = [a.inputs[0].op for a in applies]
        node = FusedNode(graph, "MultiApplySGD", "fused_sgd", [a.inputs[1] for a in applies], [],
                         {"var_ops": var_ops,
                          "learning_rates": [a.attrs["learning_rate"] for a in applies],
                          "counter_increments": []},
                         control_inputs=[c for a in applies for c in a.control_inputs],
                         device=applies[0].device)
        for a in applies:
            rw.op_map[a] = node
        apply_set = set(applies)
        for op in ops:
            if (op.type == "AssignAdd" and op.inputs[1].op.type in ("Const", "Fill")
                    and op.control_inputs and set(op.control_inputs) <= apply_set
                    and not consumers.get(op.outputs[0]) and op.outputs[0] not in fed
                    and op.outputs[0] not in fetched):
                src = op.inputs[1].op
                delta = src.attrs["value"]
                node.attrs["counter_increments"].append((op.inputs[0].op, float(delta)))
                rw.op_map[op] = node
        if step_fusion and not shard_plan and not stored_layouts:
            rw.step_candidate = _match_mlp_step(graph, alive, fed, fetched, rw, heads, node,
                                                applies)

    # -- optimizer rules with slots: one multi-tensor launch per (rule, constants) ------------
    groups: Dict[Any, List[Any]] = {}
    for op in ops:
        if op.type == "ApplyOptimizer" and not consumers.get(op.outputs[0]) \
                and op.outputs[0] not in fetched:
            lr_t = op.inputs[op.attrs["num_refs"]] if op.attrs.get("lr_input") else None
            groups.setdefault((op.attrs["rule"], op.attrs["hyper"], lr_t), []).append(op)
    for (rule, hyper, lr_t), members in groups.items():
        n_slots = {"rmsprop": 2, "adam": 2, "adadelta": 2, "sgd": 0}.get(rule, 1)
        refs = [[t.op for t in a.inputs[:a.attrs["num_refs"]]] for a in members]
        power_ops = refs[0][1 + n_slots:]
        if any(r[1 + n_slots:] != power_ops for r in refs):
            continue                          # (two Adam instances: leave their ops alone)
        node = FusedNode(graph, "MultiApplyOptimizer", "fused_" + rule,
                         [a.inputs[-1] for a in members] + ([lr_t] if lr_t is not None else []),
                         [],
                         {"var_ops": [r[0] for r in refs], "lr_input": lr_t is not None,
                          "slot_ops": [r[1:1 + n_slots] for r in refs],
                          "power_ops": power_ops, "rule": rule, "hyper": hyper,
                          "scalar_updates": []},
                         control_inputs=[c for a in members for c in a.control_inputs],
                         device=members[0].device)
        for a in members:
            rw.op_map[a] = node
        if len(groups) > 1 or applies:
            continue
        # a single rule updates the whole step: the scalars that advance behind it (running
        # powers, the step counter) ride on the node's second, one-wave launch
        member_set = set(members)
        tails = [op for op in ops if op.type == "AffineUpdateVariable" and op.control_inputs
                 and set(op.control_inputs) <= member_set
                 and not consumers.get(op.outputs[0]) and op.outputs[0] not in fetched]
        for op in tails:
            node.attrs["scalar_updates"].append(
                (op.inputs[0].op, float(op.attrs["factor"]), float(op.attrs["increment"])))
            rw.op_map[op] = node
        done = member_set | set(tails)
        for op in ops:
            if (op.type == "AssignAdd" and op.inputs[1].op.type in ("Const", "Fill")
                    and op.control_inputs and set(op.control_inputs) <= done
                    and not consumers.get(op.outputs[0]) and op.outputs[0] not in fed
                    and op.outputs[0] not in fetched
                    and op.inputs[0].op.outputs[0].dtype.name == "float32"):
                node.attrs["scalar_updates"].append(
                    (op.inputs[0].op, 1.0, float(op.inputs[1].op.attrs["value"])))
                rw.op_map[op] = node
    # -- moving averages: every shadow of a step in one multi-tensor launch -------------------
    # shadow -= rate * (shadow - variable), built by ExponentialMovingAverage.apply as
    # AssignSub(shadow, Mul(Sub(shadow, variable), rate)) per variable
    ema_groups: Dict[Any, List[Any]] = {}
    for op in ops:
        if op.type != "AssignSub" or consumers.get(op.outputs[0]) or op.outputs[0] in fetched:
            continue
        shadow_t, delta_t = op.inputs
        mul = delta_t.op
        if shadow_t.op.type != "VariableV2" or mul not in alive or mul.type != "Mul" \
                or not only_consumer(delta_t, op) or delta_t in fed or delta_t in fetched:
            continue
        sub_t, rate_t = mul.inputs
        sub = sub_t.op
        if sub not in alive or sub.type != "Sub" or sub.inputs[0] is not shadow_t \
                or not only_consumer(sub_t, mul) or sub_t in fed or sub_t in fetched:
            continue
        var_t = sub.inputs[1]
        if var_t.op.type != "VariableV2" or var_t.dtype != dtypes.float32 \
                or shadow_t.dtype != dtypes.float32 or var_t.shape != shadow_t.shape:
            continue
        if rate_t.shape not in ((), None) and tuple(rate_t.shape) != ():
            continue
        if rate_t.op.type == "Const" and rate_t not in fed:
            import numpy as _np
            key = ("const", float(_np.asarray(rate_t.op.attrs["value"]).reshape(-1)[0]))
        else:
            key = ("tensor", rate_t)
        ema_groups.setdefault(key, []).append((op, shadow_t.op, var_t.op, mul, sub))
    for (kind, rate), members in ema_groups.items():
        ctrl = []
        for a, _, _, mul, sub in members:
            for c in list(a.control_inputs) + list(mul.control_inputs) + list(sub.control_inputs):
                if c not in ctrl:
                    ctrl.append(c)
        node = FusedNode(graph, "MultiMovingAverage", "fused_moving_average",
                         [rate] if kind == "tensor" else [], [],
                         {"shadow_ops": [m[1] for m in members],
                          "var_ops": [m[2] for m in members],
                          "rate": rate if kind == "const" else None},
                         control_inputs=ctrl, device=members[0][0].device)
        for a, *_ in members:
            rw.op_map[a] = node

    if grads_fusion and rw.step_candidate is None and not shard_plan and not stored_layouts:
        every = list(applies) + [a for members in groups.values() for a in members]
        rw.step_candidate = _match_mlp_grads(graph, alive, fed, fetched, rw, heads, every)
    return rw
