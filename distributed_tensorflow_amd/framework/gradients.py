"""Reverse-mode automatic differentiation over the deferred graph.

``GradientDescentOptimizer.minimize`` (reference ``tf_distributed.py:76``) expands to
``gradients(loss, trainable_variables)`` followed by one update op per variable. The
backward ops are ordinary graph ops appended after the forward ones, so creation order stays a
valid topological order. Every ``tensor -> d(loss)/d(tensor)`` pair is recorded in
``graph.grad_map`` so the session's fusion pass can swap whole backward chains for fused
kernels.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Set

from . import ops as F
from .graph import Operation, Tensor
from .op_registry import get_opdef

_NON_DIFF_DTYPES = ("int32", "int64", "bool", "string")


def gradients(ys, xs, grad_ys=None, name: str = "gradients") -> List[Optional[Tensor]]:
    ys = [F.convert_to_tensor(y) for y in (ys if isinstance(ys, (list, tuple)) else [ys])]
    xs_in = xs if isinstance(xs, (list, tuple)) else [xs]
    xs = [F.convert_to_tensor(x) for x in xs_in]
    graph = ys[0].graph
    if grad_ys is None:
        grad_ys = [None] * len(ys)

    # ops reachable backwards from ys ...
    reach_from_y: Set[Operation] = set()
    stack = [y.op for y in ys]
    while stack:
        op = stack.pop()
        if op in reach_from_y:
            continue
        reach_from_y.add(op)
        stack.extend(t.op for t in op.inputs)
    # ... that also depend on some x
    x_ops = {x.op for x in xs}
    depends_on_x: Dict[Operation, bool] = {}
    for op in sorted(reach_from_y, key=lambda o: o.id):
        depends_on_x[op] = op in x_ops or any(depends_on_x.get(t.op, False) for t in op.inputs)

    pending: Dict[Tensor, List[Tensor]] = {}
    with graph.name_scope(name):
        for y, gy in zip(ys, grad_ys):
            if gy is None:
                gy = F.ones_like(y, name="grad_ys")
            pending.setdefault(y, []).append(F.convert_to_tensor(gy))

        for op in sorted((o for o in reach_from_y if depends_on_x.get(o)),
                         key=lambda o: -o.id):
            out_grads: List[Optional[Tensor]] = []
            for out in op.outputs:
                parts = pending.get(out)
                if not parts:
                    out_grads.append(None)
                    continue
                total = parts[0] if len(parts) == 1 else F.add_n(parts)
                pending[out] = [total]
                graph.grad_map[out] = total
                out_grads.append(total)
            if all(g is None for g in out_grads):
                continue
            if not op.inputs or not any(depends_on_x.get(t.op, False) for t in op.inputs):
                continue   # a source, or one of the xs themselves: nothing upstream to feed
            if op.type == "StopGradient":
                continue
            grad_fn = get_opdef(op.type).grad_fn
            if grad_fn is None:
                if any(t.dtype.name not in _NON_DIFF_DTYPES and depends_on_x.get(t.op, False)
                       for t in op.inputs):
                    raise LookupError("no gradient defined for op type %r (op %r)"
                                      % (op.type, op.name))
                continue
            with graph.name_scope(op.name.split("/")[-1] + "_grad"):
                grads = out_grads if len(op.outputs) > 1 else out_grads[:1]
                in_grads = grad_fn(op, *grads)
            if len(in_grads) != len(op.inputs):
                raise ValueError("gradient of %r returned %d values for %d inputs"
                                 % (op.type, len(in_grads), len(op.inputs)))
            for t, g in zip(op.inputs, in_grads):
                if g is not None and depends_on_x.get(t.op, False):
                    pending.setdefault(t, []).append(g)

        results: List[Optional[Tensor]] = []
        for x in xs:
            parts = pending.get(x)
            if not parts:
                results.append(None)
                continue
            total = parts[0] if len(parts) == 1 else F.add_n(parts)
            graph.grad_map[x] = total
            results.append(total)
    return results
