"""Operation definitions: per-type kernels and gradient builders.

Every op type has exactly one *reference kernel* written with plain torch ops (used for the
CPU plumbing configuration and as the numerical oracle in tests) and optionally a *HIP kernel*
(hand-written gfx950 code reached through ``ops.hip``) that the executor uses when the
session's device is a GPU. A gradient builder takes the op and the gradients of its outputs
and returns symbolic gradients for its inputs.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence

Kernel = Callable[[Any, Sequence[Any], Any], List[Any]]
GradFn = Callable[..., Sequence[Optional[Any]]]


class OpDef:
    __slots__ = ("type", "ref_kernel", "hip_kernel", "grad_fn", "stateful", "host_only")

    def __init__(self, op_type: str):
        self.type = op_type
        self.ref_kernel: Optional[Kernel] = None
        self.hip_kernel: Optional[Kernel] = None
        self.grad_fn: Optional[GradFn] = None
        self.stateful = False      # never pruned as dead / never constant-folded
        self.host_only = False     # produces host values (strings): no device kernel


_REGISTRY: Dict[str, OpDef] = {}


def opdef(op_type: str) -> OpDef:
    if op_type not in _REGISTRY:
        _REGISTRY[op_type] = OpDef(op_type)
    return _REGISTRY[op_type]


def get_opdef(op_type: str) -> OpDef:
    try:
        return _REGISTRY[op_type]
    except KeyError:
        raise KeyError("unknown op type %r" % (op_type,))


def ref_kernel(op_type: str, stateful: bool = False, host_only: bool = False):
    def deco(fn: Kernel) -> Kernel:
        d = opdef(op_type)
        d.ref_kernel = fn
        d.stateful = d.stateful or stateful
        d.host_only = d.host_only or host_only
        return fn
    return deco


def hip_kernel(op_type: str):
    def deco(fn: Kernel) -> Kernel:
        opdef(op_type).hip_kernel = fn
        return fn
    return deco


def gradient(op_type: str):
    def deco(fn: GradFn) -> GradFn:
        opdef(op_type).grad_fn = fn
        return fn
    return deco


def not_differentiable(*op_types: str) -> None:
    for t in op_types:
        opdef(t).grad_fn = None


def registered_types() -> List[str]:
    return sorted(_REGISTRY)
