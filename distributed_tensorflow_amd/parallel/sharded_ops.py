"""Kernels of the plan-time nodes a sharded matrix product is made of.

``ShardedMatMul`` replaces a ``tf.matmul`` of two variables (reference
``tf_distributed_1000Matrix.py:43-48``) when the session runs under a collective runtime with
more than one worker and ``ConfigProto(matmul_sharding=...)`` asks for it (the plan comes from
``parallel/planner.py``). Every rank multiplies its block on the hand-written MFMA GEMM and one
collective over xGMI (RCCL; gloo on CPUs) assembles the result:

=====  =============================  ===========================  =========================
mode   local product                  collective                   bytes into each rank
=====  =============================  ===========================  =========================
``m``  A[rows r] @ B                  all-gather of C row blocks   (w-1)/w * M*N
``k``  A[:, cols r] @ B[rows r]       all-reduce (sum) of C        ~2 (w-1)/w * M*N (fp32)
``2d`` A[rows pr] @ B[:, cols pc]     all-gather of C blocks       (w-1)/w * M*N
=====  =============================  ===========================  =========================

In mode ``m`` the product of a row block runs with the K split the whole product would use, so
the result is bit-identical to the single-rank one (fp32: each output element is the same
k-ordered FMA chain, whatever the tiling). The result buffers are allocated once per node and
keep their addresses, so a step that contains the node can be captured into a HIP graph
together with its collective.

``ShardedVariableInit`` is the initialiser of a variable that is stored sharded and whose
initial value is a seeded counter-based random tensor: the rank generates exactly its window of
the global Philox stream. ``GatherShardedVariable`` is what a read of such a variable from
outside a sharded product becomes: an all-gather of the blocks.
"""
from __future__ import annotations

from typing import Any, List, Tuple

import torch

from ..framework.op_registry import hip_kernel, ref_kernel
from ..ops import philox
from .planner import VarLayout, rank_grid


def _comm(ctx):
    rt = ctx.session._runtime
    if rt is None or getattr(rt, "comm", None) is None:
        raise RuntimeError("a sharded op needs a collective runtime")
    return rt.comm


def _buffer(ctx, op, tag: str, shape: Tuple[int, ...], dtype: torch.dtype) -> torch.Tensor:
    """Per-node scratch that keeps its address from step to step (HIP-graph capturable)."""
    key = (op, tag)
    buf = ctx.buffers.get(key)
    if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
        buf = ctx.buffers[key] = torch.empty(shape, dtype=dtype, device=ctx.device)
    return buf


def _block(t: torch.Tensor, axis: int, parts: int, index: int, full: int) -> torch.Tensor:
    """Block ``index`` of ``t`` along ``axis`` -- or ``t`` itself when it already is that block
    (a variable stored sharded)."""
    size = full // parts
    if t.shape[axis] == size and parts > 1:
        return t
    if t.shape[axis] != full:
        raise RuntimeError("operand has %d along axis %d, expected %d (whole) or %d (block)"
                           % (t.shape[axis], axis, full, size))
    return t.narrow(axis, index * size, size)


def _gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, full_mnk) -> None:
    """``out = a @ b`` (a [m,k], b [k,n], both possibly row-strided views). On a GPU: the fp32 /
    bf16 MFMA kernels, with the K split of the WHOLE product ``full_mnk`` so a block's rows come
    out bit-identical to the unsharded result."""
    if a.is_cuda:
        from ..ops import hip
        if a.dtype == torch.float32:
            hip.gemm_f32(a, b, out=out,
                         plan_mnk=full_mnk if full_mnk[2] == a.shape[1] else None)
        elif a.dtype == torch.bfloat16:
            hip.gemm_bf16(a, b, out=out)
        else:
            raise TypeError("sharded matmul: unsupported dtype %s" % a.dtype)
        return
    if a.dtype == torch.float32:
        torch.matmul(a, b, out=out)
    else:
        out.copy_((a.float() @ b.float()).to(out.dtype))


def _copy_block(dst: torch.Tensor, src: torch.Tensor) -> None:
    """2-D block copy between row-strided views (the kernel library's ``copy_2d`` on a GPU)."""
    if dst.is_cuda:
        from ..ops import hip
        hip.copy_2d(src, dst)
    else:
        dst.copy_(src)


def _sharded_matmul(op, inputs, ctx) -> List[Any]:
    a, b = inputs
    spec = op.attrs["shard"]
    comm = _comm(ctx)
    w, r = comm.world_size, comm.rank
    if w != op.attrs["world"]:
        raise RuntimeError("plan was made for %d ranks, communicator has %d" % (op.attrs["world"], w))
    m, n, k = spec.m, spec.n, spec.k
    dt = a.dtype
    rt = ctx.session._runtime
    if spec.mode == "m":
        mb = m // w
        c = _buffer(ctx, op, "c", (m, n), dt)
        mine = _buffer(ctx, op, "mine", (mb, n), dt)
        _gemm(_block(a, 0, w, r, m), b, mine, (m, n, k))
        comm.all_gather(c, mine)
        rt.bytes_gathered += c.numel() * c.element_size()
        return [c]
    if spec.mode == "k":
        c = _buffer(ctx, op, "c", (m, n), dt)
        if dt == torch.float32:
            _gemm(_block(a, 1, w, r, k), _block(b, 0, w, r, k), c, (m, n, k))
            comm.all_reduce(c, "sum")
            rt.bytes_reduced += c.numel() * 4
            return [c]
        # bf16 operands: the partial products are summed in fp32 and rounded once
        part = _buffer(ctx, op, "partial", (m, n), torch.float32)
        a_blk, b_blk = _block(a, 1, w, r, k), _block(b, 0, w, r, k)
        if a.is_cuda:
            from ..ops import hip
            hip.gemm_bf16(a_blk, b_blk, out=part, out_dtype=torch.float32)
        else:
            torch.matmul(a_blk.float(), b_blk.float(), out=part)
        comm.all_reduce(part, "sum")
        rt.bytes_reduced += part.numel() * 4
        if a.is_cuda:
            from ..ops import hip
            hip.cast(part, dt, out=c)
        else:
            c.copy_(part)
        return [c]
    # 2d: rank (pr, pc) of the P x Q grid computes block (pr, pc) of C
    p, q = spec.grid
    pr, pc = divmod(r, q)
    mb, nb = m // p, n // q
    stage = _buffer(ctx, op, "stage", (w, mb, nb), dt)
    mine = _buffer(ctx, op, "mine", (mb, nb), dt)
    _gemm(_block(a, 0, p, pr, m), _block(b, 1, q, pc, n), mine, (m, n, k))
    comm.all_gather(stage, mine)
    rt.bytes_gathered += stage.numel() * stage.element_size()
    c = _buffer(ctx, op, "c", (m, n), dt)
    for i in range(p):
        for j in range(q):
            _copy_block(c[i * mb:(i + 1) * mb, j * nb:(j + 1) * nb], stage[i * q + j])
    return [c]


ref_kernel("ShardedMatMul", stateful=True)(_sharded_matmul)
hip_kernel("ShardedMatMul")(_sharded_matmul)


def local_window(lay: VarLayout, rank: int, world: int, device: torch.device) -> torch.Tensor:
    """This rank's block of a variable whose initial value is the counter-based random tensor
    recorded in ``lay.init``: generated in place, nothing is communicated."""
    normal, seed, op_seed, a, b = lay.init
    r0, c0, rows, cols = lay.window(rank, world)
    dt = lay.dtype.torch
    if device.type == "cuda":
        from ..ops import hip
        return hip.philox_window(rows, cols, seed, op_seed, r0, c0, lay.shape[1], dt, device,
                                 normal, a, b)
    host = torch.from_numpy(philox.window(rows, cols, seed, op_seed, r0, c0, lay.shape[1], normal))
    return (host * b + a).to(device=device, dtype=dt)


def _sharded_variable_init(op, inputs, ctx) -> List[Any]:
    lay: VarLayout = op.attrs["layout"]
    comm = _comm(ctx)
    value = local_window(lay, comm.rank, comm.world_size, ctx.device)
    return [ctx.var_assign(op.attrs["var_op"], value)]


ref_kernel("ShardedVariableInit", stateful=True)(_sharded_variable_init)
hip_kernel("ShardedVariableInit")(_sharded_variable_init)


def gather_blocks(comm, lay: VarLayout, block: torch.Tensor) -> torch.Tensor:
    """All ranks' blocks of a sharded variable assembled into the whole tensor (collective:
    every rank calls it)."""
    w = comm.world_size
    rows, cols = lay.shape
    full = torch.empty((rows, cols), dtype=block.dtype, device=block.device)
    stage = torch.empty((w,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    comm.all_gather(stage, block.contiguous())
    for rank in range(w):
        r0, c0, nr, nc = lay.window(rank, w)
        _copy_block(full[r0:r0 + nr, c0:c0 + nc], stage[rank])
    return full


def _gather_sharded_variable(op, inputs, ctx) -> List[Any]:
    lay: VarLayout = op.attrs["layout"]
    block = inputs[0]
    if tuple(block.shape) == lay.shape:          # not (yet) resharded: already whole
        return [block]
    return [gather_blocks(_comm(ctx), lay, block)]


ref_kernel("GatherShardedVariable", stateful=True)(_gather_sharded_variable)
hip_kernel("GatherShardedVariable")(_gather_sharded_variable)


__all__ = ["gather_blocks", "local_window", "rank_grid"]
