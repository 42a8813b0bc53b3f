"""Kernels of the plan-time nodes a sharded matrix product is made of.

``ShardedMatMul`` replaces a ``tf.matmul`` of two variables (reference
``tf_distributed_1000Matrix.py:43-48``) when the session runs under a collective runtime with
more than one worker and ``ConfigProto(matmul_sharding=...)`` asks for it (the plan comes from
``parallel/planner.py``). The node is a thin adapter: the product itself -- panel-pipelined local
GEMMs on the hand-written MFMA kernels with the collective of panel p in flight while panel p+1
is computed -- is run by the same ``ShardedMatmul`` engine (``parallel/sharded_matmul.py``) that
``bench.py`` times, bound to this rank's blocks of the two variables:

=====  =============================  ===========================  =========================
mode   engine strategy                collective                   bytes into each rank
=====  =============================  ===========================  =========================
``m``  ``m_shard_allgather``          all-gather of C row panels   (w-1)/w * M*N
       (``_direct``: peer memory)
``k``  ``k_shard_allreduce``          all-reduce (sum) of C        ~2 (w-1)/w * M*N (fp32)
``2d`` ``block_2d_allgather``         all-gather of C blocks       (w-1)/w * M*N
=====  =============================  ===========================  =========================

fp32 products run with tile and K split of the WHOLE product (``plan_mnk``), so modes ``m`` and
``2d`` reproduce the single-rank result bit for bit (each output element is the same k-ordered
FMA chain, whatever the tiling). bf16 and fp8 products read a K-contiguous image of the
right-hand block; it is made once and re-made only when the variable was written
(``LocalVariableStore.version``) or, for a variable some kernel of the graph updates in place,
every step. bf16 partial sums of mode ``k`` travel and are summed in fp32. Every buffer is
allocated once per node and keeps its address, so a step that contains the node can be captured
into a HIP graph together with its collectives.

``ConfigProto(matmul_collective="peer")`` (or ``DTF_MATMUL_COLLECTIVE=peer``) selects the
peer-memory all-gather for mode ``m`` on the GPUs of one node.

``ShardedVariableInit`` is the initialiser of a variable that is stored sharded and whose
initial value is a seeded counter-based random tensor: the rank generates exactly its window of
the global Philox stream. ``GatherShardedVariable`` is what a read of such a variable from
outside a sharded product becomes: an all-gather of the blocks.
"""
from __future__ import annotations

import os
from typing import Any, List, Tuple

import torch

from ..framework.op_registry import hip_kernel, ref_kernel
from ..ops import philox
from .planner import VarLayout, rank_grid

_MODE_STRATEGY = {"m": "m_shard_allgather", "k": "k_shard_allreduce", "2d": "block_2d_allgather"}
_IN_PLACE_WRITERS = frozenset(("AssignAdd", "AssignSub", "ApplyGradientDescent", "MultiApplySGD",
                               "ApplyOptimizer", "MultiApplyOptimizer", "AffineUpdateVariable",
                               "MLPTrainStep", "MultiMovingAverage"))


def _comm(ctx):
    rt = ctx.session._runtime
    if rt is None or getattr(rt, "comm", None) is None:
        raise RuntimeError("a sharded op needs a collective runtime")
    return rt.comm


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


def _copy_block(dst: torch.Tensor, src: torch.Tensor) -> None:
    """2-D block copy between row-strided views (the kernel library's ``copy_2d`` on a GPU)."""
    if dst.is_cuda:
        from ..ops import hip
        hip.copy_2d(src, dst)
    else:
        dst.copy_(src)


def _written_in_place(graph, var_op) -> bool:
    """Does any op of the graph update ``var_op``'s storage without going through the variable
    store's ``assign`` (optimizer launches, ``assign_add``)? Cached per graph version."""
    cache = graph.__dict__.setdefault("_dtf_in_place_writers", {})
    hit = cache.get(var_op)
    if hit is not None and hit[0] == graph.version:
        return hit[1]
    t = var_op.outputs[0]
    written = False
    for op in graph.get_operations():
        if op.type in _IN_PLACE_WRITERS and (
                t in op.inputs or var_op in op.attrs.get("var_ops", ())):
            written = True
            break
    cache[var_op] = (graph.version, written)
    return written


def _operand_blocks(spec, a: torch.Tensor, b: torch.Tensor, rank: int, world: int):
    """This rank's blocks of the two operands (views of replicas, or the stored blocks)."""
    m, n, k = spec.m, spec.n, spec.k
    nk = spec.b_layout == "nk"
    if spec.mode == "m":
        return _block(a, 0, world, rank, m), b
    if spec.mode == "k":
        return _block(a, 1, world, rank, k), _block(b, 1 if nk else 0, world, rank, k)
    p, q = spec.grid
    pr, pc = divmod(rank, q)
    return _block(a, 0, p, pr, m), _block(b, 0 if nk else 1, q, pc, n)


def _engine_for(op, ctx, comm, dtype: torch.dtype):
    key = (op, "engine")
    eng = ctx.buffers.get(key)
    if eng is not None and eng.dtype == dtype and eng.world == comm.world_size:
        return eng
    from .sharded_matmul import ShardedMatmul
    spec = op.attrs["shard"]
    strategy = _MODE_STRATEGY[spec.mode]
    cfg = getattr(ctx.session, "config", None)
    collective = getattr(cfg, "matmul_collective", None) or \
        os.environ.get("DTF_MATMUL_COLLECTIVE") or "rccl"
    if collective == "peer" and spec.mode == "m" and ctx.device.type == "cuda":
        strategy += "_direct"
    narrow = dtype in (torch.bfloat16, torch.float8_e4m3fn)
    eng = ShardedMatmul(
        spec.m, spec.n, spec.k, comm, dtype=dtype, strategy=strategy, device=ctx.device,
        b_layout=spec.b_layout, panels=getattr(cfg, "matmul_panels", None),
        # fp32: tile and K split of the whole product -> a block's rows are bit-identical to the
        # single-rank result; narrow operands: one sequential K order whatever the block shape
        plan_mnk=(spec.m, spec.n, spec.k) if dtype == torch.float32 and spec.mode != "k"
        else None,
        k_splits=1 if narrow and spec.mode != "k" else None,
        reduce_dtype=torch.float32 if (dtype == torch.bfloat16 and spec.mode == "k") else None)
    ctx.buffers[key] = eng
    return eng


def _sharded_matmul(op, inputs, ctx) -> List[Any]:
    a, b = inputs
    spec = op.attrs["shard"]
    comm = _comm(ctx)
    w, r = comm.world_size, comm.rank
    if w != op.attrs["world"]:
        raise RuntimeError("plan was made for %d ranks, communicator has %d" % (op.attrs["world"], w))
    rt = ctx.session._runtime
    eng = _engine_for(op, ctx, comm, a.dtype)
    a_blk, b_blk = _operand_blocks(spec, a, b, r, w)
    eng.bind_operands(a_blk, b_blk)
    if spec.b_layout == "kn" and a.dtype != torch.float32 and a.is_cuda:
        # the K-contiguous image of the right-hand block: still valid?
        b_op = op.inputs[1].op
        store = ctx.session._store_for(b_op)
        if _written_in_place(ctx.session.graph, b_op) or not hasattr(store, "version"):
            eng.refresh_images()
        else:
            seen = ctx.buffers.get((op, "b_version"))
            now = store.version(b_op.name)
            if seen != now:
                eng.refresh_images()
                ctx.buffers[(op, "b_version")] = now
                store.watch_writes(b_op.name)
    eng.tracer = getattr(ctx, "tracer", None)     # a traced run: one bracket per panel product
    try:
        eng.step()
    finally:
        eng.tracer = None
    c = eng.result_rowmajor()
    nbytes = c.numel() * c.element_size()
    if spec.mode == "k":
        rt.bytes_reduced += c.numel() * eng.reduce_dtype.itemsize
    else:
        rt.bytes_gathered += nbytes
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


def gather_blocks(comm, lay: VarLayout, block: torch.Tensor, buffers=None) -> torch.Tensor:
    """All ranks' blocks of a sharded variable assembled into the whole tensor (collective:
    every rank calls it). ``buffers``: a ``(whole, stage)`` pair to reuse (a node that gathers
    every step keeps one); without it both are allocated."""
    w = comm.world_size
    rows, cols = lay.shape
    if buffers is None:
        buffers = (torch.empty((rows, cols), dtype=block.dtype, device=block.device),
                   torch.empty((w,) + tuple(block.shape), dtype=block.dtype,
                               device=block.device))
    full, stage = buffers
    if lay.axis == 0 and lay.group == "world":
        # row blocks in rank order ARE the whole tensor: gather straight into it
        comm.all_gather(full, block if block.is_contiguous() else block.contiguous())
        return full
    comm.all_gather(stage, block if block.is_contiguous() else block.contiguous())
    for rank in range(w):
        r0, c0, nr, nc = lay.window(rank, w)
        _copy_block(full[r0:r0 + nr, c0:c0 + nc], stage[rank])
    return full


def _gather_sharded_variable(op, inputs, ctx) -> List[Any]:
    lay: VarLayout = op.attrs["layout"]
    block = inputs[0]
    if tuple(block.shape) == lay.shape:          # not (yet) resharded: already whole
        return [block]
    key = (op, "gather")
    bufs = ctx.buffers.get(key)
    comm = _comm(ctx)
    if bufs is None or bufs[0].dtype != block.dtype or bufs[0].device != block.device \
            or tuple(bufs[1].shape[1:]) != tuple(block.shape):
        bufs = ctx.buffers[key] = (
            torch.empty(lay.shape, dtype=block.dtype, device=block.device),
            torch.empty((comm.world_size,) + tuple(block.shape), dtype=block.dtype,
                        device=block.device))
    return [gather_blocks(comm, lay, block, bufs)]


ref_kernel("GatherShardedVariable", stateful=True)(_gather_sharded_variable)
hip_kernel("GatherShardedVariable")(_gather_sharded_variable)


__all__ = ["gather_blocks", "local_window", "rank_grid"]
