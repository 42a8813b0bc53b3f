"""Placement resolver and communication planner for sharded matrix products.

In the reference a ``tf.matmul`` whose operands are variables placed on ``/job:ps`` by
``replica_device_setter`` (``tf_distributed_1000Matrix.py:33-48``) is cut by TensorFlow's graph
partitioner into a ps subgraph (the variable reads) and a worker subgraph (the product), joined
by ``_Send/_Recv`` pairs: every worker pulls both operands in full (8 MB) each step and computes
all of ``C``. Under a collective runtime the same graph is planned differently: the variables
are *sharded* over the worker ranks, each rank multiplies only its shard, and the cross-device
edge becomes one collective on the result instead of two transfers of the operands.

``plan_matmul_sharding`` inspects the whole graph once and returns a :class:`ShardPlan`:

* which ``MatMul`` ops run sharded, and how (``m``: row blocks of ``A`` + all-gather of ``C``;
  ``k``: inner-dimension blocks of ``A`` and ``B`` + all-reduce of the partial products; ``2d``:
  row blocks of ``A`` x column blocks of ``B`` on a P x Q rank grid + all-gather of the blocks);
* which variables are *stored* sharded: a variable whose every reader in the graph is a sharded
  product that wants the same block of it keeps only that block in HBM (a 65536 x 65536 operand
  is then M/world rows per GPU, not a replica). Any other reader (an optimizer update, a fetch
  of a derived tensor, a second product in another role) keeps the variable replicated; the
  product then multiplies a view of the replica's block, without a copy;
* for sharded variables whose initial value is a seeded counter-based random op, the recipe
  that lets every rank generate exactly its own window of the global stream, so initialisation
  moves no data at all (``CollectiveRuntime.after_init``).

The session's fusion pass turns the planned ``MatMul`` ops into ``ShardedMatMul`` nodes
(``parallel/sharded_ops.py`` executes them); reads of a sharded variable from outside a sharded
product (a direct fetch, the Saver) all-gather it.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple

MODES = ("none", "m", "k", "2d")

_REF_ONLY_TYPES = frozenset(("Assign", "IsVariableInitialized", "ReportUninitializedVariables"))
_RANDOM_INITS = {"RandomStandardNormal": True, "RandomUniform": False}     # -> normal?


def rank_grid(world: int) -> Tuple[int, int]:
    """The most square P x Q factorisation of ``world`` with P <= Q (8 ranks -> 2 x 4)."""
    p = max(d for d in range(1, int(math.isqrt(world)) + 1) if world % d == 0)
    return p, world // p


def block_of(total: int, parts: int, index: int) -> Tuple[int, int]:
    """(start, size) of block ``index`` of ``total`` cut into ``parts`` equal blocks."""
    size = total // parts
    return index * size, size


class VarLayout:
    """How one variable is cut over the ranks. ``axis``: 0 = row blocks, 1 = column blocks;
    ``parts``: number of blocks along that axis; ``group``: 'world' (block index = rank),
    'grid_row' (= rank // Q) or 'grid_col' (= rank % Q) on the P x Q grid. ``stored``: the
    variable's storage holds only the local block. ``init``: ``None`` or the recipe
    ``(normal, seed, op_seed, a, b)`` of a counter-based initial value ``a + b * g``."""

    __slots__ = ("name", "shape", "axis", "parts", "group", "stored", "init", "dtype")

    def __init__(self, name: str, shape: Tuple[int, int], axis: int, parts: int, group: str,
                 dtype) -> None:
        self.name, self.shape, self.axis, self.parts = name, tuple(shape), axis, parts
        self.group = group
        self.stored = True
        self.init: Optional[Tuple[bool, int, int, float, float]] = None
        self.dtype = dtype

    def key(self) -> Tuple[int, int, str]:
        return (self.axis, self.parts, self.group)

    def block_index(self, rank: int, world: int) -> int:
        if self.group == "world":
            return rank
        _, q = rank_grid(world)
        return rank // q if self.group == "grid_row" else rank % q

    def window(self, rank: int, world: int) -> Tuple[int, int, int, int]:
        """(row0, col0, rows, cols) of this rank's block."""
        idx = self.block_index(rank, world)
        rows, cols = self.shape
        if self.axis == 0:
            r0, n = block_of(rows, self.parts, idx)
            return r0, 0, n, cols
        c0, n = block_of(cols, self.parts, idx)
        return 0, c0, rows, n

    def local_shape(self) -> Tuple[int, int]:
        rows, cols = self.shape
        return (rows // self.parts, cols) if self.axis == 0 else (rows, cols // self.parts)

    def slice_of(self, full, rank: int, world: int):
        r0, c0, rows, cols = self.window(rank, world)
        return full[r0:r0 + rows, c0:c0 + cols]


class MatMulShard:
    """One planned product. ``b_layout``: ``"kn"`` = the right-hand operand is stored [K, N] as
    ``tf.matmul(a, b)`` has it, ``"nk"`` = [N, K] (``transpose_b=True``: already the K-contiguous
    layout the bf16 / fp8 kernels read)."""

    __slots__ = ("mode", "m", "n", "k", "grid", "b_layout", "dtype")

    def __init__(self, mode: str, m: int, n: int, k: int, grid: Tuple[int, int],
                 b_layout: str = "kn", dtype=None):
        self.mode, self.m, self.n, self.k, self.grid = mode, m, n, k, grid
        self.b_layout, self.dtype = b_layout, dtype

    def bytes_per_rank(self, world: int) -> Dict[str, int]:
        """HBM bytes one rank holds for this product: its blocks of the operands as stored, the
        K-contiguous image of a [K, N] right-hand block (bf16 / fp8 only), the full result and
        the staging of its collective."""
        es = self.dtype.size if self.dtype is not None else 4
        narrow = es < 4
        out_es = 2 if narrow else 4                   # bf16 results for bf16 / fp8 operands
        m, n, k = self.m, self.n, self.k
        p, q = self.grid
        if self.mode == "m":
            a, b = m // world * k, k * n
            stage = m // world * n * out_es
        elif self.mode == "k":
            a, b = m * (k // world), (k // world) * n
            stage = m * n * 4 if narrow else 0        # fp32 partial sums of narrow operands
        else:
            a, b = m // p * k, k * (n // q)
            stage = (m // p) * (n // q) * out_es + m * n * out_es
        out = {"a": a * es, "b": b * es,
               "b_image": b * es if (narrow and self.b_layout == "kn") else 0,
               "c": m * n * out_es, "staging": stage}
        out["total"] = sum(out.values())
        return out


class ShardPlan:
    def __init__(self, mode: str, world: int):
        self.mode, self.world = mode, world
        self.matmuls: Dict[Any, MatMulShard] = {}        # MatMul op -> how it is sharded
        self.layouts: Dict[str, VarLayout] = {}          # variable op name -> layout
        self.init_of: Dict[Any, VarLayout] = {}          # Assign op -> layout it initialises

    def stored(self, name: str) -> Optional[VarLayout]:
        lay = self.layouts.get(name)
        return lay if lay is not None and lay.stored else None

    def __bool__(self) -> bool:
        return bool(self.matmuls)


def _static2(t) -> Optional[Tuple[int, int]]:
    s = t.shape
    if s is None or len(s) != 2 or s[0] is None or s[1] is None:
        return None
    return int(s[0]), int(s[1])


def _wanted_layouts(mode: str, world: int, m: int, n: int, k: int, b_nk: bool = False):
    """(layout key of A, layout key of B) for one product, or None when the shape does not
    divide. A key is (axis, parts, group); ``None`` = the operand is needed whole. ``b_nk``: the
    right-hand variable is stored [N, K] (``transpose_b``), so its K blocks are column blocks and
    its N blocks row blocks."""
    k_axis, n_axis = (1, 0) if b_nk else (0, 1)
    if mode == "m":
        return ((0, world, "world"), None) if m % world == 0 else None
    if mode == "k":
        return ((1, world, "world"), (k_axis, world, "world")) if k % world == 0 else None
    if mode == "2d":
        p, q = rank_grid(world)
        if m % p or n % q:
            return None
        return ((0, p, "grid_row") if p > 1 else None,
                (n_axis, q, "grid_col") if q > 1 else None)
    return None


def plan_matmul_sharding(graph, world: int, mode: str, shard_variables: bool = True) -> ShardPlan:
    """Decide, for the whole graph, which products run sharded and which variables are stored as
    shards (see the module docstring). ``world`` is the number of worker ranks."""
    if mode not in MODES:
        raise ValueError("matmul sharding mode must be one of %s, got %r" % (MODES, mode))
    plan = ShardPlan(mode, world)
    if mode == "none" or world < 2:
        return plan
    ops = graph.get_operations()
    consumers: Dict[Any, List[Any]] = {}
    for op in ops:
        for t in op.inputs:
            consumers.setdefault(t, []).append(op)

    want: Dict[str, List[Optional[Tuple[int, int, str]]]] = {}     # var name -> wanted layouts
    var_tensor: Dict[str, Any] = {}
    for op in ops:
        if op.type != "MatMul" or op.attrs["transpose_a"]:
            continue
        a, b = op.inputs
        if a.op.type != "VariableV2" or b.op.type != "VariableV2" or a.dtype != b.dtype:
            continue
        if a.dtype.name not in ("float32", "bfloat16", "float8_e4m3fn"):
            continue
        b_nk = bool(op.attrs["transpose_b"])
        sa, sb = _static2(a), _static2(b)
        if sa is None or sb is None:
            continue
        if b_nk:
            sb = (sb[1], sb[0])                          # as [K, N]
        if sa[1] != sb[0]:
            continue
        m, k, n = sa[0], sa[1], sb[1]
        keys = _wanted_layouts(mode, world, m, n, k, b_nk)
        if keys is None:
            continue
        plan.matmuls[op] = MatMulShard(mode, m, n, k, rank_grid(world),
                                       "nk" if b_nk else "kn", a.dtype)
        for t, key in ((a, keys[0]), (b, keys[1])):
            want.setdefault(t.op.name, []).append(key)
            var_tensor[t.op.name] = t

    for name, keys in want.items():
        t = var_tensor[name]
        first = keys[0]
        if first is None or any(kk != first for kk in keys):
            continue                    # needed whole somewhere, or in two different cuts
        lay = VarLayout(name, _static2(t), first[0], first[1], first[2], t.dtype)
        readers = [c for c in consumers.get(t, []) if c.type not in _REF_ONLY_TYPES]
        lay.stored = bool(shard_variables) and all(c in plan.matmuls for c in readers)
        plan.layouts[name] = lay
        if not lay.stored:
            continue
        assigns = [c for c in consumers.get(t, []) if c.type == "Assign" and c.inputs[0] is t]
        if len(assigns) != 1:
            continue                    # re-assigned somewhere: initial values travel whole
        src = assigns[0].inputs[1].op
        if src.type in _RANDOM_INITS and src.attrs.get("deterministic", True) \
                and tuple(src.attrs["shape"]) == lay.shape:
            if src.type == "RandomStandardNormal":
                a_, b_ = float(src.attrs["mean"]), float(src.attrs["stddev"])
            else:
                lo, hi = float(src.attrs["minval"]), float(src.attrs["maxval"])
                a_, b_ = lo, hi - lo
            lay.init = (_RANDOM_INITS[src.type], int(src.attrs["seed"]), int(src.attrs["seed2"]),
                        a_, b_)
            plan.init_of[assigns[0]] = lay
    return plan
