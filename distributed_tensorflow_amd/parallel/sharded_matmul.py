"""Sharded matrix multiply: ``C[M,N] = A[M,K] @ B[K,N]`` with operands distributed over ranks.

The reference's matmul workload (``tf_distributed_1000Matrix.py:43-48``) puts ``A`` and ``B``
on the single parameter server and has *every* worker pull both (8 MB per step over gRPC) and
compute the whole product. Here the operands live sharded in HBM and each rank computes only
its share; the collectives run over xGMI through RCCL. Strategies:

=========================  =====================  ================  ==========================
strategy                   A / B shard per rank   communication     result
=========================  =====================  ================  ==========================
``replicated``             whole / whole          none              full C on every rank
                                                                    (the reference's layout)
``m_shard``                row block / whole      none              row block of C
``m_shard_allgather``      row block / whole      all-gather of C   full C on every rank
``k_shard_allreduce``      col block / row block  all-reduce        full C on every rank
``k_shard_reduce_scatter`` col block / row block  reduce-scatter    row block of C
``summa_2d``               2-D block / 2-D block  row/col broadcast 2-D block of C
``block_2d_allgather``     row block (grid row) / all-gather of C   full C on every rank
                           column block (grid     blocks
                           column), all of K
=========================  =====================  ================  ==========================

``block_2d_allgather`` is the 2-D decomposition for operands that fit a GPU many times over (a
65536 x 65536 fp8 operand is 4 GiB of 288): rank (i, j) of the P x Q grid keeps the rows of A of
grid row i and the columns of B of grid column j with ALL of K, so no operand panel ever
travels; it computes block (i, j) of C and the blocks are all-gathered. ``summa_2d`` cuts K as
well (operands 1/world per rank) and pays for that with the panel broadcasts.

``m_shard_allgather_direct`` and ``k_shard_allreduce_direct`` are the same data layouts with the
collective done over peer-mapped memory (``parallel/peer_mem.py``): every rank pushes its rows
to all peers at once instead of passing them round a ring. GPU only, one node only.

``peer_transport="epilogue"`` (``m_shard_allgather_direct``, bf16): there is no transfer step at
all -- the GEMM kernel's epilogue stores every result vector into the same rows of all peers'
result buffers next to its own (``hip.gemm_bf16_nt_mirrored``), tile by tile as the product runs,
so the rows leave at the rate the matrix cores produce them and C is never read back for a copy.

``recompute_rows=x`` (all-gather strategies): a rank also computes the first ``x`` rows of every
*peer's* row block instead of receiving them, so only ``M/world - x`` rows per block travel.
With full C wanted on every rank each xGMI link has to carry ``M*N/world`` result elements per
step however the collective is organised (one link per peer pair), which for 16384^3 bf16 takes
longer than a rank's share of the product at every world size; recomputing a slice of the
neighbours' rows moves the balance point: ``(M/world + (world-1)x)`` rows of compute against
``M/world - x`` rows per link. ``balanced_recompute_rows`` solves for ``x`` from a measured
compute rate and step time. The heads of all blocks are one contiguous [world*x, N] region of
the result buffer (one GEMM launch), followed by the transferred bodies; ``result_dense()``
returns plain row order.

``pipelined=True`` (all-gather strategies): repeated steps are independent products, so two
result buffers alternate and a step returns without waiting for its own collectives -- the tail
of step s's all-gather runs under the first panel of step s+1. ``wait()`` completes everything.

``summa_2d`` runs on a P x Q process grid (the most square factorisation of the world size:
8 ranks -> 2 x 4). K is cut into L = lcm(P, Q) panels; A's block (i, j) holds the L/Q panels of
grid column j, Bt's block the L/P panels of grid row i, each stored contiguously. Step l
broadcasts A's panel along grid rows and B's panel along grid columns -- two collectives on
disjoint xGMI links -- and the product of panel l runs while the broadcasts of panel l+1 are in
flight (two receive buffers per operand). The fp32 accumulation across panels happens in the
GEMM epilogue (``beta = 1``), not in a separate add.

Operand dtypes: fp32 (exact MFMA), bf16, and OCP fp8 e4m3 (block-scaled MFMA with unit scales;
results in bf16, or fp32 for ``summa_2d``).

xGMI is a point-to-point mesh (7 links of ~153 GB/s per GPU), so the bytes a strategy moves
decide everything: for 16384^3 bf16 on 8 GPUs a rank computes for ~0.75 ms, an all-gather of C
moves 448 MiB into each rank and an all-reduce of fp32 partial sums twice 896 MiB. Every
communicating strategy is therefore *panel-pipelined*: the local product is cut into row
panels, and the collective of panel p runs on the communication stream while the matrix
cores compute panel p+1. Partial sums travel in the result dtype (bf16 for bf16 operands)
unless ``reduce_dtype`` asks for fp32.

Operands are generated in place by the counter-based Philox kernels: each rank fills exactly
its window of the *global* random matrices, so results do not depend on the world size and a
sharded run can be checked against a single-GPU one.

The right-hand operand is stored K-contiguous (``Bt = B^T``, [N, K]) -- the layout the bf16
MFMA kernel consumes directly.

The same engine runs the ``tf.matmul`` of two variables that the session's planner shards
(``parallel/sharded_ops.py``): ``bind_operands`` hands it the rank's blocks of the variables
instead of generated ones, ``b_layout="kn"`` says that the right-hand operand lies as the graph
has it ([K, N]; bf16 and fp8 products then run on a K-contiguous image of it that is re-made only
when the variable's storage changed), ``plan_mnk`` pins tile and K split of the fp32 kernel to
those of the whole product so that a block's rows are bit-identical to the single-rank result,
and ``result_rowmajor()`` is the full C in plain row order whatever the panel layout.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch

from ..ops import philox
from .comm import Communicator, SingleComm

STRATEGIES = ("replicated", "m_shard", "m_shard_allgather", "k_shard_allreduce",
              "k_shard_reduce_scatter", "summa_2d", "m_shard_allgather_direct",
              "k_shard_allreduce_direct", "block_2d_allgather")


def _split(total: int, parts: int, index: int) -> Tuple[int, int]:
    """Start and size of block ``index`` when ``total`` is cut into ``parts`` near-equal blocks
    (the first ``total % parts`` blocks get one extra element)."""
    base, extra = divmod(total, parts)
    start = index * base + min(index, extra)
    return start, base + (1 if index < extra else 0)


def balanced_recompute_rows(block_rows: int, world: int, compute_secs_per_block: float,
                            step_secs: float, granule: Optional[int] = None) -> List[int]:
    """Candidate ``recompute_rows`` values around the balance point of compute and transfer.

    ``compute_secs_per_block``: time of a rank's own M/world-row product; ``step_secs``: measured
    time of a whole all-gather step without recomputation. When the step takes longer than the
    product, the excess is transfer time that scales with the rows sent, ``step * (b - x) / b``,
    while the product grows to ``compute * (b + (world-1) x) / b``; the two meet at
    ``x = b (step - compute) / ((world-1) compute + step)``. Returns the multiples of
    ``granule`` just below and above that point (empty when the step is already compute-bound).
    The granule defaults to one 256-row output tile, half of one from 8 ranks on (where the
    balance point of a 16384-row product lies below 256 rows).
    """
    b, w = int(block_rows), int(world)
    if granule is None:
        granule = 128 if w >= 8 else 256
    if w < 2 or compute_secs_per_block <= 0 or step_secs <= 1.03 * compute_secs_per_block:
        return []
    x = b * (step_secs - compute_secs_per_block) / ((w - 1) * compute_secs_per_block + step_secs)
    lo = int(x // granule) * granule
    cands = [c for c in (lo, lo + granule) if 0 < c <= b // 2]
    return sorted(set(cands))


class ShardedMatmul:
    def __init__(self, m: int, n: int, k: int, comm: Optional[Communicator] = None,
                 dtype: torch.dtype = torch.bfloat16, strategy: str = "m_shard_allgather",
                 panels: Optional[int] = None, seed: int = 1,
                 device: Optional[torch.device] = None,
                 reduce_dtype: Optional[torch.dtype] = None, distribution: str = "uniform",
                 pipelined: bool = False, recompute_rows: int = 0,
                 peer_transport: str = "copy", scaling: str = "tensor",
                 k_splits: Optional[int] = None, b_layout: str = "nk",
                 plan_mnk: Optional[Tuple[int, int, int]] = None):
        if strategy not in STRATEGIES:
            raise ValueError("unknown strategy %r (choose from %s)" % (strategy, STRATEGIES))
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.comm = comm or SingleComm(device or torch.device("cpu"))
        self.rank, self.world = self.comm.rank, self.comm.world_size
        self.device = torch.device(device) if device is not None else self.comm.device
        self.dtype = dtype
        # K splits of the local products: None = the kernel library's plan for each shape (small
        # tile grids are cut along K and their fp32 partial planes summed, which changes the
        # summation order with the shape); 1 = every product sums K in one sequential order, so
        # a row of C has the same bits whichever panel, head block or world size computed it
        self.k_splits = k_splits
        # "nk": the right-hand operand is stored [N, K] (K-contiguous, what the bf16 / fp8
        # kernels read); "kn": [K, N] as a graph's tf.matmul has it
        if b_layout not in ("nk", "kn"):
            raise ValueError("b_layout is 'nk' or 'kn'")
        self.b_layout = b_layout
        self.plan_mnk = tuple(int(v) for v in plan_mnk) if plan_mnk is not None else None
        self._image = None          # (key, K-contiguous image) of a bound [K, N] operand
        self.tracer = None          # utils.tracing.StepTracer while a traced step runs
        self._c_dense: Optional[torch.Tensor] = None
        # fp8 operands: "tensor" = unit scales (the generated operands fit e4m3 as they are),
        # "mx" = block-scaled (one E8M0 scale per 32 k of a row, applied by the matrix
        # instruction): operands are generated in bf16 and quantised by the MX kernel
        if scaling not in ("tensor", "mx"):
            raise ValueError("scaling is 'tensor' or 'mx'")
        self.scaling = scaling
        if scaling == "mx":
            if dtype != torch.float8_e4m3fn:
                raise ValueError("scaling='mx' applies to fp8 operands")
            if strategy == "summa_2d" or recompute_rows or b_layout != "nk":
                raise ValueError("scaling='mx': operands must stay on their rank as quantised "
                                 "[.., K] images (no summa_2d, no recompute_rows, b_layout='nk')")
        # fp8 operands produce bf16 results (fp32 accumulate inside the kernel)
        self.out_dtype = torch.bfloat16 if dtype == torch.float8_e4m3fn else dtype
        self.strategy = strategy
        self.seed = seed
        self.distribution = distribution
        # pipelined (all-gather strategies): two result buffers alternate and a step does not
        # wait for its own collectives, so the tail of step s's all-gather overlaps the first
        # panel of step s+1 (the steps are independent products). Results are only guaranteed
        # complete after wait().
        self.pipelined = bool(pipelined) and strategy in ("m_shard_allgather",
                                                          "m_shard_allgather_direct")
        self._steps_done = 0
        self._set_works: List[list] = [[], []]
        self.reduce_dtype = reduce_dtype or self.out_dtype
        self.on_gpu = self.device.type == "cuda"
        if self.on_gpu:
            from ..ops import hip
            self._hip = hip
            hip.lib()
        w = self.world
        if strategy == "summa_2d":
            p_rows = max(d for d in range(1, int(math.isqrt(w)) + 1) if w % d == 0)
            self.grid_rows, self.grid_cols = p_rows, w // p_rows
            self.k_panels = math.lcm(self.grid_rows, self.grid_cols)
            if self.k % self.k_panels:
                raise ValueError("K=%d must be divisible by lcm(grid)=%d"
                                 % (self.k, self.k_panels))
            self.pr, self.pc = divmod(self.rank, self.grid_cols)
            self.row_comm = self.comm.split(self.pr, self.pc)     # ranks of my grid row
            self.col_comm = self.comm.split(self.pc, self.pr)     # ranks of my grid column
        if strategy == "block_2d_allgather":
            p_rows = max(d for d in range(1, int(math.isqrt(w)) + 1) if w % d == 0)
            self.grid_rows, self.grid_cols = p_rows, w // p_rows
            self.pr, self.pc = divmod(self.rank, self.grid_cols)
            if self.m % self.grid_rows or self.n % self.grid_cols:
                raise ValueError("M=%d / N=%d must be divisible by the %d x %d rank grid"
                                 % (self.m, self.n, self.grid_rows, self.grid_cols))
        if strategy in ("m_shard", "m_shard_allgather", "k_shard_reduce_scatter",
                        "m_shard_allgather_direct", "k_shard_allreduce_direct") \
                and self.m % w:
            raise ValueError("M=%d must be divisible by the world size %d" % (self.m, w))
        if strategy.startswith("k_shard") and self.k % w:
            raise ValueError("K=%d must be divisible by the world size %d" % (self.k, w))
        rows_local = self.m // w if strategy.startswith("m_shard") else self.m
        if strategy == "block_2d_allgather":
            rows_local = self.m // self.grid_rows
        # *_direct strategies: "copy" | "kernel" | "epilogue" (the last one only where the product
        # itself can store into the peers: the bf16 all-gather; elsewhere it means "copy")
        if peer_transport == "epilogue":
            if strategy != "m_shard_allgather_direct":
                peer_transport = "copy"
            elif dtype != torch.bfloat16 or b_layout != "nk":
                raise ValueError("peer_transport='epilogue': the bf16 product with a K-contiguous "
                                 "right-hand operand is the one whose epilogue stores into peers")
        self.peer_transport = peer_transport
        self.recompute_rows = int(recompute_rows)
        if self.recompute_rows:
            if strategy not in ("m_shard_allgather", "m_shard_allgather_direct") or w == 1:
                raise ValueError("recompute_rows applies to the all-gather strategies on more "
                                 "than one rank")
            if not 0 < self.recompute_rows < rows_local:
                raise ValueError("recompute_rows=%d must lie inside the %d-row block"
                                 % (self.recompute_rows, rows_local))
            rows_local -= self.recompute_rows          # rows per block that still travel
        if panels is None:
            # a panel's product must still fill the chip: at least one 256x256 output tile per
            # CU (256 of them), otherwise the pipelining costs more MFMA time than the overlap
            # hides. 16384 columns -> panels of >= 1024 rows (8 ranks: 2 panels of the 2048-row
            # shard; 2 ranks: 8 panels of the 8192-row shard).
            panels = 1
            if w > 1 and strategy not in ("replicated", "m_shard", "summa_2d"):
                n_local = self.n // self.grid_cols if strategy == "block_2d_allgather" else self.n
                col_tiles = max(1, (n_local + 255) // 256)
                min_rows = 256 * max(1, (256 + col_tiles - 1) // col_tiles)
                panels = max(1, min(8, rows_local // min_rows))
        self.panels = max(1, int(panels))
        if strategy in ("m_shard_allgather", "m_shard_allgather_direct", "block_2d_allgather") \
                and rows_local % 256 == 0 and rows_local // 256 >= self.panels:
            # panels of whole 256-row output tiles (a ragged last tile row costs a full tile's
            # time), as even as the tile count allows; the same split on every rank
            units = rows_local // 256
            self._panel_rows = [256 * (units // self.panels + (1 if i < units % self.panels else 0))
                                for i in range(self.panels)]
        else:
            while rows_local % self.panels or (strategy == "k_shard_reduce_scatter"
                                               and (self.m // self.panels) % w):
                self.panels -= 1
            self._panel_rows = [rows_local // self.panels] * self.panels
        self._panel_start = [sum(self._panel_rows[:i]) for i in range(self.panels)]
        self.a: Optional[torch.Tensor] = None
        self.bt: Optional[torch.Tensor] = None
        self.c: Optional[torch.Tensor] = None
        self._partial: Optional[torch.Tensor] = None
        self.flops_per_step = 2.0 * self.m * self.n * self.k

    # -- operands ------------------------------------------------------------------------
    def _window(self, which: str, rows: int, cols: int, row0: int, col0: int,
                global_cols: int) -> torch.Tensor:
        """Window of global A ([M,K], stream op_seed 1) or global Bt ([N,K], op_seed 2)."""
        op_seed = 1 if which == "a" else 2
        normal = self.distribution == "normal"
        a, b = (0.0, 1.0) if normal else (-1.0, 2.0)     # N(0,1) or U(-1,1)
        if self.on_gpu:
            if self.scaling == "mx":
                src = self._hip.philox_window(rows, cols, self.seed, op_seed, row0, col0,
                                              global_cols, torch.bfloat16, self.device, normal,
                                              a, b)
                return self._hip.quantize_mx_fp8(src, 0 if which == "a" else 1)
            return self._hip.philox_window(rows, cols, self.seed, op_seed, row0, col0,
                                           global_cols, self.dtype, self.device, normal, a, b)
        host = philox.window(rows, cols, self.seed, op_seed, row0, col0, global_cols, normal)
        return (torch.from_numpy(host) * b + a).to(self.device, self.dtype)

    def init_operands(self) -> None:
        m, n, k, w, r = self.m, self.n, self.k, self.world, self.rank
        s = self.strategy
        if s == "replicated":
            self.a = self._window("a", m, k, 0, 0, k)
            self.bt = self._window("b", n, k, 0, 0, k)
        elif s.startswith("m_shard"):
            ml = m // w
            self.a = self._window("a", ml, k, r * ml, 0, k)
            self.bt = self._window("b", n, k, 0, 0, k)
            x = self.recompute_rows
            if x:
                # the first x rows of EVERY block, in rank order, as one contiguous operand
                self.a_heads = torch.empty((w * x, k), dtype=self.dtype, device=self.device)
                for q in range(w):
                    self.a_heads[q * x:(q + 1) * x].copy_(
                        self.a[:x] if q == r else self._window("a", x, k, q * ml, 0, k))
        elif s.startswith("k_shard"):
            kl = k // w
            self.a = self._window("a", m, kl, 0, r * kl, k)
            self.bt = self._window("b", n, kl, 0, r * kl, k)
        elif s == "block_2d_allgather":
            mb, nb = m // self.grid_rows, n // self.grid_cols
            self.a = self._window("a", mb, k, self.pr * mb, 0, k)
            self.bt = self._window("b", nb, k, self.pc * nb, 0, k)
        else:
            # summa_2d: rank (pr, pc) owns rows M_pr of A for the K panels of grid column pc and
            # rows N_pc of Bt for the K panels of grid row pr; one contiguous tensor per panel
            P, Q, L = self.grid_rows, self.grid_cols, self.k_panels
            m0, ml = _split(m, P, self.pr)
            n0, nl = _split(n, Q, self.pc)
            kp = k // L
            self.a_panels = {l: self._window("a", ml, kp, m0, l * kp, k)
                             for l in range(L) if l * Q // L == self.pc}
            self.b_panels = {l: self._window("b", nl, kp, n0, l * kp, k)
                             for l in range(L) if l * P // L == self.pr}
            self.a = next(iter(self.a_panels.values()))
            self.bt = next(iter(self.b_panels.values()))
        self._alloc_result()

    def operand_shapes(self) -> Tuple[Tuple[int, int], Tuple[int, int]]:
        """Shapes of the blocks ``bind_operands`` expects: (left, right); the right one in this
        engine's ``b_layout``."""
        m, n, k, w, s = self.m, self.n, self.k, self.world, self.strategy
        if s == "replicated":
            ml, nl, kl = m, n, k
        elif s.startswith("m_shard"):
            ml, nl, kl = m // w, n, k
        elif s.startswith("k_shard"):
            ml, nl, kl = m, n, k // w
        elif s == "block_2d_allgather":
            ml, nl, kl = m // self.grid_rows, n // self.grid_cols, k
        else:
            raise ValueError("bind_operands does not apply to %s (operands are K panels)" % s)
        return (ml, kl), ((nl, kl) if self.b_layout == "nk" else (kl, nl))

    def bind_operands(self, a: torch.Tensor, b: torch.Tensor) -> None:
        """Run on these tensors -- this rank's blocks of the operands, e.g. the storage of two
        variables -- instead of generated ones. They are read in place every step (row-strided
        views are fine); result buffers are allocated on the first call and kept."""
        want_a, want_b = self.operand_shapes()
        if tuple(a.shape) != want_a or tuple(b.shape) != want_b:
            raise ValueError("%s on %d rank(s) wants blocks %s and %s, got %s and %s"
                             % (self.strategy, self.world, want_a, want_b, tuple(a.shape),
                                tuple(b.shape)))
        if self.recompute_rows:
            raise ValueError("recompute_rows needs the peers' rows of A: generated operands only")
        first = self.c is None
        self.a, self.bt = a, b
        if first:
            self._alloc_result()

    def _right(self, b: torch.Tensor) -> torch.Tensor:
        """The right-hand operand as the bf16 / fp8 kernels read it ([N, K]). A bound [K, N]
        operand is transposed once into an image that is kept while the operand keeps its
        storage and ``refresh_images`` has not been called (its owner wrote to it)."""
        if self.b_layout == "nk":
            return b
        key = (b.data_ptr(), tuple(b.shape), tuple(b.stride()))
        if self._image is None or self._image[0] != key:
            self._image = (key, self._hip.transpose2d(b))
        return self._image[1]

    def refresh_images(self) -> None:
        self._image = None

    def _alloc_result(self) -> None:
        m, n, w = self.m, self.n, self.world
        s = self.strategy
        dev, dt = self.device, self.out_dtype
        if s in ("m_shard_allgather_direct", "k_shard_allreduce_direct"):
            from .peer_mem import SymmetricBuffer
            # (with recompute_rows the m rows are [world*x heads][world x (m/world - x) bodies])
            self._sym_c = SymmetricBuffer(self.comm, (m, n), dt, dev, self.peer_transport)
            self.c = self._sym_c.local
            self._sym_sets = [self._sym_c]
            if self.pipelined:
                self._sym_sets.append(SymmetricBuffer(self.comm, (m, n), dt, dev,
                                                      self.peer_transport))
            if s == "k_shard_allreduce_direct":
                self._sym_partial = SymmetricBuffer(self.comm, (m, n), self.reduce_dtype, dev)
                self._stage = None             # fp32 rows of a mixed-precision reduction
        elif s in ("replicated", "m_shard_allgather", "k_shard_allreduce"):
            self.c = torch.empty((m, n), dtype=dt, device=dev)
        elif s in ("m_shard", "k_shard_reduce_scatter"):
            self.c = torch.empty((m // w, n), dtype=dt, device=dev)
        elif s == "block_2d_allgather":
            mb, nb = m // self.grid_rows, n // self.grid_cols
            self.c = torch.empty((m, n), dtype=dt, device=dev)
            self._local = torch.empty((mb, nb), dtype=dt, device=dev)
            # one gathered image per panel: [world, rows of the panel, nb]
            self._stages = [torch.empty((w, rows, nb), dtype=dt, device=dev)
                            for rows in self._panel_rows]
        else:
            _, ml = _split(m, self.grid_rows, self.pr)
            _, nl = _split(n, self.grid_cols, self.pc)
            kp = self.k // self.k_panels
            self.c = torch.empty((ml, nl), dtype=torch.float32, device=dev)
            # two receive buffers per operand: panel l+1 lands while panel l is multiplied
            self._a_recv = [torch.empty((ml, kp), dtype=self.dtype, device=dev) for _ in (0, 1)]
            self._b_recv = [torch.empty((nl, kp), dtype=self.dtype, device=dev) for _ in (0, 1)]
        if s == "k_shard_reduce_scatter" or (s == "k_shard_allreduce"
                                             and self.reduce_dtype != dt):
            self._partial = torch.empty((m, n), dtype=self.reduce_dtype, device=dev)
        if s == "m_shard_allgather":
            body = m // w - self.recompute_rows
            self._local = torch.empty((body, n), dtype=dt, device=dev)
            self._ag_sets = [(self.c, self._local)]
            if self.pipelined:
                self._ag_sets.append((torch.empty((m, n), dtype=dt, device=dev),
                                      torch.empty((body, n), dtype=dt, device=dev)))

    # -- local product -------------------------------------------------------------------
    def _gemm(self, a: torch.Tensor, bt: torch.Tensor, out: torch.Tensor,
              accumulate: bool = False, mirrors: Optional[torch.Tensor] = None) -> None:
        """out (+)= a @ op(bt) (a [m,k]; bt [n,k], or [k,n] under ``b_layout="kn"``) on the
        hand-written kernels (GPU) or torch (CPU). ``mirrors``: device table of further
        destinations the bf16 kernel's epilogue stores the result into as well."""
        kn = self.b_layout == "kn"
        if self.tracer is not None:
            tracer, self.tracer = self.tracer, None
            try:
                with tracer.span("panel_product[%dx%dx%d]" % (a.shape[0], out.shape[1],
                                                            a.shape[1]), "panel_product"):
                    return self._gemm(a, bt, out, accumulate, mirrors)
            finally:
                self.tracer = tracer
        if mirrors is not None:
            self._hip.gemm_bf16_nt_mirrored(a, bt, out, mirrors)
            return
        if self.on_gpu:
            beta = 1.0 if accumulate else 0.0
            if self.scaling == "mx":
                self._hip.gemm_fp8_mx_nt(a, bt, out.dtype, out=out, beta=beta)
            elif a.dtype == torch.bfloat16:
                self._hip.gemm_bf16_nt(a, self._right(bt), out.dtype, out=out, beta=beta,
                                       splits=self.k_splits)
            elif a.dtype == torch.float8_e4m3fn:
                self._hip.gemm_fp8_nt(a, self._right(bt), out.dtype, out=out, beta=beta)
            elif a.dtype == torch.float32:
                plan = self.plan_mnk if (self.plan_mnk and self.plan_mnk[2] == a.shape[1]) \
                    else None
                splits = None if plan is not None else self.k_splits
                if accumulate:
                    out.add_(self._hip.gemm_f32(a, bt, trans_b=not kn, splits=splits,
                                                plan_mnk=plan))
                else:
                    self._hip.gemm_f32(a, bt, trans_b=not kn, out=out, splits=splits,
                                       plan_mnk=plan)
            else:
                raise TypeError("unsupported GEMM dtype %s" % a.dtype)
        elif a.dtype == torch.float32 and out.dtype == torch.float32 and not accumulate:
            torch.matmul(a, bt if kn else bt.t(), out=out)
        else:
            prod = a.float() @ (bt.float() if kn else bt.float().t())
            if accumulate:
                out.add_(prod.to(out.dtype))
            else:
                out.copy_(prod.to(out.dtype))

    # -- one step ------------------------------------------------------------------------
    def step(self) -> torch.Tensor:
        if self.a is None:
            self.init_operands()
        s = self.strategy
        if s in ("replicated", "m_shard"):
            self._gemm(self.a, self.bt, self.c)
            return self.c
        if s == "m_shard_allgather":
            return self._step_m_allgather()
        if s == "k_shard_allreduce":
            return self._step_k_allreduce()
        if s == "k_shard_reduce_scatter":
            return self._step_k_reduce_scatter()
        if s == "m_shard_allgather_direct":
            return self._step_m_allgather_direct()
        if s == "k_shard_allreduce_direct":
            return self._step_k_allreduce_direct()
        if s == "block_2d_allgather":
            return self._step_block_2d()
        return self._step_summa()

    def _step_block_2d(self) -> torch.Tensor:
        """Rank (i, j) of the P x Q grid multiplies its rows of A by its columns of B in row
        panels; the all-gather of panel p's blocks runs while panel p+1 is computed. The
        gathered blocks are laid into row-major C by block copies once their panel has landed."""
        P, Q = self.grid_rows, self.grid_cols
        mb, nb = self.m // P, self.n // Q
        cols = self.bt
        works = []
        for idx, (start, rows) in enumerate(zip(self._panel_start, self._panel_rows)):
            local = self._local[start:start + rows]
            self._gemm(self.a[start:start + rows], cols, local)
            works.append(self.comm.all_gather(self._stages[idx], local, async_op=True))
        for idx, (start, rows) in enumerate(zip(self._panel_start, self._panel_rows)):
            works[idx].wait()
            stage = self._stages[idx]
            for q in range(self.world):
                i, j = divmod(q, Q)
                dst = self.c[i * mb + start:i * mb + start + rows, j * nb:(j + 1) * nb]
                if self.on_gpu:
                    self._hip.copy_2d(stage[q], dst)
                else:
                    dst.copy_(stage[q])
        return self.c

    def _step_m_allgather_direct(self) -> torch.Tensor:
        """Row panels of the local block are computed straight into this rank's rows of the
        symmetric result buffer; the push of panel p to all peers (one copy per peer, each on
        its own stream and link) is queued behind panel p's product and runs while panel p+1
        is computed. The exchange is completed on the device (sequence-number flags written
        behind the copies, one wait kernel on the compute stream); pipelined steps publish
        their flags at once and wait for the peers' only when the buffer set comes round again,
        so the compute stream never idles behind a transfer. Without the flag page: one host
        synchronisation + barrier per step."""
        w, x = self.world, self.recompute_rows
        body = self.m // w - x                 # rows of a block that travel
        base = w * x + self.rank * body        # (x = 0: plain row order)
        which = self._steps_done % len(self._sym_sets)
        sym = self._sym_sets[which]
        if self.pipelined and self._set_works[which]:
            sym.finish()                       # the exchange that used this set two steps ago
            self._set_works[which] = []
        self.c = sym.local
        sym.tracer = self.tracer
        fused = self.peer_transport == "epilogue" and w > 1
        for start, rows in zip(self._panel_start, self._panel_rows):
            r0 = base + start
            if fused:
                sym.begin_epilogue_push()
                self._gemm(self.a[x + start:x + start + rows], self.bt, self.c[r0:r0 + rows],
                           mirrors=sym.mirror_table(r0, rows))
            else:
                self._gemm(self.a[x + start:x + start + rows], self.bt, self.c[r0:r0 + rows])
                sym.push_rows(r0, rows)
        if x:
            # heads of all blocks, computed here while the links carry the bodies
            self._gemm(self.a_heads, self.bt, self.c[:w * x])
        if self.pipelined:
            if sym.flags is not None:
                sym.publish()                  # peers learn of the rows now; we wait later
            self._set_works[which] = [True]
        else:
            sym.finish()
        self._steps_done += 1
        return self.c

    def _step_k_allreduce_direct(self) -> torch.Tensor:
        """Two-shot all-reduce over peer memory: every rank pulls its M/world rows of all
        ranks' partial products (N-1 concurrent copies), sums them in rank order, and pushes
        the finished rows to all peers."""
        from .peer_mem import reduce_scatter_rows
        w, r = self.world, self.rank
        per = self.m // w
        part = self._sym_partial
        part.before_overwrite()                # peers have read the previous step's partials
        self._gemm(self.a, self.bt, part.local)
        part.quiesce()
        mine = self.c[r * per:(r + 1) * per]
        if self.reduce_dtype == self.out_dtype:
            reduce_scatter_rows(part, mine, r * per, per)
        else:
            if self._stage is None or self._stage.shape != (per, self.n):
                self._stage = torch.empty((per, self.n), dtype=self.reduce_dtype,
                                          device=self.device)
            reduce_scatter_rows(part, self._stage, r * per, per)
            mine.copy_(self._stage)
        self._sym_c.push_rows(r * per, per)
        self._sym_c.finish()
        return self.c

    def _step_m_allgather(self) -> torch.Tensor:
        """Row panels of the local block: GEMM panel p, then all-gather it while panel p+1 is
        computed. ``c`` is stored panel-major ([panel][rank][rows][N], behind the recomputed
        heads if any); ``result_dense`` undoes the permutation."""
        w, x = self.world, self.recompute_rows
        which = self._steps_done % len(self._ag_sets)
        self.c, local_buf = self._ag_sets[which]
        for wk in self._set_works[which]:      # pipelined: collectives of two steps ago
            wk.wait()
        self._set_works[which] = []
        works = []
        for start, rows in zip(self._panel_start, self._panel_rows):
            local = local_buf[start:start + rows]
            self._gemm(self.a[x + start:x + start + rows], self.bt, local)
            works.append(self.comm.all_gather(self._gathered_panel(start, rows), local,
                                              async_op=True))
        if x:
            # heads of all blocks, computed here while the links carry the bodies
            self._gemm(self.a_heads, self.bt, self.c[:w * x])
        if self.pipelined:
            self._set_works[which] = works
        else:
            for wk in works:
                wk.wait()
        self._steps_done += 1
        return self.c

    def _gathered_panel(self, start: int, rows: int) -> torch.Tensor:
        """[world, rows, N] view of ``c`` that the all-gather of the panel starting at local
        body row ``start`` fills."""
        w = self.world
        off = w * self.recompute_rows + w * start
        return self.c[off:off + w * rows].view(w, rows, self.n)

    def _step_k_allreduce(self) -> torch.Tensor:
        p = self.panels
        rows = self.m // p
        target = self._partial if self._partial is not None else self.c
        works = []
        for i in range(p):
            panel = target[i * rows:(i + 1) * rows]
            self._gemm(self.a[i * rows:(i + 1) * rows], self.bt, panel)
            works.append(self.comm.all_reduce(panel, "sum", async_op=True))
        for wk in works:
            wk.wait()
        if target is not self.c:
            if self.on_gpu:
                self._hip.cast(target, self.c.dtype, out=self.c)
            else:
                self.c.copy_(target)
        return self.c

    def _step_k_reduce_scatter(self) -> torch.Tensor:
        """Panel p of the partial product is reduce-scattered over ranks: rank r ends up with
        rows [r*rows/w, (r+1)*rows/w) of every panel (``c`` stored [panel][rows/w][N])."""
        w, p = self.world, self.panels
        rows = self.m // p
        out = self.c.view(p, rows // w, self.n)
        works = []
        for i in range(p):
            panel = self._partial[i * rows:(i + 1) * rows]
            self._gemm(self.a[i * rows:(i + 1) * rows], self.bt, panel)
            if self.reduce_dtype == self.dtype:
                works.append(self.comm.reduce_scatter(out[i], panel, "sum", async_op=True))
            else:
                tmp = torch.empty((rows // w, self.n), dtype=self.reduce_dtype,
                                  device=self.device)
                self.comm.reduce_scatter(tmp, panel, "sum")
                out[i].copy_(tmp)
        for wk in works:
            wk.wait()
        return self.c

    def _step_summa(self) -> torch.Tensor:
        """SUMMA on the P x Q grid: for K panel l, A's panel is broadcast along grid rows from
        grid column ``l*Q/L`` and Bt's panel along grid columns from grid row ``l*P/L``; every
        rank accumulates its C block in fp32. The broadcasts of panel l+1 are issued before
        the product of panel l so the links and the matrix cores work at the same time."""
        P, Q, L = self.grid_rows, self.grid_cols, self.k_panels

        def issue(l: int):
            a_src, b_src = l * Q // L, l * P // L
            a_blk = self.a_panels[l] if self.pc == a_src else self._a_recv[l & 1]
            b_blk = self.b_panels[l] if self.pr == b_src else self._b_recv[l & 1]
            wa = self.row_comm.broadcast(a_blk, src=a_src, async_op=True) if Q > 1 else None
            wb = self.col_comm.broadcast(b_blk, src=b_src, async_op=True) if P > 1 else None
            return a_blk, b_blk, wa, wb

        nxt = issue(0)
        for l in range(L):
            a_blk, b_blk, wa, wb = nxt
            if wa is not None:
                wa.wait()
            if wb is not None:
                wb.wait()
            if l + 1 < L:
                nxt = issue(l + 1)
            self._gemm(a_blk, b_blk, self.c, accumulate=(l > 0))
        return self.c

    def time_local_product(self, iters: int = 3) -> float:
        """Seconds for this rank's own row-block product alone (no communication), GPU only:
        the compute side of the balance ``balanced_recompute_rows`` solves."""
        rows = self.a.shape[0]
        scratch = torch.empty((rows, self.n), dtype=self.out_dtype, device=self.device)
        self._gemm(self.a, self.bt, scratch)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            self._gemm(self.a, self.bt, scratch)
        stop.record()
        stop.synchronize()
        return start.elapsed_time(stop) * 1e-3 / iters

    def wait(self) -> None:
        """Complete every collective a pipelined step left in flight (no-op otherwise): after
        it, ``c`` / ``result_dense()`` of the latest step are whole on every rank."""
        if not self.pipelined:
            return
        if self.strategy == "m_shard_allgather_direct":
            for which, sym in enumerate(self._sym_sets):
                if self._set_works[which]:
                    sym.finish()
                    self._set_works[which] = []
            if any(sym.flags is not None for sym in self._sym_sets):
                # device-side completion: the waits are queued, not run -- drain and look at
                # the status words (a peer that never signalled shows up here as CommError)
                torch.cuda.current_stream(self.device).synchronize()
                for sym in self._sym_sets:
                    sym.check()
        else:
            for which in range(len(self._set_works)):
                for wk in self._set_works[which]:
                    wk.wait()
                self._set_works[which] = []

    # -- checking ------------------------------------------------------------------------
    def result_dense(self) -> torch.Tensor:
        """This rank's result in plain row-major order (undoes the panel-major layouts)."""
        s, w, p, x = self.strategy, self.world, self.panels, self.recompute_rows
        if s == "m_shard_allgather" and (p > 1 or x):
            bodies = torch.cat([self._gathered_panel(start, rows) for start, rows
                                in zip(self._panel_start, self._panel_rows)], dim=1)
        elif x:
            bodies = self.c[w * x:].view(w, self.m // w - x, self.n)
        else:
            bodies = None
        if bodies is not None:
            if x:
                bodies = torch.cat([self.c[:w * x].view(w, x, self.n), bodies], dim=1)
            return bodies.reshape(self.m, self.n)
        if s == "k_shard_reduce_scatter":
            return self.c   # [panel][rows/w][N]; rows_of_result() gives the global row ids
        return self.c

    def result_rowmajor(self) -> torch.Tensor:
        """The latest step's full C in plain row order in a buffer that keeps its address (for
        the strategies that leave full C on every rank). The RCCL all-gather with several panels
        stores C panel-major; its chunks -- one per (panel, rank), each contiguous -- are moved
        into row order by one multi-tensor copy launch. Everything else already is row-major."""
        s, w = self.strategy, self.world
        if s not in ("replicated", "m_shard_allgather", "k_shard_allreduce",
                     "m_shard_allgather_direct", "k_shard_allreduce_direct",
                     "block_2d_allgather"):
            raise ValueError("%s leaves C sharded" % s)
        if self.recompute_rows:
            return self.result_dense()
        if s != "m_shard_allgather" or self.panels == 1:
            return self.c
        if self._c_dense is None:
            self._c_dense = torch.empty((self.m, self.n), dtype=self.out_dtype,
                                        device=self.device)
        mb = self.m // w
        src, dst = [], []
        for start, rows in zip(self._panel_start, self._panel_rows):
            gathered = self._gathered_panel(start, rows)             # [w, rows, N]
            for q in range(w):
                src.append(gathered[q])
                dst.append(self._c_dense[q * mb + start:q * mb + start + rows])
        if self.on_gpu:
            self._hip.multi_copy(src, dst)
        else:
            for a, b in zip(src, dst):
                b.copy_(a)
        return self._c_dense

    def rows_of_result(self) -> np.ndarray:
        """Global row index of every row of ``result_dense()``."""
        s, w, r, p = self.strategy, self.world, self.rank, self.panels
        if s in ("replicated", "m_shard_allgather", "k_shard_allreduce",
                 "m_shard_allgather_direct", "k_shard_allreduce_direct", "block_2d_allgather"):
            return np.arange(self.m)
        if s == "m_shard":
            return np.arange(self.m // w) + r * (self.m // w)
        if s == "k_shard_reduce_scatter":
            rows = self.m // p
            per = rows // w
            return np.concatenate([i * rows + r * per + np.arange(per) for i in range(p)])
        m0, ml = _split(self.m, self.grid_rows, self.pr)
        return np.arange(ml) + m0

    def cols_of_result(self) -> np.ndarray:
        if self.strategy == "summa_2d":
            n0, nl = _split(self.n, self.grid_cols, self.pc)
            return np.arange(nl) + n0
        return np.arange(self.n)

    def reference(self, rows: np.ndarray, cols: np.ndarray) -> torch.Tensor:
        """fp64 reference of C[rows][:, cols] computed on the host from the global streams
        (use small slices: it regenerates whole rows of A and Bt)."""
        normal = self.distribution == "normal"
        a, b = (0.0, 1.0) if normal else (-1.0, 2.0)
        a_rows = np.stack([philox.window(1, self.k, self.seed, 1, int(i), 0, self.k, normal)[0]
                           for i in rows])
        b_rows = np.stack([philox.window(1, self.k, self.seed, 2, int(j), 0, self.k, normal)[0]
                           for j in cols])
        if self.scaling == "mx":
            # what the MX quantiser keeps: bf16 source, per 32 k the smallest power-of-two scale
            # under which the block fits e4m3, elements rounded to e4m3
            def mx(x):
                x = x.to(torch.bfloat16).float().view(x.shape[0], -1, 32)
                amax = x.abs().amax(dim=2, keepdim=True)
                e = torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / 448.0))
                scale = torch.where(amax > 0, torch.pow(2.0, e), torch.ones_like(amax))
                q = (x / scale).to(torch.float8_e4m3fn).float() * scale
                return q.view(q.shape[0], -1).double()
            at = mx(torch.from_numpy(a_rows) * b + a)
            bt = mx(torch.from_numpy(b_rows) * b + a)
            return at @ bt.t()
        at = (torch.from_numpy(a_rows) * b + a).to(self.dtype).double()
        bt = (torch.from_numpy(b_rows) * b + a).to(self.dtype).double()
        # (torch's fp8 cast rounds to nearest even like the device fill; U(-1,1) and N(0,1)
        # samples are far below the e4m3 saturation point, where the two would differ)
        return at @ bt.t()
