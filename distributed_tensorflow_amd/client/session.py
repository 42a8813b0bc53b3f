"""Session: prune -> plan -> execute for ``run(fetches, feed_dict)``.

The reference's hot loop is one blocking
``sess.run([train_op, cross_entropy, summary_op, global_step], feed_dict={x:..., y_:...})``
per step (``tf_distributed.py:111``) plus an eval-only ``sess.run(accuracy, ...)`` with a
different fetch signature (``:126``). ``Session.run`` therefore:

1. flattens the fetch structure (ops fetch as ``None``, ``None`` entries stay ``None``);
2. looks up / builds an ``ExecutionPlan`` keyed by *(fetches, fed tensors)*: the subgraph
   reachable from the fetches (stopping at fed tensors), rewritten by the fusion pass and
   topologically ordered, with variables grouped by the store that owns them;
3. executes the plan. On an MI355X the plan's device phase is launched eagerly once (warm-up),
   then captured into a HIP graph keyed by the feed shapes, so that a steady-state step is one
   graph launch + one packed device->host copy of the fetched values;
4. runs the host phase (summary protobuf encoding) on the fetched scalars.

Variable placement decides *where state lives*: ops on ``/job:ps/task:K`` are served by the
store bound to that task (a remote parameter server in CPU plumbing mode; the local HBM store
in single-process and collective modes).
"""
from __future__ import annotations

import contextlib
import functools
import os
import threading
import time
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch

from ..framework import dtypes
from ..framework.device import DeviceSpec
from ..framework.graph import Graph, Operation, Tensor, get_default_graph, value_inputs
from ..framework.op_registry import get_opdef
from ..ops import ref_kernels  # noqa: F401  (registers the reference kernels)
from ..parallel import sharded_ops  # noqa: F401  (registers the sharded-product kernels)
from . import fusion
from .variable_store import FailedPreconditionError, LocalVariableStore

_VAR_TYPES = ("VariableV2",)
# op types that need no kernel of their own on a GPU: views, variable reads, one-time constant
# uploads and host bookkeeping. Everything else must have a hand-written kernel.
_NO_KERNEL_NEEDED = frozenset(("Const", "Identity", "StopGradient", "Reshape", "ReshapeLike",
                               "NoOp", "Placeholder", "VariableV2", "IsVariableInitialized"))
_TABLE_HOST_BYTES = 1 << 20      # host inputs up to this size are read by the gather kernel
_SPIN_SECS = 300e-6              # how long a step's completion word is polled before blocking
_NO_SPIN_FETCH = bool(os.environ.get("DTF_NO_SPIN_FETCH"))
_NO_FETCH_AHEAD = bool(os.environ.get("DTF_NO_FETCH_AHEAD"))
_NO_DIRECT_STEP = bool(os.environ.get("DTF_NO_DIRECT_STEP"))
# a captured step of at most this many library calls is re-issued call by call rather than
# replayed as a HIP graph: a graph launch costs about 15 us of host time plus a gap of several
# microseconds on the GPU between one replay and the next, a direct launch 3-5 us
_DIRECT_MAX_CALLS = int(os.environ.get("DTF_DIRECT_MAX_CALLS", "4"))
_HIP_KERNEL_NODE = 0


def _graph_node_types(graph) -> Optional[List[int]]:
    """hipGraphNodeType of every node of a captured graph that was created with
    ``keep_graph=True`` (None when the runtime cannot be asked)."""
    import ctypes
    try:
        rt = ctypes.CDLL("libamdhip64.so")
        raw = ctypes.c_void_p(graph.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        if rt.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * max(1, n.value))()
        if rt.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
            return None
        types = []
        for i in range(n.value):
            t = ctypes.c_int(-1)
            if rt.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
                return None
            types.append(t.value)
        return types
    except (OSError, AttributeError, RuntimeError):
        return None
# ops that write a variable in place (a gather must not move across them)
_MUTATING_OPS = frozenset(("Assign", "AssignAdd", "AssignSub", "ApplyGradientDescent",
                           "MultiApplySGD", "ApplyOptimizer", "MultiApplyOptimizer",
                           "AffineUpdateVariable", "MultiMovingAverage"))
_PACK_WORDS = 256     # fp32 words of pinned staging per captured step for small fetches


_GC_PAUSE_LOCK = threading.Lock()
_GC_PAUSE_DEPTH = 0
_GC_WAS_ENABLED = False


@contextlib.contextmanager
def _gc_paused():
    """Cyclic garbage collection off for the whole process while ANY thread is inside the block.

    A collection that runs during stream capture can finalise a dropped graph, event or tensor,
    and the allocator / runtime calls that makes are illegal in a capturing thread. The pause is
    a counted guard: the first thread in disables collection and remembers whether it was on,
    the last one out restores it -- so a Supervisor service thread or a second session leaving
    its own capture cannot switch collection back on under a thread that is still capturing."""
    import gc
    global _GC_PAUSE_DEPTH, _GC_WAS_ENABLED
    if os.environ.get("DTF_NO_GC_PAUSE"):       # A/B knob of scripts/gc_during_capture.py only
        yield
        return
    with _GC_PAUSE_LOCK:
        if _GC_PAUSE_DEPTH == 0:
            _GC_WAS_ENABLED = gc.isenabled()
            gc.disable()
        _GC_PAUSE_DEPTH += 1
    try:
        yield
    finally:
        with _GC_PAUSE_LOCK:
            _GC_PAUSE_DEPTH -= 1
            if _GC_PAUSE_DEPTH == 0 and _GC_WAS_ENABLED:
                gc.enable()


class SessionClosedError(RuntimeError):
    """``run`` on a session that has been closed."""


class ConfigProto:
    """Session options (the subset that means something here).

    ``graph_fusion``   -- enable the plan-time fusion pass (default on).
    ``hip_graphs``     -- capture steady-state steps into HIP graphs on GPU (default on).
    ``device``         -- torch device string; default: ``cuda:<LOCAL_RANK>`` when a GPU is
                          visible, else ``cpu``.
    ``strategy`` / ``grad_aggregation`` -- collective runtime options (see ``parallel.runtime``).
    ``capture_collectives`` -- under a collective runtime, capture the step INCLUDING its RCCL
                          calls into the HIP graph (RCCL collectives enqueued during stream
                          capture become graph nodes), so a multi-rank step is one graph launch
                          like a single-GPU one. Default (``None``): on when the communicator's
                          backend is RCCL (``nccl``), off otherwise (gloo and the loopback world
                          move data on the host); ``DTF_CAPTURE_COLLECTIVES=0/1`` overrides.
    ``grad_bucket_bytes`` -- under ``allreduce_sync``: gradients are all-reduced in buckets of
                          about this many bytes, each issued as soon as the backward pass has
                          produced its members, so the collectives overlap the rest of the
                          backward pass (default: the runtime's 32 MiB; a model whose
                          gradients fit one bucket -- the reference MLP's 318 KB -- is reduced
                          in a single collective after the backward pass).
    ``matmul_sharding`` -- under a collective runtime with several workers: how a
                          ``tf.matmul`` of two variables is distributed -- ``"none"`` (every
                          rank multiplies replicas, the reference's layout), ``"m"`` (row blocks
                          of A, all-gather of C), ``"k"`` (inner-dimension blocks, all-reduce of
                          the partial products) or ``"2d"`` (row blocks of A x column blocks of B
                          on the rank grid). Default: ``DTF_MATMUL_SHARDING`` or ``"none"``.
    ``shard_variables`` -- store a variable that only sharded products read as its local block
                          (default on); off keeps replicas and multiplies views of them.
    ``matmul_collective`` -- how a row-sharded product's result travels: ``"rccl"`` (default:
                          panel-wise all-gather through the communicator) or ``"peer"`` (every
                          rank stores its rows into all peers' result buffers over peer-mapped
                          memory; GPUs of one node). ``matmul_panels``: row panels per product
                          (default: as many as keep every panel's grid filling the chip).
    ``rendezvous_timeout_s`` -- how long a task started by hand waits for the other tasks of
                          its ``protocol="rccl"`` cluster (default ``DTF_RENDEZVOUS_TIMEOUT`` or
                          600 s).
    ``allow_torch_fallback`` -- GPU sessions only. Off (default): an op, element type or layout
                          without a hand-written kernel raises when it is planned / first run.
                          On (or ``DTF_ALLOW_TORCH_FALLBACK=1``): such a case runs the torch
                          reference kernel and is counted in ``stats["torch_fallbacks"]`` and
                          listed in ``Session.fallback_log``.
    ``step_fusion``    -- let the planner replace a whole training step it recognises (the
                          two-layer classifier of the reference: both layers, loss, backward and
                          the SGD update) by one node of three launches. Default (``None``): on
                          for GPU sessions; needs ``graph_fusion`` and variables in plain local
                          storage (no parameter server, no gradient exchange).
    ``deferred_fetches`` -- captured GPU steps only. Off (default): ``run`` returns when the
                          step's small results have reached the host, as TensorFlow does. On (or
                          ``DTF_DEFERRED_FETCHES=1``): ``run`` returns as soon as the step is
                          enqueued and the small fp32 results (loss, step counter, accuracy) come
                          back as ``DeferredValue`` objects that wait for the step only when they
                          are looked at (``float(v)``, ``np.asarray(v)``, arithmetic, printing):
                          the host prepares the following steps while the GPU runs this one, up
                          to a ring of 16 steps ahead.
    ``allow_soft_placement`` / ``log_device_placement`` and any other TensorFlow option
    (``device_filters``, ``gpu_options`` ...) -- accepted for compatibility, kept in
    ``tensorflow_only``.
    """

    def __init__(self, graph_fusion: bool = True, hip_graphs: bool = True,
                 device: Optional[str] = None, allow_soft_placement: bool = True,
                 log_device_placement: bool = False, strategy: Optional[str] = None,
                 grad_aggregation: Optional[str] = None,
                 capture_collectives: Optional[bool] = None,
                 grad_bucket_bytes: Optional[int] = None, matmul_sharding: Optional[str] = None,
                 shard_variables: bool = True, allow_torch_fallback: Optional[bool] = None,
                 step_fusion: Optional[bool] = None, deferred_fetches: Optional[bool] = None,
                 matmul_collective: Optional[str] = None, matmul_panels: Optional[int] = None,
                 rendezvous_timeout_s: Optional[float] = None,
                 **tensorflow_only):
        # TensorFlow options without a meaning here (device_filters, gpu_options,
        # intra_op_parallelism_threads, ...): kept, not interpreted, so that a script's
        # ConfigProto(...) call runs unchanged
        self.tensorflow_only = dict(tensorflow_only)
        self.step_fusion = step_fusion
        self.deferred_fetches = bool(os.environ.get("DTF_DEFERRED_FETCHES")) \
            if deferred_fetches is None else bool(deferred_fetches)
        self.allow_torch_fallback = bool(os.environ.get("DTF_ALLOW_TORCH_FALLBACK")) \
            if allow_torch_fallback is None else bool(allow_torch_fallback)
        self.matmul_sharding = matmul_sharding
        if matmul_collective not in (None, "rccl", "peer"):
            raise ValueError("matmul_collective is 'rccl' or 'peer'")
        self.matmul_collective = matmul_collective
        self.matmul_panels = matmul_panels
        self.rendezvous_timeout_s = rendezvous_timeout_s
        self.shard_variables = shard_variables
        self.grad_bucket_bytes = grad_bucket_bytes
        self.grad_aggregation = grad_aggregation
        self.capture_collectives = capture_collectives
        self.graph_fusion = graph_fusion
        self.hip_graphs = hip_graphs
        self.device = device
        self.allow_soft_placement = allow_soft_placement
        self.log_device_placement = log_device_placement
        self.strategy = strategy


class GPUOptions:
    """``tf.GPUOptions(allow_growth=..., per_process_gpu_memory_fraction=...)``: accepted and
    kept; device memory here is managed by the caching allocator of the torch runtime."""

    def __init__(self, **options):
        self.__dict__.update(options)


class RunOptions:
    NO_TRACE, FULL_TRACE = 0, 3

    def __init__(self, trace_level: int = 0):
        self.trace_level = trace_level


class RunMetadata:
    """Receives per-op timings when ``RunOptions(trace_level=FULL_TRACE)`` is passed: one entry
    per plan step and per collective, timed on the device by HIP events and on the host by the
    enqueueing code (``utils/tracing.py``)."""

    def __init__(self):
        self.step_stats: List[Dict[str, Any]] = []

    def to_chrome_trace(self) -> Dict[str, Any]:
        from ..utils.tracing import chrome_trace
        return chrome_trace(self.step_stats)

    def summary(self) -> Dict[str, Dict[str, float]]:
        from ..utils.tracing import summarize
        return summarize(self.step_stats)


def default_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))
                            % max(1, torch.cuda.device_count()))
    return torch.device("cpu")


# ---------------------------------------------------------------------------------------
# execution context: what kernels see
# ---------------------------------------------------------------------------------------
class ExecContext:
    def __init__(self, session: "Session"):
        self.session = session
        self.device = session.device
        self.const_cache: Dict[Any, torch.Tensor] = {}
        self.buffers: Dict[Any, torch.Tensor] = {}     # per-node scratch with stable addresses
        self.out_hint: Optional[Dict[int, torch.Tensor]] = None   # placed outputs of this step
        # while a step is being captured: its ring (ops.hip.RingArgs), the table entry of every
        # fed tensor that is read through the ring instead of being gathered, and the request of
        # the session to have the step's small results gathered by the node that ends the step
        self.ring = None
        self.ring_feeds: Dict[int, int] = {}
        self.pack_request = None
        self.direct_call = None     # set by a node whose whole work is one prepared library call
        self._nd_seed = int.from_bytes(os.urandom(4), "little")

    tracer = None     # the StepTracer of a traced run (``RunOptions.FULL_TRACE``), else None

    def note_fallback(self, op, why: str = "") -> None:
        """A GPU kernel wrapper has no native code for this case (see ``ops/hip_kernels``)."""
        sess = self.session
        entry = (op.type, op.name, why)
        if self.device.type == "cuda" and not sess.config.allow_torch_fallback:
            from ..ops.hip import HipKernelError
            raise HipKernelError(
                "op %s (%s) has no gfx950 kernel for %s; refusing to run torch kernels on the GPU "
                "(ConfigProto(allow_torch_fallback=True) permits it)" % (op.name, op.type, why))
        sess.stats["torch_fallbacks"] += 1
        if len(sess.fallback_log) < 256 and entry not in sess.fallback_log:
            sess.fallback_log.append(entry)

    def nondeterministic_seed(self) -> int:
        self._nd_seed = (self._nd_seed * 1103515245 + 12345) & 0x7FFFFFFF
        return self._nd_seed

    def _store(self, var_op):
        return self.session._store_for(var_op)

    def var_read(self, var_op) -> torch.Tensor:
        return self._store(var_op).read(var_op.name)

    def var_assign(self, var_op, value: torch.Tensor) -> torch.Tensor:
        return self._store(var_op).assign(var_op.name, value)

    def var_assign_add(self, var_op, delta) -> torch.Tensor:
        return self._store(var_op).assign_add(var_op.name, delta)

    def var_apply_sgd(self, var_op, lr: float, grad: torch.Tensor) -> torch.Tensor:
        return self._store(var_op).apply_sgd(var_op.name, lr, grad)

    def var_is_initialized(self, var_op) -> bool:
        return self._store(var_op).is_initialized(var_op.name)

    def uninitialized(self, var_ops) -> List[str]:
        """Names of the given variables that hold no value, one query per owning store."""
        groups: Dict[int, Tuple[Any, List[str]]] = {}
        for v in var_ops:
            st = self._store(v)
            groups.setdefault(id(st), (st, []))[1].append(v.name)
        missing: List[str] = []
        for st, names in groups.values():
            missing.extend(st.uninitialized(names))
        order = {v.name: i for i, v in enumerate(var_ops)}
        return sorted(missing, key=lambda n: order.get(n, 0))

    # -- gradient synchronisation hooks (synchronous strategies) --------------------------
    def reduce_gradients(self, grads, var_ops=None):
        rt = self.session._runtime
        if rt is None:
            return list(grads)
        if var_ops is not None and hasattr(rt, "grad_view"):
            return rt.reduce_gradients(grads, [v.name for v in var_ops])
        return rt.reduce_gradients(grads)

    def sharded_sgd(self, var_ops, grads, lrs) -> bool:
        rt = self.session._runtime
        fn = getattr(rt, "sharded_sgd_step", None)
        return bool(fn and fn([v.name for v in var_ops], grads, lrs))

    def sharded_apply(self, rule, hyper, var_ops, grads, slot_ops=None, powers=(), lr=None,
                      lrs=None, counters=(), deltas=()):
        """Owner-shard update of the runtime (``reduce_scatter_sync``): indices of the variables
        it updated, ``None`` when the runtime has no such path. ``counters_applied()`` says
        whether that call also bumped the ``counters`` handed to it."""
        fn = getattr(self.session._runtime, "sharded_apply", None)
        if fn is None:
            return None
        return fn(rule, hyper, [v.name for v in var_ops], list(grads),
                  [[s.name for s in ops] for ops in slot_ops] if slot_ops is not None else None,
                  powers, lr, lrs, counters, deltas)

    def counters_applied(self) -> bool:
        return bool(getattr(self.session._runtime, "counters_applied", False))


# ---------------------------------------------------------------------------------------
# plans
# ---------------------------------------------------------------------------------------
class _Step:
    __slots__ = ("node", "kernel", "in_slots", "out_slots", "host", "buckets", "grad_views")

    def __init__(self, node, kernel, in_slots, out_slots, host):
        self.buckets = None       # [(bucket id, gradient slots)] complete once this step ran
        self.grad_views = None    # {output index: variable name}: outputs that ARE gradients
        self.node = node
        self.kernel = kernel
        self.in_slots = in_slots
        self.out_slots = out_slots
        self.host = host


class ExecutionPlan:
    def __init__(self):
        self.steps: List[_Step] = []          # device phase, topological
        self.host_steps: List[_Step] = []     # host phase (string-valued ops)
        self.num_slots = 0
        self.feed_slots: List[Tuple[Tensor, int]] = []
        self.fetch_slots: List[Optional[int]] = []   # per flattened fetch; None for ops
        self.host_input_slots: List[int] = []        # device values the host phase needs
        self.var_ops: List[Any] = []                 # VariableV2 ops read by the plan
        self.mutates_state = False
        self.capturable = True
        self.grad_buckets = 0                        # gradient buckets issued during backward
        self.graphs: Dict[Tuple, "_CapturedStep"] = {}
        self.runs = 0


class _StepRing:
    """Host side of a captured step's ring (``csrc/kernels/step_ring.h``): per slot the feed
    addresses of one replay (pinned table the kernels read) and its packed small results
    (pinned buffer the step's last kernel writes and stamps). ``launched`` counts replays and is,
    in stream order, what the device's own counter holds: replay ``n`` uses slot ``n % SLOTS``."""

    SLOTS = 16

    def __init__(self, device: torch.device):
        from ..ops import hip
        n = self.SLOTS
        self.table = torch.zeros(n * hip.RING_FEEDS, dtype=torch.int64).pin_memory()
        self.table_np = self.table.numpy().reshape(n, hip.RING_FEEDS)
        self.out = torch.zeros(n * hip.RING_WORDS, dtype=torch.float32).pin_memory()
        self.out_np = self.out.numpy().reshape(n, hip.RING_WORDS)
        self.ctr = torch.zeros(hip.RING_CTR_WORDS, dtype=torch.int32, device=device)
        self.args = hip.RingArgs(self.table, self.ctr, self.out, n)
        self.words = hip.RING_WORDS
        self.launched = 0
        self.pending: List[Optional["_PendingStep"]] = [None] * n
        self.live: List[Any] = [None] * n         # feed sources of the replay in each slot
        self.skip_spin = 0                        # replays left that block without polling

    @staticmethod
    def stamp(n: int) -> float:
        return float((n & 0x7fffff) + 1)


class _PendingStep:
    """The small results of replay ``n``: still in the ring until ``resolve`` copies them out."""

    __slots__ = ("ring", "n", "slot", "words", "snap", "lock")

    def __init__(self, ring: _StepRing, n: int, slot: int, words: int):
        self.ring, self.n, self.slot, self.words = ring, n, slot, words
        self.snap = None
        self.lock = threading.Lock()

    def resolve(self):
        if self.snap is not None:
            return self.snap
        with self.lock:
            if self.snap is not None:
                return self.snap
            ring = self.ring
            row, want = ring.out_np[self.slot], _StepRing.stamp(self.n)
            last = ring.words - 1
            done = row[last] == want
            if ring.skip_spin > 0:
                ring.skip_spin -= 1              # a long step: block right away for a while
            elif not done and not _NO_SPIN_FETCH:
                # the step's last kernel stamps the slot behind the packed values: poll it for a
                # bounded time (a short step finishes well inside it; the wake-up of a stream
                # synchronisation costs several microseconds)
                deadline = time.perf_counter() + _SPIN_SECS
                while True:
                    if row[last] == want:
                        done = True
                        break
                    if time.perf_counter() > deadline:
                        ring.skip_spin = 64
                        break
            if not done:
                torch.cuda.synchronize(ring.ctr.device)
                if row[last] != want:
                    raise RuntimeError("captured step %d finished without stamping its results "
                                       "(slot %d holds %r, expected %r)"
                                       % (self.n, self.slot, float(row[last]), want))
            self.snap = row[:self.words].copy()  # one copy out of the pinned buffer
            return self.snap

    def value(self, off: int, n: int, shape: tuple):
        snap = self.resolve()
        return snap[off:off + n].reshape(shape) if shape else snap[off]


class _PackRequest:
    """What the session asks of a node that ends a captured step and can gather the step's small
    results itself: ``parts_for`` lists them (fp32, at most 64 words each), ``claim`` says the
    node does it -- the session then adds no gather launch of its own."""

    def __init__(self, want: List[int], vals: List[Any], step: "_Step", limit: int):
        self.want, self.vals, self.step, self.limit = want, vals, step, limit
        self.claimed = False
        self.packed: List[Tuple[int, int, tuple, torch.Tensor]] = []
        self._candidate: List[Tuple[int, int, tuple, torch.Tensor]] = []
        self._unpacked: List[Any] = []
        self.adds_fn = None

    def parts_for(self, node, outs: Sequence[Any]) -> Optional[List[torch.Tensor]]:
        if node is not self.step.node:
            return None
        mine = dict(zip(self.step.out_slots, outs))
        pairs = [(sl, self.vals[sl] if self.vals[sl] is not None else mine.get(sl))
                 for sl in dict.fromkeys(self.want)]
        self._candidate = _packable(pairs, self.limit)
        taken = {p[0] for p in self._candidate}
        self._unpacked = [v for sl, v in pairs if sl not in taken]
        return [p[3] for p in self._candidate]

    def addends(self, parts: Sequence[torch.Tensor]) -> Optional[List[float]]:
        """For a node that gathers AHEAD of the step's update launch: what to add to each part
        (a counter that launch bumps is fetched as old value + increment); ``None`` when a
        fetched value overlaps something the update writes. Without an update behind the node:
        zeros."""
        if self.adds_fn is None:
            return [0.0] * len(parts)
        return self.adds_fn(list(parts), [p[3] for p in self._candidate] + self._unpacked)

    def claim(self) -> None:
        self.packed, self.claimed = self._candidate, True


def _packable(pairs, limit: int):
    """(slot, offset, shape, tensor) of the values that travel in a ring slot: small contiguous
    fp32 GPU tensors, up to 16 of them, ``limit`` words in all."""
    packed, off = [], 0
    for sl, v in pairs:
        if (isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32
                and v.is_contiguous() and 0 < v.numel() <= 64 and off + v.numel() <= limit
                and len(packed) < 16):
            packed.append((sl, off, tuple(v.shape), v))
            off += v.numel()
    return packed


class _CapturedStep:
    """One HIP-graph instantiation of a plan for fixed feed shapes."""

    def __init__(self):
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.ring: Optional[_StepRing] = None
        self.static_feeds: List[torch.Tensor] = []
        self.use_table = False                            # feeds travel as addresses in the ring
        self.pinned: List[Any] = []                       # staging of host feeds
        self.h2d_done: Optional[torch.cuda.Event] = None
        self.host_words = 0
        self.layout: List[Tuple[int, int, tuple, int]] = []   # (slot, offset, shape, numel)
        self.slot_values: List[Any] = []
        self.generation = 0                               # variable-store generation at capture
        self.direct = None          # (library function, prepared argument block) of a step that
                                    # is a single library call: issued directly, not replayed
        self.calls = None           # [(library function, arguments without the stream)] of a
                                    # short step: re-issued in order instead of replayed
        self.watch = None           # communicator's watch(event, what) when the graph holds
        self.watch_events: List[Any] = []                 # captured collectives (one per slot)


class _DeferredHost:
    """The host phase (summaries) of a step whose inputs are still on their way."""

    def __init__(self, session: "Session", plan: ExecutionPlan, slot_vals: Dict[int, Any]):
        self.session, self.plan, self.slot_vals = session, plan, slot_vals
        self.out: Optional[Dict[int, Any]] = None
        self.lock = threading.Lock()

    def get(self, sl: int):
        with self.lock:
            if self.out is None:
                self.out = self.session._run_host_phase(self.plan, self.slot_vals, None)
                self.slot_vals = None
        return self.out[sl]


_DEFERRABLE_HOST_OPS = frozenset(("ScalarSummary", "MergeSummary"))
# node types that read fed tensors through the ring
_RING_READERS = frozenset(("MLPTrainStep", "MLPStepGrads"))
# node types that can gather a captured step's small results on their own last launch although
# the step's update launch still follows them
_PACK_RIDERS = frozenset(("MLPStepGrads", "MSELossStep"))


_default_session = threading.local()


def get_default_session() -> Optional["Session"]:
    stack = getattr(_default_session, "stack", None)
    return stack[-1] if stack else None


class Session:
    def __init__(self, target: str = "", graph: Optional[Graph] = None,
                 config: Optional[ConfigProto] = None):
        self.graph = graph or get_default_graph()
        self.config = config or ConfigProto()
        self.target = target or ""
        self._closed = False
        self._close_hooks: List[Any] = []
        self._plans: Dict[Tuple, ExecutionPlan] = {}
        self._call_cache: Dict[Tuple, Tuple] = {}
        self._eager_buf = self._eager_seq = self._eager_np = None   # pinned staging of _fetch_small
        self._eager_count = 0
        self._fetch_lock = threading.Lock()
        self._lock = threading.RLock()
        # the device phase of run() is serialised per session: the Supervisor's service threads
        # (checkpoint, summaries, step rate) call run() next to the training loop, and a step's
        # collectives, gradient buckets and HIP-graph capture must not interleave with another's
        self._run_lock = threading.RLock()
        self._shard_cache: Tuple[int, Any] = (-1, None)
        self.device = torch.device(self.config.device) if self.config.device \
            else default_device()
        self._runtime = None
        self._stores: Dict[Tuple[str, int], Any] = {}
        if self.target:
            from ..train.server import lookup_server
            server = lookup_server(self.target)
            self._runtime = server.runtime
            if self.config.device is None:
                self.device = self._runtime.device
        if self.device.type == "cuda":
            if self.device.index is None:       # "cuda" = this process's current GPU
                self.device = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.set_device(self.device)
            from ..ops import hip_kernels  # noqa: F401  (registers HIP kernels; fails loudly)
        self._local_store = (self._runtime.local_store if self._runtime is not None
                             else LocalVariableStore(self.device))
        self._ctx = ExecContext(self)
        self.stats = {"runs": 0, "graph_replays": 0, "eager_runs": 0, "captures": 0,
                      "torch_fallbacks": 0}
        self.fallback_log: List[Tuple[str, str, str]] = []

    # -- context-manager / default-session protocol ---------------------------------------
    def __enter__(self) -> "Session":
        stack = getattr(_default_session, "stack", None)
        if stack is None:
            stack = _default_session.stack = []
        stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        _default_session.stack.pop()
        self.close()

    @contextlib.contextmanager
    def as_default(self):
        stack = getattr(_default_session, "stack", None)
        if stack is None:
            stack = _default_session.stack = []
        stack.append(self)
        try:
            yield self
        finally:
            stack.pop()

    def add_close_hook(self, fn) -> None:
        """``fn(session)`` runs at the start of ``close()`` while the session is still usable
        (the Supervisor's final checkpoint hangs here)."""
        self._close_hooks.append(fn)

    def close(self) -> None:
        if self._closed:
            return
        hooks, self._close_hooks = self._close_hooks, []
        for fn in hooks:
            fn(self)
        self._closed = True
        for plan in self._plans.values():
            for cap in plan.graphs.values():
                self._retire(cap)        # results still in a ring outlive the session
            plan.graphs.clear()
        self._plans.clear()
        self._call_cache.clear()

    # -- variable routing ------------------------------------------------------------------
    def _store_for(self, var_op):
        """The store that owns ``var_op``: decided by the variable's device (its ps task)."""
        if self._runtime is None:
            return self._local_store
        return self._runtime.store_for_device(var_op.device)

    @property
    def variable_store(self) -> LocalVariableStore:
        return self._local_store

    # -- fetch handling --------------------------------------------------------------------
    def _flatten(self, fetches) -> Tuple[List[Any], Any]:
        flat: List[Any] = []

        def walk(f):
            if f is None:
                return ("none",)
            if isinstance(f, (list, tuple)):
                return ("seq", type(f), [walk(x) for x in f])
            if isinstance(f, dict):
                return ("dict", [(k, walk(v)) for k, v in f.items()])
            elem = self.graph.as_graph_element(f)
            flat.append(elem)
            return ("leaf", len(flat) - 1)

        return flat, walk(fetches)

    @staticmethod
    def _unflatten(structure, values: List[Any]):
        kind = structure[0]
        if kind == "none":
            return None
        if kind == "leaf":
            return values[structure[1]]
        if kind == "seq":
            items = [Session._unflatten(s, values) for s in structure[2]]
            return items if structure[1] is list else tuple(items)
        return {k: Session._unflatten(s, values) for k, s in structure[1]}

    # -- planning --------------------------------------------------------------------------
    def _build_plan(self, flat: List[Any], fed: List[Tensor]) -> ExecutionPlan:
        fed_set: Set[Tensor] = set(fed)
        fetched_tensors = {f for f in flat if isinstance(f, Tensor)}
        # fetched tensors that are themselves fed need no producer
        live = [f for f in flat if not (isinstance(f, Tensor) and f in fed_set)]

        def prune(rw: fusion.Rewrite) -> List[Any]:
            """Post-order DFS from the fetches over the (rewritten) graph = topological order
            of exactly the nodes the fetches need."""
            order: List[Any] = []
            state: Dict[Any, int] = {}
            for f in live:
                root = rw.resolve(f).op if isinstance(f, Tensor) else rw.resolve_op(f)
                stack = [(root, False)]
                while stack:
                    node, done = stack.pop()
                    if done:
                        order.append(node)
                        continue
                    if state.get(node):
                        continue
                    state[node] = 1
                    stack.append((node, True))
                    deps = []
                    raw = node.attrs.get("_raw_inputs")
                    for _i, t in value_inputs(node):
                        if t in fed_set:
                            continue
                        rt = t if raw else rw.resolve(t)
                        if rt not in fed_set:
                            deps.append(rt.op)
                    deps.extend(rw.resolve_op(c) for c in node.control_inputs)
                    for d in reversed(deps):
                        if not state.get(d):
                            stack.append((d, False))
            return order

        rw = fusion.Rewrite()
        order = prune(rw)
        shard_plan = self._shard_plan()
        stored = getattr(self._runtime, "var_layouts", None)
        if self.config.graph_fusion or shard_plan or stored:
            rw = fusion.build_rewrite(self.graph, sorted(order, key=lambda o: o.id), fed_set,
                                      fetched_tensors, shard_plan=shard_plan,
                                      stored_layouts=stored, fuse=self.config.graph_fusion,
                                      step_fusion=self._step_fusion(),
                                      grads_fusion=self._grads_fusion())
            if rw.tensor_map or rw.op_map:
                order = prune(rw)
            if rw.step_candidate is not None:
                # a whole-step node stands only if it makes every node it replaces unreachable
                # (nothing fetched, and no other op, still needs one of their values)
                rw2 = rw.with_candidate()
                order2 = prune(rw2)
                redundant = set(rw.step_candidate[3])
                if not any(n in redundant for n in order2):
                    rw, order = rw2, order2

        plan = ExecutionPlan()
        slot_of: Dict[Tensor, int] = {}

        def slot(t: Tensor) -> int:
            if t not in slot_of:
                slot_of[t] = plan.num_slots
                plan.num_slots += 1
            return slot_of[t]

        for t in fed:
            plan.feed_slots.append((t, slot(t)))
        use_hip = self.device.type == "cuda"
        for node in order:
            d = get_opdef(node.type)
            kernel = d.hip_kernel if (use_hip and d.hip_kernel is not None) else d.ref_kernel
            if kernel is None:
                raise NotImplementedError("op type %r has no kernel" % node.type)
            if use_hip and d.hip_kernel is None and not d.host_only \
                    and node.type not in _NO_KERNEL_NEEDED:
                if not self.config.allow_torch_fallback:
                    raise NotImplementedError(
                        "op %s (%s) has no gfx950 kernel; refusing to plan a torch kernel on the "
                        "GPU (ConfigProto(allow_torch_fallback=True) permits it)"
                        % (node.name, node.type))
                self.stats["torch_fallbacks"] += 1
                self.fallback_log.append((node.type, node.name, "no kernel"))
            ins = [-1] * len(node.inputs)          # -1: a variable reference, not a value
            raw = node.attrs.get("_raw_inputs")
            for i, t in value_inputs(node):
                rt = t if (t in fed_set or raw) else rw.resolve(t)
                ins[i] = slot(rt)
            outs = [slot(o) for o in node.outputs]
            step = _Step(node, kernel, ins, outs, d.host_only)
            if d.host_only:
                plan.host_steps.append(step)
            else:
                plan.steps.append(step)
            if node.type in _VAR_TYPES:
                plan.var_ops.append(node)
            if d.stateful and node.type not in ("VariableV2", "NoOp", "Placeholder"):
                plan.mutates_state = True
        host_nodes = {s.node for s in plan.host_steps}
        needed: List[int] = []
        for s in plan.host_steps:
            for t, sl in zip(s.node.inputs, s.in_slots):
                if sl < 0:
                    continue
                producer = (t if t in fed_set else rw.resolve(t)).op
                if producer not in host_nodes and sl not in needed:
                    needed.append(sl)
        plan.host_input_slots = needed
        for f in flat:
            if isinstance(f, Tensor):
                rt = f if f in fed_set else rw.resolve(f)
                plan.fetch_slots.append(slot(rt))
            else:
                plan.fetch_slots.append(None)
        # anything that talks to another process, or initialises storage, stays eager
        for s in plan.steps:
            if s.node.type in ("Assign", "RandomStandardNormal", "RandomUniform",
                               "TruncatedNormal", "IsVariableInitialized",
                               "ShardedVariableInit", "GatherShardedVariable"):
                plan.capturable = False
        if self._runtime is not None and not self._runtime.capturable and not (
                self._capture_collectives() and
                getattr(self._runtime, "capturable_with_collectives", False)):
            plan.capturable = False
        if not plan.capturable:
            self._plan_gradient_buckets(plan, {sl: t for t, sl in slot_of.items()})
        self._plan_gradient_views(plan)
        return plan

    def _plan_gradient_views(self, plan: ExecutionPlan) -> None:
        """Under a multi-rank collective runtime the gradients the optimizer step consumes are
        PRODUCED in their slots of the runtime's flat gradient arena (the buffer the collectives
        run on): the producing step is told which of its outputs go where, kernels with a
        placeable output write there directly, and the bucketing copy per gradient disappears.
        Only gradients nobody else reads or fetches are placed (the collective reduces the arena
        in place)."""
        rt = self._runtime
        if rt is None or not hasattr(rt, "grad_view") or plan.grad_buckets:
            return
        # (a world of one reduces nothing -- except under reduce_scatter_sync, whose owner-update
        # path runs on the arena whatever the world size)
        if rt.comm.world_size < 2 and getattr(rt, "strategy", "") != "reduce_scatter_sync":
            return
        producer = {sl: (st, idx) for st in plan.steps for idx, sl in enumerate(st.out_slots)}
        readers: Dict[int, int] = {}
        for st in plan.steps + plan.host_steps:
            for sl in st.in_slots:
                if sl >= 0:
                    readers[sl] = readers.get(sl, 0) + 1
        fetched = {sl for sl in plan.fetch_slots if sl is not None} | set(plan.host_input_slots)
        for st in plan.steps:
            if st.node.type in ("MultiApplySGD", "MultiApplyOptimizer"):
                pairs = [(v.name, sl) for v, sl in zip(st.node.attrs["var_ops"], st.in_slots)]
            elif st.node.type in ("ApplyGradientDescent", "ApplyOptimizer"):
                pairs = [(st.node.inputs[0].op.name, st.in_slots[-1])]
            else:
                continue
            for name, sl in pairs:
                hit = producer.get(sl)
                if hit is None or sl < 0 or readers.get(sl, 0) != 1 or sl in fetched:
                    continue
                src, idx = hit
                if src.node.type in _VAR_TYPES or src.node.type == "Placeholder":
                    continue
                if src.grad_views is None:
                    src.grad_views = {}
                src.grad_views[idx] = name

    def _step_fusion(self) -> bool:
        """Whole-step nodes update variables in place with their own kernels: only where every
        variable is plain local storage (no parameter server, no gradient exchange)."""
        want = self.config.step_fusion
        if want is None:
            want = self.device.type == "cuda"
        if not (self.config.graph_fusion and want):
            return False
        if self._runtime is None:
            return True
        from ..parallel.runtime import CollectiveRuntime, LocalRuntime
        rt = self._runtime
        if type(rt) is LocalRuntime:
            return True
        # a collective world of ONE rank under all-reduce exchanges nothing: its variables are
        # plain local storage too (views of the parameter arena)
        return (type(rt) is CollectiveRuntime and rt.comm.world_size == 1
                and rt.strategy == "allreduce_sync" and not rt.has_sharded_storage())

    def _grads_fusion(self) -> bool:
        """Forward + backward of a recognised model as one node that ends in the gradients (the
        update stays with the optimizer node): wherever the whole-step node cannot stand --
        several ranks, sharded owners, optimizer rules with state -- as long as every variable is
        storage of this process."""
        want = self.config.step_fusion
        if want is None:
            want = self.device.type == "cuda"
        if not (self.config.graph_fusion and want):
            return False
        if self._runtime is None:
            return True
        from ..parallel.runtime import CollectiveRuntime, LocalRuntime
        return type(self._runtime) in (LocalRuntime, CollectiveRuntime)

    def _capture_collectives(self) -> bool:
        want = self.config.capture_collectives
        env = os.environ.get("DTF_CAPTURE_COLLECTIVES")
        if want is None and env is not None:
            want = env not in ("0", "", "false", "False")
        if want is None:
            comm = getattr(self._runtime, "comm", None)
            want = getattr(comm, "backend", "") in ("nccl", "rccl_native") \
                and self.device.type == "cuda"
        return bool(want)

    def _shard_plan(self):
        """The graph-wide sharding plan of variable x variable products (``parallel/planner``),
        recomputed when the graph has grown; ``None`` without a multi-rank collective runtime or
        with ``matmul_sharding="none"``. The runtime is told about it so that the initial-state
        exchange (``after_init``) lays the variables out accordingly."""
        rt = self._runtime
        comm = getattr(rt, "comm", None)
        if rt is None or comm is None or comm.world_size < 2 \
                or not hasattr(rt, "set_shard_plan"):
            return None
        mode = (self.config.matmul_sharding or getattr(rt, "matmul_sharding", None)
                or os.environ.get("DTF_MATMUL_SHARDING") or "none")
        if mode == "none":
            return None
        version = self.graph.version
        if self._shard_cache[0] != version:
            from ..parallel.planner import plan_matmul_sharding
            plan = plan_matmul_sharding(self.graph, comm.world_size, mode,
                                        self.config.shard_variables)
            rt.set_shard_plan(plan)
            self._shard_cache = (version, plan)
        return self._shard_cache[1]

    def _plan_gradient_buckets(self, plan: ExecutionPlan, tensor_of: Dict[int, Tensor]) -> None:
        """Bucketed gradient all-reduce overlapped with the backward pass. The gradients the
        optimizer step consumes are ordered by the plan position of their producers (the
        backward pass yields the last layer's first) and cut into buckets of about
        ``grad_bucket_bytes``; the step that produces a bucket's last member triggers that
        bucket's asynchronous all-reduce, and the optimizer step waits for all of them. Plans
        whose gradients fit one bucket keep the single collective at the optimizer step."""
        rt = self._runtime
        begin = getattr(rt, "begin_gradient_bucket", None)
        if begin is None or rt.comm.world_size == 1 or rt.strategy != "allreduce_sync":
            return
        limit = int(self.config.grad_bucket_bytes or rt.bucket_bytes)
        grad_slots: List[int] = []
        for s in plan.steps:
            if s.node.type in ("MultiApplySGD", "MultiApplyOptimizer"):
                n_grads = len(s.node.attrs["var_ops"])      # (a learning-rate input follows)
                grad_slots.extend(sl for sl in s.in_slots[:n_grads] if sl >= 0)
            elif s.node.type in ("ApplyGradientDescent", "ApplyOptimizer"):
                grad_slots.append(s.in_slots[-1])
        producer = {sl: i for i, s in enumerate(plan.steps) for sl in s.out_slots}
        sized = []
        for sl in grad_slots:
            t = tensor_of.get(sl)
            dims = None if t is None or t.shape is None else list(t.shape)
            if sl not in producer or dims is None or any(d is None for d in dims):
                return
            n = 1
            for d in dims:
                n *= int(d)
            sized.append((producer[sl], sl, n * 4))
        if sum(b for _, _, b in sized) <= limit:
            return
        sized.sort()
        buckets, cur, cur_bytes = [], [], 0
        for idx, sl, nbytes in sized:
            cur.append((idx, sl))
            cur_bytes += nbytes
            if cur_bytes >= limit:
                buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            buckets.append(cur)
        for bid, members in enumerate(buckets):
            trigger = plan.steps[max(i for i, _ in members)]
            if trigger.buckets is None:
                trigger.buckets = []
            trigger.buckets.append((bid, [sl for _, sl in members]))
        plan.grad_buckets = len(buckets)

    # -- execution -------------------------------------------------------------------------
    def _to_device(self, t: Tensor, value: Any) -> torch.Tensor:
        if isinstance(value, torch.Tensor):
            out = value.to(device=self.device, dtype=t.dtype.torch)
        else:
            arr = np.asarray(value)
            if t.dtype.numpy is not None and arr.dtype != np.dtype(t.dtype.numpy):
                arr = arr.astype(t.dtype.numpy)
            # (np.ascontiguousarray would turn a 0-d value into shape (1,))
            out = torch.from_numpy(arr if arr.flags.c_contiguous else arr.copy())
            if t.dtype.numpy is None:
                out = out.to(t.dtype.torch)
            out = out.to(self.device, non_blocking=False)
        if t.shape is not None:
            if len(t.shape) != out.dim() or any(
                    d is not None and d != s for d, s in zip(t.shape, out.shape)):
                raise ValueError("Cannot feed value of shape %s for Tensor %r, which has shape %s"
                                 % (tuple(out.shape), t.name, t.shape))
        return out.contiguous()

    def _prep_feed(self, t: Tensor, value: Any):
        """Feed value for the HIP-graph path: a device tensor is used as is; host data becomes
        a contiguous ndarray of the placeholder's dtype that ``_run_captured`` stages through
        pinned memory (one asynchronous H2D copy straight into the graph's input buffer)."""
        if isinstance(value, torch.Tensor):
            if value.device.type == "cuda" or t.dtype.numpy is None:
                if (value.device == self.device and value.dtype == t.dtype.torch
                        and value.is_contiguous() and t.shape is not None
                        and len(t.shape) == value.dim()
                        and all(d is None or d == s for d, s in zip(t.shape, value.shape))):
                    return value            # the common case: a batch view of a resident data set
                return self._to_device(t, value)
            value = value.detach().numpy()
        if t.dtype.numpy is None:
            return self._to_device(t, value)
        arr = np.asarray(value)
        if arr.dtype != np.dtype(t.dtype.numpy):
            arr = arr.astype(t.dtype.numpy)
        if t.shape is not None:
            if len(t.shape) != arr.ndim or any(
                    d is not None and d != s for d, s in zip(t.shape, arr.shape)):
                raise ValueError("Cannot feed value of shape %s for Tensor %r, which has shape %s"
                                 % (tuple(arr.shape), t.name, t.shape))
        return arr

    def _run_steps(self, steps: Sequence[_Step], vals: List[Any], trace=None) -> None:
        ctx = self._ctx
        for s in steps:
            ins = [vals[i] if i >= 0 else None for i in s.in_slots]
            rec = trace.begin(s.node.name, s.node.type) if trace is not None else None
            if s.grad_views is None:
                outs = s.kernel(s.node, ins, ctx)
            else:
                outs = self._run_placed(s, ins, ctx)
            if rec is not None:
                trace.end(rec)
            for sl, o in zip(s.out_slots, outs):
                vals[sl] = o
            if s.buckets is not None:
                for bid, slots in s.buckets:
                    self._runtime.begin_gradient_bucket(bid, [vals[sl] for sl in slots])

    def _run_placed(self, s: _Step, ins: List[Any], ctx: ExecContext) -> List[Any]:
        """Run a step some of whose outputs are gradients with a slot in the runtime's flat
        arena: the kernel sees the slots as placement hints; an output that did not land there
        (a kernel without a placeable output) is moved by one block copy."""
        views = {}
        for idx, name in s.grad_views.items():
            v = self._runtime.grad_view(name)
            if v is not None:
                views[idx] = v
        ctx.out_hint = views or None
        try:
            outs = list(s.kernel(s.node, ins, ctx))
        finally:
            ctx.out_hint = None
        for idx, view in views.items():
            o = outs[idx]
            if not isinstance(o, torch.Tensor) or o.dtype != view.dtype \
                    or o.numel() != view.numel():
                continue
            if o.data_ptr() != view.data_ptr():
                dst = view.view(o.shape)
                if o.is_cuda and o.is_contiguous() and o.numel():
                    from ..ops import hip
                    hip.copy_2d(o.reshape(1, -1), dst.view(1, -1))
                else:
                    dst.copy_(o)
                outs[idx] = dst
            elif o.shape != view.shape:
                outs[idx] = o
        return outs

    @staticmethod
    def _to_host(value: Any) -> Any:
        if isinstance(value, torch.Tensor):
            if value.dtype in (torch.bfloat16, torch.float8_e4m3fn):
                value = value.float()
            on_host = value.device.type == "cpu"
            arr = value.detach().cpu().numpy()
            if on_host:
                arr = arr.copy()      # never hand out live storage (variables, result buffers)
            return arr[()] if arr.ndim == 0 else arr
        return value

    def run(self, fetches, feed_dict: Optional[Dict[Any, Any]] = None,
            options: Optional[RunOptions] = None,
            run_metadata: Optional[RunMetadata] = None):
        if self._closed:
            raise SessionClosedError("Attempted to use a closed Session.")
        feed_dict = feed_dict or {}
        # A training loop calls run() with the same fetch list and the same placeholders every
        # step: what they resolve to (graph elements, feed order, plan, result structure) is
        # remembered under the identities of the objects passed. The entry keeps those objects
        # alive, so an identity cannot come back meaning something else.
        ck = None
        tf_ = type(fetches)
        if tf_ is list or tf_ is tuple:
            ck = (tf_, tuple(map(id, fetches)), tuple(map(id, feed_dict)))
        ent = self._call_cache.get(ck) if ck is not None else None
        if ent is not None and ent[0] == (self.graph.version if not self.graph.finalized else -1):
            _, _keep, feed_order, plan, structure = ent
            fed_pairs = [(t, feed_dict[k]) for k, t in feed_order]
        else:
            fed_pairs = []
            keys = {}
            for k, v in feed_dict.items():
                t = self.graph.as_graph_element(k, allow_operation=False)
                fed_pairs.append((t, v))
                keys[t] = k
            fed_pairs.sort(key=lambda p: (p[0].op.id, p[0].value_index))
            flat, structure = self._flatten(fetches)
            version = self.graph.version if not self.graph.finalized else -1
            key = (tuple(id(f) for f in flat), tuple(id(t) for t, _ in fed_pairs), version)
            with self._lock:
                plan = self._plans.get(key)
                if plan is None:
                    plan = self._build_plan(flat, [t for t, _ in fed_pairs])
                    self._plans[key] = plan
                if ck is not None and all(not isinstance(f, (list, tuple, dict)) and f is not None
                                          for f in fetches):
                    if len(self._call_cache) > 256:
                        self._call_cache.clear()
                    self._call_cache[ck] = (version, (list(fetches), list(feed_dict)),
                                            [(keys[t], t) for t, _ in fed_pairs], plan, structure)
        self.stats["runs"] += 1
        trace = None
        if options is not None and options.trace_level:
            from ..utils.tracing import StepTracer
            trace = StepTracer(self.device)      # HIP events per step: no sync between steps

        self._run_lock.acquire()
        token = None
        try:
            token = self._begin_step(plan)
            use_graph = (self.device.type == "cuda" and self.config.hip_graphs
                         and plan.capturable and trace is None and plan.runs >= 1)
            if use_graph:
                slot_vals = self._run_captured(plan, [self._prep_feed(t, v)
                                                      for t, v in fed_pairs])
            else:
                feeds = [self._to_device(t, v) for t, v in fed_pairs]
                vals: List[Any] = [None] * plan.num_slots
                for (_, sl), v in zip(plan.feed_slots, feeds):
                    vals[sl] = v
                if trace is not None:
                    self._ctx.tracer = trace      # nodes with inner phases record them too
                    try:
                        with trace.collectives(getattr(self._runtime, "comm", None)):
                            self._run_steps(plan.steps, vals, trace)
                    finally:
                        self._ctx.tracer = None
                else:
                    self._run_steps(plan.steps, vals, trace)
                self.stats["eager_runs"] += 1
                slot_vals = {sl: vals[sl] for sl in plan.fetch_slots if sl is not None}
                slot_vals.update((sl, vals[sl]) for sl in plan.host_input_slots)
                if self.device.type == "cuda" and trace is None:
                    self._fetch_small(slot_vals)
            plan.runs += 1
        finally:
            try:
                self._end_step(plan, token)
            finally:
                self._run_lock.release()

        host_out: Dict[int, Any] = {}
        if plan.host_steps:
            from .deferred import DeferredValue
            if (any(isinstance(slot_vals[sl], DeferredValue) for sl in plan.host_input_slots)
                    and trace is None
                    and all(st.node.type in _DEFERRABLE_HOST_OPS for st in plan.host_steps)):
                lazy = _DeferredHost(self, plan, slot_vals)
                for st in plan.host_steps:
                    for sl in st.out_slots:
                        host_out[sl] = DeferredValue(functools.partial(lazy.get, sl))
            else:
                host_out = self._run_host_phase(plan, slot_vals, trace)
        if trace is not None:
            stats = trace.finish()               # the one synchronisation of a traced run
            if run_metadata is not None:
                run_metadata.step_stats.extend(stats)
        results = [None if sl is None else
                   (host_out[sl] if sl in host_out else self._to_host(slot_vals[sl]))
                   for sl in plan.fetch_slots]
        return self._unflatten(structure, results)

    def _run_host_phase(self, plan: ExecutionPlan, slot_vals: Dict[int, Any],
                        trace) -> Dict[int, Any]:
        from .deferred import resolve
        hv: List[Any] = [None] * plan.num_slots
        for sl in plan.host_input_slots:
            hv[sl] = self._to_host(resolve(slot_vals[sl]))
        self._run_steps(plan.host_steps, hv, trace)
        return {sl: hv[sl] for st in plan.host_steps for sl in st.out_slots}

    def _fetch_ahead_addends(self, sgd_node, parts: List[torch.Tensor],
                             fetched: List[Any]) -> Optional[List[float]]:
        """Addend per packed part when the gather may run ahead of the optimizer launch
        ``sgd_node``: the increment for a part that IS one of the counters it bumps, 0 for parts
        it does not touch; ``None`` when any fetched value (packed or not) overlaps a tensor the
        launch writes, or a variable is not local."""
        def local(var_op):
            fast = getattr(self._store_for(var_op), "local_tensor", None)
            return fast(var_op.name) if fast is not None else None

        def span(t):
            return t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()

        written, counters = [], {}
        state = [s for slots in sgd_node.attrs.get("slot_ops", []) for s in slots] + \
            list(sgd_node.attrs.get("power_ops", []))        # (rules with per-variable state)
        owner_sharded = getattr(self._runtime, "_slot_info", {})
        for var_op in list(sgd_node.attrs["var_ops"]) + state:
            if var_op.name in owner_sharded:
                continue          # state kept as owner shards: nothing fetched can alias it
            t = local(var_op)
            if t is None or not t.is_cuda:
                return None
            written.append(span(t))
        bumps = [(v, 1.0, d) for v, d in sgd_node.attrs.get("counter_increments", [])] + \
            list(sgd_node.attrs.get("scalar_updates", []))
        for var_op, factor, delta in bumps:
            t = local(var_op)
            if t is None or not t.is_cuda or t.numel() != 1 or t.dtype != torch.float32:
                return None
            if factor == 1.0:                    # a counter: its new value is old + increment
                counters[t.data_ptr()] = float(delta)
            written.append(span(t))
        packed_ids = {id(p) for p in parts}
        for v in fetched:                        # values that bypass the gather must be untouched
            if isinstance(v, torch.Tensor) and v.is_cuda and id(v) not in packed_ids:
                lo, hi = span(v)
                if any(lo < b and a < hi for a, b in written):
                    return None
        adds = []
        for p in parts:
            if p.numel() == 1 and p.data_ptr() in counters:
                adds.append(counters[p.data_ptr()])
                continue
            lo, hi = span(p)
            if any(lo < b and a < hi for a, b in written):
                return None
            adds.append(0.0)
        return adds

    def _fetch_small(self, slot_vals: Dict[int, Any]) -> None:
        """Eager steps: the small fp32 results among ``slot_vals`` (loss, step counter,
        summaries) reach the host through ONE gather launch into pinned memory whose completion
        word is polled, instead of one synchronising device-to-host copy per value; they are
        replaced by numpy values in place."""
        small = [(sl, v) for sl, v in slot_vals.items()
                 if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32
                 and 0 < v.numel() <= 64 and v.is_contiguous()][:16]
        if len(small) < 2 or sum(v.numel() for _, v in small) >= _PACK_WORDS:
            return
        from ..ops import hip
        with self._fetch_lock:               # one staging buffer per session
            self._fetch_small_locked(small, slot_vals, hip)

    def _fetch_small_locked(self, small, slot_vals, hip) -> None:
        if self._eager_buf is None:
            self._eager_buf = torch.zeros(_PACK_WORDS, dtype=torch.float32).pin_memory()
            self._eager_seq = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._eager_np = (self._eager_buf.numpy(), self._eager_seq.numpy())
        buf_np, seq_np = self._eager_np
        self._eager_count = self._eager_count % 1000000 + 1
        seq_np[0] = want = self._eager_count
        offsets = hip.pack_f32([v for _, v in small], self._eager_buf,
                               signal=(self._eager_seq, _PACK_WORDS - 1))
        deadline = time.perf_counter() + (0.0 if _NO_SPIN_FETCH else _SPIN_SECS)
        while buf_np[_PACK_WORDS - 1] != want:
            if time.perf_counter() > deadline:
                torch.cuda.current_stream(self.device).synchronize()
                break
        for (sl, v), off in zip(small, offsets):
            n = v.numel()
            slot_vals[sl] = buf_np[off:off + n].reshape(tuple(v.shape)).copy() if v.dim() \
                else buf_np[off].copy()

    # -- step brackets for remote stores ---------------------------------------------------
    def _begin_step(self, plan: ExecutionPlan):
        if self._runtime is None:
            return None
        return self._runtime.begin_step(plan)

    def _end_step(self, plan: ExecutionPlan, token) -> None:
        if self._runtime is not None:
            self._runtime.end_step(plan, token)

    # -- HIP-graph path --------------------------------------------------------------------
    def _run_captured(self, plan: ExecutionPlan, feeds: List[Any]):
        # every input already in GPU memory (or small enough to be read over PCIe from pinned
        # staging): the step finds them through the ring's address table -- no runtime copy in
        # front of the launch; large host inputs keep the DMA path
        use_table = (0 < len(feeds) <= 8 and not os.environ.get("DTF_NO_FEED_TABLE")
                     and all((f.numel() > 0) if isinstance(f, torch.Tensor)
                             else (0 < f.nbytes <= _TABLE_HOST_BYTES) for f in feeds))
        sig = tuple((tuple(f.shape), f.dtype, isinstance(f, torch.Tensor)) for f in feeds) \
            + (use_table,)
        cap = plan.graphs.get(sig)
        generation = self._local_store.generation
        if cap is not None and cap.generation != generation:
            self._retire(cap)           # a variable moved to new storage since the capture
            cap = None
        if cap is None:
            dev_feeds = [f if isinstance(f, torch.Tensor) else
                         torch.from_numpy(np.ascontiguousarray(f).reshape(f.shape)).to(self.device)
                         for f in feeds]
            cap = self._capture(plan, dev_feeds, use_table)
            plan.graphs[sig] = cap
            self.stats["captures"] += 1
        ring = cap.ring
        n = ring.launched
        slot = n % ring.SLOTS
        prev = ring.pending[slot]
        if prev is not None:
            prev.resolve()                   # replay n - SLOTS is done with this slot: keep its results
            ring.pending[slot] = None
        if cap.use_table:
            row = ring.table_np[slot]
            for i, f in enumerate(feeds):
                if isinstance(f, torch.Tensor):
                    row[i] = f.data_ptr()
                    continue
                pins = cap.pinned[i]
                if pins is None:
                    pins = cap.pinned[i] = [None] * ring.SLOTS
                pin = pins[slot]
                if pin is None:
                    static = cap.static_feeds[i]
                    t = torch.empty(static.shape, dtype=static.dtype, pin_memory=True)
                    pin = pins[slot] = (t, t.numpy(), t.data_ptr())
                np.copyto(pin[1], f)
                row[i] = pin[2]
            ring.live[slot] = feeds          # the sources stay alive until the slot is reused
        elif feeds:
            if cap.h2d_done is not None:
                cap.h2d_done.synchronize()   # last step's copies have left the pinned buffers
            staged = False
            for i, (static, f) in enumerate(zip(cap.static_feeds, feeds)):
                if isinstance(f, torch.Tensor):
                    static.copy_(f, non_blocking=True)
                    continue
                pin = cap.pinned[i]
                if pin is None:
                    pin = cap.pinned[i] = torch.empty(static.shape, dtype=static.dtype,
                                                      pin_memory=True)
                np.copyto(pin.numpy(), f)
                static.copy_(pin, non_blocking=True)
                staged = True
            if staged:
                if cap.h2d_done is None:
                    cap.h2d_done = torch.cuda.Event()
                cap.h2d_done.record()
        if cap.direct is not None:
            # the whole step is one prepared call into the kernel library (its launches cost
            # less host time issued directly than as a graph of so few nodes): the batch's
            # addresses and the replay number go into the argument block, so the kernels start
            # without a round trip to the ring's table
            fn, ref, block, xi, li = cap.direct
            block.x, block.labels = int(row[xi]), int(row[li])
            block.host_seq = n & 0x7fffffff
            rc = fn(ref, torch._C._cuda_getCurrentRawStream(self.device.index))
            if rc != 0:
                raise RuntimeError("captured step: direct launch failed with code %d" % rc)
        elif cap.calls is not None:
            # a handful of launches: cheaper issued one by one than as a graph
            s = torch._C._cuda_getCurrentRawStream(self.device.index)
            for fn, args in cap.calls:
                rc = fn(*args, s)
                if rc != 0:
                    raise RuntimeError("captured step: direct launch failed with code %d" % rc)
            self.stats["direct_steps"] = self.stats.get("direct_steps", 0) + 1
        else:
            cap.graph.replay()
            if cap.watch is not None:
                # the graph holds captured collectives: their completion is bounded by the
                # communicator's watchdog through an event behind the replay
                ev = cap.watch_events[slot]
                if ev is None:
                    ev = cap.watch_events[slot] = torch.cuda.Event()
                ev.record()
                cap.watch(ev, "captured step %d" % n)
        ring.launched = n + 1
        self.stats["graph_replays"] += 1
        pend = _PendingStep(ring, n, slot, cap.host_words)
        ring.pending[slot] = pend
        out: Dict[int, Any] = {}
        if cap.layout:
            if self.config.deferred_fetches:
                from .deferred import DeferredValue
                for slot_idx, off, shape, cnt in cap.layout:
                    out[slot_idx] = DeferredValue(functools.partial(pend.value, off, cnt, shape))
            else:
                for slot_idx, off, shape, cnt in cap.layout:
                    out[slot_idx] = pend.value(off, cnt, shape)
        for sl in list(plan.fetch_slots) + plan.host_input_slots:
            if sl is not None and sl not in out:
                out[sl] = cap.slot_values[sl]
        return out

    def _direct_call_list(self, graph, calls, capture_stream):
        """The captured step as a list of library calls, or None when it has to stay a graph.

        ``calls`` is everything the capturing thread asked of the kernel library; the launchers
        among them are the calls whose last argument was the capture stream. The list may stand
        for the graph only if it accounts for ALL of it: the list is captured a second time into
        a scratch graph, and both graphs must consist of kernel nodes only and of equally many
        (a torch kernel, a copy or a memset in the step -- none of which the recorder sees --
        leaves the counts different)."""
        launchers = [(fn, args[:-1]) for fn, args in calls
                     if args and capture_stream and args[-1] == capture_stream]
        if not launchers or len(launchers) > _DIRECT_MAX_CALLS:
            return None
        main = _graph_node_types(graph)
        if not main or any(t != _HIP_KERNEL_NODE for t in main):
            return None
        try:
            scratch = torch.cuda.CUDAGraph(keep_graph=True)
            with _gc_paused(), torch.cuda.graph(scratch, capture_error_mode="thread_local"):
                s = torch._C._cuda_getCurrentRawStream(self.device.index)
                for fn, args in launchers:
                    if fn(*args, s) != 0:
                        return None
            again = _graph_node_types(scratch)
        except (RuntimeError, TypeError):
            return None
        if again is None or len(again) != len(main) or any(t != _HIP_KERNEL_NODE for t in again):
            return None
        return launchers

    def _retire(self, cap: _CapturedStep) -> None:
        """Take every outstanding result out of a captured step's ring (before the step is
        dropped or the session closes)."""
        if cap.ring is None:
            return
        for i, p in enumerate(cap.ring.pending):
            if p is not None:
                p.resolve()
                cap.ring.pending[i] = None

    def _ring_reads_suffice(self, plan: ExecutionPlan, feed_of: Dict[int, torch.Tensor]) -> bool:
        """Whether every step that reads a fed tensor can take it straight from where it lies,
        through the step ring's address table, so that the feed gather can be dropped: the
        whole-step node, or the small fp32 products (``FusedDense`` / ``MatMul`` with the fed
        tensor as their first operand) and the classifier head (the fed labels) of the chain of
        general kernels. The first of them in plan order must be one that can publish the
        replay number (the head kernel cannot)."""
        from ..ops import hip
        first_type = None
        for st in plan.steps + plan.host_steps:
            hits = [(i, sl) for i, sl in enumerate(st.in_slots) if sl in feed_of]
            if not hits:
                continue
            t = st.node.type
            if first_type is None:
                first_type = t
            if t in _RING_READERS:
                # (shapes beyond the small-step kernels run general kernels on the fed tensors
                # themselves: those need the gather)
                dims = {}
                for i, sl in hits:
                    dims[i] = tuple(feed_of[sl].shape)
                w2_shape = st.node.attrs["var_ops"][2].outputs[0].shape
                xs = dims.get(0) or dims.get(1)
                if xs is None or len(xs) != 2 or w2_shape is None or None in tuple(w2_shape) \
                        or (0 in dims and not hip.mlp_step_supported(
                            int(xs[0]), int(dims[0][1]), int(w2_shape[0]), int(w2_shape[1]))):
                    return False
                continue
            if len(hits) != 1 or st.host:
                return False
            i, sl = hits[0]
            f = feed_of[sl]
            if f.dtype != torch.float32 or f.dim() != 2 or not f.is_contiguous():
                return False
            shapes = [getattr(inp, "shape", None) for inp in st.node.inputs]
            if t in ("FusedDense", "MatMul") and i == 0:
                ta = bool(st.node.attrs.get("transpose_a")) if t == "MatMul" else False
                tb = bool(st.node.attrs.get("transpose_b")) if t == "MatMul" else False
                bs = shapes[1]
                if bs is None or len(bs) != 2:
                    return False
                m, k = (f.shape[1], f.shape[0]) if ta else f.shape
                n, kb = (bs[0], bs[1]) if tb else (bs[1], bs[0])
                if n is None or kb not in (None, k) or \
                        hip.lib().dtf_gemm_f32_tile(int(m), int(n), int(k)) != 16:
                    return False
            elif t == "MSELossStep" and i == 1:
                rows, cols = (f.shape if f.dim() == 2 else (f.numel(), 1))
                if not hip.mse_step_supported(int(rows), int(cols),
                                              bool(st.node.attrs.get("emit_bias_grad"))):
                    return False
            elif t == "ClassifierHeadTrain" and i == 3:
                a_shape = shapes[0]
                if a_shape is None or len(a_shape) != 2 or a_shape[1] is None or \
                        not hip.mlp_head_supported(f.shape[0], int(a_shape[1]), f.shape[1]):
                    return False
            else:
                return False
        return first_type in _RING_READERS or first_type in ("FusedDense", "MatMul")

    def _capture(self, plan: ExecutionPlan, feeds: List[torch.Tensor],
                 use_table: bool = False) -> _CapturedStep:
        from ..ops import hip
        cap = _CapturedStep()
        cap.static_feeds = [torch.empty_like(f) for f in feeds]
        cap.pinned = [None] * len(feeds)
        cap.use_table = use_table
        for static, f in zip(cap.static_feeds, feeds):
            static.copy_(f)
        ring = cap.ring = _StepRing(self.device)
        comm_watch = getattr(getattr(self._runtime, "comm", None), "watch", None)
        if comm_watch is not None and self._capture_collectives():
            cap.watch, cap.watch_events = comm_watch, [None] * ring.SLOTS
        want = [sl for sl in plan.fetch_slots if sl is not None] + plan.host_input_slots
        # feeds that only ring-aware nodes read are not gathered at all: those nodes take the
        # batch straight from where it lies, through the ring's table
        feed_slots = [sl for _, sl in plan.feed_slots]
        direct = bool(use_table and feeds and not os.environ.get("DTF_NO_RING_READS")
                      and not any(sl in want for sl in feed_slots)
                      and self._ring_reads_suffice(plan, dict(zip(feed_slots, feeds))))
        ctx = self._ctx
        torch.cuda.synchronize(self.device)
        try:
            graph = torch.cuda.CUDAGraph(keep_graph=True)     # its nodes are counted below
        except TypeError:
            graph = torch.cuda.CUDAGraph()
        vals: List[Any] = [None] * plan.num_slots
        packed: List[Tuple[int, int, tuple, torch.Tensor]] = []
        limit = ring.words - 1
        recorder = None
        capture_stream = None
        # (thread-local capture mode: a service thread's host allocation or copy during this
        # capture must not invalidate it)
        try:
            ctx.ring = ring.args
            ctx.ring_feeds = {t.data_ptr(): i for i, t in enumerate(cap.static_feeds)} \
                if direct else {}
            if direct:
                self.stats["ring_read_captures"] = self.stats.get("ring_read_captures", 0) + 1
            # (no cyclic garbage collection while the stream is capturing: a collector run that
            # finalises an old graph, event or device allocation inside the capture aborts the
            # process)
            with _gc_paused(), torch.cuda.graph(graph, capture_error_mode="thread_local"), \
                    hip.record_calls() as recorder:
                capture_stream = torch._C._cuda_getCurrentRawStream(self.device.index)
                if use_table and not direct:
                    hip.gather_feeds_ring(ring.args, cap.static_feeds)
                for (_, sl), v in zip(plan.feed_slots, cap.static_feeds):
                    vals[sl] = v
                steps = plan.steps
                # trailing variable reads return the live storage; they may run ahead of a step
                # that ends the plan by updating those variables
                k = max((i for i, st in enumerate(steps) if st.node.type != "VariableV2"),
                        default=-1)
                reads = steps[k + 1:]

                def live_reads() -> bool:
                    for st in reads:
                        fast = getattr(self._store_for(st.node), "local_tensor", None)
                        live = fast(st.node.name) if fast is not None else None
                        out_t = vals[st.out_slots[0]] if st.out_slots else None
                        if live is None or not isinstance(out_t, torch.Tensor) \
                                or out_t.data_ptr() != live.data_ptr():
                            return False
                    return True

                done = False
                if k >= 0 and steps[k].node.type in _RING_READERS and not _NO_FETCH_AHEAD:
                    # a whole-step node ends the plan: it gathers the small results itself,
                    # in the workgroup that finishes loss, accuracy and the step counters
                    self._run_steps(steps[:k], vals)
                    self._run_steps(reads, vals)
                    if live_reads():
                        req = ctx.pack_request = _PackRequest(want, vals, steps[k], limit)
                        try:
                            self._run_steps([steps[k]], vals)
                        finally:
                            ctx.pack_request = None
                        if req.claimed:
                            packed = req.packed
                            call = ctx.direct_call
                            if (call is not None and direct and not _NO_DIRECT_STEP and all(
                                    st.node.type == "VariableV2" for st in steps[:k])):
                                cap.direct = call
                                call[2].x_index = call[2].labels_index = -1
                        else:
                            packed = _packable([(sl, vals[sl]) for sl in dict.fromkeys(want)],
                                               limit)
                            hip.pack_f32_ring([p[3] for p in packed], ring.args)
                    else:
                        self._run_steps([steps[k]], vals)
                        self._run_steps(reads, vals)
                        packed = _packable([(sl, vals[sl]) for sl in dict.fromkeys(want)], limit)
                        hip.pack_f32_ring([p[3] for p in packed], ring.args)
                    done = True
                elif k >= 0 and steps[k].node.type in ("MultiApplySGD", "MultiApplyOptimizer") \
                        and not _NO_FETCH_AHEAD:
                    # A step that ends in the fused optimizer launch: nothing that is fetched
                    # depends on it except the step counters it bumps, and their new value is
                    # known beforehand (old + increment). The gather of the small results -- with
                    # that increment added -- then runs BEFORE the optimizer launch: the host has
                    # its results while the GPU still applies the update. The gather can move
                    # further up: behind the last step that produces something fetched, provided
                    # nothing between there and the optimizer writes in place.
                    from_reads = {sl for st in reads for sl in st.out_slots}
                    made_at = {sl: i for i, st in enumerate(steps[:k]) for sl in st.out_slots}
                    j = 1 + max((made_at.get(sl, -1) for sl in want if sl not in from_reads),
                                default=-1)
                    if any(st.node.type in _MUTATING_OPS for st in steps[j:k]):
                        j = k
                    rode = False
                    if j >= 1 and steps[j - 1].node.type in _PACK_RIDERS:
                        # the node that produces the last fetched value can gather the step's
                        # small results itself, on its last launch: no gather launch at all
                        self._run_steps(steps[:j - 1], vals)
                        self._run_steps(reads, vals)
                        if live_reads():
                            req = ctx.pack_request = _PackRequest(want, vals, steps[j - 1], limit)
                            req.adds_fn = functools.partial(self._fetch_ahead_addends,
                                                            steps[k].node)
                            try:
                                self._run_steps([steps[j - 1]], vals)
                            finally:
                                ctx.pack_request = None
                            if req.claimed:
                                packed = req.packed
                                self._run_steps(steps[j:k + 1], vals)
                                done = rode = True
                        else:
                            self._run_steps([steps[j - 1]], vals)
                    else:
                        self._run_steps(steps[:j], vals)
                        self._run_steps(reads, vals)
                    adds = None
                    if not rode and live_reads():
                        packed = _packable([(sl, vals[sl]) for sl in dict.fromkeys(want)], limit)
                        adds = self._fetch_ahead_addends(steps[k].node, [p[3] for p in packed],
                                                         [vals[sl] for sl in dict.fromkeys(want)])
                    if rode:
                        pass
                    elif adds is not None:
                        hip.pack_f32_ring([p[3] for p in packed], ring.args, adds=adds)
                        self._run_steps(steps[j:k + 1], vals)
                        done = True
                    else:             # something fetched is touched by the update: usual order
                        self._run_steps(steps[j:k + 1], vals)
                        self._run_steps(reads, vals)
                else:
                    self._run_steps(steps, vals)
                if not done:
                    packed = _packable([(sl, vals[sl]) for sl in dict.fromkeys(want)], limit)
                    hip.pack_f32_ring([p[3] for p in packed], ring.args)
        finally:
            ctx.ring = None
            ctx.ring_feeds = {}
            ctx.pack_request = None
            ctx.direct_call = None
        cap.host_words = sum(p[3].numel() for p in packed)
        cap.layout = [(sl, off, shape, int(v.numel())) for sl, off, shape, v in packed]
        if cap.direct is None and recorder is not None and not _NO_DIRECT_STEP:
            cap.calls = self._direct_call_list(graph, recorder.calls, capture_stream)
        cap.graph = graph
        cap.slot_values = vals
        cap.generation = self._local_store.generation
        return cap

    # -- conveniences ----------------------------------------------------------------------
    def make_callable(self, fetches, feed_list: Optional[Sequence[Any]] = None,
                      accept_options: bool = False):
        """``f = sess.make_callable(fetches, feed_list)``; ``f(*feed_values)`` is
        ``sess.run(fetches, dict(zip(feed_list, feed_values)))`` with the lookups of ``run`` (graph
        elements, feed order, plan) done once. With ``accept_options`` the callable also takes
        ``options=`` and ``run_metadata=``."""
        feed_list = list(feed_list or [])
        fetches_t = tuple(fetches) if isinstance(fetches, (list, tuple)) else fetches
        keys = tuple(feed_list)
        as_list = isinstance(fetches, list)

        def call(*values, options=None, run_metadata=None):
            if len(values) != len(keys):
                raise TypeError("expected %d feed values, got %d" % (len(keys), len(values)))
            if not accept_options and (options is not None or run_metadata is not None):
                raise TypeError("this callable was made without accept_options")
            out = self.run(fetches_t, dict(zip(keys, values)), options=options,
                           run_metadata=run_metadata)
            return list(out) if as_list else out
        return call

    def list_devices(self) -> List[str]:
        spec = DeviceSpec(job="localhost", replica=0, task=0,
                          device_type="GPU" if self.device.type == "cuda" else "CPU",
                          device_index=self.device.index or 0)
        return [spec.to_string()]


class InteractiveSession(Session):
    def __init__(self, target: str = "", graph: Optional[Graph] = None,
                 config: Optional[ConfigProto] = None):
        super().__init__(target, graph, config)
        self.__enter__()
