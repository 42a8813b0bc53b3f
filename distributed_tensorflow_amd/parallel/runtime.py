"""Cluster runtimes: how a session's variables and gradients move between tasks.

The reference has exactly one distribution strategy -- asynchronous data-parallel training
with between-graph replication and one parameter server (``tf_distributed.py:33-36,75-76``) --
implemented by TensorFlow's placer, graph partitioner and gRPC rendezvous. A *runtime* here is
the equivalent of those pieces for one task: it decides which store owns each variable
(from the variable's ``/job:ps/task:K`` placement), brackets every ``Session.run`` with the
transfers that placement implies, and synchronises gradients when the strategy is
synchronous.

* ``PsRuntime``         -- ``strategy="ps_async"``: variables live in ps tasks reached over TCP;
  per step one PULL and one APPLY per ps task; updates are asynchronous and lock-free exactly
  as in the reference. Runs without a GPU.
* ``CollectiveRuntime`` -- ``strategy="allreduce_sync"`` / ``"reduce_scatter_sync"``: one process
  per GPU, no ps processes. Variable pull/push become collectives over xGMI (RCCL):
  ``allreduce_sync`` keeps every variable replicated and all-reduces the gradients in flat
  buckets -- one collective after the backward pass when they fit a bucket (32 MiB; the
  reference MLP has 318 KB), otherwise one asynchronous collective per bucket, issued by the
  session as soon as the backward pass has produced the bucket's members;
  ``reduce_scatter_sync`` keeps the parameter-server idea of a single owner per parameter shard
  -- gradients are reduce-scattered to the owners, each owner applies the optimizer's rule
  (plain gradient descent, momentum, Adagrad, RMSProp, Adadelta, Adam) to its shard, and the
  updated shards are all-gathered (= broadcast from every owner). The rule's per-element state
  (slot variables) exists ONLY on the owner: ``1 / world`` of it per rank.
  Semantic difference from the reference, by construction: updates are bulk-synchronous.
  Gradients are *summed* across workers (each worker's batch-mean gradient applied with the
  same learning rate, as the asynchronous ps would apply them one by one) unless
  ``grad_aggregation="mean"``; ``global_step`` advances by one per synchronous step.
* ``LocalRuntime``      -- a server with no peers (``Server.create_local_server``).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch

from ..client.variable_store import LocalVariableStore
from ..framework.device import DeviceSpec
from .comm import Communicator, SingleComm


class LocalRuntime:
    strategy = "local"
    capturable = True

    def __init__(self, device: torch.device):
        self.device = torch.device(device)
        self.local_store = LocalVariableStore(self.device)
        self.is_chief = True
        self.comm: Communicator = SingleComm(self.device)

    def store_for_device(self, device: str):
        return self.local_store

    def begin_step(self, plan) -> Any:
        return None

    def end_step(self, plan, token) -> None:
        pass

    def reduce_gradients(self, grads: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        return list(grads)

    def after_init(self, session, var_names: Sequence[str]) -> None:
        pass

    def wait_ready(self, timeout_s: Optional[float] = None) -> None:
        pass

    def all_stores(self) -> List[Any]:
        return [self.local_store]

    def shutdown(self) -> None:
        pass


class PsRuntime(LocalRuntime):
    """Worker-side runtime of the asynchronous parameter-server strategy."""

    strategy = "ps_async"
    capturable = False      # every step talks to the ps over TCP

    def __init__(self, cluster, job_name: str, task_index: int, device: torch.device,
                 connect_timeout_s: float = 60.0):
        super().__init__(device)
        from .ps_transport import RemoteVariableStore
        self.cluster = cluster
        self.job_name, self.task_index = job_name, task_index
        self.is_chief = job_name == "worker" and task_index == 0
        self._remote: Dict[int, RemoteVariableStore] = {}
        self._timeout = connect_timeout_s
        self._ps_job = "ps"

    def _ps_store(self, task: int):
        if task not in self._remote:
            from .ps_transport import RemoteVariableStore
            addr = self.cluster.task_address(self._ps_job, task)
            self._remote[task] = RemoteVariableStore(addr, self.device, self._timeout)
        return self._remote[task]

    def store_for_device(self, device: str):
        spec = DeviceSpec.from_string(device)
        if spec.job == self._ps_job and self._ps_job in self.cluster.jobs:
            return self._ps_store(spec.task or 0)
        return self.local_store

    def begin_step(self, plan) -> Any:
        by_store: Dict[int, List[str]] = {}
        for var_op in plan.var_ops:
            spec = DeviceSpec.from_string(var_op.device)
            if spec.job == self._ps_job and self._ps_job in self.cluster.jobs:
                by_store.setdefault(spec.task or 0, []).append(var_op.name)
        for task, names in by_store.items():
            self._ps_store(task).begin_step(names)       # one PULL per ps task
        return list(self._remote.values())

    def end_step(self, plan, token) -> None:
        for store in (token or []):
            store.end_step()                              # one APPLY per ps task

    def all_stores(self) -> List[Any]:
        tasks = self.cluster.task_indices(self._ps_job) if self._ps_job in self.cluster.jobs \
            else []
        return [self._ps_store(t) for t in tasks] + [self.local_store]

    def shutdown(self) -> None:
        for store in self._remote.values():
            store.close()

    def report_done(self) -> None:
        """This worker is finished: every ps task exits ``join()`` once all workers said so."""
        from .ps_transport import UnavailableError
        for t in self.cluster.task_indices(self._ps_job):
            try:
                self._ps_store(t).report_done(self.task_index)
            except UnavailableError:
                pass

    def shutdown_parameter_servers(self) -> None:
        """Ask every ps task to leave ``join()`` (the reference has no such path: its ps
        blocks forever and has to be killed, ``tf_distributed.py:31``)."""
        for t in self.cluster.task_indices(self._ps_job):
            self._ps_store(t).shutdown_server()


class _OwnerShardedStore(LocalVariableStore):
    """The local store of a collective runtime. Optimizer slots that the runtime keeps as owner
    shards (``CollectiveRuntime._slot_info``) have no replica here: they count as initialised,
    reading one is the runtime's all-gather (a COLLECTIVE: every rank reads, as every rank saves),
    assigning one keeps this rank's part, and no kernel gets in-place access to them."""

    def __init__(self, device: torch.device, runtime: "CollectiveRuntime"):
        super().__init__(device)
        self._rt = runtime

    def is_initialized(self, name: str) -> bool:
        return name in self._values or name in self._rt._slot_info

    def uninitialized(self, names):
        slots = self._rt._slot_info
        return [n for n in names if n not in self._values and n not in slots]

    def names(self) -> List[str]:
        return sorted(set(self._values) | set(self._rt._slot_info))

    def read(self, name: str) -> torch.Tensor:
        if name in self._rt._slot_info:
            return self._rt.read_variable_full(name)
        return super().read(name)

    def local_tensor(self, name: str) -> Optional[torch.Tensor]:
        if name in self._rt._slot_info:
            return None
        return super().local_tensor(name)

    def assign(self, name: str, value: torch.Tensor) -> torch.Tensor:
        if name in self._rt._slot_info:
            return self._rt.assign_variable_full(name, value)
        return super().assign(name, value)

    def assign_add(self, name: str, delta: torch.Tensor) -> torch.Tensor:
        if name in self._rt._slot_info:
            raise RuntimeError("%s is optimizer state kept as owner shards: it is updated by the "
                               "sharded optimizer step only" % name)
        return super().assign_add(name, delta)


class CollectiveRuntime(LocalRuntime):
    """Synchronous data parallelism over a communicator (RCCL on GPUs, gloo on CPUs)."""

    # collectives are issued eagerly between kernel launches unless the session asks for them
    # inside the captured step (every buffer a collective touches here -- flat arena, gradient
    # bucket, shards -- keeps its address from step to step); a world of one has no
    # collectives to issue and is captured like a local session
    capturable_with_collectives = True
    ARENA_ALIGN = 64            # fp32 elements: variables start on 256-byte boundaries

    @property
    def capturable(self) -> bool:
        return self.comm.world_size == 1

    def __init__(self, comm: Communicator, device: torch.device,
                 strategy: str = "allreduce_sync", grad_aggregation: str = "sum",
                 bucket_bytes: int = 32 << 20):
        super().__init__(device)
        self.local_store = _OwnerShardedStore(self.device, self)
        # optimizer slots kept as owner shards (reduce_scatter_sync): slot variable name ->
        # (slot key, arena offset of its variable, numel, shape); slot key -> this rank's shard
        self._slot_info: Dict[str, Any] = {}
        self._slot_shards: Dict[str, torch.Tensor] = {}
        self._slot_images: Dict[str, torch.Tensor] = {}     # gathered for a save, until next step
        self._segments: Dict[Any, Any] = {}
        self.slot_bytes_total = 0          # bytes of slot state the job holds (all ranks)
        # gradient bucket size for the overlapped all-reduce. xGMI collectives are bound by one
        # link per ring step, so buckets are large (few, long transfers): 32 MiB by default
        self.bucket_bytes = int(bucket_bytes)
        self._bucket_bufs: Dict[int, torch.Tensor] = {}
        self._pending: List[Any] = []      # (work, gradients, flat bucket) issued this step
        self.buckets_issued = 0
        if strategy not in ("allreduce_sync", "reduce_scatter_sync"):
            raise ValueError("unknown collective strategy %r" % strategy)
        if grad_aggregation not in ("sum", "mean"):
            raise ValueError("grad_aggregation must be 'sum' or 'mean'")
        self.comm = comm
        self.strategy = strategy
        self.grad_aggregation = grad_aggregation
        self.is_chief = comm.rank == 0
        self._flat_params: Optional[torch.Tensor] = None
        self._flat_grads: Optional[torch.Tensor] = None
        self._layout: List[Any] = []       # (name, offset, numel, shape)
        self.bytes_reduced = 0
        self.bytes_gathered = 0
        self.sharded_steps = 0             # steps that took the reduce-scatter / owner-update path
        self.counters_applied = False      # the last sharded_apply also bumped the step counters
        self.grads_in_place = 0            # gradients found already in their arena slot
        self.grads_copied = 0              # gradients that had to be copied into it
        self._grad_views: Optional[Dict[str, torch.Tensor]] = None
        self._extent_cache: Dict[Any, Any] = {}     # names of a step -> (arena, gradient span)
        self._g_shard: Optional[torch.Tensor] = None
        self._p_stage: Optional[torch.Tensor] = None
        self.shard_plan = None             # parallel.planner.ShardPlan of the session's graph
        self.var_layouts: Dict[str, Any] = {}   # variables stored as blocks: name -> VarLayout
        self.matmul_sharding: Optional[str] = None

    # -- sharded products (parallel/planner.py) ------------------------------------------
    def set_shard_plan(self, plan) -> None:
        """The session's graph-wide plan of sharded products; ``after_init`` lays the variables
        it marks as stored-sharded out as blocks."""
        self.shard_plan = plan

    def has_sharded_storage(self) -> bool:
        return bool(self.var_layouts) or bool(self._slot_info)

    def slot_bytes_local(self) -> int:
        """Bytes of optimizer slot state this rank holds as owner shards."""
        return sum(t.numel() * t.element_size() for t in self._slot_shards.values())

    def _shard_len(self) -> int:
        return self._flat_params.numel() // self.comm.world_size

    def read_variable_full(self, name: str) -> torch.Tensor:
        """The whole value of ``name``. For a variable stored as blocks, and for an optimizer
        slot kept as owner shards, this is an all-gather: every rank must call it (the Saver
        does, through this method)."""
        info = self._slot_info.get(name)
        if info is not None:
            key, off, numel, shape = info
            image = self._slot_images.get(key)
            if image is None:
                shard = self._slot_shards[key]
                image = torch.empty(shard.numel() * self.comm.world_size, dtype=shard.dtype,
                                    device=self.device)
                w, r, n = self.comm.world_size, self.comm.rank, shard.numel()
                if self.comm.inplace_all_gather:
                    image[r * n:(r + 1) * n].copy_(shard)
                    self.comm.all_gather(image, image[r * n:(r + 1) * n])
                else:
                    self.comm.all_gather(image, shard)
                self.bytes_gathered += image.numel() * 4
                self._slot_images[key] = image
            return image[off:off + numel].view(shape).clone()
        value = self.local_store.read(name)
        lay = self.var_layouts.get(name)
        if lay is None or tuple(value.shape) == tuple(lay.shape):
            return value
        from .sharded_ops import gather_blocks
        return gather_blocks(self.comm, lay, value)

    def assign_variable_full(self, name: str, value: torch.Tensor) -> torch.Tensor:
        """Assign a whole value; a variable stored as blocks keeps this rank's block of it."""
        info = self._slot_info.get(name)
        if info is not None:
            key, off, numel, shape = info
            if tuple(value.shape) != tuple(shape):
                raise ValueError("%s: value of shape %s for a slot of shape %s"
                                 % (name, tuple(value.shape), tuple(shape)))
            n, r = self._shard_len(), self.comm.rank
            lo, hi = max(off, r * n), min(off + numel, (r + 1) * n)
            if lo < hi:
                self._slot_shards[key][lo - r * n:hi - r * n].copy_(
                    value.reshape(-1)[lo - off:hi - off].to(self.device, torch.float32))
            self._slot_images.pop(key, None)
            return value
        lay = self.var_layouts.get(name)
        if lay is not None and tuple(value.shape) == tuple(lay.shape):
            value = lay.slice_of(value, self.comm.rank, self.comm.world_size).contiguous()
        return LocalVariableStore.assign(self.local_store, name, value.to(self.device))

    # -- initial state: chief's values -> everyone -----------------------------------------
    def after_init(self, session, var_names: Sequence[str]) -> None:
        """Collective: the chief's initial (or restored) values reach every rank.

        * replicated fp32 variables: packed into ONE flat arena, one broadcast; afterwards the
          variables are views of the arena (the flat buffer the gradient collectives run on);
        * variables the shard plan stores as blocks: when the chief's value already is a block
          (its initialiser generated only the local window of a seeded random stream) every
          other rank generates its own window -- no transfer at all; otherwise (restored from a
          checkpoint, or a constant initial value) the whole tensor is broadcast once and every
          rank keeps its block;
        * anything else (other dtypes): one broadcast each.
        """
        store = self.local_store
        plan = self.shard_plan
        w, r = self.comm.world_size, self.comm.rank

        # a re-initialisation (recovery) starts from replicated slots again
        self._slot_info, self._slot_shards, self._slot_images = {}, {}, {}
        self._segments = {}
        self.slot_bytes_total = 0
        graph = getattr(session, "graph", None)

        def slot_of(n):
            """(variable, slot key) when ``n`` is per-element optimizer state that this strategy
            keeps on the owner of the variable's shard only."""
            if self.strategy != "reduce_scatter_sync" or graph is None:
                return None
            try:
                attrs = graph.get_operation_by_name(n).attrs
            except (KeyError, ValueError):
                return None
            if "slot_of" not in attrs:
                return None
            return attrs["slot_of"], attrs["slot_key"]

        def describe(n):
            t = LocalVariableStore.read(store, n)
            lay = plan.stored(n) if plan else None
            state = "replicated"
            if lay is not None and w > 1:
                state = "block" if (tuple(t.shape) == tuple(lay.local_shape())
                                    and lay.init is not None) else "whole"
            return (n, tuple(t.shape), str(t.dtype), state)

        present = self.comm.broadcast_object(
            [describe(n) for n in var_names] if self.is_chief else None, src=0)
        # slots of replicated fp32 variables become owner shards (the test is made of graph
        # attributes and the chief's table only, so every rank decides alike)
        table = {n: (shape, dt, state) for n, shape, dt, state in present}
        slots = []
        for n, shape, dt, state in present:
            hit = slot_of(n)
            if hit is None or state != "replicated" or dt != "torch.float32":
                continue
            primary = table.get(hit[0])
            if primary is None or primary[1] != "torch.float32" or primary[2] != "replicated" \
                    or tuple(primary[0]) != tuple(shape):
                continue
            slots.append((n, hit[0], hit[1], shape))
        slot_names = {n for n, _, _, _ in slots}
        plain = [(n, shape, dt) for n, shape, dt, state in present
                 if state == "replicated" and n not in slot_names]
        f32 = [(n, shape) for n, shape, dt in plain if dt == "torch.float32"]
        other = [(n, shape, dt) for n, shape, dt in plain if dt != "torch.float32"]
        # every variable starts on a 256-byte boundary of the arena: the GEMM kernels take their
        # 16-byte vector / buffer-load paths only for aligned operands (a scalar such as
        # global_step in front of a matrix would otherwise push it onto the element-wise path:
        # 1.65 ms instead of 1.06 ms for the 4096^3 fp32 product)
        align = self.ARENA_ALIGN
        offsets, total = [], 0
        for _, shape in f32:
            offsets.append(total)
            numel = int(torch.Size(shape).numel())
            total += ((numel + align - 1) // align) * align
        # every rank's shard starts on a 256-byte boundary as well: the parts of a variable on
        # either side of a shard boundary stay on the kernels' 16-byte paths
        padded = ((total + w * align - 1) // (w * align)) * (w * align) if total else 0
        flat = torch.zeros(padded, dtype=torch.float32, device=self.device)
        self._layout = []
        for (n, shape), off in zip(f32, offsets):
            numel = int(torch.Size(shape).numel())
            if self.is_chief:
                flat[off:off + numel].copy_(store.read(n).reshape(-1))
            self._layout.append((n, off, numel, shape))
        if padded:
            self.comm.broadcast(flat, src=0)
        for n, o, numel, shape in self._layout:
            store.adopt(n, flat[o:o + numel].view(shape))    # variables are views of the arena
        for n, shape, dt in other:
            t = store.read(n).clone() if self.is_chief else \
                torch.empty(shape, dtype=getattr(torch, dt.split(".")[-1]), device=self.device)
            self.comm.broadcast(t, src=0)
            store.assign(n, t)
        for n, shape, dt, state in present:
            if state == "replicated":
                continue
            lay = plan.stored(n)
            if state == "block":
                if not self.is_chief:
                    from .sharded_ops import local_window
                    store.assign(n, local_window(lay, r, w, self.device))
            else:
                whole = store.read(n) if self.is_chief else \
                    torch.empty(shape, dtype=getattr(torch, dt.split(".")[-1]), device=self.device)
                self.comm.broadcast(whole, src=0)
                store.assign(n, lay.slice_of(whole, r, w).clone())
            self.var_layouts[n] = lay
        self._flat_params = flat
        self._flat_grads = torch.zeros_like(flat)
        self._grad_views = None
        self._extent_cache = {}
        if slots:
            self._distribute_slots(store, slots)
        self.comm.barrier()

    def _distribute_slots(self, store, slots) -> None:
        """Optimizer slots -> owner shards. For every slot key (``Adam``, ``Adam_1``, ...) rank q
        receives the chief's values for the elements of ITS parameter shard: one broadcast of a
        shard-sized buffer per rank and key, so no rank ever holds more than a shard of slot
        state beyond what the chief's initialiser produced (and that is dropped here)."""
        w, r, n = self.comm.world_size, self.comm.rank, self._shard_len()
        where = {name: (off, numel) for name, off, numel, _ in self._layout}
        by_key: Dict[str, List[Any]] = {}
        for name, primary, key, shape in slots:
            off, numel = where[primary]
            by_key.setdefault(key, []).append((name, off, numel, shape))
        stage = torch.empty(n, dtype=torch.float32, device=self.device)
        for key in sorted(by_key):
            mine = torch.zeros(n, dtype=torch.float32, device=self.device)
            for q in range(w):
                if self.is_chief:
                    stage.zero_()
                    for name, off, numel, _ in by_key[key]:
                        lo, hi = max(off, q * n), min(off + numel, (q + 1) * n)
                        if lo < hi:
                            full = LocalVariableStore.read(store, name).reshape(-1)
                            stage[lo - q * n:hi - q * n].copy_(full[lo - off:hi - off])
                if w > 1:
                    self.comm.broadcast(stage, src=0)
                if q == r:
                    mine.copy_(stage)
            self._slot_shards[key] = mine
            for name, off, numel, shape in by_key[key]:
                self._slot_info[name] = (key, off, numel, tuple(shape))
                self.slot_bytes_total += numel * 4
        with store._lock:
            for name in self._slot_info:
                store._values.pop(name, None)       # the chief's full copies
            store.generation += 1

    def wait_ready(self, timeout_s: Optional[float] = None) -> None:
        self.comm.barrier()

    def begin_step(self, plan) -> Any:
        self._pending = []          # buckets of a step that was abandoned half-way
        return None

    # -- gradient synchronisation --------------------------------------------------------
    def begin_gradient_bucket(self, bucket_id: int, grads: Sequence[torch.Tensor]) -> None:
        """Start the all-reduce of one gradient bucket while the backward pass continues (the
        session calls this right after the step that produced the bucket's last member).
        ``reduce_gradients`` later hands out the reduced views."""
        grads = list(grads)
        total = sum(g.numel() for g in grads)
        buf = self._bucket_bufs.get(bucket_id)
        if buf is None or buf.numel() != total or buf.dtype != grads[0].dtype:
            buf = torch.empty(total, dtype=grads[0].dtype, device=grads[0].device)
            self._bucket_bufs[bucket_id] = buf
        self._pack(grads, buf)
        work = self.comm.all_reduce(buf, "avg" if self.grad_aggregation == "mean" else "sum",
                                    async_op=True)
        self._pending.append((work, grads, buf))
        self.bytes_reduced += buf.numel() * buf.element_size()
        self.buckets_issued += 1

    @staticmethod
    def _pack(grads: Sequence[torch.Tensor], buf: torch.Tensor) -> None:
        """The members of a bucket, one behind the other, into its flat buffer: one multi-tensor
        launch on a GPU (``dtf_multi_copy``), slice copies on a CPU."""
        if buf.is_cuda and all(g.is_cuda and g.is_contiguous() and g.dtype == buf.dtype
                               for g in grads):
            from ..ops import hip
            hip.pack_into(list(grads), buf)
            return
        off = 0
        for g in grads:
            buf[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()

    def _collect_buckets(self) -> Dict[int, torch.Tensor]:
        """Wait for the buckets issued during this step; reduced view per gradient object."""
        done: Dict[int, torch.Tensor] = {}
        for work, grads, buf in self._pending:
            if work is not None:
                work.wait()
            off = 0
            for g in grads:
                done[id(g)] = buf[off:off + g.numel()].view(g.shape)
                off += g.numel()
        self._pending = []
        return done

    def grad_view(self, name: str) -> Optional[torch.Tensor]:
        """The slot of variable ``name``'s gradient in the flat gradient arena (same offsets as
        the parameter arena), shaped like the variable; ``None`` for a variable outside it."""
        views = self._grad_views
        if views is None:
            if self._flat_grads is None:
                return None
            views = self._grad_views = {
                n: self._flat_grads[off:off + numel].view(shape)
                for n, off, numel, shape in self._layout}
        return views.get(name)

    def _grad_runs(self, names: Sequence[str]) -> List[Any]:
        """The spans ``(start, end)`` of the flat gradient arena that hold the gradients of
        ``names`` and nothing else: variables that follow each other in the arena (only
        alignment padding, which is never written, in between) share a span. A step's trainable
        variables lie first and together, so this is one span -- but the slot of a variable the
        step does NOT cover (another train op's, written by an earlier step) is never part of
        one: reduced in place it would be multiplied by the world size every step."""
        key = tuple(names)
        cache = self._extent_cache
        hit = cache.get(key)
        if hit is None or hit[0] is not self._flat_grads:
            wanted = set(names)
            runs: List[List[int]] = []
            open_run = False
            for n, off, numel, _ in self._layout:        # arena order
                if n in wanted:
                    if open_run:
                        runs[-1][1] = off + numel
                    else:
                        runs.append([off, off + numel])
                        open_run = True
                else:
                    open_run = False
            hit = cache[key] = (self._flat_grads, [tuple(r) for r in runs])
        return hit[1]

    def _place(self, grads: Sequence[torch.Tensor], names: Optional[Sequence[str]]) -> bool:
        """Make sure every gradient sits in its arena slot (a gradient the session already had
        produced there is left alone; anything else is copied in). False when a gradient has no
        slot or the arena does not exist."""
        if names is None or self._flat_grads is None:
            return False
        for n, g in zip(names, grads):
            slot = self.grad_view(n)
            if slot is None or slot.numel() != g.numel() or g.dtype != torch.float32:
                return False
        for n, g in zip(names, grads):
            slot = self.grad_view(n)
            if g.data_ptr() == slot.data_ptr():
                self.grads_in_place += 1
                continue
            if g.is_cuda and g.is_contiguous() and g.numel():
                from ..ops import hip
                hip.copy_2d(g.reshape(1, -1), slot.view(1, -1))
            else:
                slot.copy_(g.reshape(slot.shape))
            self.grads_copied += 1
        return True

    def reduce_gradients(self, grads: Sequence[torch.Tensor],
                         names: Optional[Sequence[str]] = None) -> List[torch.Tensor]:
        """All-reduce a list of gradients; returns reduced views. Gradients whose bucket was
        already issued during the backward pass are only waited for. The rest travel as ONE
        collective over the flat gradient arena when their variables (``names``) all have a slot
        there -- gradients the session had its kernels produce in place need no staging at all --
        and as one packed bucket otherwise."""
        grads = list(grads)
        if self.comm.world_size == 1 or not grads:
            return grads
        op = "avg" if self.grad_aggregation == "mean" else "sum"
        if self._pending:
            done = self._collect_buckets()
            rest = [(i, g) for i, g in enumerate(grads) if id(g) not in done]
            reduced_rest = iter(self.reduce_gradients(
                [g for _, g in rest], [names[i] for i, _ in rest] if names else None)) \
                if rest else iter(())
            return [done[id(g)] if id(g) in done else next(reduced_rest) for g in grads]
        if self._place(grads, names):
            # only the spans of the arena that hold this step's gradients travel: the session
            # manager lays trainable variables out first, so that is one collective, and
            # optimizer state, counters and variables of other train ops are left out
            for start, end in self._grad_runs(names):
                flat = self._flat_grads[start:end]
                self.comm.all_reduce(flat, op)
                self.bytes_reduced += flat.numel() * 4
            return [self.grad_view(n) for n in names]
        total = sum(g.numel() for g in grads)
        buf = self._bucket_bufs.get(-1)
        if buf is None or buf.numel() < total or buf.dtype != grads[0].dtype:
            buf = self._bucket_bufs[-1] = torch.empty(total, dtype=grads[0].dtype,
                                                      device=grads[0].device)
        bucket = buf[:total]
        self._pack(grads, bucket)
        self.comm.all_reduce(bucket, op)
        self.bytes_reduced += bucket.numel() * bucket.element_size()
        out, off = [], 0
        for g in grads:
            out.append(bucket[off:off + g.numel()].view(g.shape))
            off += g.numel()
        return out

    def sharded_sgd_step(self, names: Sequence[str], grads: Sequence[torch.Tensor],
                         lrs: Sequence[float]) -> bool:
        """Plain gradient descent on the owner shards (see ``sharded_apply``); True when every
        variable was updated this way."""
        done = self.sharded_apply("sgd", (), names, grads, lrs=lrs)
        return done is not None and len(done) == len(names)

    def _segments_for(self, rule: str, names: Sequence[str],
                      slot_names: Sequence[Sequence[str]]):
        """This rank's part of every variable of ``names``: views of the parameter shard, of the
        reduce-scattered gradient shard and of the slot shards that cover the same elements
        (a variable crosses at most one shard boundary on each side, so at most one part per
        variable). Cached: every buffer involved keeps its address."""
        key = (rule, tuple(names), tuple(tuple(sn) for sn in slot_names))
        hit = self._segments.get(key)
        if hit is not None and hit[0] is self._flat_params:
            return hit[1]
        n, r = self._shard_len(), self.comm.rank
        where = {name: (off, numel) for name, off, numel, _ in self._layout}
        segs = []
        for i, name in enumerate(names):
            off, numel = where[name]
            lo, hi = max(off, r * n), min(off + numel, (r + 1) * n)
            if lo >= hi:
                continue
            a, b = lo - r * n, hi - r * n
            slots = [self._slot_shards[self._slot_info[sn][0]][a:b] for sn in slot_names[i]]
            segs.append((i, self._flat_params[lo:hi], self._g_shard[a:b], slots))
        self._segments[key] = (self._flat_params, segs)
        return segs

    def sharded_apply(self, rule: str, hyper: Sequence[float], names: Sequence[str],
                      grads: Sequence[torch.Tensor],
                      slot_names: Optional[Sequence[Sequence[str]]] = None,
                      powers: Sequence[torch.Tensor] = (), lr: Optional[torch.Tensor] = None,
                      lrs: Optional[Sequence[float]] = None,
                      counters: Sequence[torch.Tensor] = (),
                      deltas: Sequence[float] = ()) -> Optional[List[int]]:
        """``reduce_scatter_sync``: reduce-scatter the flat gradient arena to the shard owners,
        apply the optimizer's rule on the owned shard only, all-gather the updated parameters --
        the collective image of "variables live on the parameter server, gradients are pushed to
        it, updated values are pulled back" (reference ``tf_distributed.py:34-36,75-76``), for
        every rule of ``optim.hip`` (ZeRO stage 1: the rule's per-element state exists only on
        the owner, ``_slot_shards``). ``rule == "sgd"`` with ``lrs``: one constant rate per
        variable and no state.

        Only the parts of THIS step's variables that fall into the owned shard are touched: a
        variable of the arena the step does not cover (``global_step``, running powers, another
        train op's variables) is left alone whatever its gradient slot holds from earlier steps.
        Scalars that advance with the step (powers, counters) stay replicated; the caller
        advances them on every rank alike.

        ``counters`` / ``deltas`` (plain gradient descent on a GPU): replicated step counters the
        update launch bumps on its way (``counter += delta`` on every rank alike, before the
        all-gather, so the owner's copy and everybody else's agree); ``counters_applied`` tells
        the caller whether that happened.

        Returns the indices of the variables updated here (the caller sends the others down the
        all-reduce path), ``None`` when the strategy or the storage does not allow it."""
        self.counters_applied = False
        if self.strategy != "reduce_scatter_sync" or self._flat_params is None or not grads:
            return None
        slot_names = [()] * len(names) if slot_names is None else list(slot_names)
        take = []
        for i, (n, g) in enumerate(zip(names, grads)):
            slot = self.grad_view(n)
            if slot is None or slot.numel() != g.numel() or g.dtype != torch.float32:
                continue
            if any(sn not in self._slot_info for sn in slot_names[i]):
                continue
            take.append(i)
        if not take:
            return None
        t_names = [names[i] for i in take]
        if not self._place([grads[i] for i in take], t_names):
            return None
        flat_g, flat_p = self._flat_grads, self._flat_params
        w, r = self.comm.world_size, self.comm.rank
        shard = flat_p.numel() // w
        if self._g_shard is None or self._g_shard.numel() != shard:
            self._g_shard = torch.empty(shard, dtype=torch.float32, device=self.device)
            self._segments = {}
        self._slot_images.clear()
        g_shard = self._g_shard
        self.comm.reduce_scatter(g_shard, flat_g,
                                 "avg" if self.grad_aggregation == "mean" else "sum")
        segs = self._segments_for(rule, t_names, [slot_names[i] for i in take])
        if self.device.type == "cuda" and rule == "sgd" and lr is None and counters \
                and len(segs) <= 16:
            from ..ops import hip
            hip.multi_sgd([s[1] for s in segs], [s[2] for s in segs],
                          [lrs[take[s[0]]] for s in segs], list(counters), list(deltas))
            self.counters_applied = True
        elif segs:
            ps = [s[1] for s in segs]
            gs = [s[2] for s in segs]
            if self.device.type == "cuda":
                from ..ops import hip
                if rule == "sgd" and lr is None:
                    hip.multi_sgd(ps, gs, [lrs[take[s[0]]] for s in segs])
                else:
                    hip.multi_optim(rule, ps, gs, [s[3][0] for s in segs if s[3]],
                                    [s[3][1] for s in segs if len(s[3]) > 1], hyper, powers,
                                    lr=lr)
            else:
                from ..ops.ref_kernels import optimizer_rule_step
                for idx, p, g, slots in segs:
                    if rule == "sgd" and lr is None:
                        p.add_(g, alpha=-float(lrs[take[idx]]))
                        continue
                    h = tuple(hyper) if lr is None else (float(lr),) + tuple(hyper[1:])
                    new_p, s0, s1 = optimizer_rule_step(
                        rule, h, p, g, slots[0] if slots else None,
                        slots[1] if len(slots) > 1 else None, powers or None)
                    p.copy_(new_p)
                    for dst, src in zip(slots, (s0, s1)):
                        dst.copy_(src)
        p_shard = flat_p[r * shard:(r + 1) * shard]
        if self.comm.inplace_all_gather:
            self.comm.all_gather(flat_p, p_shard)            # the shard already is its own slot
        else:
            if self._p_stage is None or self._p_stage.numel() != shard:
                self._p_stage = torch.empty(shard, dtype=torch.float32, device=self.device)
            self._p_stage.copy_(p_shard)
            self.comm.all_gather(flat_p, self._p_stage)
        self.bytes_reduced += flat_g.numel() * 4
        self.bytes_gathered += flat_p.numel() * 4
        self.sharded_steps += 1
        return take

    def shutdown(self) -> None:
        pass
