#!/usr/bin/env python3
"""Headline benchmark: sharded GEMM TFLOP/s (whole job) and step latency on N MI355X.

Workload: ``C = A @ B`` with A, B 16384 x 16384 bf16 (synthetic: counter-based Philox
N(0,1) -- the reference's ``tf.random_normal`` operands -- generated in place on each rank), fp32 accumulation on the hand-written MFMA kernel,
result bf16 materialised in full on every rank. One *step* = one complete product including
its collectives -- the analogue of one ``sess.run(C)`` of the reference's matmul script
(``tf_distributed_1000Matrix.py:48,74``), at the 16384^3 bf16 size of the sharded-GEMM
configuration. Total work is fixed as N grows (strong scaling).

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Steps are independent products, so the all-gather strategies run them pipelined: two result
buffers alternate and the tail of step s's all-gather overlaps the first panel of step s+1;
every step's result is complete before the timed region ends (``--no-pipeline`` turns it off).

Rank 0 prints ONE JSON line. On more than one GPU ``--strategy auto`` (default) is a measured
choice in two phases:

0. the product with no data exchange (every rank computes all of C), kept as a fallback line:
   reported only if no sharded strategy completes (first contact with hardware this code has not
   run on must not end without a number). Next to it, for information only
   (``config.row_blocks_only_tflops``): the row-sharded product with C left in row blocks, i.e.
   what the same ranks reach when nothing has to travel.
A. the plain RCCL strategies -- row-sharded A + all-gather of C, K-sharded operands +
   all-reduce of partial sums -- are each measured in full (W warm-up steps, K timed steps
   between barriers). From here on a valid result line exists.
B. optional candidates, inside a time box (``--probe-budget``): the all-gather over peer-mapped
   memory, and -- when the fastest all-gather variant is transfer-bound -- variants in which
   every rank recomputes the first rows of its peers' blocks instead of receiving them
   (``config.recompute_rows``). A candidate must reproduce the RCCL variant's FULL result bit for
   bit, every stage (construction, first step, probe, timed run) ends in a vote of all ranks over
   a separate gloo group with a bounded wait, and a failure on any rank drops the candidate on
   all of them. If the ranks cannot agree any more or the box is exceeded (a rank stuck in a
   collective), every rank's watchdog reports phase A's line and exits 0. A candidate that is
   faster in the probe is timed for the same K steps and reported only if that measurement wins.

The choice is reported in ``config.parallelism``; each candidate's time goes to stderr. TFLOP/s
always count 2*M*N*K per step, never the recomputed rows. ``DTF_BENCH_INJECT_FAULT=<candidate
prefix>:<rank>`` makes that rank fail in that candidate's first step (a rehearsal of phase B's
recovery).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.parallel.comm import SingleComm, TorchDistComm  # noqa: E402
from distributed_tensorflow_amd.parallel.sharded_matmul import (  # noqa: E402
    ShardedMatmul, balanced_recompute_rows)

DTYPES = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp8": torch.float8_e4m3fn}
FULL_RESULT_STRATEGIES = ("m_shard_allgather", "k_shard_allreduce")
DIRECT_STRATEGIES = ("m_shard_allgather_direct",)     # peer-memory collectives (one node)


def timed_steps(engine: ShardedMatmul, comm, steps: int) -> float:
    """Wall seconds for ``steps`` steps, bracketed by barrier + device sync on both sides."""
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        engine.step()
    engine.wait()                 # pipelined strategies: drain the collectives still in flight
    torch.cuda.synchronize()
    comm.barrier()
    return time.perf_counter() - t0


PARALLELISM = {"replicated": "replicated%d", "m_shard": "m_shard%d",
               "m_shard_allgather": "m_shard%d+allgather",
               "k_shard_allreduce": "k_shard%d+allreduce",
               "m_shard_allgather_direct": "m_shard%d+allgather(peer memory)",
               "k_shard_allreduce_direct": "k_shard%d+allreduce(peer memory)",
               "k_shard_reduce_scatter": "k_shard%d+reduce_scatter", "summa_2d": "summa2d_%d"}


def result_line(args, world: int, strategy: str, engine: ShardedMatmul, elapsed: float) -> str:
    ms_per_step = elapsed * 1e3 / args.steps
    tflops = engine.flops_per_step / (ms_per_step * 1e-3) / 1e12
    par = "single_gpu" if (world == 1 and strategy == "replicated") else PARALLELISM[strategy] % world
    if engine.recompute_rows:
        par += "+recompute%d" % engine.recompute_rows
    return json.dumps({
        "metric": "sharded_gemm_tflops",
        "value": round(tflops, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic (Philox %s operands generated on device, random init)"
                % ("N(0,1)" if args.distribution == "normal" else "U(-1,1)"),
        "config": {"model": "gemm_%dx%dx%d_%s" % (args.m, args.n, args.k, args.dtype),
                   "global_batch": args.m, "seq_len": args.k,
                   "m": args.m, "n": args.n, "k": args.k,
                   "parallelism": par, "panels": engine.panels,
                   "pipelined_steps": bool(engine.pipelined),
                   "recompute_rows": engine.recompute_rows,
                   "fp8_scaling": engine.scaling if args.dtype == "fp8" else None,
                   "peer_transport": engine.peer_transport
                   if strategy.endswith("_direct") else None,
                   # what the same ranks reach when every rank keeps only its own row block of C
                   # (no data exchange): the ceiling the links take the step away from
                   "row_blocks_only_tflops": getattr(args, "row_blocks_only_tflops", None),
                   "result": "full C on every rank" if strategy in
                   ("replicated", "m_shard_allgather", "k_shard_allreduce",
                    "m_shard_allgather_direct", "k_shard_allreduce_direct")
                   else "C left sharded"},
    })


class Secured:
    """The best complete measurement so far, and the way out if anything later goes wrong.

    ``--strategy auto`` first measures the plain RCCL strategies in full (warm-up + K timed
    steps). From then on a valid result line exists; the optional candidates (peer-memory
    collectives, recomputation) are tried under a deadline. ``bail`` -- called by the watchdog
    thread when the deadline passes, or by the main thread when the ranks can no longer agree --
    prints that line on rank 0 and ends this rank with exit code 0 without touching the
    communicator again (a peer may be stuck inside a collective)."""

    def __init__(self, rank: int, result_fd: int):
        self.rank, self.result_fd = rank, result_fd
        self.line = None
        self.elapsed = None
        self.fallback = None          # reported only if no sharded measurement was completed
        self.lock = threading.Lock()
        self.deadline = None
        self.done = False

    def offer(self, line: str, elapsed: float) -> bool:
        with self.lock:
            if self.elapsed is None or elapsed < self.elapsed:
                self.line, self.elapsed = line, elapsed
                return True
            return False

    def offer_fallback(self, line: str) -> None:
        with self.lock:
            self.fallback = line

    @property
    def have_result(self) -> bool:
        return self.line is not None or self.fallback is not None

    def emit(self) -> None:
        with self.lock:
            if self.done:
                return
            self.done = True
            line = self.line if self.line is not None else self.fallback
            if self.rank == 0 and line is not None:
                os.write(self.result_fd, (line + "\n").encode())

    def bail(self, why: str) -> None:
        print("note: rank %d leaves the optional candidates (%s); reporting the measurement "
              "already taken" % (self.rank, why), file=sys.stderr)
        sys.stderr.flush()
        self.emit()
        os._exit(0 if self.have_result else 3)

    def arm(self, seconds: float) -> None:
        self.deadline = time.monotonic() + seconds

        def watch():
            while not self.done:
                if time.monotonic() > self.deadline:
                    self.bail("time box of %.0f s exceeded" % seconds)
                time.sleep(0.25)
        threading.Thread(target=watch, name="bench-watchdog", daemon=True).start()


def through_session(args, comm, device, world: int, secured, measure) -> None:
    """The product as ``sess.run`` of a ``tf.matmul`` of two variables (the reference's graph,
    ``tf_distributed_1000Matrix.py:43-48``, at the benchmark's size and dtype): variables placed
    by ``replica_device_setter``, initialised from the seeded random stream, the plan captured
    (with its collectives) and replayed. The fetched object is the product's OP, so C stays in
    HBM as in the engine's step; TFLOP/s count 2*M*N*K."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.train.server import _free_port
    tf_dtype = {"bf16": tf.bfloat16, "fp32": tf.float32, "fp8": tf.float8_e4m3fn}[args.dtype]
    # the engine on the same operands' shapes, for the comparison line
    strategy = "replicated" if world == 1 else {
        "m": "m_shard_allgather", "k": "k_shard_allreduce", "2d": "block_2d_allgather"}[args.shard]
    engine, engine_elapsed = measure(strategy)
    engine_ms = engine_elapsed * 1e3 / args.steps
    del engine
    torch.cuda.empty_cache()
    g = tf.Graph()
    with g.as_default():
        cfg = tf.ConfigProto(device=str(device), matmul_sharding=args.shard if world > 1 else "none",
                             matmul_panels=args.panels)
        ports = comm.broadcast_object([_free_port() for _ in range(world)], src=0)
        cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % p for p in ports]})
        server = tf.train.Server(cluster, job_name="worker", task_index=comm.rank,
                                 protocol="rccl", config=cfg, comm=comm)
        with tf.device(tf.train.replica_device_setter(
                worker_device="/job:worker/task:%d" % comm.rank, cluster=cluster)):
            tf.set_random_seed(1)
            rand = tf.random_normal if args.distribution == "normal" else \
                (lambda shape, dtype: tf.random_uniform(shape, -1.0, 1.0, dtype=dtype))
            A = tf.Variable(rand([args.m, args.k], dtype=tf_dtype))
            B = tf.Variable(rand([args.k, args.n], dtype=tf_dtype))
            C = tf.matmul(A, B)
            init = tf.global_variables_initializer()
        sv = tf.train.Supervisor(is_chief=(comm.rank == 0), init_op=init, global_step=None,
                                 summary_op=None, logdir=None)
        with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
            for _ in range(args.warmup):
                sess.run(C.op)
            comm.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sess.run(C.op)
            torch.cuda.synchronize()
            comm.barrier()
            elapsed = time.perf_counter() - t0
            stats = dict(sess.stats)
        sv.stop()
    if world > 1:
        t = torch.tensor([elapsed], device=device)
        comm.all_reduce(t, "max")
        elapsed = t.item()
    ms = elapsed * 1e3 / args.steps
    rec = {
        "metric": "sharded_gemm_tflops",
        "value": round(2.0 * args.m * args.n * args.k / (ms * 1e-3) / 1e12, 2),
        "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic (Philox %s operands generated on device, random init)"
                % ("N(0,1)" if args.distribution == "normal" else "U(-1,1)"),
        "config": {"model": "gemm_%dx%dx%d_%s" % (args.m, args.n, args.k, args.dtype),
                   "global_batch": args.m, "seq_len": args.k,
                   "m": args.m, "n": args.n, "k": args.k,
                   "parallelism": "single_gpu" if world == 1 else
                   "tf.matmul sharded '%s' over %d ranks" % (args.shard, world),
                   "through_session": True, "step": "sess.run(tf.matmul(A, B).op)",
                   "engine_ms_per_step": round(engine_ms, 4),
                   "session_over_engine": round(ms / engine_ms, 4),
                   "session_stats": {k: stats.get(k) for k in
                                     ("graph_replays", "direct_steps", "eager_runs",
                                      "torch_fallbacks")},
                   "result": "full C on every rank"},
    }
    secured.offer(json.dumps(rec), elapsed)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--k", type=int, default=16384)
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES))
    ap.add_argument("--fp8-scaling", default="tensor", choices=("tensor", "mx"),
                    help="--dtype fp8 only: 'mx' = block-scaled operands (one E8M0 scale per 32 k, "
                         "applied by the matrix instruction)")
    ap.add_argument("--strategy", default="auto")
    ap.add_argument("--panels", type=int, default=None)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="all-gather strategies: wait for a step's collectives before the next "
                         "step starts (default: two result buffers alternate, so the tail of one "
                         "step's all-gather overlaps the first panel of the next step)")
    ap.add_argument("--no-recompute", action="store_true",
                    help="all-gather strategies: never trade recomputed rows of the peers' blocks "
                         "for transfer (default: tried when the step is transfer-bound)")
    ap.add_argument("--no-direct", action="store_true",
                    help="do not try the peer-memory collectives")
    ap.add_argument("--peer-transport", default="copy", choices=("copy", "kernel", "epilogue"),
                    help="peer-memory all-gather: one runtime copy per peer on the copy engines "
                         "(default) or one kernel that stores the rows into all peers itself")
    ap.add_argument("--recompute-rows", type=int, default=None,
                    help="time this many recomputed rows per peer block instead of the values "
                         "around the measured balance point")
    ap.add_argument("--phase-a-budget", type=float, default=300.0,
                    help="seconds the fully measured RCCL strategies may take before the job "
                         "reports the measurement it already has")
    ap.add_argument("--probe-budget", type=float, default=60.0,
                    help="seconds the optional candidates of --strategy auto may take in total; "
                         "past it every rank reports the RCCL measurement already taken")
    ap.add_argument("--through-session", action="store_true",
                    help="time the same product as a user of the facade gets it: A and B are "
                         "tf.Variables under replica_device_setter, the step is sess.run of "
                         "tf.matmul(A, B) on the session's captured plan (sharded by the planner "
                         "on more than one rank: --shard m|k|2d). The engine's own step time is "
                         "measured next to it and reported as config.engine_ms_per_step")
    ap.add_argument("--shard", default="m", choices=("m", "k", "2d"),
                    help="--through-session on several ranks: ConfigProto(matmul_sharding=...)")
    ap.add_argument("--distribution", default="normal", choices=("normal", "uniform"),
                    help="operand values: N(0,1) as the reference's tf.random_normal "
                         "(tf_distributed_1000Matrix.py:43-44), or U(-1,1)")
    args = ap.parse_args()

    # stdout carries exactly one line, the result: everything else that writes to descriptor 1
    # while the job runs (RCCL prints a version banner there from C) is sent to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("DTF_BENCH_DEBUG"):        # dump every thread's stack if a run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["DTF_BENCH_DEBUG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    use_dist = world > 1 or os.environ.get("DTF_FORCE_DIST") == "1"   # world-1 RCCL smoke
    # DTF_COMM_BACKEND=gloo: control plane for rehearsing several ranks on ONE GPU (RCCL refuses
    # two ranks per device); the driver's runs use the default, RCCL
    comm = TorchDistComm(device=device, backend=os.environ.get("DTF_COMM_BACKEND") or None) \
        if use_dist else SingleComm(device)
    rank = comm.rank
    if world != args.gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world),
              file=sys.stderr)

    dtype = DTYPES[args.dtype]
    if args.dtype != "fp8":
        args.fp8_scaling = "tensor"
    if args.fp8_scaling == "mx":
        args.no_recompute = True          # block-scaled operands stay on the rank that made them
    secured = Secured(rank, result_fd)

    def say(msg: str) -> None:
        if rank == 0:
            print(msg, file=sys.stderr)
            sys.stderr.flush()

    def make(strategy: str, recompute_rows: int = 0) -> ShardedMatmul:
        eng = ShardedMatmul(args.m, args.n, args.k, comm, dtype=dtype, strategy=strategy,
                            panels=args.panels, seed=1, device=device,
                            distribution=args.distribution, pipelined=not args.no_pipeline,
                            recompute_rows=recompute_rows, peer_transport=args.peer_transport,
                            scaling=args.fp8_scaling,
                            # candidates are compared bit for bit: every product sums K in one
                            # sequential order whatever its shape (none of the 16384-wide panels
                            # of the default plan would be split anyway)
                            k_splits=1 if world > 1 else None)
        eng.init_operands()
        return eng

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], device=device)
        comm.all_reduce(t, "max")
        return t.item()

    def measure(strategy: str, recompute_rows: int = 0):
        """Warm-up + K timed steps of one strategy; the maximum over ranks."""
        eng = make(strategy, recompute_rows)
        for _ in range(args.warmup):
            eng.step()
        eng.wait()
        return eng, max_over_ranks(timed_steps(eng, comm, args.steps))

    if args.through_session:
        through_session(args, comm, device, world, secured, measure)
        secured.emit()
        comm.close()
        return

    if world == 1 or args.strategy != "auto" or args.dtype == "fp8":
        # one fixed strategy. fp8 is the 2-D sharded configuration (65536^2 sized for 288 GB
        # HBM): SUMMA on the P x Q grid, C left 2-D sharded
        strategy = args.strategy
        if strategy == "auto":
            # (block-scaled operands stay on the rank that quantised them: row blocks + all-gather)
            strategy = "replicated" if world == 1 else (
                "m_shard_allgather" if args.fp8_scaling == "mx" else "summa_2d")
        engine, elapsed = measure(strategy)
        secured.offer(result_line(args, world, strategy, engine, elapsed), elapsed)
        secured.emit()
        comm.close()
        return

    # ---- world > 1, measured choice -----------------------------------------------------------
    # Phase 0: the product with no data exchange at all (every rank computes all of C): a valid,
    # if unsharded, measurement that exists before the first collective carrying data is issued.
    # Whatever happens to a communicating strategy on hardware this code has not met before --
    # an error on one rank, a collective that never completes -- the job still reports a number.
    try:
        eng, elapsed = measure("replicated")
        say("candidate %-28s %8.3f ms/step (%d timed steps)"
            % ("replicated", elapsed * 1e3 / args.steps, args.steps))
        secured.offer_fallback(result_line(args, world, "replicated", eng, elapsed))
        del eng
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        say("note: replicated baseline failed: %s" % str(e)[:200])
    # (informational, same guard: the row-sharded product with C left in row blocks -- no
    # collective carries data -- shows what the transfer of the full result costs)
    try:
        eng, elapsed = measure("m_shard")
        args.row_blocks_only_tflops = round(
            eng.flops_per_step / (elapsed / args.steps) / 1e12, 2)
        say("candidate %-28s %8.3f ms/step (%d timed steps; C left in row blocks, reported as "
            "config.row_blocks_only_tflops only)" % ("m_shard", elapsed * 1e3 / args.steps,
                                                     args.steps))
        del eng
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        say("note: m_shard measurement failed: %s" % str(e)[:200])
    secured.arm(args.phase_a_budget)

    # Phase A: the plain RCCL strategies, measured in full. After it a sharded result exists.
    best_ag = None                    # (engine, elapsed) of the RCCL all-gather variant
    for cand in FULL_RESULT_STRATEGIES:
        try:
            eng, elapsed = measure(cand)
        except Exception as e:  # noqa: BLE001 -- peers may be inside the collective: do not rejoin
            if not secured.have_result:
                raise
            secured.bail("%s failed on rank %d: %s" % (cand, rank, str(e)[:160]))
        say("candidate %-28s %8.3f ms/step (%d timed steps)"
            % (cand, elapsed * 1e3 / args.steps, args.steps))
        secured.offer(result_line(args, world, cand, eng, elapsed), elapsed)
        if cand == "m_shard_allgather":
            best_ag = (eng, elapsed)
        else:
            del eng
            torch.cuda.empty_cache()
    if best_ag is None:
        secured.bail("no all-gather measurement")

    # Phase B: optional candidates under a deadline. Every stage ends in a vote over a separate
    # gloo group (bounded wait): a failure on any rank drops the candidate on all of them; ranks
    # that cannot agree any more, or a stage that outlasts the time box, fall back to phase A's
    # line and exit. The reference result of the RCCL variant is kept whole: a candidate is
    # eligible only if its full C is bit-identical (same kernel, same panels, same K order).
    import datetime
    import torch.distributed as dist
    ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(
        seconds=float(os.environ.get("DTF_BENCH_VOTE_TIMEOUT", "20"))))
    secured.arm(args.probe_budget)
    inject = os.environ.get("DTF_BENCH_INJECT_FAULT", "")     # "<candidate prefix>:<rank>"

    def agree(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        try:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctrl)
        except Exception as e:  # noqa: BLE001 -- a peer never arrived
            secured.bail("vote failed: %s" % str(e)[:120])
        return bool(flag.item())

    def stage(label: str, what: str, fn):
        """Run ``fn`` on this rank, then vote. Returns (everyone succeeded, this rank's value)."""
        val, ok = None, True
        try:
            if inject and label.startswith(inject.split(":")[0]) \
                    and int(inject.split(":")[1]) == rank and what == "first step":
                raise RuntimeError("injected fault (DTF_BENCH_INJECT_FAULT)")
            val = fn()
        except Exception as e:  # noqa: BLE001
            ok = False
            print("note: rank %d: %s failed at %s: %s" % (rank, label, what, str(e)[:300]),
                  file=sys.stderr)
        return agree(ok), val

    ag_engine, ag_elapsed = best_ag
    ag_engine.step()
    ag_engine.wait()
    torch.cuda.synchronize()
    reference = ag_engine.result_dense().clone()

    def consider(cand: str, recompute_rows: int = 0):
        label = cand + ("+recompute%d" % recompute_rows if recompute_rows else "")
        holder = {}

        def build():
            holder["eng"] = make(cand, recompute_rows)

        def first():
            eng = holder["eng"]
            eng.step()
            eng.wait()
            torch.cuda.synchronize()
            if not torch.equal(eng.result_dense(), reference):
                raise RuntimeError("result differs from the RCCL variant's")

        def probe():
            return timed_steps(holder["eng"], comm, max(2, args.warmup))

        def full():
            return timed_steps(holder["eng"], comm, args.steps)

        for what, fn in (("construction", build), ("first step", first)):
            ok, _ = stage(label, what, fn)
            if not ok:
                say("candidate %-28s dropped (%s failed on some rank)" % (label, what))
                holder.clear()
                torch.cuda.empty_cache()
                return None
        ok, secs = stage(label, "probe", probe)
        if not ok:
            say("candidate %-28s dropped (probe failed)" % label)
            return None
        per_step = max_over_ranks(secs) / max(2, args.warmup)
        if per_step * args.steps >= secured.elapsed:
            say("candidate %-28s %8.3f ms/step in the probe: not faster, not timed"
                % (label, per_step * 1e3))
            return holder["eng"], per_step, False
        ok, secs = stage(label, "timed run", full)
        if not ok:
            say("candidate %-28s dropped (timed run failed)" % label)
            return None
        elapsed = max_over_ranks(secs)
        say("candidate %-28s %8.3f ms/step (%d timed steps)"
            % (label, elapsed * 1e3 / args.steps, args.steps))
        secured.offer(result_line(args, world, cand, holder["eng"], elapsed), elapsed)
        return holder["eng"], elapsed / args.steps, True

    base_cand, base_eng, base_step = "m_shard_allgather", ag_engine, ag_elapsed / args.steps
    if not args.no_direct:
        for cand in DIRECT_STRATEGIES:
            got = consider(cand)
            if got is not None and got[1] < base_step:
                base_cand, base_eng, base_step = cand, got[0], got[1]
    # Full C on every rank puts M*N/world result elements on every link per step, however the
    # all-gather is organised. When the fastest all-gather variant is transfer-bound, let every
    # rank recompute the first x rows of its peers' blocks instead of receiving them: x from the
    # measured product time and step time (agreed across ranks); the two multiples of the row
    # granule around the balance point are tried like any other candidate.
    if not args.no_recompute:
        ok, tc = stage(base_cand, "product timing", base_eng.time_local_product)
        if ok:
            tc = max_over_ranks(tc)
            xs = [args.recompute_rows] if args.recompute_rows else balanced_recompute_rows(
                args.m // world, world, tc, base_step)
            say("note: %s: product %.3f ms, step %.3f ms -> recompute_rows candidates %s"
                % (base_cand, tc * 1e3, base_step * 1e3, xs))
            for x in xs:
                consider(base_cand, x)
    secured.emit()
    secured.deadline = time.monotonic() + 30.0     # closing the communicator may not hang either
    try:
        comm.close()
    except Exception:  # noqa: BLE001
        pass
    os._exit(0)


if __name__ == "__main__":
    main()
