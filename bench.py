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

Rank 0 prints ONE JSON line. ``--strategy auto`` (default) times the full-result strategies
(row-sharded A + all-gather of C; K-sharded operands + all-reduce of partial sums) during
warm-up -- plus the all-gather over peer-mapped memory when that is available and bit-identical
-- and runs the fastest; the choice is reported in ``config.parallelism``. If the fastest
all-gather variant is transfer-bound, variants in which every rank recomputes the first rows of
its peers' blocks instead of receiving them are timed too (``config.recompute_rows``); the
TFLOP/s always count 2*M*N*K per step, never the recomputed rows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.parallel.comm import SingleComm, TorchDistComm  # noqa: E402
from distributed_tensorflow_amd.parallel.peer_mem import PeerMemError  # noqa: E402
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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--k", type=int, default=16384)
    ap.add_argument("--dtype", default="bf16", choices=sorted(DTYPES))
    ap.add_argument("--strategy", default="auto")
    ap.add_argument("--panels", type=int, default=None)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="all-gather strategies: wait for a step's collectives before the next "
                         "step starts (default: two result buffers alternate, so the tail of one "
                         "step's all-gather overlaps the first panel of the next step)")
    ap.add_argument("--no-recompute", action="store_true",
                    help="all-gather strategies: never trade recomputed rows of the peers' blocks "
                         "for transfer (default: tried when the step is transfer-bound)")
    ap.add_argument("--peer-transport", default="copy", choices=("copy", "kernel"),
                    help="peer-memory all-gather: one runtime copy per peer on the copy engines "
                         "(default) or one kernel that stores the rows into all peers itself")
    ap.add_argument("--recompute-rows", type=int, default=None,
                    help="time this many recomputed rows per peer block instead of the values "
                         "around the measured balance point")
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
    device = torch.device("cuda", local_rank)
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

    def make(strategy: str, recompute_rows: int = 0) -> ShardedMatmul:
        eng = ShardedMatmul(args.m, args.n, args.k, comm, dtype=dtype, strategy=strategy,
                            panels=args.panels, seed=1, device=device,
                            distribution=args.distribution, pipelined=not args.no_pipeline,
                            recompute_rows=recompute_rows, peer_transport=args.peer_transport)
        eng.init_operands()
        return eng

    if world == 1:
        strategy = "replicated" if args.strategy == "auto" else args.strategy
        engine = make(strategy)
        for _ in range(args.warmup):
            engine.step()
    elif args.strategy != "auto" or args.dtype == "fp8":
        # fp8 is the 2-D sharded configuration (65536^2 sized for 288 GB HBM): SUMMA on the
        # P x Q grid, C left 2-D sharded
        strategy = "summa_2d" if args.strategy == "auto" else args.strategy
        engine = make(strategy)
        for _ in range(args.warmup):
            engine.step()
    else:
        # measured choice, inside the (untimed) warm-up: run each candidate, keep the fastest.
        # The peer-memory variant is only eligible if its buffers could be mapped on every rank
        # and its result is bit-identical to the RCCL variant's (same kernel, same panels).
        best = None
        probe = max(2, args.warmup)
        baseline_rows = None
        sample = torch.arange(0, args.m, max(1, args.m // 64), device=device)
        def consider(cand: str, recompute_rows: int = 0) -> None:
            nonlocal best, baseline_rows
            label = cand + ("+recompute%d" % recompute_rows if recompute_rows else "")
            try:
                eng = make(cand, recompute_rows)
            except PeerMemError as e:                    # raised on every rank or on none
                if rank == 0:
                    print("note: %s unavailable (%s)" % (label, str(e)[:200]), file=sys.stderr)
                return
            eng.step()                                   # first-touch / communicator setup
            eng.wait()
            torch.cuda.synchronize()
            if cand == "m_shard_allgather" and not recompute_rows:
                baseline_rows = eng.result_dense().index_select(0, sample).clone()
            if cand in DIRECT_STRATEGIES or recompute_rows:
                same = baseline_rows is not None and torch.equal(
                    eng.result_dense().index_select(0, sample), baseline_rows)
                ok = torch.tensor([1.0 if same else 0.0], device=device)
                comm.all_reduce(ok, "min")
                if ok.item() < 1.0:
                    if rank == 0:
                        print("note: %s result differs from the RCCL variant; not used" % label,
                              file=sys.stderr)
                    del eng
                    torch.cuda.empty_cache()
                    return
            secs = timed_steps(eng, comm, probe)
            t = torch.tensor([secs], device=device)
            comm.all_reduce(t, "max")
            if best is None or t.item() < best[0]:
                best = (t.item(), cand, eng)
            else:
                del eng
                torch.cuda.empty_cache()

        for cand in FULL_RESULT_STRATEGIES + DIRECT_STRATEGIES:
            consider(cand)
        # Full C on every rank puts M*N/world result elements on every link per step, however
        # the all-gather is organised. When the fastest all-gather variant is transfer-bound,
        # let every rank recompute the first x rows of its peers' blocks instead of receiving
        # them: x from the measured product time and step time (agreed across ranks), the two
        # multiples of 128 rows around the balance point are timed like any other candidate.
        if best[1].startswith("m_shard_allgather") and not args.no_recompute:
            t_step, base_cand, base_eng = best
            tc = torch.tensor([base_eng.time_local_product()], device=device)
            comm.all_reduce(tc, "max")
            xs = [args.recompute_rows] if args.recompute_rows else balanced_recompute_rows(
                args.m // world, world, tc.item(), t_step / probe)
            if rank == 0:
                print("note: %s: product %.3f ms, step %.3f ms -> recompute_rows candidates %s"
                      % (base_cand, tc.item() * 1e3, t_step / probe * 1e3, xs), file=sys.stderr)
            for x in xs:
                consider(base_cand, x)
        _, strategy, engine = best

    elapsed = timed_steps(engine, comm, args.steps)
    if world > 1:
        t = torch.tensor([elapsed], device=device)
        comm.all_reduce(t, "max")
        elapsed = t.item()

    ms_per_step = elapsed * 1e3 / args.steps
    tflops = engine.flops_per_step / (ms_per_step * 1e-3) / 1e12
    if rank == 0:
        par = {"replicated": "single_gpu" if world == 1 else "replicated%d" % world,
               "m_shard": "m_shard%d" % world,
               "m_shard_allgather": "m_shard%d+allgather" % world,
               "k_shard_allreduce": "k_shard%d+allreduce" % world,
               "m_shard_allgather_direct": "m_shard%d+allgather(peer memory)" % world,
               "k_shard_allreduce_direct": "k_shard%d+allreduce(peer memory)" % world,
               "k_shard_reduce_scatter": "k_shard%d+reduce_scatter" % world,
               "summa_2d": "summa2d_%d" % world}[strategy]
        if engine.recompute_rows:
            par += "+recompute%d" % engine.recompute_rows
        line = json.dumps({
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
                       "peer_transport": engine.peer_transport
                       if strategy.endswith("_direct") else None,
                       "result": "full C on every rank" if strategy in
                       ("replicated", "m_shard_allgather", "k_shard_allreduce",
                        "m_shard_allgather_direct", "k_shard_allreduce_direct")
                       else "C left sharded"},
        })
        sys.stdout.flush()
        os.write(result_fd, (line + "\n").encode())
    comm.close()


if __name__ == "__main__":
    main()
