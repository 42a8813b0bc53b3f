"""Rank body for the 2-process gloo tests (started by ``torch.distributed.run``).

Checks, over a real process group: TorchDistComm collectives, every ShardedMatmul strategy
against a single-process product of the same Philox operands, and the two synchronous
runtimes keeping replicas bit-identical."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.parallel.comm import SingleComm, TorchDistComm  # noqa: E402
from distributed_tensorflow_amd.parallel.sharded_matmul import ShardedMatmul  # noqa: E402


def main() -> int:
    dev = torch.device("cpu")
    comm = TorchDistComm(device=dev, backend="gloo")
    rank, world = comm.rank, comm.world_size
    out = {"rank": rank, "world": world}

    # collectives
    t = torch.full((8,), float(rank + 1))
    comm.all_reduce(t, "sum")
    out["allreduce"] = t[0].item()
    parts = torch.arange(4 * world, dtype=torch.float32) * (rank + 1)
    shard = torch.empty(4)
    comm.reduce_scatter(shard, parts, "sum")
    out["reduce_scatter"] = shard.tolist()
    gathered = torch.empty(2 * world)
    comm.all_gather(gathered, torch.full((2,), float(rank)))
    out["all_gather"] = gathered.tolist()
    out["bcast_obj"] = comm.broadcast_object({"x": 7} if rank == 0 else None, src=0)

    # sharded matmul vs the single-process product
    m = n = k = 64
    ref = ShardedMatmul(m, n, k, SingleComm(dev), dtype=torch.float32, strategy="replicated",
                        seed=3, device=dev)
    ref.init_operands()
    want = ref.step().clone()
    errs = {}
    for strategy in ("m_shard_allgather", "k_shard_allreduce", "k_shard_reduce_scatter", "m_shard"):
        eng = ShardedMatmul(m, n, k, comm, dtype=torch.float32, strategy=strategy, seed=3,
                            device=dev)
        eng.init_operands()
        got = eng.step()
        if strategy in ("m_shard", "k_shard_reduce_scatter"):
            rows = m // world
            expect = want[rank * rows:(rank + 1) * rows]
        else:
            expect = want
        errs[strategy] = (got.float() - expect.float()).abs().max().item()
    out["matmul_err"] = errs
    # pipelined all-gather: steps do not wait for their own collectives; wait() drains them
    eng = ShardedMatmul(m, n, k, comm, dtype=torch.float32, strategy="m_shard_allgather",
                        seed=3, device=dev, pipelined=True, panels=2)
    eng.init_operands()
    for _ in range(5):
        got = eng.step()
    eng.wait()
    out["pipelined_err"] = (eng.result_dense().float() - want.float()).abs().max().item()

    # synchronous data-parallel training through the facade
    strategy = os.environ.get("DTF_STRATEGY", "allreduce_sync")
    cluster = tf.train.ClusterSpec.from_launcher_env()
    job, idx = tf.train.task_from_env()
    server = tf.train.Server(cluster, job_name=job, task_index=idx, protocol="rccl",
                             config=tf.ConfigProto(device="cpu", strategy=strategy), comm=comm)
    bucket_bytes = int(os.environ.get("DTF_BUCKET_BYTES", "0")) or None
    with tf.device(tf.train.replica_device_setter(cluster=cluster)):
        step = tf.train.get_or_create_global_step()
        x = tf.placeholder(tf.float32, [None, 4])
        y = tf.placeholder(tf.float32, [None, 1])
        w = tf.Variable(tf.zeros([4, 1]), name="w")
        b = tf.Variable(tf.zeros([1]), name="b")
        loss = tf.reduce_mean(tf.square(tf.matmul(x, w) + b - y))
        train = tf.train.GradientDescentOptimizer(0.05).minimize(loss, global_step=step)
        init = tf.global_variables_initializer()
    sv = tf.train.Supervisor(is_chief=(idx == 0), global_step=step, init_op=init,
                             recovery_wait_secs=0.1)
    g = torch.Generator().manual_seed(100 + rank)          # every rank its own data
    true_w = torch.tensor([[1.0], [-2.0], [0.5], [3.0]])
    run_config = tf.ConfigProto(device="cpu", strategy=strategy, grad_bucket_bytes=bucket_bytes)
    with sv.prepare_or_wait_for_session(server.target, config=run_config) as sess:
        first = last = None
        for _ in range(60):
            bx = torch.randn(16, 4, generator=g)
            by = bx @ true_w + 0.25
            _, l = sess.run([train, loss], {x: bx.numpy(), y: by.numpy()})
            first = l if first is None else first
            last = l
        out["loss_first"], out["loss_last"] = float(first), float(last)
        out["w"] = [float(v) for v in sess.run(w).reshape(-1)]
        out["step"] = float(sess.run(step))
        out["buckets_issued"] = getattr(server.runtime, "buckets_issued", 0)
        out["sharded_steps"] = getattr(server.runtime, "sharded_steps", 0)
        out["grads_placed"] = (getattr(server.runtime, "grads_in_place", 0)
                               + getattr(server.runtime, "grads_copied", 0))
    sv.stop()
    server.stop()
    with open(os.path.join(os.environ["DTF_RESULT_DIR"], "rank%d.json" % rank), "w") as fh:
        json.dump(out, fh)
    comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
