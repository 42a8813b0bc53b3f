"""Rank body for the GPU rehearsal of the sharded ``tf.matmul`` node: N processes that all use
GPU 0 (control plane gloo, as RCCL refuses two ranks per device). Each rank builds the
reference's graph -- ``tf.matmul`` of two variables under ``replica_device_setter`` -- at
``DTF_SHAPE`` in ``DTF_DTYPE``, runs it sharded (mode ``DTF_SHARD``; ``DTF_COLLECTIVE=peer`` sends
the row panels over peer-mapped memory), compares with the same product computed by this rank
alone, traces one step, and writes ``<out_dir>/rank<r>.json`` (rank 0 also ``trace.json``)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.parallel.comm import TorchDistComm  # noqa: E402


def main() -> int:
    out_dir = sys.argv[1]
    m, n, k = (int(v) for v in os.environ.get("DTF_SHAPE", "2048,2048,2048").split(","))
    dtype = {"fp32": tf.float32, "bf16": tf.bfloat16, "fp8": tf.float8_e4m3fn}[
        os.environ.get("DTF_DTYPE", "fp32")]
    mode = os.environ.get("DTF_SHARD", "m")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    comm = TorchDistComm(device=dev, backend="gloo")
    rank, world = comm.rank, comm.world_size

    def graph():
        g = tf.Graph()
        with g.as_default():
            tf.set_random_seed(1)
            A = tf.Variable(tf.random_normal([m, k], dtype=dtype))
            B = tf.Variable(tf.random_normal([k, n], dtype=dtype))
            C = tf.matmul(A, B)
            init = tf.global_variables_initializer()
        return g, C, init

    # this rank alone: the whole product
    g, C, init = graph()
    with g.as_default(), tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
        sess.run(init)
        alone = sess.run(C)
        fallbacks_alone = sess.stats["torch_fallbacks"]

    g, C, init = graph()
    res = {"rank": rank, "world": world, "mode": mode}
    with g.as_default():
        cfg = tf.ConfigProto(device="cuda:0", matmul_sharding=mode,
                             matmul_collective=os.environ.get("DTF_COLLECTIVE") or None,
                             matmul_panels=int(os.environ.get("DTF_PANELS", "0")) or None,
                             capture_collectives=False)
        cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % (20000 + i)
                                                   for i in range(world)]})
        server = tf.train.Server(cluster, job_name="worker", task_index=rank, protocol="rccl",
                                 config=cfg, comm=comm)
        sv = tf.train.Supervisor(is_chief=(rank == 0), init_op=init, global_step=None,
                                 summary_op=None, logdir=None)
        with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
            for _ in range(3):
                got = sess.run(C)
            md = tf.RunMetadata()
            sess.run(C.op, options=tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE),
                     run_metadata=md)
            engines = [v for key, v in sess._ctx.buffers.items()
                       if isinstance(key, tuple) and len(key) == 2 and key[1] == "engine"]
            res["strategy"] = engines[0].strategy
            res["panels"] = engines[0].panels
            res["torch_fallbacks"] = sess.stats["torch_fallbacks"] + fallbacks_alone
            res["a_local"] = list(sess.variable_store.read(
                g.get_collection("variables")[0].op.name).shape)
        sv.stop()
    a64, g64 = np.asarray(alone, dtype=np.float64), np.asarray(got, dtype=np.float64)
    res["bit_equal"] = bool(np.array_equal(np.asarray(alone), np.asarray(got)))
    res["max_err"] = float(np.abs(a64 - g64).max())
    res["scale"] = float(np.abs(a64).max())
    # overlap on the device: a transfer bracket (collective category) that runs while a panel
    # product of the same step does
    stats = md.step_stats
    products = [(s["start"], s["end"]) for s in stats if s["op"] == "panel_product"]
    moves = [(s["start"], s["end"], s["name"]) for s in stats
             if s["cat"] == "collective" and s["op"] in ("peer_push", "all_gather", "all_reduce")]
    overlap = 0.0
    for p0, p1 in products:
        for m0, m1, _ in moves:
            overlap += max(0.0, min(p1, m1) - max(p0, m0))
    res["panel_products"] = len(products)
    res["transfers"] = len(moves)
    res["overlap_us"] = overlap * 1e6
    res["product_us"] = sum(p1 - p0 for p0, p1 in products) * 1e6
    res["transfer_us"] = sum(m1 - m0 for m0, m1, _ in moves) * 1e6
    if rank == 0:
        from distributed_tensorflow_amd.utils.tracing import write_chrome_trace
        write_chrome_trace(os.path.join(out_dir, "trace.json"), stats)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump(res, fh)
    comm.barrier()
    comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
