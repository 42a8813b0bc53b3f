#!/usr/bin/env python3
"""Distributed 1000x1000 matrix-multiply benchmark -- the *intent* of the reference's
``tf_distributed_1000Matrix.py``.

The reference builds ``A``, ``B`` = ``tf.Variable(tf.random_normal([1000, 1000]))`` on the
parameter server and ``C = tf.matmul(A, B)`` on each worker
(``tf_distributed_1000Matrix.py:41-48``), but its loop was pasted from the MNIST script and
fetches names that do not exist in this file (``train_op``, ``cross_entropy``, ``x``, ``y_``,
``accuracy`` -- ``:74,89``), so it raises ``NameError`` on its first step. What it evidently
meant to do -- and what this script does -- is run ``sess.run(C)`` once per step and report the
average step time in the same stdout format, plus the derived GFLOP/s.

Roles and flags as in ``tf_distributed.py``. With ``--protocol=grpc`` (default) every step pulls
A and B (8 MB) from the ps like the reference would; with ``--protocol=rccl`` the operands live
in HBM and a step is fp32 MFMA GEMM work plus at most one collective:

* ``--shard=none``: A and B replicated, every worker computes all of C (the reference's layout);
* ``--shard=m``: worker r stores rows [r*M/W, (r+1)*M/W) of A, multiplies them by B, and the
  row blocks of C are all-gathered over RCCL -- the same graph, ``tf.matmul`` of two variables
  under ``replica_device_setter``, planned as a sharded product (``parallel/planner.py``);
* ``--shard=k``: inner-dimension blocks of A and B, all-reduce of the partial products;
* ``--shard=2d``: row blocks of A x column blocks of B on the P x Q worker grid, all-gather.

The variables are generated shard by shard from the seeded counter-based stream, so C does not
depend on the number of workers (``--shard=m``: bit for bit).
``--matrix_size``, ``--steps_per_epoch`` and ``--training_epochs`` replace the reference's
borrowed MNIST loop bounds (550 x 100).
"""
from __future__ import print_function

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter  # noqa: E402

logs_path = "/tmp/mnist/1"


def define_flags():
    tf.app.flags.define_cluster_flags("ps")     # default role ps, like the reference (:13)
    tf.app.flags.DEFINE_string("protocol", "grpc", "grpc (async parameter server) | rccl")
    tf.app.flags.DEFINE_string("strategy", "allreduce_sync", "rccl strategy")
    tf.app.flags.DEFINE_string("shard", "none", "rccl: how C = A @ B is distributed over the "
                               "workers: none | m | k | 2d")
    tf.app.flags.DEFINE_string("dump_result", "", "write the last C of this worker to this .npy "
                               "file (tests compare worlds)")
    tf.app.flags.DEFINE_integer("matrix_size", 1000, "N of the NxN operands")
    tf.app.flags.DEFINE_integer("training_epochs", 100, "epochs (reference: 100)")
    tf.app.flags.DEFINE_integer("steps_per_epoch", 550, "steps per epoch (reference: 550)")
    tf.app.flags.DEFINE_integer("max_steps", 0, "stop after this many local steps (0 = off)")
    tf.app.flags.DEFINE_string("logs_path", logs_path, "event-file directory")
    tf.app.flags.DEFINE_boolean("shutdown_ps", False, "report completion so the ps exits once all workers are done")
    tf.app.flags.DEFINE_string("device", "", "torch device of this task (default: auto)")
    tf.app.flags.DEFINE_boolean("deferred_fetches", False, "GPU tasks: sess.run returns as soon as "
                                "the step is launched; the checksum is waited for when it is "
                                "printed")
    tf.app.flags.DEFINE_boolean("capture_collectives", False, "rccl: record the step's RCCL calls "
                                "into the HIP graph together with its kernels")
    return tf.app.flags.FLAGS


def main(argv=None):
    FLAGS = tf.app.flags.FLAGS
    ps_hosts = FLAGS.ps_hosts or FLAGS.ps
    worker_hosts = FLAGS.worker_hosts or FLAGS.worker
    if worker_hosts:
        cluster = tf.train.ClusterSpec.from_host_lists(ps_hosts, worker_hosts)
    elif "WORLD_SIZE" in os.environ and FLAGS.protocol == "rccl":
        cluster = tf.train.ClusterSpec.from_launcher_env()
        FLAGS.job_name, FLAGS.task_index = tf.train.task_from_env()
    else:
        cluster = tf.train.ClusterSpec({"ps": ["localhost:2222"],
                                        "worker": ["localhost:%d" % (2223 + i) for i in range(7)]})
    config = tf.ConfigProto(device=FLAGS.device or None, strategy=FLAGS.strategy,
                            capture_collectives=FLAGS.capture_collectives,
                            matmul_sharding=FLAGS.shard, deferred_fetches=FLAGS.deferred_fetches)
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index,
                             protocol=FLAGS.protocol, config=config)
    if FLAGS.job_name == "ps":
        server.join()
        return 0

    n = FLAGS.matrix_size
    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)):
        # count the number of steps
        global_step = tf.get_variable("global_step", [], initializer=tf.constant_initializer(0),
                                      trainable=False)
        tf.set_random_seed(1)
        with tf.name_scope("weights"):
            A = tf.Variable(tf.random_normal([n, n]))
            B = tf.Variable(tf.random_normal([n, n]))
        # matrix multiplication
        with tf.name_scope("matmul"):
            C = tf.matmul(A, B)
            # a scalar that depends on every element of C: fetching it proves the product ran
            # without copying 4 MB back to the host each step
            checksum = tf.reduce_mean(C)
        # merge all summaries into a single "operation" (none here, so this is None)
        summary_op = tf.summary.merge_all()
        init_op = tf.global_variables_initializer()
        print("Variables initialized ...")

    sv = tf.train.Supervisor(is_chief=(FLAGS.task_index == 0), global_step=global_step,
                             init_op=init_op, recovery_wait_secs=1.0)
    begin_time = time.time()
    frequency = 100
    flops = 2.0 * n * n * n
    local_step = 0
    with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
        writer = tf.summary.FileWriter(FLAGS.logs_path, graph=tf.get_default_graph())
        reporter = StepTimeReporter(frequency)
        for name, lay in sorted(getattr(server.runtime, "var_layouts", {}).items()):
            print("Sharded variable %s: %s blocks of %s -> local %s" % (
                name, "row" if lay.axis == 0 else "column", list(lay.shape),
                list(sess.variable_store.read(name).shape)))
        done = False
        value = float("nan")
        for epoch in range(FLAGS.training_epochs):
            batch_count = FLAGS.steps_per_epoch
            reporter.start_epoch()
            for i in range(batch_count):
                value, summary = sess.run([checksum, summary_op])
                writer.add_summary(summary, local_step)
                ms = reporter.step(local_step, epoch, i, batch_count, value)
                if ms:
                    print("GFLOPS: %.1f" % (flops / (ms * 1e-3) / 1e9))
                local_step += 1
                if FLAGS.max_steps and local_step >= FLAGS.max_steps:
                    done = True
                    break
            print("Total Time: %3.2fs" % float(time.time() - begin_time))
            print("Final Cost: %.4f" % value)
            if done:
                break
        if FLAGS.dump_result:
            import numpy as np
            np.save(FLAGS.dump_result.replace("{rank}", str(FLAGS.task_index)), sess.run(C))
        writer.close()
        if FLAGS.shutdown_ps and hasattr(server.runtime, "report_done"):
            server.runtime.report_done()      # the ps exits join() once every worker has
    sv.stop()
    server.stop()
    print("done")
    return 0


if __name__ == "__main__":
    define_flags()
    tf.app.run(main=main)
