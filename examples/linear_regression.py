#!/usr/bin/env python3
"""Between-graph replicated linear regression with SGD.

The same cluster / role / Supervisor structure as ``tf_distributed.py`` on the smallest possible
model. Meant for the collective runtimes -- e.g. 8 workers, one per MI355X, parameters owned in
shards (``reduce_scatter_sync``: gradients reduce-scattered to the owners, updated shards
all-gathered back)::

    python -m distributed_tensorflow_amd.launch --num_workers 8 --protocol rccl \
        --strategy reduce_scatter_sync examples/linear_regression.py -- --max_steps 500

but it runs unchanged as 1 ps + N workers over the asynchronous parameter server
(``--protocol grpc``, CPU).
"""
from __future__ import print_function

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import linear_regression as lr_model  # noqa: E402
from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter  # noqa: E402


def define_flags():
    tf.app.flags.define_cluster_flags("worker")
    tf.app.flags.DEFINE_string("protocol", "grpc", "grpc (async parameter server) | rccl")
    tf.app.flags.DEFINE_string("strategy", "reduce_scatter_sync",
                               "rccl only: allreduce_sync | reduce_scatter_sync")
    tf.app.flags.DEFINE_integer("features", 1024, "input dimension")
    tf.app.flags.DEFINE_integer("outputs", 1, "output dimension")
    tf.app.flags.DEFINE_integer("batch_size", 256, "per-worker batch")
    tf.app.flags.DEFINE_float("learning_rate", 0.05, "SGD step (per-worker gradients are summed "
                              "under the collective strategies: scale by 1/workers there)")
    tf.app.flags.DEFINE_integer("max_steps", 300, "local steps")
    tf.app.flags.DEFINE_boolean("shutdown_ps", False, "report completion to the ps tasks")
    tf.app.flags.DEFINE_string("device", "", "torch device of this task (default: auto)")
    tf.app.flags.DEFINE_boolean("device_data", True, "GPU tasks: generate the batches in GPU memory")
    tf.app.flags.DEFINE_boolean("capture_collectives", False, "rccl: record the step's RCCL calls "
                                "into the HIP graph together with its kernels")
    return tf.app.flags.FLAGS


def main(argv=None):
    FLAGS = tf.app.flags.FLAGS
    worker_hosts = FLAGS.worker_hosts or FLAGS.worker
    if worker_hosts:
        cluster = tf.train.ClusterSpec.from_host_lists(FLAGS.ps_hosts or FLAGS.ps, worker_hosts)
    elif "WORLD_SIZE" in os.environ and FLAGS.protocol == "rccl":
        cluster = tf.train.ClusterSpec.from_launcher_env()
        FLAGS.job_name, FLAGS.task_index = tf.train.task_from_env()
    else:
        cluster = tf.train.ClusterSpec({"worker": ["localhost:2223"]})
    config = tf.ConfigProto(device=FLAGS.device or None, strategy=FLAGS.strategy,
                            capture_collectives=FLAGS.capture_collectives)
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index,
                             protocol=FLAGS.protocol, config=config)
    if FLAGS.job_name == "ps":
        server.join()
        return 0

    workers = cluster.num_tasks("worker")
    step_size = FLAGS.learning_rate / (workers if FLAGS.protocol == "rccl" else 1)
    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)):
        model = lr_model.build_model(FLAGS.features, FLAGS.outputs, step_size)
    data = lr_model.SyntheticRegressionData(FLAGS.features, FLAGS.outputs, FLAGS.task_index)
    if FLAGS.device_data:
        data.to_device(server.device)       # GPU tasks: batches are generated in HBM
    sv = tf.train.Supervisor(is_chief=(FLAGS.task_index == 0), global_step=model.global_step,
                             init_op=model.init_op, recovery_wait_secs=1.0)
    begin = time.time()
    with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
        reporter = StepTimeReporter(100)
        reporter.start_epoch()
        loss = float("nan")
        for i in range(FLAGS.max_steps):
            bx, by = data.next_batch(FLAGS.batch_size)
            _, loss, step = sess.run([model.train_op, model.loss, model.global_step],
                                     feed_dict={model.x: bx, model.y_: by})
            reporter.step(step, 0, i, FLAGS.max_steps, loss)
        w = sess.run(model.w)
        err = float(np.abs(w - data.w).max())
        print("Weight-Error: %.5f" % err)
        print("Total Time: %3.2fs" % float(time.time() - begin))
        print("Final Cost: %.6f" % loss)
        if FLAGS.shutdown_ps and hasattr(server.runtime, "report_done"):
            server.runtime.report_done()
    sv.stop()
    server.stop()
    print("done")
    return 0


if __name__ == "__main__":
    define_flags()
    tf.app.run(main=main)
