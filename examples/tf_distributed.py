#!/usr/bin/env python3
"""Distributed MNIST MLP training -- the workload of the reference's ``tf_distributed.py``,
written against this framework's ``tf.train``-compatible facade.

One process per task, started by hand (or by ``launch.py``)::

    python examples/tf_distributed.py --job_name=ps --task_index=0
    python examples/tf_distributed.py --job_name=worker --task_index=0
    python examples/tf_distributed.py --job_name=worker --task_index=1
    ...

Same structure as the reference (``tf_distributed.py:8-131``): ClusterSpec -> Server ->
``replica_device_setter`` placement -> model -> Supervisor -> ``sess.run`` loop with a
step-time report every 100 steps and a test-accuracy line per epoch. Differences, all
deliberate:

* the cluster is configurable (``--ps_hosts`` / ``--worker_hosts``, or the reference's
  ``--ps`` / ``--worker`` spellings that its argparse block declared but never used,
  ``tf_distributed.py:137-148``); the default is 1 ps + 5 workers like the reference's, on
  localhost ports instead of 10.0.1.x hosts;
* ``--protocol=grpc`` (default) is the asynchronous parameter server of the reference and runs
  without a GPU; ``--protocol=rccl`` runs the same script as one worker per MI355X with the
  parameter pull/push replaced by RCCL collectives (``--strategy``);
* MNIST comes from IDX files in ``--data_dir`` when present and is synthetic otherwise;
* ``AvgTime`` divides by the number of steps in the window (the reference under-reports the
  50-step tail window of each epoch by 2x); ``--max_steps`` bounds a run for tests;
* with ``--shutdown_ps`` every worker reports completion and the ps leaves ``join()`` once all
  of them have; the reference's ps has to be killed.
"""
from __future__ import print_function

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.data import mnist as input_data  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402
from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter  # noqa: E402

# defaults of this variant (tf_distributed_worker.py overrides them)
DEFAULT_JOB = "ps"
DEFAULT_NUM_WORKERS = 5
DEFAULT_EPOCHS = 20

# config
batch_size = 100
learning_rate = 0.0005
logs_path = "/tmp/mnist/1"


def make_optimizer(name: str, lr: float):
    """The optimizer named on the command line; ``None`` = the reference's
    ``GradientDescentOptimizer(learning_rate)`` (built by the model itself)."""
    if name == "sgd":
        return None
    if name in ("momentum", "nesterov"):
        return tf.train.MomentumOptimizer(lr, 0.9, use_nesterov=(name == "nesterov"))
    if name == "adagrad":
        return tf.train.AdagradOptimizer(lr * 20)
    if name == "adadelta":
        return tf.train.AdadeltaOptimizer(1.0)
    if name == "rmsprop":
        return tf.train.RMSPropOptimizer(lr)
    if name == "adam":
        return tf.train.AdamOptimizer(lr * 2)
    raise ValueError("unknown --optimizer %r" % name)


def define_flags(default_job=DEFAULT_JOB, default_epochs=DEFAULT_EPOCHS):
    tf.app.flags.define_cluster_flags(default_job)
    tf.app.flags.DEFINE_string("protocol", "grpc", "grpc (async parameter server) | rccl")
    tf.app.flags.DEFINE_string("strategy", "allreduce_sync",
                               "rccl only: allreduce_sync | reduce_scatter_sync")
    tf.app.flags.DEFINE_integer("training_epochs", default_epochs, "epochs")
    tf.app.flags.DEFINE_integer("max_steps", 0, "stop after this many local steps (0 = off)")
    tf.app.flags.DEFINE_string("data_dir", "MNIST_data", "directory with the MNIST IDX files")
    tf.app.flags.DEFINE_string("logs_path", logs_path, "event-file directory")
    tf.app.flags.DEFINE_string("metrics_path", "", "also append every report as a JSON line here")
    tf.app.flags.DEFINE_string("checkpoint_dir", "", "Supervisor logdir (enables checkpoints)")
    tf.app.flags.DEFINE_boolean("shutdown_ps", False, "report completion so the ps exits once all workers are done")
    tf.app.flags.DEFINE_string("device", "", "torch device of this task (default: auto)")
    tf.app.flags.DEFINE_string("optimizer", "sgd", "sgd (the reference's plain gradient descent) | "
                               "momentum | nesterov | adagrad | adadelta | rmsprop | adam")
    tf.app.flags.DEFINE_boolean("capture_collectives", False, "rccl: record the step's RCCL calls "
                                "into the HIP graph together with its kernels")
    tf.app.flags.DEFINE_boolean("device_data", True, "keep the data set resident in GPU memory "
                                "(GPU tasks only) so feeds are device-to-device copies")
    tf.app.flags.DEFINE_integer("data_seed", -1, "seed of the per-epoch shuffle (-1: unseeded, as "
                                "the reference's read_data_sets: every run walks the data in "
                                "another order, so single-batch costs differ from run to run)")
    tf.app.flags.DEFINE_boolean("hip_graphs", True, "GPU tasks: capture steps into HIP graphs "
                                "(false: every step is issued kernel by kernel)")
    tf.app.flags.DEFINE_boolean("deferred_fetches", False, "GPU tasks: sess.run returns as soon as "
                                "the step is enqueued; cost / summary / step are waited for only "
                                "when looked at (every 100 steps here)")
    return tf.app.flags.FLAGS


def build_cluster(FLAGS, default_num_workers=DEFAULT_NUM_WORKERS):
    ps_hosts = FLAGS.ps_hosts or FLAGS.ps
    worker_hosts = FLAGS.worker_hosts or FLAGS.worker
    if worker_hosts:
        return tf.train.ClusterSpec.from_host_lists(ps_hosts, worker_hosts)
    if "WORLD_SIZE" in os.environ and FLAGS.protocol == "rccl":
        return tf.train.ClusterSpec.from_launcher_env()
    # cluster specification (the reference hard-codes 10.0.1.2 .. 10.0.1.7, port 2222)
    parameter_servers = ["localhost:2222"]
    workers = ["localhost:%d" % (2223 + i) for i in range(default_num_workers)]
    return tf.train.ClusterSpec({"ps": parameter_servers, "worker": workers})


def main(argv=None, default_num_workers=DEFAULT_NUM_WORKERS):
    FLAGS = tf.app.flags.FLAGS
    cluster = build_cluster(FLAGS, default_num_workers)
    if "WORLD_SIZE" in os.environ and FLAGS.protocol == "rccl" and not FLAGS.worker_hosts:
        FLAGS.job_name, FLAGS.task_index = tf.train.task_from_env()
    config = tf.ConfigProto(device=FLAGS.device or None, strategy=FLAGS.strategy,
                            capture_collectives=FLAGS.capture_collectives,
                            deferred_fetches=FLAGS.deferred_fetches, hip_graphs=FLAGS.hip_graphs)

    # start a server for a specific task
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index,
                             protocol=FLAGS.protocol, config=config)

    # load mnist data set (every task loads the full set and shuffles independently)
    mnist = input_data.read_data_sets(FLAGS.data_dir, one_hot=True,
                                      seed=None if FLAGS.data_seed < 0 else FLAGS.data_seed)

    if FLAGS.job_name == "ps":
        server.join()
        return 0

    # Between-graph replication
    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)):
        model = mnist_mlp.build_model(learning_rate=learning_rate, seed=1,
                                      optimizer=make_optimizer(FLAGS.optimizer, learning_rate))
        print("Variables initialized ...")

    sv = tf.train.Supervisor(is_chief=(FLAGS.task_index == 0), global_step=model.global_step,
                             init_op=model.init_op, logdir=FLAGS.checkpoint_dir or None,
                             summary_op=None, recovery_wait_secs=1.0)

    test_images, test_labels = mnist.test.images, mnist.test.labels
    if FLAGS.device_data and server.device.type == "cuda":
        mnist.train.to_device(server.device)
        mnist.test.to_device(server.device)
        test_images, test_labels = mnist.test.device_images, mnist.test.device_labels

    begin_time = time.time()
    frequency = 100
    local_step = 0
    with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
        writer = tf.summary.FileWriter(FLAGS.logs_path, graph=tf.get_default_graph())
        sink = None
        if FLAGS.metrics_path:
            from distributed_tensorflow_amd.utils.metrics import JsonLinesSink
            sink = JsonLinesSink(FLAGS.metrics_path, common={"job": FLAGS.job_name,
                                                             "task": FLAGS.task_index})
        reporter = StepTimeReporter(frequency, sink=sink)
        cost = float("nan")
        done = False
        # perform training cycles
        for epoch in range(FLAGS.training_epochs):
            # number of batches in one epoch
            batch_count = int(mnist.train.num_examples / batch_size)
            reporter.start_epoch()
            for i in range(batch_count):
                batch_x, batch_y = mnist.train.next_batch(batch_size)
                # perform the operations we defined earlier on batch
                _, cost, summary, step = sess.run(
                    [model.train_op, model.loss, model.summary_op, model.global_step],
                    feed_dict={model.x: batch_x, model.y_: batch_y})
                writer.add_summary(summary, step)
                reporter.step(step, epoch, i, batch_count, cost)
                local_step += 1
                if FLAGS.max_steps and local_step >= FLAGS.max_steps:
                    done = True
                    break
            test_accuracy = sess.run(
                model.accuracy, feed_dict={model.x: test_images, model.y_: test_labels})
            print("Test-Accuracy: %2.2f" % test_accuracy)
            print("Total Time: %3.2fs" % float(time.time() - begin_time))
            print("Final Cost: %.4f" % cost)
            if sink is not None:
                sink.write("epoch", epoch=epoch + 1, test_accuracy=float(test_accuracy),
                           total_secs=float(time.time() - begin_time), final_cost=float(cost))
            if done:
                break
        writer.close()
        if sink is not None:
            sink.close()
        if FLAGS.shutdown_ps and hasattr(server.runtime, "report_done"):
            server.runtime.report_done()      # the ps exits join() once every worker has
    sv.stop()
    server.stop()
    print("done")
    return 0


if __name__ == "__main__":
    define_flags()
    tf.app.run(main=main)
