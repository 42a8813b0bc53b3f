#!/usr/bin/env python3
"""The reference's MNIST job in the later ``tf.train`` idiom: ``tf.layers.dense`` layers,
integer labels, an optimizer with state wrapped in ``SyncReplicasOptimizer``, and a
``MonitoredTrainingSession`` whose hooks stop the job, write checkpoints and summaries and log
the loss -- instead of the ``Supervisor`` + hand-written loop of ``tf_distributed.py:91-128``.

    python -m distributed_tensorflow_amd.launch --protocol rccl examples/mnist_monitored.py
    python -m distributed_tensorflow_amd.launch --protocol rccl --num_workers 2 \\
        examples/mnist_monitored.py -- --device cpu --last_step 300

Cluster flags, launcher environment and data loading are those of ``tf_distributed.py``.
Synchronous training needs the collective strategies (``--protocol rccl``); with
``--nosync_replicas`` the script also runs on the asynchronous parameter server.
"""
from __future__ import print_function

import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402

import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.data import mnist as input_data  # noqa: E402

import tf_distributed as base  # noqa: E402  (flags, cluster construction, optimizer factory)

batch_size = 100


def main(argv=None):
    FLAGS = tf.app.flags.FLAGS
    cluster = base.build_cluster(FLAGS, default_num_workers=1)
    config = tf.ConfigProto(device=FLAGS.device or None, strategy=FLAGS.strategy,
                            capture_collectives=True if FLAGS.capture_collectives else None)
    server = tf.train.Server(cluster, job_name=FLAGS.job_name, task_index=FLAGS.task_index,
                             protocol=FLAGS.protocol, config=config)
    if FLAGS.job_name == "ps":
        server.join()
        return 0
    mnist = input_data.read_data_sets(FLAGS.data_dir, one_hot=False)
    is_chief = FLAGS.task_index == 0
    num_workers = cluster.num_tasks("worker")

    with tf.device(tf.train.replica_device_setter(
            worker_device="/job:worker/task:%d" % FLAGS.task_index, cluster=cluster)):
        global_step = tf.train.get_or_create_global_step()
        x = tf.placeholder(tf.float32, [None, 784], name="x")
        y = tf.placeholder(tf.int64, [None], name="labels")
        tf.set_random_seed(1)
        hidden = tf.layers.dense(x, FLAGS.hidden_units, activation=tf.nn.sigmoid, name="hidden")
        logits = tf.layers.dense(hidden, 10, name="logits")
        loss = tf.reduce_mean(tf.nn.sparse_softmax_cross_entropy_with_logits(
            labels=y, logits=logits))
        correct = tf.equal(tf.argmax(logits, 1), y)
        accuracy = tf.reduce_mean(tf.cast(correct, tf.float32))
        tf.summary.scalar("cost", loss)
        tf.summary.scalar("accuracy", accuracy)
        opt = base.make_optimizer(FLAGS.optimizer, FLAGS.learning_rate) or \
            tf.train.GradientDescentOptimizer(FLAGS.learning_rate)
        hooks = [tf.train.StopAtStepHook(last_step=FLAGS.last_step),
                 tf.train.NanTensorHook(loss)]
        if FLAGS.sync_replicas:
            opt = tf.train.SyncReplicasOptimizer(opt, replicas_to_aggregate=num_workers,
                                                 total_num_replicas=num_workers)
            hooks.append(opt.make_session_run_hook(is_chief))
        train_op = opt.minimize(loss, global_step=global_step)
    logger = tf.train.LoggingTensorHook(
        {"step": global_step, "cost": loss}, every_n_iter=100,
        formatter=lambda v: "Step: %d,  Cost: %.4f" % (int(v["step"]), float(v["cost"])))

    test_x, test_y = mnist.test.images, mnist.test.labels.astype(np.int64)
    if FLAGS.device_data and server.device.type == "cuda":
        # the data set lives in HBM: a batch is a view of it, nothing is uploaded per step
        mnist.train.to_device(server.device)
        mnist.test.to_device(server.device)
        test_x, test_y = mnist.test.device_images, mnist.test.device_labels
    begin = time.time()
    steps = 0
    with tf.train.MonitoredTrainingSession(
            master=server.target, is_chief=is_chief, checkpoint_dir=FLAGS.checkpoint_dir or None,
            hooks=hooks + [logger], config=config, save_checkpoint_secs=None,
            save_checkpoint_steps=FLAGS.save_checkpoint_steps, save_summaries_steps=100,
            log_step_count_steps=100) as sess:
        while not sess.should_stop():
            bx, by = mnist.train.next_batch(batch_size)
            sess.run(train_op, feed_dict={x: bx, y: by})
            steps += 1
        raw = sess.raw_session()
        acc, step = raw.run([accuracy, global_step], feed_dict={x: test_x, y: test_y})
        elapsed = time.time() - begin
        print("Test-Accuracy: %2.2f" % acc)
        print("Global Step: %d  (local steps: %d)" % (int(step), steps))
        print("AvgTime: %3.2fms" % (elapsed * 1e3 / max(1, steps)))
        print("Total Time: %3.2fs" % elapsed)
    server.stop()
    print("done")
    return 0


if __name__ == "__main__":
    base.define_flags(default_job="worker", default_epochs=1)
    tf.app.flags.DEFINE_integer("last_step", 1000, "stop when global_step reaches this value")
    tf.app.flags.DEFINE_integer("hidden_units", 100, "width of the hidden layer")
    tf.app.flags.DEFINE_float("learning_rate", 0.005, "learning rate")
    tf.app.flags.DEFINE_boolean("sync_replicas", True, "wrap the optimizer in "
                                "SyncReplicasOptimizer (mean of the workers' gradients)")
    tf.app.flags.DEFINE_integer("save_checkpoint_steps", 500, "checkpoint interval in steps "
                                "(needs --checkpoint_dir)")
    tf.app.run(main=main)
