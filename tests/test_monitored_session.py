"""``MonitoredTrainingSession``, its hooks and ``SyncReplicasOptimizer`` -- the hook-driven form
of the reference's Supervisor loop (``tf_distributed.py:91-128``)."""
import os
import threading

import numpy as np
import pytest

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel.comm import LoopbackComm
from distributed_tensorflow_amd.summary import read_events


def _batch(seed=0, n=50):
    rng = np.random.RandomState(seed)
    return (rng.rand(n, 784).astype(np.float32),
            np.eye(10, dtype=np.float32)[rng.randint(0, 10, n)])


def test_hooks_share_the_training_run_and_stop_it(tmp_path):
    bx, by = _batch()
    lines = []
    with tf.Graph().as_default():
        m = mnist_mlp.build_model(learning_rate=0.05, seed=1)
        logger = tf.train.LoggingTensorHook({"loss": m.loss, "step": m.global_step},
                                            every_n_iter=5, print_fn=lines.append)
        final = tf.train.FinalOpsHook(m.global_step)
        with tf.train.MonitoredTrainingSession(
                checkpoint_dir=str(tmp_path), config=tf.ConfigProto(device="cpu"),
                hooks=[tf.train.StopAtStepHook(last_step=23), logger, final,
                       tf.train.NanTensorHook(m.loss)],
                save_checkpoint_steps=10, save_checkpoint_secs=None, save_summaries_steps=7,
                log_step_count_steps=5) as sess:
            runs = 0
            while not sess.should_stop():
                out = sess.run([m.train_op, m.loss], {m.x: bx, m.y_: by})
                assert out[0] is None and np.isfinite(out[1])     # the caller's fetches only
                runs += 1
            raw = sess.raw_session()
            # every hook's fetches travelled with the caller's: one Session.run per step
            assert raw.stats["runs"] <= runs + 6, raw.stats
        assert runs == 23 and float(final.final_ops_values) == 23.0
        assert len(lines) == 5 and lines[0].startswith("loss = ")
    state = tf.train.get_checkpoint_state(str(tmp_path))
    assert state.model_checkpoint_path.endswith("model.ckpt-23")
    saved = sorted(int(p.rsplit("-", 1)[1]) for p in state.all_model_checkpoint_paths)
    assert saved == [0, 10, 20, 23]
    events = [e for f in sorted(os.listdir(tmp_path)) if "tfevents" in f
              for e in read_events(os.path.join(str(tmp_path), f))]
    tags = {t for e in events for t, _ in e.get("summary", [])}
    assert {"cost", "accuracy", "global_step/sec"} <= tags

    # a second session in the same directory resumes from the checkpoint
    with tf.Graph().as_default():
        m = mnist_mlp.build_model(learning_rate=0.05, seed=1)
        with tf.train.MonitoredTrainingSession(
                checkpoint_dir=str(tmp_path), config=tf.ConfigProto(device="cpu"),
                hooks=[tf.train.StopAtStepHook(num_steps=2)], save_checkpoint_secs=None,
                save_summaries_steps=None, log_step_count_steps=None) as sess:
            steps = []
            while not sess.should_stop():
                steps.append(float(sess.run([m.train_op, m.global_step],
                                            {m.x: bx, m.y_: by})[1]))
        assert steps == [24.0, 25.0]


def test_nan_hook_raises_and_no_final_checkpoint_is_written(tmp_path):
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [])
        gs = tf.train.get_or_create_global_step()
        bump = gs.assign_add(1.0)
        loss = tf.log(x)
        with pytest.raises(tf.train.NanLossDuringTrainingError):
            with tf.train.MonitoredTrainingSession(
                    checkpoint_dir=str(tmp_path), config=tf.ConfigProto(device="cpu"),
                    hooks=[tf.train.NanTensorHook(loss)], save_checkpoint_secs=1000,
                    save_summaries_steps=None, log_step_count_steps=None) as sess:
                sess.run([bump, loss], {x: 2.0})
                sess.run([bump, loss], {x: -1.0})
        state = tf.train.get_checkpoint_state(str(tmp_path))
        assert state.model_checkpoint_path.endswith("model.ckpt-0")     # only the initial one


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sync_job(cluster, task, comm, results, strategy, world):
    with tf.Graph().as_default():
        config = tf.ConfigProto(device="cpu", strategy=strategy)
        server = tf.train.Server(cluster, job_name="worker", task_index=task, protocol="rccl",
                                 config=config, comm=comm)
        with tf.device(tf.train.replica_device_setter(
                worker_device="/job:worker/task:%d" % task, cluster=cluster)):
            opt = tf.train.SyncReplicasOptimizer(tf.train.GradientDescentOptimizer(0.05),
                                                 replicas_to_aggregate=world)
            m = mnist_mlp.build_model(seed=1, optimizer=opt, with_summaries=False)
        bx, by = _batch(seed=task)
        with tf.train.MonitoredTrainingSession(
                master=server.target, is_chief=(task == 0), config=config,
                hooks=[opt.make_session_run_hook(task == 0),
                       tf.train.StopAtStepHook(last_step=6)]) as sess:
            while not sess.should_stop():
                sess.run(m.train_op, {m.x: bx, m.y_: by})
            raw = sess.raw_session()
            results[task] = {"w1": raw.run(m.w1), "step": float(raw.run(m.global_step)),
                             "aggregation": server.runtime.grad_aggregation}
        server.stop()


@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_sync_replicas_optimizer_averages_over_the_collective_world(strategy):
    world = 2
    cluster = tf.train.ClusterSpec({"ps": ["127.0.0.1:%d" % _free_port()],
                                    "worker": ["127.0.0.1:%d" % _free_port()
                                               for _ in range(world)]})
    comms = LoopbackComm.create_world(world, timeout_s=30)
    results = {}
    threads = [threading.Thread(target=_sync_job,
                                args=(cluster, t, comms[t], results, strategy, world))
               for t in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert set(results) == {0, 1}
    assert results[0]["aggregation"] == results[1]["aggregation"] == "mean"
    assert results[0]["step"] == results[1]["step"] == 6.0
    assert np.array_equal(results[0]["w1"], results[1]["w1"])

    # the mean of the two replicas' gradients = one worker on the concatenated batch
    with tf.Graph().as_default():
        m = mnist_mlp.build_model(learning_rate=0.05, seed=1, with_summaries=False)
        (x0, y0), (x1, y1) = _batch(seed=0), _batch(seed=1)
        bx, by = np.concatenate([x0, x1]), np.concatenate([y0, y1])
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(m.init_op)
            for _ in range(6):
                sess.run(m.train_op, {m.x: bx, m.y_: by})
            single = sess.run(m.w1)
    assert np.allclose(results[0]["w1"], single, rtol=1e-4, atol=1e-5)


def test_sync_replicas_optimizer_refuses_what_it_cannot_honour():
    with pytest.raises(ValueError, match="backup replicas"):
        tf.train.SyncReplicasOptimizer(tf.train.GradientDescentOptimizer(0.1), 2, 3)
    with tf.Graph().as_default():
        opt = tf.train.SyncReplicasOptimizer(tf.train.GradientDescentOptimizer(0.1), 2)
        m = mnist_mlp.build_model(seed=1, optimizer=opt, with_summaries=False)
        with pytest.raises(ValueError, match="in a job of 1"):
            with tf.train.MonitoredTrainingSession(
                    config=tf.ConfigProto(device="cpu"),
                    hooks=[opt.make_session_run_hook(True)]):
                pass
