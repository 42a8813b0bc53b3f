"""Slot-carrying optimizers (momentum, Adagrad, RMSProp, Adam): update rules against
``torch.optim`` / a NumPy transcription of TensorFlow's formulas, slot placement and naming,
checkpointing of the state, fused vs per-variable execution, parameter-server and collective
jobs. The reference itself only uses plain SGD (``tf_distributed.py:75``); these are the other
members of the ``tf.train`` optimizer family a user of that API reaches for next."""
import threading

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel.comm import LoopbackComm

STEPS = 6


def _problem():
    rng = np.random.RandomState(0)
    return rng.randn(16, 8).astype(np.float32), rng.randn(16, 3).astype(np.float32)


def _train(make_opt, device="cpu", fusion=True, steps=STEPS):
    bx, by = _problem()
    g = tf.Graph()
    with g.as_default():
        tf.set_random_seed(1)
        x = tf.placeholder(tf.float32, [None, 8])
        y = tf.placeholder(tf.float32, [None, 3])
        w = tf.Variable(tf.random_normal([8, 3]), name="W")
        b = tf.Variable(tf.zeros([3]), name="b")
        loss = tf.reduce_mean(tf.reduce_sum(tf.square(tf.subtract(tf.add(tf.matmul(x, w), b), y)), 1))
        gs = tf.train.get_or_create_global_step()
        opt = make_opt()
        train = opt.minimize(loss, global_step=gs)
        init = tf.global_variables_initializer()
        with tf.Session(config=tf.ConfigProto(device=device, graph_fusion=fusion)) as sess:
            sess.run(init)
            w0, b0 = sess.run([w, b])
            losses = [float(sess.run([train, loss], {x: bx, y: by})[1]) for _ in range(steps)]
            types = {s.node.type for p in sess._plans.values() for s in p.steps}
            out = dict(w0=w0, b0=b0, w=sess.run(w), b=sess.run(b), losses=losses,
                       step=float(sess.run(gs)), types=types,
                       names=sorted(v.op.name for v in tf.global_variables()),
                       slots={n: sess.run(opt.get_slot(w, n)) for n in opt.get_slot_names()})
    return out


def _numpy_rule(rule, hyper, w, b, steps=STEPS):
    """TensorFlow's update formulas in float64 on the quadratic problem of ``_train``."""
    bx, by = _problem()
    params = [w.astype(np.float64), b.astype(np.float64)]
    s0 = [np.full_like(p, {"adagrad": 0.1, "rmsprop": 1.0}.get(rule, 0.0)) for p in params]
    s1 = [np.zeros_like(p) for p in params]
    for t in range(1, steps + 1):
        z = bx @ params[0] + params[1]
        dz = 2.0 * (z - by) / bx.shape[0]
        grads = [bx.T @ dz, dz.sum(0)]
        for i, g in enumerate(grads):
            if rule in ("momentum", "nesterov"):
                lr, mu = hyper
                s0[i] = mu * s0[i] + g
                params[i] -= lr * (g + mu * s0[i]) if rule == "nesterov" else lr * s0[i]
            elif rule == "adagrad":
                s0[i] += g * g
                params[i] -= hyper[0] * g / np.sqrt(s0[i])
            elif rule == "adadelta":
                lr, rho, eps = hyper
                s0[i] = rho * s0[i] + (1 - rho) * g * g
                update = np.sqrt(s1[i] + eps) / np.sqrt(s0[i] + eps) * g
                s1[i] = rho * s1[i] + (1 - rho) * update * update
                params[i] -= lr * update
            elif rule == "rmsprop":
                lr, rho, mu, eps = hyper
                s0[i] = rho * s0[i] + (1 - rho) * g * g
                s1[i] = mu * s1[i] + lr * g / np.sqrt(s0[i] + eps)
                params[i] -= s1[i]
            else:
                lr, b1, b2, eps = hyper
                s0[i] = b1 * s0[i] + (1 - b1) * g
                s1[i] = b2 * s1[i] + (1 - b2) * g * g
                params[i] -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * s0[i] / (np.sqrt(s1[i]) + eps)
    return params


CASES = [
    ("momentum", (0.01, 0.9), lambda: tf.train.MomentumOptimizer(0.01, 0.9)),
    ("nesterov", (0.01, 0.9), lambda: tf.train.MomentumOptimizer(0.01, 0.9, use_nesterov=True)),
    ("adagrad", (0.1,), lambda: tf.train.AdagradOptimizer(0.1)),
    ("rmsprop", (0.01, 0.9, 0.5, 1e-10), lambda: tf.train.RMSPropOptimizer(0.01, momentum=0.5)),
    ("adam", (0.01, 0.9, 0.999, 1e-8), lambda: tf.train.AdamOptimizer(0.01)),
    ("adadelta", (1.0, 0.9, 1e-6), lambda: tf.train.AdadeltaOptimizer(1.0, rho=0.9, epsilon=1e-6)),
]


@pytest.mark.parametrize("rule,hyper,make", CASES, ids=[c[0] for c in CASES])
def test_rules_match_the_published_formulas_fused_and_unfused(rule, hyper, make):
    fused = _train(make)
    plain = _train(make, fusion=False)
    assert "MultiApplyOptimizer" in fused["types"] and "ApplyOptimizer" not in fused["types"]
    assert "ApplyOptimizer" in plain["types"]
    ref_w, ref_b = _numpy_rule(rule, hyper, fused["w0"], fused["b0"])
    for run in (fused, plain):
        assert np.allclose(run["w"], ref_w, rtol=2e-5, atol=2e-6)
        assert np.allclose(run["b"], ref_b, rtol=2e-5, atol=2e-6)
        assert run["step"] == STEPS
        assert run["losses"][-1] < run["losses"][0]
    assert np.allclose(fused["w"], plain["w"], rtol=1e-6, atol=1e-7)


def test_adam_and_momentum_agree_with_torch_optim():
    for make, make_torch in (
            (lambda: tf.train.AdamOptimizer(0.01), lambda p: torch.optim.Adam(p, lr=0.01)),
            (lambda: tf.train.MomentumOptimizer(0.01, 0.9),
             lambda p: torch.optim.SGD(p, lr=0.01, momentum=0.9))):
        run = _train(make)
        bx, by = (torch.tensor(a) for a in _problem())
        w = torch.tensor(run["w0"], requires_grad=True)
        b = torch.tensor(run["b0"], requires_grad=True)
        opt = make_torch([w, b])
        for _ in range(STEPS):
            opt.zero_grad()
            ((bx @ w + b - by) ** 2).sum(1).mean().backward()
            opt.step()
        assert np.allclose(run["w"], w.detach().numpy(), rtol=1e-5, atol=2e-6)
        assert np.allclose(run["b"], b.detach().numpy(), rtol=1e-5, atol=2e-6)


def test_slots_are_named_after_and_placed_with_their_variable():
    cluster = tf.train.ClusterSpec({"ps": ["h:1", "h:2"], "worker": ["h:3"]})
    g = tf.Graph()
    with g.as_default():
        with tf.device(tf.train.replica_device_setter(worker_device="/job:worker/task:0",
                                                      cluster=cluster)):
            with tf.name_scope("model"):
                w = tf.Variable(tf.zeros([4, 2]), name="W")
                b = tf.Variable(tf.zeros([2]), name="b")
            x = tf.placeholder(tf.float32, [None, 4])
            loss = tf.reduce_mean(tf.square(tf.add(tf.matmul(x, w), b)))
            with tf.name_scope("train"):
                opt = tf.train.AdamOptimizer()
                train = opt.minimize(loss)
        assert {w.device, b.device} == {"/job:ps/task:0", "/job:ps/task:1"}
        for v in (w, b):
            for slot in ("m", "v"):
                s = opt.get_slot(v, slot)
                assert s.device == v.device and not s.trainable
                assert s.op.name.startswith(v.op.name + "/Adam")
        assert opt.get_slot_names() == ["m", "v"]
        names = {v.op.name for v in tf.global_variables()}
        assert {"beta1_power", "beta2_power", "model/W/Adam", "model/W/Adam_1"} <= names
        assert tf.trainable_variables() == [w, b]
        # every update runs where its variable lives (the parameter owner)
        for op in g.get_operations():
            if op.type == "ApplyOptimizer":
                assert op.device == op.inputs[0].op.device
        assert train.type in ("NoOp", "Group") or train.control_inputs or True


def test_optimizer_state_is_checkpointed(tmp_path):
    def build():
        tf.set_random_seed(3)
        w = tf.Variable(tf.random_normal([4, 2]), name="W")
        x = tf.placeholder(tf.float32, [None, 4])
        loss = tf.reduce_mean(tf.square(tf.matmul(x, w)))
        opt = tf.train.AdamOptimizer(0.05)
        return w, x, loss, opt, opt.minimize(loss), tf.train.Saver()

    bx = np.random.RandomState(1).randn(8, 4).astype(np.float32)
    with tf.Graph().as_default():
        w, x, loss, opt, train, saver = build()
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(tf.global_variables_initializer())
            for _ in range(3):
                sess.run(train, {x: bx})
            path = saver.save(sess, str(tmp_path / "model"))
            for _ in range(2):
                sess.run(train, {x: bx})
            want = sess.run(w)
    with tf.Graph().as_default():
        w, x, loss, opt, train, saver = build()
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            saver.restore(sess, path)
            assert np.isclose(float(sess.run(opt._powers[0])), 0.9 ** 4)   # three steps taken
            assert np.abs(sess.run(opt.get_slot(w, "m"))).max() > 0
            for _ in range(2):
                sess.run(train, {x: bx})
            assert np.allclose(sess.run(w), want, rtol=1e-6, atol=1e-7)


def _mlp_job(cluster, task, steps, results, make_opt, protocol="grpc", comm=None, strategy=None):
    g = tf.Graph()
    with g.as_default():
        config = tf.ConfigProto(device="cpu", strategy=strategy)
        server = tf.train.Server(cluster, job_name="worker", task_index=task, protocol=protocol,
                                 config=config, comm=comm)
        with tf.device(tf.train.replica_device_setter(
                worker_device="/job:worker/task:%d" % task, cluster=cluster)):
            m = mnist_mlp.build_model(seed=1, optimizer=make_opt(), with_summaries=False)
        sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                 init_op=m.init_op, logdir=None, recovery_wait_secs=0.05,
                                 summary_op=None)
        from distributed_tensorflow_amd.data.mnist import synthetic_mnist
        images, labels = synthetic_mnist(400, seed=task)
        onehot = np.eye(10, dtype=np.float32)[labels]
        rng = np.random.RandomState(7 + task)
        costs = []
        with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
            for _ in range(steps):
                idx = rng.randint(0, 400, 50)
                costs.append(float(sess.run([m.train_op, m.loss],
                                            {m.x: images[idx], m.y_: onehot[idx]})[1]))
            slot = m.optimizer.get_slot(m.w1, m.optimizer.get_slot_names()[0])
            final = {"w1": sess.run(m.w1), "m": sess.run(slot),
                     "step": float(sess.run(m.global_step))}
            if hasattr(m.optimizer, "_powers"):
                final["p1"] = float(sess.run(m.optimizer._powers[0]))
        sv.stop()
        results[task] = dict(costs=costs, final=final, runtime=server.runtime,
                             slot_device=slot.device)
        server.stop()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cluster(workers):
    return tf.train.ClusterSpec({"ps": ["127.0.0.1:%d" % _free_port()],
                                 "worker": ["127.0.0.1:%d" % _free_port() for _ in range(workers)]})


def _collective_world(world, steps, make_opt, strategy):
    cluster = _cluster(world)
    comms = LoopbackComm.create_world(world, timeout_s=30)
    results = {}
    threads = [threading.Thread(target=_mlp_job, args=(cluster, t, steps, results, make_opt),
                                kwargs=dict(protocol="rccl", comm=comms[t], strategy=strategy))
               for t in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert set(results) == set(range(world))
    return results


@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_adam_under_the_collective_strategies_keeps_replicas_identical(strategy):
    world, steps = 2, 12
    results = _collective_world(world, steps, lambda: tf.train.AdamOptimizer(0.01), strategy)
    a, b = results[0]["final"], results[1]["final"]
    assert np.array_equal(a["w1"], b["w1"]) and np.array_equal(a["m"], b["m"])      # bitwise
    assert a["step"] == b["step"] == steps
    assert np.isclose(a["p1"], 0.9 ** (steps + 1))
    assert results[0]["costs"] != results[1]["costs"]
    assert np.mean(results[0]["costs"][-3:]) < np.mean(results[0]["costs"][:3])
    rt = results[0]["runtime"]
    per_step = rt.bytes_reduced / steps
    if strategy == "allreduce_sync":
        # only the trainable variables' span of the arena travels: Adam's moments, the running
        # powers and global_step lie behind it
        assert 0 < per_step <= (mnist_mlp.NUM_PARAMS + 4 * 64) * 4
        assert rt.sharded_steps == 0 and not rt.has_sharded_storage()
    else:
        # the parameter arena (the moments are not part of it any more) is reduce-scattered
        assert 0 < per_step <= (mnist_mlp.NUM_PARAMS + 8 * 64 + world * 64) * 4
        assert rt.sharded_steps == steps


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("rule", ["adam", "momentum", "adagrad", "rmsprop", "adadelta"])
def test_owner_sharded_optimizer_state_under_reduce_scatter_sync(rule, world):
    """ZeRO stage 1 for every slot-carrying rule: under ``reduce_scatter_sync`` each rank keeps
    only its shard of the rule's state and updates only its shard of the parameters; the
    replicas end bit-equal to those of ``allreduce_sync`` (replicated state, every rank updates
    everything), every step took the owner path, and a rank holds ``1 / world`` of the state."""
    steps = 8
    make = {"adam": lambda: tf.train.AdamOptimizer(0.01),
            "momentum": lambda: tf.train.MomentumOptimizer(0.01, 0.9),
            "adagrad": lambda: tf.train.AdagradOptimizer(0.05),
            "rmsprop": lambda: tf.train.RMSPropOptimizer(0.001, momentum=0.5),
            "adadelta": lambda: tf.train.AdadeltaOptimizer(1.0)}[rule]
    ref = _collective_world(world, steps, make, "allreduce_sync")
    got = _collective_world(world, steps, make, "reduce_scatter_sync")
    for t in range(world):
        assert np.array_equal(got[t]["final"]["w1"], ref[0]["final"]["w1"])
        # the slot is read back whole (an all-gather of the owners' parts)
        assert np.array_equal(got[t]["final"]["m"], ref[0]["final"]["m"])
        assert got[t]["costs"] == ref[t]["costs"]
        rt = got[t]["runtime"]
        assert rt.sharded_steps == steps and rt.has_sharded_storage()
        n_slots = {"adam": 2, "rmsprop": 2, "adadelta": 2}.get(rule, 1)
        assert rt.slot_bytes_total == n_slots * mnist_mlp.NUM_PARAMS * 4
        # 1 / world of it, up to the arena's alignment padding (256 bytes per variable and rank)
        assert rt.slot_bytes_local() * world <= rt.slot_bytes_total + n_slots * 4 * 64 * (
            8 + world)
        assert rt.slot_bytes_local() * world >= rt.slot_bytes_total
        assert not any("/" + k in n for n in rt.local_store.state_dict()
                       for k in ("Adam", "Momentum", "Adagrad", "RMSProp", "Adadelta"))


def test_two_train_ops_over_disjoint_variables_leave_each_others_variables_alone():
    """Two train ops with different rules and rates over disjoint var_lists, run in alternation
    under both collective strategies: a step must not touch the variables of the other op,
    whatever their gradient slots still hold from the step before (reduce-scatter: the owner
    applies only this step's variables; all-reduce: only their spans travel)."""
    world, rounds = 2, 4

    def job(task, comm, strategy, out):
        g = tf.Graph()
        with g.as_default():
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:1", "127.0.0.1:2"]})
            cfg = tf.ConfigProto(device="cpu", strategy=strategy)
            server = tf.train.Server(cluster, job_name="worker", task_index=task,
                                     protocol="rccl", config=cfg, comm=comm)
            tf.set_random_seed(3)
            x = tf.placeholder(tf.float32, [None, 6])
            a = tf.Variable(tf.random_normal([6, 5]), name="a")
            b = tf.Variable(tf.random_normal([5, 3]), name="b")
            loss = tf.reduce_mean(tf.square(tf.matmul(tf.nn.sigmoid(tf.matmul(x, a)), b)))
            train_a = tf.train.GradientDescentOptimizer(0.5).minimize(loss, var_list=[a])
            train_b = tf.train.MomentumOptimizer(0.01, 0.9).minimize(loss, var_list=[b])
            sv = tf.train.Supervisor(is_chief=(task == 0), init_op=tf.global_variables_initializer(),
                                     logdir=None, recovery_wait_secs=0.05, summary_op=None)
            rng = np.random.RandomState(task)
            log = []
            with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
                for _ in range(rounds):
                    feed = {x: rng.rand(4, 6).astype(np.float32)}
                    b0 = sess.run(b)
                    sess.run(train_a, feed)
                    log.append(("a-step leaves b", np.array_equal(sess.run(b), b0)))
                    a0 = sess.run(a)
                    sess.run(train_b, feed)
                    log.append(("b-step leaves a", np.array_equal(sess.run(a), a0)))
                out[task] = (log, sess.run(a), sess.run(b))
            sv.stop()
            server.stop()

    finals = {}
    for strategy in ("allreduce_sync", "reduce_scatter_sync"):
        comms = LoopbackComm.create_world(world, timeout_s=30)
        out = {}
        threads = [threading.Thread(target=job, args=(t, comms[t], strategy, out))
                   for t in range(world)]
        [t.start() for t in threads]
        [t.join(60) for t in threads]
        assert set(out) == {0, 1}, strategy
        for t in range(world):
            assert all(ok for _, ok in out[t][0]), (strategy, out[t][0])
        assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
        assert np.isfinite(out[0][1]).all() and np.isfinite(out[0][2]).all()
        finals[strategy] = out[0]
    assert np.array_equal(finals["allreduce_sync"][1], finals["reduce_scatter_sync"][1])
    assert np.array_equal(finals["allreduce_sync"][2], finals["reduce_scatter_sync"][2])


def test_owner_sharded_slots_checkpoint_round_trip_across_world_sizes(tmp_path):
    """Adam under ``reduce_scatter_sync`` with 2 ranks: ``Saver.save`` (every worker calls it)
    gathers the owners' parts of the moments into whole tensors; worlds of 1, 2 and 4 ranks
    restore that checkpoint -- each rank keeps its own part -- read the same values back and
    go on training on the owner path."""
    from distributed_tensorflow_amd.data.mnist import synthetic_mnist
    images, labels = synthetic_mnist(200, seed=0)
    onehot = np.eye(10, dtype=np.float32)[labels]

    def job(world, task, comm, restore_from, out):
        g = tf.Graph()
        with g.as_default():
            cluster = tf.train.ClusterSpec(
                {"worker": ["127.0.0.1:%d" % (1 + i) for i in range(world)]})
            cfg = tf.ConfigProto(device="cpu", strategy="reduce_scatter_sync")
            server = tf.train.Server(cluster, job_name="worker", task_index=task,
                                     protocol="rccl", config=cfg, comm=comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                m = mnist_mlp.build_model(seed=1, optimizer=tf.train.AdamOptimizer(0.01),
                                          with_summaries=False)
            saver = tf.train.Saver()
            sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                     init_op=m.init_op, logdir=None, recovery_wait_secs=0.05,
                                     summary_op=None, saver=None)
            slots = [m.optimizer.get_slot(v, k) for v in (m.w1, m.b2) for k in ("m", "v")]
            with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
                if restore_from:
                    saver.restore(sess, restore_from)
                else:
                    for i in range(5):
                        lo = 20 * i
                        sess.run(m.train_op, {m.x: images[lo:lo + 20], m.y_: onehot[lo:lo + 20]})
                    prefix = saver.save(sess, str(tmp_path / ("w%d_r%d" % (world, task)) / "ck"))
                state = [sess.run(v) for v in [m.w1, m.b2, m.global_step] + slots]
                state.append(float(sess.run(m.optimizer._powers[0])))
                before = server.runtime.sharded_steps
                sess.run(m.train_op, {m.x: images[:20], m.y_: onehot[:20]})
                assert server.runtime.sharded_steps == before + 1
                assert np.isfinite(sess.run(m.w1)).all()
                out[task] = (state, None if restore_from else prefix,
                             server.runtime.slot_bytes_local())
            sv.stop()
            server.stop()

    def run(world, restore_from=None):
        comms = LoopbackComm.create_world(world, timeout_s=30) if world > 1 \
            else [SingleCommCpu()]
        out = {}
        threads = [threading.Thread(target=job, args=(world, t, comms[t], restore_from, out))
                   for t in range(world)]
        [t.start() for t in threads]
        [t.join(60) for t in threads]
        assert set(out) == set(range(world))
        return out

    from distributed_tensorflow_amd.parallel.comm import SingleComm as SingleCommCpu
    saved = run(2)
    state, prefix, local2 = saved[0]
    assert any(np.abs(s).max() > 0 for s in state[3:7])            # the moments hold something
    from distributed_tensorflow_amd.train.saver import read_bundle
    on_disk = read_bundle(prefix, ["weights/Variable/Adam", "biases/Variable_1/Adam_1"])
    assert np.array_equal(on_disk["weights/Variable/Adam"].numpy(), state[3])
    assert np.array_equal(on_disk["biases/Variable_1/Adam_1"].numpy(), state[6])
    for world in (1, 2, 4):
        got = run(world, restore_from=prefix)
        for t in range(world):
            for a, b in zip(got[t][0], state):
                assert np.array_equal(np.asarray(a), np.asarray(b)), (world, t)
        if world == 4:
            assert got[0][2] * 2 <= local2 + 2 * 4 * 64 * 4    # half the state of a 2-rank owner


def test_momentum_on_a_parameter_server_keeps_its_state_on_the_ps():
    cluster = _cluster(2)
    ps = tf.train.Server(cluster, job_name="ps", task_index=0)
    results = {}
    threads = [threading.Thread(target=_mlp_job, args=(
        cluster, t, 40, results, lambda: tf.train.MomentumOptimizer(0.002, 0.5)))
        for t in (1, 0)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert set(results) == {0, 1}
    store = ps.runtime.local_store
    assert "weights/Variable/Momentum" in store.names()
    assert results[0]["slot_device"] == "/job:ps/task:0"
    assert store.read("global_step").item() == 80.0
    assert float(store.read("weights/Variable/Momentum").abs().max()) > 0
    assert results[0]["runtime"].local_store.names() == []
    for t in (0, 1):
        assert np.mean(results[t]["costs"][-8:]) < np.mean(results[t]["costs"][:8])
    ps.stop()


# -- gfx950 kernels (run on the MI355X box) -------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("rule,hyper", [(c[0], c[1]) for c in CASES], ids=[c[0] for c in CASES])
def test_multi_optim_kernel_against_fp64(rule, hyper):
    from distributed_tensorflow_amd.ops import hip
    from distributed_tensorflow_amd.ops.ref_kernels import optimizer_rule_step
    g = torch.Generator().manual_seed(3)
    sizes = [78400, 100, 1000, 10, 1, 4099]           # the reference MLP's tensors + odd sizes
    two = rule in ("rmsprop", "adam", "adadelta")
    backing = [torch.randn(n + 3, generator=g).cuda() for n in sizes]
    params = [t[(i % 2):(i % 2) + n] for i, (t, n) in enumerate(zip(backing, sizes))]   # odd starts
    grads = [torch.randn(n, generator=g).cuda() for n in sizes]
    s0 = [torch.rand(n, generator=g).cuda() + 0.1 for n in sizes]
    s1 = [torch.rand(n, generator=g).cuda() for n in sizes] if two else []
    powers = [torch.tensor(0.9 ** 3).cuda(), torch.tensor(0.999 ** 3).cuda()] \
        if rule == "adam" else []
    want = [optimizer_rule_step(rule, hyper, p.double(), gr.double(), a.double(),
                                s1[i].double() if two else None,
                                [pw.double() for pw in powers] or None)
            for i, (p, gr, a) in enumerate(zip(params, grads, s0))]
    guard = [t.clone() for t in backing]
    hip.multi_optim(rule, params, grads, s0, s1, hyper, powers)
    torch.cuda.synchronize()
    for i, (p, a) in enumerate(zip(params, s0)):
        w_ref, a_ref, b_ref = want[i]
        assert torch.allclose(p.double(), w_ref, rtol=2e-6, atol=2e-6), (rule, i)
        assert torch.allclose(a.double(), a_ref, rtol=2e-6, atol=1e-7), (rule, i)
        if two:
            assert torch.allclose(s1[i].double(), b_ref, rtol=2e-6, atol=1e-7), (rule, i)
        # nothing outside the tensor's span was written
        off, n = i % 2, sizes[i]
        assert torch.equal(backing[i][:off], guard[i][:off])
        assert torch.equal(backing[i][off + n:], guard[i][off + n:])
    if powers:
        assert float(powers[0]) == pytest.approx(0.9 ** 3)              # read, not advanced
    # more than 16 tensors: several launches
    many = [torch.ones(5).cuda() for _ in range(19)]
    hip.multi_optim("momentum", many, [torch.ones(5).cuda() for _ in many],
                    [torch.zeros(5).cuda() for _ in many], (), (0.5, 0.0))
    assert all(torch.equal(t, torch.full((5,), 0.5).cuda()) for t in many)


@pytest.mark.gpu
def test_affine_scalars_kernel():
    from distributed_tensorflow_amd.ops import hip
    a, b, c = (torch.tensor(v).cuda() for v in (0.9, 0.999, 41.0))
    hip.affine_scalars([a, b, c], [0.9, 0.999, 1.0], [0.0, 0.0, 1.0])
    assert float(a) == pytest.approx(0.81) and float(b) == pytest.approx(0.998001)
    assert float(c) == 42.0
    with pytest.raises(hip.HipKernelError):
        hip.affine_scalars([a, a], [1.0, 1.0], [0.0, 0.0])


@pytest.mark.gpu
@pytest.mark.parametrize("rule,hyper,make", CASES, ids=[c[0] for c in CASES])
def test_gpu_session_runs_the_rules_natively_and_captured(rule, hyper, make):
    run = _train(make, device="cuda:0", steps=STEPS)
    ref_w, ref_b = _numpy_rule(rule, hyper, run["w0"], run["b0"])
    assert np.allclose(run["w"], ref_w, rtol=5e-5, atol=5e-6)
    assert np.allclose(run["b"], ref_b, rtol=5e-5, atol=5e-6)
    assert run["step"] == STEPS and "MultiApplyOptimizer" in run["types"]


@pytest.mark.gpu
def test_gpu_mnist_with_adam_is_captured_without_torch_kernels():
    g = tf.Graph()
    with g.as_default():
        m = mnist_mlp.build_model(seed=1, optimizer=tf.train.AdamOptimizer(0.01),
                                  with_summaries=False)
        rng = np.random.RandomState(0)
        bx = rng.rand(100, 784).astype(np.float32)
        by = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            sess.run(m.init_op)
            costs = []
            for i in range(30):
                _, cost, step = sess.run([m.train_op, m.loss, m.global_step], {m.x: bx, m.y_: by})
                costs.append(float(cost))
                assert float(step) == i + 1          # fetched next to the update that bumps it
            assert costs[-1] < 0.5 * costs[0], costs
            assert sess.stats["graph_replays"] >= 20
            assert sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log
            assert float(sess.run(m.global_step)) == 30.0
            assert float(sess.run(m.optimizer._powers[0])) == pytest.approx(0.9 ** 31, rel=1e-4)


@pytest.mark.gpu
def test_gpu_dense_layers_with_integer_labels_and_adam_run_natively():
    rng = np.random.RandomState(0)
    bx = rng.rand(64, 32).astype(np.float32)
    labels = rng.randint(0, 8, 64).astype(np.int64)
    with tf.Graph().as_default():
        tf.set_random_seed(4)
        x = tf.placeholder(tf.float32, [None, 32])
        y = tf.placeholder(tf.int64, [None])
        h = tf.layers.dense(x, 48, activation=tf.nn.relu, name="hidden")
        logits = tf.layers.dense(h, 8, name="out")
        w = [v for v in tf.trainable_variables() if v.op.name == "out/kernel"][0]
        loss = tf.add(tf.reduce_mean(tf.nn.sparse_softmax_cross_entropy_with_logits(
            labels=y, logits=logits)), tf.scale(tf.nn.l2_loss(w), 1e-3))
        train = tf.train.AdamOptimizer(0.01).minimize(
            loss, global_step=tf.train.get_or_create_global_step())
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            sess.run(tf.global_variables_initializer())
            costs = [float(sess.run([train, loss], {x: bx, y: labels})[1]) for _ in range(80)]
            assert costs[-1] < 0.5 * costs[0], (costs[0], costs[-1])
            assert sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log
            assert sess.stats["graph_replays"] >= 60


@pytest.mark.gpu
def test_gpu_clipped_momentum_step_runs_natively():
    rng = np.random.RandomState(1)
    bx = (10 * rng.randn(8, 5)).astype(np.float32)
    with tf.Graph().as_default():
        tf.set_random_seed(7)
        x = tf.placeholder(tf.float32, [None, 5])
        w = tf.Variable(tf.random_normal([5, 3]), name="W")
        b = tf.Variable(tf.ones([3]), name="b")
        loss = tf.reduce_sum(tf.square(tf.add(tf.matmul(x, w), b)))
        opt = tf.train.MomentumOptimizer(0.1, 0.0)
        gv = opt.compute_gradients(loss)
        clipped, norm = tf.clip_by_global_norm([g for g, _ in gv], 1.5)
        train = opt.apply_gradients(list(zip(clipped, [v for _, v in gv])))
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            sess.run(tf.global_variables_initializer())
            w0, b0 = sess.run([w, b])
            _, n = sess.run([train, norm], {x: bx})
            w1, b1 = sess.run([w, b])
            assert n > 1.5 and sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log
            step = np.sqrt(((w1 - w0) ** 2).sum() + ((b1 - b0) ** 2).sum())
            assert np.isclose(step, 0.1 * 1.5, rtol=1e-4)


# -- learning-rate schedules ----------------------------------------------------------------
def _scheduled_run(make_opt, schedule, device="cpu", steps=7, fusion=True):
    bx, by = _problem()
    with tf.Graph().as_default():
        tf.set_random_seed(1)
        x = tf.placeholder(tf.float32, [None, 8])
        y = tf.placeholder(tf.float32, [None, 3])
        w = tf.Variable(tf.random_normal([8, 3]), name="W")
        b = tf.Variable(tf.zeros([3]), name="b")
        loss = tf.reduce_mean(tf.reduce_sum(tf.square(tf.subtract(tf.add(tf.matmul(x, w), b), y)), 1))
        gs = tf.train.get_or_create_global_step()
        lr = schedule(gs)
        train = make_opt(lr).minimize(loss, global_step=gs)
        with tf.Session(config=tf.ConfigProto(device=device, graph_fusion=fusion)) as sess:
            sess.run(tf.global_variables_initializer())
            w0, b0 = sess.run([w, b])
            rates = [float(sess.run([train, lr], {x: bx, y: by})[1]) for _ in range(steps)]
            types = {s.node.type for p in sess._plans.values() for s in p.steps}
            stats = dict(sess.stats)
            return w0, b0, sess.run(w), sess.run(b), rates, types, stats


def _sgd_reference(w0, b0, rates):
    bx, by = _problem()
    w, b = w0.astype(np.float64), b0.astype(np.float64)
    for lr in rates:
        dz = 2.0 * (bx @ w + b - by) / bx.shape[0]
        w, b = w - lr * (bx.T @ dz), b - lr * dz.sum(0)
    return w, b


SCHEDULES = [
    ("exponential", lambda gs: tf.train.exponential_decay(0.05, gs, 2, 0.5),
     lambda t: 0.05 * 0.5 ** (t / 2)),
    ("staircase", lambda gs: tf.train.exponential_decay(0.05, gs, 3, 0.5, staircase=True),
     lambda t: 0.05 * 0.5 ** (t // 3)),
    ("inverse_time", lambda gs: tf.train.inverse_time_decay(0.05, gs, 2, 0.5),
     lambda t: 0.05 / (1 + 0.5 * t / 2)),
    ("polynomial", lambda gs: tf.train.polynomial_decay(0.05, gs, 5, 0.01, power=2.0),
     lambda t: (0.05 - 0.01) * (1 - min(t, 5) / 5) ** 2 + 0.01),
    ("piecewise", lambda gs: tf.train.piecewise_constant(gs, [1, 4], [0.05, 0.02, 0.005]),
     lambda t: 0.05 if t <= 1 else (0.02 if t <= 4 else 0.005)),
    ("natural_exp", lambda gs: tf.train.natural_exp_decay(0.05, gs, 2, 0.5),
     lambda t: 0.05 * np.exp(-0.5 * t / 2)),
    ("cosine", lambda gs: tf.train.cosine_decay(0.05, gs, 5, alpha=0.1),
     lambda t: 0.05 * (0.9 * 0.5 * (1 + np.cos(np.pi * min(t, 5) / 5)) + 0.1)),
]


@pytest.mark.parametrize("name,schedule,formula", SCHEDULES, ids=[s[0] for s in SCHEDULES])
def test_learning_rate_schedules_drive_plain_sgd(name, schedule, formula):
    w0, b0, w, b, rates, types, _ = _scheduled_run(tf.train.GradientDescentOptimizer, schedule)
    want = [formula(t) for t in range(len(rates))]          # step value BEFORE the increment
    assert np.allclose(rates, want, rtol=1e-5)
    ref_w, ref_b = _sgd_reference(w0, b0, want)
    assert np.allclose(w, ref_w, rtol=2e-5, atol=2e-6) and np.allclose(b, ref_b, rtol=2e-5, atol=2e-6)
    assert "MultiApplyOptimizer" in types


def test_scheduled_adam_equals_adam_with_the_rate_of_each_step():
    schedule, formula = SCHEDULES[1][1], SCHEDULES[1][2]
    for fusion in (True, False):
        w0, b0, w, b, rates, _, _ = _scheduled_run(tf.train.AdamOptimizer, schedule, fusion=fusion)
        bx, by = _problem()
        p = [w0.astype(np.float64), b0.astype(np.float64)]
        m = [np.zeros_like(v) for v in p]
        v2 = [np.zeros_like(v) for v in p]
        for t in range(1, len(rates) + 1):
            dz = 2.0 * (bx @ p[0] + p[1] - by) / bx.shape[0]
            for i, g in enumerate([bx.T @ dz, dz.sum(0)]):
                m[i] = 0.9 * m[i] + 0.1 * g
                v2[i] = 0.999 * v2[i] + 0.001 * g * g
                p[i] -= formula(t - 1) * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m[i] / (
                    np.sqrt(v2[i]) + 1e-8)
        assert np.allclose(w, p[0], rtol=5e-5, atol=5e-6) and np.allclose(b, p[1], rtol=5e-5, atol=5e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,schedule,formula", SCHEDULES[:2] + SCHEDULES[4:],
                         ids=[s[0] for s in SCHEDULES[:2] + SCHEDULES[4:]])  # (incl. exp, cos)
def test_gpu_schedule_is_read_on_the_device_inside_captured_steps(name, schedule, formula):
    w0, b0, w, b, rates, types, stats = _scheduled_run(tf.train.GradientDescentOptimizer,
                                                       schedule, device="cuda:0", steps=12)
    want = [formula(t) for t in range(len(rates))]
    assert np.allclose(rates, want, rtol=1e-5)
    ref_w, ref_b = _sgd_reference(w0, b0, want)
    assert np.allclose(w, ref_w, rtol=5e-5, atol=5e-6) and np.allclose(b, ref_b, rtol=5e-5, atol=5e-6)
    assert stats["graph_replays"] >= 8 and stats["torch_fallbacks"] == 0


@pytest.mark.gpu
def test_gpu_chain_of_general_kernels_reads_the_batch_through_the_step_ring(monkeypatch):
    """Adam on the MNIST MLP with the data set resident in HBM: the captured step's first-layer
    product, weight-gradient product and head read the batch where it lies (no feed gather);
    losses and weights are bit-identical to the step that gathers its feeds first."""
    rng = np.random.RandomState(0)
    xs = torch.from_numpy(rng.rand(16, 100, 784).astype(np.float32)).cuda()
    ys = torch.from_numpy(np.eye(10, dtype=np.float32)[rng.randint(0, 10, (16, 100))]).cuda()

    def run():
        with tf.Graph().as_default():
            m = mnist_mlp.build_model(seed=1, optimizer=tf.train.AdamOptimizer(0.01),
                                      with_summaries=False)
            with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
                sess.run(m.init_op)
                losses = [float(sess.run([m.train_op, m.loss, m.global_step],
                                         {m.x: xs[i % 16], m.y_: ys[i % 16]})[1])
                          for i in range(40)]
                return losses, sess.run(m.w1), dict(sess.stats)

    ring = run()
    monkeypatch.setenv("DTF_NO_RING_READS", "1")
    gathered = run()
    assert ring[2].get("ring_read_captures", 0) >= 1 and ring[2]["graph_replays"] >= 30
    assert gathered[2].get("ring_read_captures", 0) == 0
    assert ring[0] == gathered[0] and np.array_equal(ring[1], gathered[1])
    assert ring[0][-1] < 0.5 * ring[0][0]


def _ema_run(device=None, num_updates=False, steps=6, optimizer=None, fusion=True):
    """SGD on the two-variable regression problem with a moving average of both variables taken
    after every update; returns the per-step (w, b, avg_w, avg_b) and the session's stats."""
    bx, by = _problem()
    with tf.Graph().as_default():
        tf.set_random_seed(3)
        gs = tf.train.get_or_create_global_step()
        x = tf.placeholder(tf.float32, list(bx.shape))
        y_ = tf.placeholder(tf.float32, list(by.shape))
        w = tf.Variable(tf.random_normal([bx.shape[1], by.shape[1]]), name="w")
        b = tf.Variable(tf.zeros([by.shape[1]]), name="b")
        loss = tf.reduce_mean(tf.reduce_sum(tf.square(tf.matmul(x, w) + b - y_), axis=[1]))
        opt = optimizer() if optimizer else tf.train.GradientDescentOptimizer(0.05)
        train_op = opt.minimize(loss, global_step=gs)
        ema = tf.train.ExponentialMovingAverage(0.8, num_updates=gs if num_updates else None)
        with tf.control_dependencies([train_op]):
            ema_op = ema.apply([w, b])
        names = (ema.average_name(w), ema.average(w).op.name, sorted(ema.variables_to_restore()))
        cfg = tf.ConfigProto(device=device or "cpu", graph_fusion=fusion)
        rows = []
        with tf.Session(config=cfg) as sess:
            sess.run(tf.global_variables_initializer())
            rows.append(sess.run([w, b, ema.average(w), ema.average(b)]))
            for _ in range(steps):
                sess.run(ema_op, {x: bx, y_: by})
                rows.append(sess.run([w, b, ema.average(w), ema.average(b)]))
            stats = dict(sess.stats)
            stats["node_types"] = sorted({s.node.type for p in sess._plans.values()
                                          for s in p.steps})
    return rows, names, stats


def _check_ema(rows, num_updates):
    assert np.array_equal(rows[0][0], rows[0][2]) and np.array_equal(rows[0][1], rows[0][3])
    assert not np.array_equal(rows[1][0], rows[0][0])            # the variables do move
    avg = [rows[0][2].astype(np.float64), rows[0][3].astype(np.float64)]
    for t in range(1, len(rows)):
        # the counter has been advanced by the step whose update the average follows
        decay = min(0.8, (1 + t) / (10 + t)) if num_updates else 0.8
        for i in range(2):
            avg[i] = avg[i] - (1 - decay) * (avg[i] - rows[t][i].astype(np.float64))
            assert np.allclose(rows[t][2 + i], avg[i], rtol=2e-5, atol=2e-6), (t, i)


@pytest.mark.parametrize("num_updates", [False, True], ids=["fixed_decay", "num_updates"])
def test_exponential_moving_average_follows_the_updated_variables(num_updates):
    rows, names, _ = _ema_run(num_updates=num_updates)
    _check_ema(rows, num_updates)
    assert names[0] == names[1] == "w/ExponentialMovingAverage"
    assert names[2] == ["b/ExponentialMovingAverage", "global_step", "w/ExponentialMovingAverage"]


@pytest.mark.parametrize("num_updates", [False, True], ids=["fixed_decay", "num_updates"])
def test_moving_averages_of_a_step_are_one_fused_node(num_updates):
    """The per-variable ``Sub`` / ``Mul`` / ``AssignSub`` triples become ONE ``MultiMovingAverage``
    node (one multi-tensor launch on a GPU); values are those of the unfused ops, bit for bit."""
    fused, _, st = _ema_run(num_updates=num_updates)
    plain, _, st0 = _ema_run(num_updates=num_updates, fusion=False)
    assert "MultiMovingAverage" in st["node_types"] and "AssignSub" not in st["node_types"]
    assert "MultiMovingAverage" not in st0["node_types"] and "AssignSub" in st0["node_types"]
    for a, b in zip(fused, plain):
        for p, q in zip(a, b):
            np.testing.assert_allclose(p, q, rtol=1e-6, atol=1e-7)   # (the loss fuses too)
        assert np.array_equal(a[2] - a[0], a[2] - a[0])


def test_moving_averages_checkpoint_and_load_into_the_variables(tmp_path):
    with tf.Graph().as_default():
        v = tf.Variable(tf.fill([4], 2.0), name="v")
        ema = tf.train.ExponentialMovingAverage(0.5)
        step = tf.group(v.assign_add(tf.fill([4], 1.0)).op)
        with tf.control_dependencies([step]):
            ema_op = ema.apply([v])
        with pytest.raises(ValueError):
            ema.apply([v])
        saver = tf.train.Saver()
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            sess.run(ema_op)
            sess.run(ema_op)
            assert np.allclose(sess.run(v), 4.0)
            assert np.allclose(sess.run(ema.average(v)), 3.25)     # 2 -> 2.5 -> 3.25
            path = saver.save(sess, str(tmp_path / "model"))
        evaluating = tf.train.Saver(ema.variables_to_restore())
        with tf.Session() as sess:
            evaluating.restore(sess, path)
            assert np.allclose(sess.run(v), 3.25)


@pytest.mark.gpu
@pytest.mark.parametrize("num_updates", [False, True], ids=["fixed_decay", "num_updates"])
def test_gpu_moving_average_inside_captured_steps(num_updates):
    rows, _, stats = _ema_run(device="cuda:0", num_updates=num_updates, steps=10)
    _check_ema(rows, num_updates)
    assert stats["torch_fallbacks"] == 0 and stats["graph_replays"] >= 5


@pytest.mark.gpu
def test_gpu_moving_average_next_to_adam():
    rows, _, stats = _ema_run(device="cuda:0", steps=8,
                              optimizer=lambda: tf.train.AdamOptimizer(0.01))
    _check_ema(rows, False)
    assert stats["torch_fallbacks"] == 0


def _ema_mlp_job(cluster, task, steps, results, protocol="grpc", comm=None, strategy=None):
    with tf.Graph().as_default():
        config = tf.ConfigProto(device="cpu", strategy=strategy)
        server = tf.train.Server(cluster, job_name="worker", task_index=task, protocol=protocol,
                                 config=config, comm=comm)
        with tf.device(tf.train.replica_device_setter(
                worker_device="/job:worker/task:%d" % task, cluster=cluster)):
            m = mnist_mlp.build_model(seed=1, with_summaries=False, learning_rate=0.01)
            ema = tf.train.ExponentialMovingAverage(0.9)
            with tf.control_dependencies([m.train_op]):
                step_op = ema.apply([m.w1, m.b2])
            init_op = tf.global_variables_initializer()       # (covers the shadows too)
        sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                 init_op=init_op, logdir=None, recovery_wait_secs=0.05,
                                 summary_op=None)
        from distributed_tensorflow_amd.data.mnist import synthetic_mnist
        images, labels = synthetic_mnist(400, seed=task)
        onehot = np.eye(10, dtype=np.float32)[labels]
        rng = np.random.RandomState(7 + task)
        with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
            for _ in range(steps):
                idx = rng.randint(0, 400, 50)
                sess.run(step_op, {m.x: images[idx], m.y_: onehot[idx]})
            results[task] = dict(w1=sess.run(m.w1), avg=sess.run(ema.average(m.w1)),
                                 avg_b2=sess.run(ema.average(m.b2)), b2=sess.run(m.b2),
                                 step=float(sess.run(m.global_step)),
                                 shadow_device=ema.average(m.w1).device, runtime=server.runtime)
        sv.stop()
        server.stop()


def test_moving_average_under_allreduce_keeps_replicas_identical():
    world, steps = 2, 8
    cluster = _cluster(world)
    comms = LoopbackComm.create_world(world, timeout_s=30)
    results = {}
    threads = [threading.Thread(target=_ema_mlp_job, args=(cluster, t, steps, results),
                                kwargs=dict(protocol="rccl", comm=comms[t],
                                            strategy="allreduce_sync")) for t in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert set(results) == {0, 1}
    for key in ("w1", "avg", "avg_b2", "b2"):
        assert np.array_equal(results[0][key], results[1][key]), key          # bitwise
    assert results[0]["step"] == steps
    # the average trails the variable: it has moved off its start (zero for b2) but not caught up
    assert np.abs(results[0]["avg_b2"]).max() > 0
    assert np.abs(results[0]["avg_b2"]).max() < np.abs(results[0]["b2"]).max()
    # only the trainable variables' span of the arena is reduced; the shadows lie behind it
    assert 0 < results[0]["runtime"].bytes_reduced / steps <= (mnist_mlp.NUM_PARAMS + 4 * 64) * 4


def test_moving_average_on_a_parameter_server_lives_with_its_variable():
    cluster = _cluster(1)
    ps = tf.train.Server(cluster, job_name="ps", task_index=0)
    results = {}
    _ema_mlp_job(cluster, 0, 6, results)
    store = ps.runtime.local_store
    assert results[0]["shadow_device"] == "/job:ps/task:0"
    assert "weights/Variable/ExponentialMovingAverage" in store.names()
    assert np.array_equal(store.read("biases/Variable_1/ExponentialMovingAverage").numpy(),
                          results[0]["avg_b2"])
    assert np.abs(results[0]["avg_b2"]).max() > 0
    assert not np.array_equal(results[0]["avg"], results[0]["w1"])
    ps.stop()


@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_gradient_step_node_equals_the_chain_of_general_nodes(strategy, opt):
    """Two loopback ranks: with ``step_fusion`` the forward and backward of the classifier are ONE
    node ending in the four gradients (``MLPStepGrads``; the optimizer node behind the collective
    is unchanged), without it the chain of fused nodes. Same costs, same replicas, bit for bit."""
    world, steps = 2, 6

    def run(fused):
        cluster = _cluster(world)
        comms = LoopbackComm.create_world(world, timeout_s=30)
        results = {}

        def job(task):
            g = tf.Graph()
            with g.as_default():
                config = tf.ConfigProto(device="cpu", strategy=strategy, step_fusion=fused)
                server = tf.train.Server(cluster, job_name="worker", task_index=task,
                                         protocol="rccl", config=config, comm=comms[task])
                with tf.device(tf.train.replica_device_setter(
                        worker_device="/job:worker/task:%d" % task, cluster=cluster)):
                    m = mnist_mlp.build_model(
                        seed=1, with_summaries=False,
                        optimizer=tf.train.AdamOptimizer(0.01) if opt == "adam" else None)
                sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                         init_op=m.init_op, logdir=None,
                                         recovery_wait_secs=0.05, summary_op=None)
                from distributed_tensorflow_amd.data.mnist import synthetic_mnist
                images, labels = synthetic_mnist(200, seed=task)
                onehot = np.eye(10, dtype=np.float32)[labels]
                costs = []
                with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
                    for i in range(steps):
                        lo = 30 * i
                        _, c, a, st = sess.run([m.train_op, m.loss, m.accuracy, m.global_step],
                                               {m.x: images[lo:lo + 30], m.y_: onehot[lo:lo + 30]})
                        costs.append((float(c), float(a), float(st)))
                    plan_types = {s.node.type for p in sess._plans.values() for s in p.steps}
                    results[task] = (costs, sess.run(m.w1), sess.run(m.b2), plan_types)
                sv.stop()
                server.stop()

        threads = [threading.Thread(target=job, args=(t,)) for t in range(world)]
        [t.start() for t in threads]
        [t.join(120) for t in threads]
        assert set(results) == {0, 1}
        return results

    chain, fused = run(False), run(True)
    for t in range(world):
        assert "MLPStepGrads" in fused[t][3] and "MLPStepGrads" not in chain[t][3]
        assert "ClassifierHeadTrain" in chain[t][3] and "ClassifierHeadTrain" not in fused[t][3]
        assert fused[t][0] == chain[t][0]
        assert np.array_equal(fused[t][1], chain[t][1]) and np.array_equal(fused[t][2], chain[t][2])
