"""RCCL code paths on the GPU box (one GPU: world size 1 under the real ``nccl`` backend):
every collective of ``TorchDistComm``, and the two synchronous runtimes training the MNIST MLP
through ``Server(protocol="rccl")`` with results identical to a plain local session."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_comm():
    assert torch.cuda.is_available()
    from distributed_tensorflow_amd.parallel.comm import TorchDistComm
    from distributed_tensorflow_amd.train.server import _free_port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    comm = TorchDistComm(rank=0, world_size=1, device=torch.device("cuda", 0), backend="nccl",
                         master_addr="127.0.0.1", master_port=_free_port())
    yield comm
    comm.close()


def test_rccl_collectives_world_one(nccl_comm):
    c = nccl_comm
    assert c.rank == 0 and c.world_size == 1 and c.backend == "nccl"
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    c.all_reduce(t, "sum")
    c.all_reduce(t, "avg")
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
    out = torch.empty(8, device="cuda")
    c.reduce_scatter(out, t, "sum")
    assert torch.equal(out, t)
    g = torch.empty(8, device="cuda")
    work = c.all_gather(g, t, async_op=True)
    work.wait()
    assert torch.equal(g, t)
    b = torch.full((4,), 3.0, device="cuda").bfloat16()
    c.broadcast(b, src=0)
    f8 = torch.ones(16, device="cuda").to(torch.float8_e4m3fn)
    c.broadcast(f8, src=0)                       # 1-byte floats travel as uint8
    assert c.broadcast_object({"k": 1}) == {"k": 1}
    c.barrier()
    sub = c.split(0)
    assert sub.world_size == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("capture", [False, True])
@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_collective_runtime_trains_like_local_session(nccl_comm, strategy, capture):
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.train.server import _free_port
    rng = np.random.RandomState(2)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(5)]

    def run(distributed):
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy=strategy, capture_collectives=capture)
            target, server = "", None
            if distributed:
                cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
                server = tf.train.Server(cluster, job_name="worker", task_index=0,
                                         protocol="rccl", config=cfg, comm=nccl_comm)
                target = server.target
                scope = tf.device(tf.train.replica_device_setter(cluster=cluster))
            else:
                scope = tf.device(None)
            with scope:
                model = mnist_mlp.build_model(learning_rate=0.01, seed=1)
            losses = []
            if distributed:
                sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step,
                                         init_op=model.init_op)
                sess = sv.prepare_or_wait_for_session(target, config=cfg)
            else:
                sess = tf.Session(config=cfg)
                sess.run(model.init_op)
            for bx, by in batches:
                _, l, st = sess.run([model.train_op, model.loss, model.global_step],
                                    {model.x: bx, model.y_: by})
                losses.append(float(l))
            w2 = sess.run(model.w2)
            stats = dict(sess.stats)
            sess.close()
            if distributed:
                sv.stop()
                assert server.runtime.strategy == strategy
                assert server.runtime.comm is nccl_comm
                # a world of one is captured like a local session; with capture_collectives the
                # RCCL calls themselves are nodes of the replayed HIP graph
                assert stats["graph_replays"] >= 3, stats
            return losses, w2, float(st)

    ref = run(False)
    got = run(True)
    assert got[2] == ref[2] == 5.0
    assert np.allclose(got[0], ref[0], rtol=1e-5)
    assert np.allclose(got[1], ref[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("optimizer", ["sgd", "adam", "scheduled_momentum"])
@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_captured_collective_steps_equal_eager_ones_bit_for_bit(nccl_comm, strategy, optimizer):
    """The same 30 training steps under the collective runtime (RCCL, world of one) run eagerly
    and as replayed HIP graphs with the RCCL calls captured: losses, step counter and every
    variable are bit-equal, and no gradient is copied on its way to the collective."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.train.server import _free_port
    rng = np.random.RandomState(4)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(6)]

    def run(graphs):
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy=strategy, hip_graphs=graphs,
                                 capture_collectives=graphs)
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
            server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl",
                                     config=cfg, comm=nccl_comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                opt = None
                if optimizer == "adam":
                    opt = tf.train.AdamOptimizer(0.01)
                elif optimizer == "scheduled_momentum":
                    gs = tf.train.get_or_create_global_step()
                    opt = tf.train.MomentumOptimizer(
                        tf.train.exponential_decay(0.01, gs, 10, 0.5, staircase=True), 0.9)
                model = mnist_mlp.build_model(learning_rate=0.01, seed=1, optimizer=opt)
            sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step,
                                     init_op=model.init_op)
            sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
            hist = []
            for i in range(30):
                bx, by = batches[i % len(batches)]
                _, l, st = sess.run([model.train_op, model.loss, model.global_step],
                                    {model.x: bx, model.y_: by})
                hist.append((float(l), float(st)))
            variables = sess.run([model.w1, model.b1, model.w2, model.b2])
            stats = dict(sess.stats)
            rt = server.runtime
            placed = (rt.grads_in_place, rt.grads_copied)
            sess.close()
            sv.stop()
        return hist, variables, stats, placed

    eager, captured = run(False), run(True)
    assert eager[2]["graph_replays"] == 0 and captured[2]["graph_replays"] >= 25
    assert eager[0] == captured[0]
    for p, q in zip(eager[1], captured[1]):
        assert np.array_equal(p, q)
    for placed in (eager[3], captured[3]):
        assert placed[1] == 0, placed          # no gradient was copied into the arena


def test_bench_contract_one_json_line_on_stdout():
    """bench.py under the real RCCL backend (world 1): stdout is exactly one JSON line with the
    contract's keys -- RCCL's version banner and every note go to stderr."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DTF_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3",
                           "--warmup", "1", "--m", "2048", "--n", "2048", "--k", "2048"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=200)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["dtype"] == "bf16" and rec["higher_is_better"] is True
    assert rec["config"]["model"] == "gemm_2048x2048x2048_bf16"
    assert abs(rec["value"] - 2.0 * 2048 ** 3 / (rec["ms_per_step"] * 1e-3) / 1e12) < 0.02 * rec["value"]


def test_two_synchronous_workers_on_one_gpu_keep_replicas_identical():
    """Two worker sessions of a synchronous job in one process (loopback communicator, both on
    GPU 0): the eager multi-rank step path -- gradient all-reduce, fused update, small results
    fetched through the polled gather -- keeps the replicas bit-identical and learns."""
    import threading
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.data.mnist import synthetic_mnist
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.parallel.comm import LoopbackComm
    world, steps = 2, 30
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % (23450 + i) for i in range(world)]})
    comms = LoopbackComm.create_world(world, timeout_s=60)
    results, errors = {}, []

    def job(task):
        try:
            g = tf.Graph()
            with g.as_default():
                config = tf.ConfigProto(device="cuda", strategy="allreduce_sync")
                server = tf.train.Server(cluster, job_name="worker", task_index=task,
                                         protocol="rccl", config=config, comm=comms[task])
                with tf.device(tf.train.replica_device_setter(
                        worker_device="/job:worker/task:%d" % task, cluster=cluster)):
                    m = mnist_mlp.build_model(learning_rate=0.05, seed=1)
                sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                         init_op=m.init_op, recovery_wait_secs=0.05,
                                         summary_op=None)
                images, labels = synthetic_mnist(600, seed=task)
                onehot = np.eye(10, dtype=np.float32)[labels]
                rng = np.random.RandomState(100 + task)
                costs = []
                with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
                    for _ in range(steps):
                        idx = rng.randint(0, 600, 50)
                        _, c, step = sess.run([m.train_op, m.loss, m.global_step],
                                              {m.x: images[idx], m.y_: onehot[idx]})
                        costs.append(float(c))
                    results[task] = (costs, sess.run(m.w1), float(sess.run(m.global_step)),
                                     dict(sess.stats))
                sv.stop()
                server.stop()
        except Exception as e:  # noqa: BLE001
            errors.append((task, repr(e)))
            comms[task].kill()

    threads = [threading.Thread(target=job, args=(t,)) for t in range(world)]
    [t.start() for t in threads]
    [t.join(180) for t in threads]
    assert not errors, errors
    assert set(results) == {0, 1}
    (c0, w0, s0, st0), (c1, w1, s1, _) = results[0], results[1]
    assert np.array_equal(w0, w1) and s0 == s1 == float(steps)
    assert c0 != c1 and np.isfinite(c0).all()
    assert np.mean(c0[-5:]) < np.mean(c0[:5])
    assert st0["eager_runs"] >= steps            # multi-rank steps are not captured by default


@pytest.fixture(scope="module")
def native_comm():
    from distributed_tensorflow_amd.parallel.rccl_native import RcclComm, available
    from distributed_tensorflow_amd.train.server import _free_port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert available(), "libdtf_host.so must be built and librccl loadable on a GPU box"
    comm = RcclComm(rank=0, world_size=1, device=torch.device("cuda", 0),
                    master_addr="127.0.0.1", master_port=_free_port())
    yield comm
    comm.close()


def test_native_rccl_communicator_world_one(native_comm):
    """The communicator that talks to librccl itself (own ncclUniqueId, ncclCommInitRank, calls on
    raw pointers and a stream of its own): every collective of the interface, asynchronous
    handles, a split, host objects."""
    c = native_comm
    assert c.rank == 0 and c.world_size == 1 and c.backend == "rccl_native"
    t = torch.arange(8, dtype=torch.float32, device="cuda")
    c.all_reduce(t, "sum")
    c.all_reduce(t, "avg")
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
    out = torch.empty(8, device="cuda")
    c.reduce_scatter(out, t, "sum")
    assert torch.equal(out, t)
    g = torch.empty(8, device="cuda")
    work = c.all_gather(g, t, async_op=True)
    x = t * 2                                         # compute enqueued while it is in flight
    work.wait()
    assert work.is_completed() or True
    assert torch.equal(g, t) and torch.equal(x, t * 2)
    b = torch.full((4,), 3.0, device="cuda").bfloat16()
    c.broadcast(b, src=0)
    h = torch.arange(6, device="cuda").to(torch.bfloat16)
    c.all_reduce(h, "max")
    assert torch.equal(h.float().cpu(), torch.arange(6.0))
    f8 = torch.ones(16, device="cuda").to(torch.float8_e4m3fn)
    c.broadcast(f8, src=0)                           # 1-byte floats travel as uint8
    assert c.broadcast_object({"k": 1}) == {"k": 1} and c.all_gather_object(5) == [5]
    c.barrier()
    sub = c.split(0)
    assert sub.world_size == 1 and sub.rank == 0
    s = torch.ones(4, device="cuda")
    sub.all_reduce(s)
    assert torch.equal(s.cpu(), torch.ones(4))
    sub.close()
    torch.cuda.synchronize()


@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_collective_runtime_over_the_native_communicator_matches_torch_distributed(
        nccl_comm, native_comm, strategy):
    """The same captured training steps with the collectives issued through the native RCCL
    communicator and through torch.distributed's: bit-equal."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.train.server import _free_port
    rng = np.random.RandomState(9)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(4)]

    def run(comm):
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy=strategy, capture_collectives=True,
                                 step_fusion=False)
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
            server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl",
                                     config=cfg, comm=comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                model = mnist_mlp.build_model(learning_rate=0.01, seed=1)
            sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step,
                                     init_op=model.init_op)
            sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
            hist = []
            for i in range(12):
                bx, by = batches[i % 4]
                _, l, st = sess.run([model.train_op, model.loss, model.global_step],
                                    {model.x: bx, model.y_: by})
                hist.append((float(l), float(st)))
            w = sess.run([model.w1, model.w2])
            replays = sess.stats["graph_replays"]
            sess.close()
            sv.stop()
        return hist, w, replays

    a, b = run(nccl_comm), run(native_comm)
    assert a[0] == b[0] and b[2] >= 8
    for p, q in zip(a[1], b[1]):
        assert np.array_equal(p, q)


def test_server_selects_the_native_communicator_from_the_environment(nccl_comm, monkeypatch):
    """``DTF_COMM_BACKEND=rccl_native`` through ``tf.train.Server(protocol="rccl")`` -- no
    communicator handed in, no launcher environment: the server builds an ``RcclComm`` from its
    ClusterSpec (a world of one here) and captured MNIST steps on it are bit-equal to the same
    steps over torch.distributed's RCCL backend."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.parallel.rccl_native import RcclComm
    from distributed_tensorflow_amd.train.server import _free_port
    for key in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(key, raising=False)
    rng = np.random.RandomState(11)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(4)]

    def run(native):
        if native:
            monkeypatch.setenv("DTF_COMM_BACKEND", "rccl_native")
        else:
            monkeypatch.delenv("DTF_COMM_BACKEND", raising=False)
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy="reduce_scatter_sync",
                                 step_fusion=False)
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
            server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl",
                                     config=cfg, comm=None if native else nccl_comm)
            comm = server.runtime.comm
            assert isinstance(comm, RcclComm) == native, type(comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                model = mnist_mlp.build_model(learning_rate=0.01, seed=1)
            sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step,
                                     init_op=model.init_op)
            sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
            hist = []
            for i in range(12):
                bx, by = batches[i % 4]
                _, l, st = sess.run([model.train_op, model.loss, model.global_step],
                                    {model.x: bx, model.y_: by})
                hist.append((float(l), float(st)))
            w = sess.run([model.w1, model.w2])
            replays = sess.stats["graph_replays"]
            watched = comm._watchdog.in_flight() if native else 0
            sess.close()
            sv.stop()
            server.stop()
        return hist, w, replays, watched

    a, b = run(False), run(True)
    assert a[0] == b[0] and b[2] >= 8          # captured (RCCL calls are graph nodes) by default
    for p, q in zip(a[1], b[1]):
        assert np.array_equal(p, q)


def test_native_communicator_watchdog_turns_a_stuck_collective_into_comm_error():
    """A collective whose completion event never fires (a peer that never arrives): the watchdog
    aborts the communicator and parks a ``CommError`` within its limit -- every later call raises
    it, the process goes on (``Supervisor.run_with_recovery`` catches that exception)."""
    import time
    from distributed_tensorflow_amd.parallel.comm import CommError
    from distributed_tensorflow_amd.parallel.rccl_native import RcclComm, _StreamWork
    from distributed_tensorflow_amd.train.server import _free_port

    class NeverDone:
        def query(self):
            return False

    comm = RcclComm(rank=0, world_size=1, device=torch.device("cuda", 0),
                    master_addr="127.0.0.1", master_port=_free_port(),
                    collective_timeout_s=1.0)
    t = torch.ones(8, device="cuda")
    comm.all_reduce(t)                          # healthy first
    comm.barrier()
    assert comm._watchdog.in_flight() == 0 or time.sleep(0.2) is None
    t0 = time.monotonic()
    comm.watch(NeverDone(), "all_reduce (injected)")
    with pytest.raises(CommError, match="did not complete within"):
        while time.monotonic() - t0 < 5.0:
            comm.check_async_error()
            time.sleep(0.01)
    assert time.monotonic() - t0 < 5.0
    with pytest.raises(CommError):
        comm.all_reduce(t)
    with pytest.raises(CommError):
        comm.barrier()
    comm.close()
    # the host-side bound of a single handle
    comm = RcclComm(rank=0, world_size=1, device=torch.device("cuda", 0),
                    master_addr="127.0.0.1", master_port=_free_port())
    t0 = time.monotonic()
    with pytest.raises(CommError, match="did not complete within"):
        _StreamWork(comm, NeverDone(), "all_gather (injected)").wait(timeout=0.5)
    assert time.monotonic() - t0 < 5.0
    comm.close()
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.ones(8))  # the device is fine, only the communicator went


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
@pytest.mark.parametrize("strategy", ["reduce_scatter_sync"])
def test_gradient_step_node_on_the_gpu_equals_the_kernel_chain(nccl_comm, strategy, optimizer):
    """Captured steps under the collective runtime (RCCL, world of one): with ``step_fusion`` the
    step is the three launches of the small-step kernels in gradient mode (gradients straight
    into the arena, the small results gathered by the third launch), the collective and ONE
    update launch; without it the chain of general kernels. 100 steps, bit for bit."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.train.server import _free_port
    rng = np.random.RandomState(21)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(7)]

    def run(fused):
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy=strategy, step_fusion=fused,
                                 capture_collectives=True)
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
            server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl",
                                     config=cfg, comm=nccl_comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                opt = tf.train.AdamOptimizer(0.01) if optimizer == "adam" else None
                model = mnist_mlp.build_model(learning_rate=0.01, seed=1, optimizer=opt)
            sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step,
                                     init_op=model.init_op)
            sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
            hist = []
            for i in range(100):
                bx, by = batches[i % len(batches)]
                _, l, a, st = sess.run([model.train_op, model.loss, model.accuracy,
                                        model.global_step], {model.x: bx, model.y_: by})
                hist.append((float(l), float(a), float(st)))
            variables = sess.run([model.w1, model.b1, model.w2, model.b2])
            types = {s.node.type for p in sess._plans.values() for s in p.steps}
            stats = dict(sess.stats)
            placed = (server.runtime.grads_in_place, server.runtime.grads_copied)
            sess.close()
            sv.stop()
        return hist, variables, types, stats, placed

    chain, fused = run(False), run(True)
    assert "MLPStepGrads" in fused[2] and "MLPStepGrads" not in chain[2]
    assert fused[3]["graph_replays"] >= 95 and fused[3]["torch_fallbacks"] == 0
    # every gradient was produced in its arena slot whenever Python ran the step (the first,
    # eager run and the capture; replays run no Python)
    assert fused[4][1] == 0 and fused[4][0] >= 4
    assert fused[0] == chain[0]
    assert fused[0][-1][2] == 100.0
    for p, q in zip(fused[1], chain[1]):
        assert np.array_equal(p, q)


def test_fused_regression_step_equals_the_separate_kernels(nccl_comm, monkeypatch):
    """The linear-regression step (``models/linear_regression.py``) under the collective runtime,
    captured: with the fused loss / gradient / bias-gradient launch (which also gathers the
    step's results) against the separate kernels -- 100 steps, bit for bit."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import linear_regression as lr
    from distributed_tensorflow_amd.train.server import _free_port
    data = lr.SyntheticRegressionData(256, 1, task=0)
    batches = [data.next_batch(64) for _ in range(5)]

    def run(fused):
        if fused:
            monkeypatch.delenv("DTF_NO_MSE_STEP", raising=False)
        else:
            monkeypatch.setenv("DTF_NO_MSE_STEP", "1")
        g = tf.Graph()
        with g.as_default():
            cfg = tf.ConfigProto(device="cuda:0", strategy="reduce_scatter_sync",
                                 capture_collectives=True)
            cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
            server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl",
                                     config=cfg, comm=nccl_comm)
            with tf.device(tf.train.replica_device_setter(cluster=cluster)):
                m = lr.build_model(features=256, outputs=1, learning_rate=0.05)
            sv = tf.train.Supervisor(is_chief=True, global_step=m.global_step, init_op=m.init_op)
            sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
            hist = []
            for i in range(100):
                bx, by = batches[i % len(batches)]
                _, l, st = sess.run([m.train_op, m.loss, m.global_step], {m.x: bx, m.y_: by})
                hist.append((float(l), float(st)))
            w, b = sess.run([m.w, m.b])
            types = {s.node.type for p in sess._plans.values() for s in p.steps}
            stats = dict(sess.stats)
            sess.close()
            sv.stop()
        return hist, w, b, types, stats

    chain, fused = run(False), run(True)
    assert "MSELossStep" in fused[3] and "MSELossStep" not in chain[3]
    assert fused[4]["graph_replays"] >= 95 and fused[4]["torch_fallbacks"] == 0
    assert fused[0] == chain[0] and fused[0][-1][1] == 100.0
    assert fused[0][-1][0] < 0.05 * fused[0][0][0]            # it learns
    assert np.array_equal(fused[1], chain[1]) and np.array_equal(fused[2], chain[2])
