"""T1 (single process, several "tasks"): communicators, parameter-server transport,
Supervisor chief/non-chief protocol, async vs sync update semantics, failure injection."""
import os
import threading
import time

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel.comm import CommError, LoopbackComm, SingleComm
from distributed_tensorflow_amd.parallel.ps_transport import (PsService, RemoteVariableStore,
                                                              UnavailableError)
from distributed_tensorflow_amd.parallel.sharded_matmul import STRATEGIES, ShardedMatmul
from distributed_tensorflow_amd.train.server import _free_port


def run_ranks(world, fn, timeout_s=20.0):
    comms = LoopbackComm.create_world(world, timeout_s=timeout_s)
    out, errs = [None] * world, []

    def body(r):
        try:
            out[r] = fn(comms[r])
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    if errs:
        raise errs[0]
    return out


def test_loopback_collectives():
    def fn(c):
        r, w = c.rank, c.world_size
        t = torch.full((6,), float(r + 1))
        c.all_reduce(t)
        assert torch.equal(t, torch.full((6,), 10.0))
        b = torch.arange(4.0) if r == 2 else torch.zeros(4)
        c.broadcast(b, src=2)
        assert torch.equal(b, torch.arange(4.0))
        full = torch.arange(8.0) * (r + 1)
        out = torch.empty(2)
        c.reduce_scatter(out, full)
        assert torch.equal(out, torch.arange(8.0)[2 * r:2 * r + 2] * 10)
        gathered = torch.empty(w, 3)
        c.all_gather(gathered, torch.full((3,), float(r)))
        assert torch.equal(gathered[:, 0], torch.arange(4.0))
        m = torch.tensor([float(r)])
        c.all_reduce(m, "max")
        assert m.item() == 3.0
        a = torch.tensor([float(r)])
        c.all_reduce(a, "avg")
        assert a.item() == 1.5
        assert c.broadcast_object({"rank": r}, src=1) == {"rank": 1}
        assert c.all_gather_object(r * r) == [0, 1, 4, 9]
        sub = c.split(color=r % 2)                       # {0,2} and {1,3}
        assert sub.world_size == 2 and sub.rank == r // 2
        s = torch.tensor([float(r)])
        sub.all_reduce(s)
        assert s.item() == (2.0 if r % 2 == 0 else 4.0)
        c.barrier()
        return True

    assert run_ranks(4, fn) == [True] * 4


def test_loopback_fault_injection_detects_dead_and_slow_ranks():
    comms = LoopbackComm.create_world(2, timeout_s=0.5)
    comms[1].kill()
    with pytest.raises(CommError, match="died"):
        comms[0].all_reduce(torch.ones(1))
    comms = LoopbackComm.create_world(2, timeout_s=0.3)
    with pytest.raises(CommError, match="timed out"):
        comms[0].barrier()                                # rank 1 never arrives
    single = SingleComm()
    t = torch.ones(3)
    single.all_reduce(t)
    single.barrier()
    assert torch.equal(t, torch.ones(3)) and single.split(0).world_size == 1


@pytest.mark.parametrize("strategy", [s for s in STRATEGIES if not s.endswith("_direct")])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_matmul_strategies_match_reference(strategy, world):
    def fn(c):
        # world 8 runs SUMMA on a 2 x 4 grid (K cut into lcm = 4 panels), world 2 on 1 x 2
        sm = ShardedMatmul(64, 48, 32, c, dtype=torch.float32, strategy=strategy,
                           panels=None if strategy == "summa_2d" else 2)
        sm.step()
        got = sm.result_dense().double()
        ref = sm.reference(sm.rows_of_result(), sm.cols_of_result())
        return (got - ref).abs().max().item()

    assert max(run_ranks(world, fn)) < 1e-5


@pytest.mark.parametrize("world,x,panels,pipelined", [(2, 8, 2, False), (4, 4, 3, True),
                                                       (8, 2, 1, True), (4, 12, None, False)])
def test_sharded_matmul_allgather_with_recomputed_heads(world, x, panels, pipelined):
    """recompute_rows: every rank computes the first x rows of all blocks itself, only the rest
    of each block is gathered; the result (in plain row order) is unchanged."""
    def fn(c):
        sm = ShardedMatmul(128, 48, 32, c, dtype=torch.float32, strategy="m_shard_allgather",
                           panels=panels, pipelined=pipelined, recompute_rows=x)
        errs = []
        for _ in range(3):                       # pipelined: both buffer sets get used
            sm.step()
        sm.wait()
        got = sm.result_dense().double()
        assert got.shape == (128, 48)
        ref = sm.reference(sm.rows_of_result(), sm.cols_of_result())
        errs.append((got - ref).abs().max().item())
        assert sm._local.shape[0] == 128 // world - x      # only the bodies travel
        return max(errs)

    assert max(run_ranks(world, fn)) < 1e-5


def test_sharded_matmul_allgather_uneven_whole_tile_panels():
    """Bodies of 5 x 256 rows in 2 panels: 768 + 512 rows, the same split on every rank."""
    def fn(c):
        sm = ShardedMatmul(2 * 1536, 16, 8, c, dtype=torch.float32, strategy="m_shard_allgather",
                           panels=2, recompute_rows=256, pipelined=True)
        assert sm._panel_rows == [768, 512] and sm._panel_start == [0, 768]
        sm.step()
        sm.step()
        sm.wait()
        got = sm.result_dense().double()
        rows = sm.rows_of_result()[::97]
        ref = sm.reference(rows, sm.cols_of_result())
        return (got[::97] - ref).abs().max().item()

    assert max(run_ranks(2, fn)) < 1e-5


def test_recompute_rows_validation_and_balance_point():
    from distributed_tensorflow_amd.parallel.sharded_matmul import balanced_recompute_rows
    comm = LoopbackComm.create_world(4)[0]
    with pytest.raises(ValueError):
        ShardedMatmul(64, 32, 32, comm, strategy="k_shard_allreduce", recompute_rows=4)
    with pytest.raises(ValueError):
        ShardedMatmul(64, 32, 32, comm, strategy="m_shard_allgather", recompute_rows=16)
    with pytest.raises(ValueError):
        ShardedMatmul(64, 32, 32, strategy="m_shard_allgather", recompute_rows=4)   # world 1
    # compute-bound step: nothing to gain
    assert balanced_recompute_rows(2048, 8, 1.0e-3, 1.0e-3) == []
    # 8 ranks, product 0.73 ms, step 1.34 ms: x = 2048 * 0.61 / (7 * 0.73 + 1.34) = 193.7
    assert balanced_recompute_rows(2048, 8, 0.73e-3, 1.34e-3) == [128, 256]
    # 2 ranks, product 2.9 ms, step 5.4 ms: x = 8192 * 2.5 / 8.3 = 2467
    assert balanced_recompute_rows(8192, 2, 2.9e-3, 5.4e-3) == [2304, 2560]
    # at the balance point both sides take the same time
    b, w, tc, t0 = 4096, 4, 1.45e-3, 2.7e-3
    x = b * (t0 - tc) / ((w - 1) * tc + t0)
    assert abs(tc * (b + (w - 1) * x) / b - t0 * (b - x) / b) < 1e-9


@pytest.mark.parametrize("strategy", ["m_shard_allgather", "summa_2d"])
def test_sharded_matmul_fp8_operands(strategy):
    """fp8 e4m3 operands (bf16 / fp32 results): the CPU path multiplies the dequantised values,
    so it pins the operand generation, the sharding and the collectives of the fp8 config."""
    def fn(c):
        sm = ShardedMatmul(32, 32, 64, c, dtype=torch.float8_e4m3fn, strategy=strategy)
        sm.step()
        assert sm.a.dtype == torch.float8_e4m3fn
        got = sm.result_dense().double()
        ref = sm.reference(sm.rows_of_result(), sm.cols_of_result())
        return (got - ref).abs().max().item()

    assert max(run_ranks(4, fn)) < (0.06 if strategy == "m_shard_allgather" else 1e-4)


def test_sharded_matmul_is_world_size_invariant():
    def fn(c):
        sm = ShardedMatmul(32, 32, 64, c, dtype=torch.float32, strategy="k_shard_allreduce")
        return sm.step().clone()

    one = run_ranks(1, fn)[0]
    four = run_ranks(4, fn)
    for c in four:
        assert torch.allclose(c, one, atol=1e-5)
    with pytest.raises(ValueError):
        ShardedMatmul(30, 32, 64, LoopbackComm.create_world(4)[0], strategy="m_shard")
    with pytest.raises(ValueError):
        ShardedMatmul(32, 32, 64, strategy="bogus")


# -- parameter-server transport -----------------------------------------------------------------
@pytest.fixture(params=["python", "native"])
def ps_service(request):
    """Both implementations of the ps task speak the same protocol: the Python one and the C++
    one (``csrc/runtime/ps_server.cc``)."""
    if request.param == "native":
        from distributed_tensorflow_amd.parallel.ps_transport import NativePsService
        try:
            svc = NativePsService("127.0.0.1", 0)
        except RuntimeError:
            pytest.skip("native host library not built")
    else:
        svc = PsService("127.0.0.1", 0)
    svc.start()
    yield svc
    svc.stop()


def test_ps_transport_pull_apply_semantics(ps_service):
    addr = "127.0.0.1:%d" % ps_service.port
    a, b = RemoteVariableStore(addr), RemoteVariableStore(addr)
    assert a.uninitialized(["w", "step"]) == ["w", "step"]
    a.assign("w", torch.ones(4, 3))
    a.assign("step", torch.tensor(0.0))
    assert b.is_initialized("w") and b.uninitialized(["w", "x"]) == ["x"]
    with pytest.raises(tf.FailedPreconditionError):
        b.read("x")
    # two workers read the same weights, push different gradients: the ps applies both
    a.begin_step(["w", "step"])
    b.begin_step(["w", "step"])
    a.apply_sgd("w", 0.5, torch.full((4, 3), 2.0))
    a.assign_add("step", torch.tensor(1.0))
    b.apply_sgd("w", 0.5, torch.full((4, 3), 4.0))
    b.assign_add("step", torch.tensor(1.0))
    assert torch.equal(a.read("w"), torch.ones(4, 3))          # snapshot semantics within a step
    a.end_step()
    b.end_step()
    a.begin_step(["w", "step"])
    assert torch.equal(a.read("w"), torch.full((4, 3), 1.0 - 1.0 - 2.0))
    assert a.read("step").item() == 2.0                        # one increment per applied gradient
    assert ps_service.store.update_count == 2 and a.ping() == 2
    assert sorted(a.names()) == ["step", "w"]
    assert a.bytes_pulled > 0 and a.bytes_pushed == 4 * 3 * 4 + 4
    state = b.state_dict()
    assert torch.equal(state["w"], a.read("w"))
    a.close()
    b.close()


def test_ps_native_service_error_paths_and_dtypes(ps_service):
    addr = "127.0.0.1:%d" % ps_service.port
    c = RemoteVariableStore(addr)
    c.assign("i", torch.tensor([1, 2, 3], dtype=torch.int64))
    c.assign("d", torch.tensor([[0.5, 1.5]], dtype=torch.float64))
    c.begin_step(["i", "d"])
    c.assign_add("i", torch.tensor([10, 10, 10], dtype=torch.int64))
    c.apply_sgd("d", 2.0, torch.tensor([[1.0, 1.0]], dtype=torch.float64))
    c.end_step()
    c.begin_step(["i", "d"])
    assert c.read("i").tolist() == [11, 12, 13] and c.read("i").dtype == torch.int64
    assert c.read("d").tolist() == [[-1.5, -0.5]]
    # re-assignment with a new shape replaces the variable
    c.assign("i", torch.zeros(2, 2, dtype=torch.float32))
    c.begin_step(["i"])
    assert c.read("i").shape == (2, 2) and c.read("i").dtype == torch.float32
    # a gradient of the wrong size is refused and the variable left untouched
    from distributed_tensorflow_amd.parallel.ps_transport import OP_APPLY
    bad = np.ones(3, np.float32)
    with pytest.raises((RuntimeError, ValueError)):
        c.client.call(OP_APPLY, {"sgd": [["i", 1.0]], "add": [],
                                 "specs": [{"dtype": "float32", "shape": [3]}]}, [bad])
    c.begin_step(["i"])
    assert torch.equal(c.read("i"), torch.zeros(2, 2))
    assert torch.equal(ps_service.store.read("d"), torch.tensor([[-1.5, -0.5]], dtype=torch.float64))
    assert sorted(ps_service.store.names()) == ["d", "i"]
    c.close()


@pytest.mark.parametrize("impl", ["python", "native"])
def test_ps_client_unavailable_and_done_protocol(impl):
    with pytest.raises(UnavailableError):
        RemoteVariableStore("127.0.0.1:%d" % _free_port(), connect_timeout_s=0.3).names()
    if impl == "native":
        from distributed_tensorflow_amd.parallel.ps_transport import NativePsService
        svc = NativePsService("127.0.0.1", 0, expected_workers=2)
    else:
        svc = PsService("127.0.0.1", 0, expected_workers=2)
    svc.start()
    addr = "127.0.0.1:%d" % svc.port
    w0, w1 = RemoteVariableStore(addr), RemoteVariableStore(addr)
    assert w0.report_done(0) == 1 and not svc.stopped
    assert w1.report_done(1) == 2 and svc.wait(2.0)            # all workers done -> join() returns
    svc.stop()


# -- whole jobs in one process: 1 ps + 2 workers (threads), reference training script flow ------
def _cluster(num_workers=2):
    return tf.train.ClusterSpec({"ps": ["127.0.0.1:%d" % _free_port()],
                                 "worker": ["127.0.0.1:%d" % _free_port()
                                            for _ in range(num_workers)]})


def _worker_job(cluster, task, steps, results, protocol="grpc", comm=None, strategy=None,
                lr=0.05, logdir=None, start_delay=0.0, bucket_bytes=None, save_model_secs=0.2,
                save_summaries_secs=120.0):
    time.sleep(start_delay)
    g = tf.Graph()
    with g.as_default():
        config = tf.ConfigProto(device="cpu", strategy=strategy, grad_bucket_bytes=bucket_bytes)
        server = tf.train.Server(cluster, job_name="worker", task_index=task, protocol=protocol,
                                 config=config, comm=comm)
        with tf.device(tf.train.replica_device_setter(
                worker_device="/job:worker/task:%d" % task, cluster=cluster)):
            m = mnist_mlp.build_model(learning_rate=lr, seed=1)
        sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                 init_op=m.init_op, logdir=logdir, recovery_wait_secs=0.05,
                                 save_model_secs=save_model_secs, summary_op=None,
                                 save_summaries_secs=save_summaries_secs)
        rng = np.random.RandomState(100 + task)
        from distributed_tensorflow_amd.data.mnist import synthetic_mnist
        images, labels = synthetic_mnist(600, seed=task)
        onehot = np.eye(10, dtype=np.float32)[labels]
        costs = []
        with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
            for i in range(steps):
                idx = rng.randint(0, 600, 50)
                _, c, step = sess.run([m.train_op, m.loss, m.global_step],
                                      {m.x: images[idx], m.y_: onehot[idx]})
                costs.append(float(c))
            final = {k: sess.run(v) for k, v in
                     (("w1", m.w1), ("b2", m.b2), ("step", m.global_step))}
            placement = (m.w1.device, m.x.device, m.train_op.device)
            types = {s.node.type for p in sess._plans.values() for s in p.steps}
            buckets = max(p.grad_buckets for p in sess._plans.values())
        sv.stop()
        results[task] = dict(costs=costs, final=final, placement=placement, types=types,
                             runtime=server.runtime, grad_buckets=buckets)
        server.stop()


def test_async_parameter_server_job_one_ps_two_workers():
    cluster = _cluster(2)
    ps = tf.train.Server(cluster, job_name="ps", task_index=0)
    assert ps.target.startswith("grpc://127.0.0.1:")
    results = {}
    threads = [threading.Thread(target=_worker_job, args=(cluster, t, 40, results),
                                kwargs=dict(start_delay=0.3 * t)) for t in (1, 0)]
    [t.start() for t in threads]      # the non-chief starts first and has to wait for the chief
    [t.join(120) for t in threads]
    assert set(results) == {0, 1}
    for t in (0, 1):
        r = results[t]
        assert r["placement"] == ("/job:ps/task:0", "/job:worker/task:%d" % t, "/job:ps/task:0")
        assert np.mean(r["costs"][-5:]) < np.mean(r["costs"][:5])          # it learns
        assert "MultiApplySGD" in r["types"]
    # state lives in the ps only: both workers read the same final variables from it
    store = ps.runtime.local_store
    assert store.read("global_step").item() == 80.0                         # every update counted
    assert store.update_count == 80 * 4
    last = max(results, key=lambda t: results[t]["final"]["step"])
    assert np.allclose(results[last]["final"]["w1"], store.read("weights/Variable").numpy())
    assert results[0]["runtime"].local_store.names() == []                  # nothing stored locally
    ps.stop()


@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_sync_collective_job_keeps_replicas_identical(strategy):
    world = 2
    cluster = tf.train.ClusterSpec({"ps": ["127.0.0.1:%d" % _free_port()],
                                    "worker": ["127.0.0.1:%d" % _free_port()
                                               for _ in range(world)]})
    comms = LoopbackComm.create_world(world, timeout_s=30)
    results = {}
    threads = [threading.Thread(target=_worker_job, args=(cluster, t, 25, results),
                                kwargs=dict(protocol="rccl", comm=comms[t], strategy=strategy))
               for t in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert set(results) == {0, 1}
    a, b = results[0]["final"], results[1]["final"]
    assert np.array_equal(a["w1"], b["w1"]) and np.array_equal(a["b2"], b["b2"])   # bitwise
    assert a["step"] == b["step"] == 25.0                # one increment per synchronous step
    assert results[0]["costs"] != results[1]["costs"]    # different data on each worker
    assert np.mean(results[0]["costs"][-5:]) < np.mean(results[0]["costs"][:5])
    assert results[0]["runtime"].bytes_reduced > 0
    for t in range(world):
        rt_t = results[t]["runtime"]
        # every gradient was produced in (or moved to) its slot of the flat arena and the arena
        # travelled as one collective; the owner-update path really ran under reduce_scatter_sync
        assert rt_t.grads_in_place + rt_t.grads_copied == 4 * 25
        assert rt_t.sharded_steps == (25 if strategy == "reduce_scatter_sync" else 0)
    # the flat arena keeps every variable on a 256-byte boundary (the scalar global_step must not
    # push the matrices off the GEMM kernels' 16-byte paths) and its length divides by the world
    rt = results[0]["runtime"]
    assert {n for n, _, _, _ in rt._layout} >= {"global_step", "weights/Variable"}
    assert all(off % 64 == 0 for _, off, _, _ in rt._layout)
    spans = sorted((off, off + numel) for _, off, numel, _ in rt._layout)
    assert all(a_end <= b_start for (_, a_end), (b_start, _) in zip(spans, spans[1:]))
    assert rt._flat_params.numel() % world == 0 and rt._flat_params.numel() >= spans[-1][1]


def test_service_threads_do_not_disturb_bucketed_collective_steps(tmp_path):
    """With a logdir the chief's checkpoint and step-rate threads call ``sess.run`` every few
    milliseconds while the training loop has gradient buckets in flight. The device phase of
    ``Session.run`` is serialised per session, so a service-thread run can neither drop the
    in-flight buckets nor slip an extra collective between them: the job finishes (no rank is
    left waiting for a collective its peer never posts), the replicas stay bit-identical and
    equal to the same job without service threads."""
    world, steps = 2, 60
    finals = {}
    for label, logdir in (("quiet", None), ("busy", str(tmp_path / "logdir"))):
        cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                                   for _ in range(world)]})
        comms = LoopbackComm.create_world(world, timeout_s=30)
        results = {}
        threads = [threading.Thread(target=_worker_job, args=(cluster, t, steps, results),
                                    kwargs=dict(protocol="rccl", comm=comms[t],
                                                strategy="allreduce_sync", bucket_bytes=4096,
                                                logdir=logdir, save_model_secs=0.01,
                                                save_summaries_secs=0.01))
                   for t in range(world)]
        [t.start() for t in threads]
        [t.join(120) for t in threads]
        assert set(results) == {0, 1}, "a rank hung or died (%s)" % label
        finals[label] = results
    busy, quiet = finals["busy"], finals["quiet"]
    assert busy[0]["grad_buckets"] == 2 and busy[0]["runtime"].buckets_issued == 2 * steps
    assert any(f.startswith("model.ckpt-") for f in os.listdir(str(tmp_path / "logdir")))
    for t in range(world):
        assert np.array_equal(busy[t]["final"]["w1"], quiet[t]["final"]["w1"])
        assert busy[t]["costs"] == quiet[t]["costs"]
    assert np.array_equal(busy[0]["final"]["w1"], busy[1]["final"]["w1"])


def test_bucketed_gradient_allreduce_overlaps_backward_and_matches_single_bucket():
    """With a small bucket size the head's gradients are all-reduced while the first layer's
    weight gradient is still being computed; the trained weights are bit-identical to the
    single-collective run."""
    world, steps = 2, 12
    finals = {}
    for label, bucket_bytes in (("one", None), ("bucketed", 4096)):
        cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                                   for _ in range(world)]})
        comms = LoopbackComm.create_world(world, timeout_s=30)
        results = {}
        threads = [threading.Thread(target=_worker_job, args=(cluster, t, steps, results),
                                    kwargs=dict(protocol="rccl", comm=comms[t],
                                                strategy="allreduce_sync",
                                                bucket_bytes=bucket_bytes))
                   for t in range(world)]
        [t.start() for t in threads]
        [t.join(120) for t in threads]
        assert set(results) == {0, 1}
        finals[label] = results
    one, many = finals["one"], finals["bucketed"]
    assert one[0]["grad_buckets"] == 0 and one[0]["runtime"].buckets_issued == 0
    assert many[0]["grad_buckets"] == 2                  # {b2, W2, b1} then {W1}
    assert many[0]["runtime"].buckets_issued == 2 * steps
    for t in range(world):
        assert np.array_equal(one[t]["final"]["w1"], many[t]["final"]["w1"])
        assert np.array_equal(one[t]["final"]["b2"], many[t]["final"]["b2"])
        assert one[t]["costs"] == many[t]["costs"]


def test_sync_sum_of_gradients_equals_sequential_application():
    """allreduce_sync with sum aggregation == applying both workers' gradients (computed at the
    same weights) one after the other with the same learning rate."""
    world = 2
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                               for _ in range(world)]})
    comms = LoopbackComm.create_world(world, timeout_s=30)
    results = {}
    threads = [threading.Thread(target=_worker_job, args=(cluster, t, 1, results),
                                kwargs=dict(protocol="rccl", comm=comms[t],
                                            strategy="allreduce_sync"))
               for t in range(world)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    # oracle: single process, same init, gradient of each worker's batch at the initial weights
    from distributed_tensorflow_amd.data.mnist import synthetic_mnist
    g = tf.Graph()
    with g.as_default():
        m = mnist_mlp.build_model(learning_rate=0.05, seed=1)
        grads = tf.gradients(m.loss, [m.w1._variable])
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(m.init_op)
            w0 = sess.run(m.w1)
            total = np.zeros_like(w0)
            for task in range(world):
                rng = np.random.RandomState(100 + task)
                images, labels = synthetic_mnist(600, seed=task)
                idx = rng.randint(0, 600, 50)
                total += sess.run(grads[0], {m.x: images[idx],
                                             m.y_: np.eye(10, dtype=np.float32)[labels][idx]})
    assert np.allclose(results[0]["final"]["w1"], w0 - 0.05 * total, atol=1e-6)


def test_supervisor_checkpoints_and_recovers(tmp_path):
    logdir = str(tmp_path / "sv")
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
    results = {}
    _worker_job(cluster, 0, 30, results, protocol="grpc", logdir=logdir)
    ckpt = tf.train.latest_checkpoint(logdir)
    assert ckpt is not None and ckpt.endswith("model.ckpt-30")        # saved again at stop()
    # a new job with the same logdir resumes from the checkpoint instead of re-initialising
    results2 = {}
    _worker_job(cluster, 0, 5, results2, protocol="grpc", logdir=logdir)
    assert results2[0]["final"]["step"] == 35.0
    events = [f for f in (tmp_path / "sv").iterdir() if f.name.startswith("events.out")]
    assert events


def test_non_chief_times_out_without_chief():
    cluster = _cluster(2)
    ps = tf.train.Server(cluster, job_name="ps", task_index=0)
    g = tf.Graph()
    with g.as_default():
        server = tf.train.Server(cluster, job_name="worker", task_index=1,
                                 config=tf.ConfigProto(device="cpu"))
        with tf.device(tf.train.replica_device_setter(cluster=cluster)):
            m = mnist_mlp.build_model()
        sv = tf.train.Supervisor(is_chief=False, global_step=m.global_step, init_op=m.init_op,
                                 recovery_wait_secs=0.05)
        with pytest.raises(TimeoutError, match="not ready"):
            sv.prepare_or_wait_for_session(server.target, max_wait_secs=0.3)
        server.stop()
    ps.stop()


def test_run_with_recovery_after_ps_restart(tmp_path):
    """The ps dies mid-job: the chief rebuilds its session, restores the last checkpoint into the
    restarted ps and carries on."""
    port = _free_port()
    cluster = tf.train.ClusterSpec({"ps": ["127.0.0.1:%d" % port],
                                    "worker": ["127.0.0.1:%d" % _free_port()]})
    ps = tf.train.Server(cluster, job_name="ps", task_index=0)
    logdir = str(tmp_path / "rec")
    g = tf.Graph()
    with g.as_default():
        server = tf.train.Server(cluster, job_name="worker", task_index=0,
                                 config=tf.ConfigProto(device="cpu"))
        with tf.device(tf.train.replica_device_setter(cluster=cluster)):
            step = tf.train.get_or_create_global_step()
            w = tf.Variable(tf.zeros([4]), name="w")
            train = tf.group(w.assign_add(tf.ones([4])), step.assign_add(1.0))
        sv = tf.train.Supervisor(is_chief=True, global_step=step, logdir=logdir,
                                 save_model_secs=0.05, recovery_wait_secs=0.05, summary_op=None)
        state = {"n": 0, "killed": False, "ps": ps}

        def step_fn(sess):
            sess.run(train)
            state["n"] += 1
            if state["n"] == 10 and not state["killed"]:
                time.sleep(0.3)                       # let a checkpoint land
                state["ps"].stop()                    # ps crashes ...
                state["killed"] = True
                state["ps"] = tf.train.Server(cluster, job_name="ps", task_index=0)   # ... restarts empty
            return state["n"] < 20

        restarts = sv.run_with_recovery(server.target, step_fn)
        assert restarts == 1
        final = state["ps"].runtime.local_store.read("global_step").item()
        assert 10 <= final <= 21                      # resumed from a checkpoint, not from zero
        server.stop()
    state["ps"].stop()


def test_local_server_and_server_validation():
    s = tf.train.Server.create_local_server()
    assert s.target.startswith("local://")
    with tf.Session(s.target) as sess:
        assert sess.run(tf.constant(3.0) * 2) == 6.0
    s.stop()
    c = tf.train.ClusterSpec({"ps": ["a:1"], "worker": ["b:1", "c:1"]})
    with pytest.raises(ValueError):
        tf.train.Server(c, job_name="evaluator", task_index=0, start=False)
    with pytest.raises(ValueError):
        tf.train.Server(c, job_name="worker", task_index=5, start=False)
    with pytest.raises(ValueError):
        tf.train.Server(c, start=False)                       # ambiguous job
    with pytest.raises(ValueError):
        tf.train.Server(c, job_name="worker", task_index=0, protocol="mpi", start=False)
    with pytest.raises(ValueError):
        tf.Session("grpc://nowhere:1")
    sd = tf.train.Server(c, job_name="worker", task_index=1, start=False).server_def
    assert sd.job_name == "worker" and sd.task_index == 1 and sd.SerializeToString()


def test_peer_memory_strategies_need_a_gpu():
    """The peer-memory strategies map GPU allocations between processes: on a CPU device they
    refuse with a clear error instead of falling back silently."""
    from distributed_tensorflow_amd.parallel.peer_mem import PeerMemError
    eng = ShardedMatmul(64, 64, 64, SingleComm(torch.device("cpu")), dtype=torch.float32,
                        strategy="m_shard_allgather_direct", device=torch.device("cpu"))
    with pytest.raises(PeerMemError):
        eng.init_operands()


def test_traced_collective_step_records_collectives_with_payload(tmp_path):
    """A traced synchronous step lists its collectives next to its ops: name, payload bytes and
    a duration each; the Chrome trace puts them on their own track."""
    world = 2
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                               for _ in range(world)]})
    comms = LoopbackComm.create_world(world, timeout_s=30)
    out = {}

    def job(task):
        g = tf.Graph()
        with g.as_default():
            config = tf.ConfigProto(device="cpu", strategy="reduce_scatter_sync")
            server = tf.train.Server(cluster, job_name="worker", task_index=task,
                                     protocol="rccl", config=config, comm=comms[task])
            with tf.device(tf.train.replica_device_setter(
                    worker_device="/job:worker/task:%d" % task, cluster=cluster)):
                m = mnist_mlp.build_model(learning_rate=0.05, seed=1)
            sv = tf.train.Supervisor(is_chief=(task == 0), global_step=m.global_step,
                                     init_op=m.init_op, recovery_wait_secs=0.05, summary_op=None)
            rng = np.random.RandomState(task)
            feed = {m.x: rng.rand(20, 784).astype(np.float32),
                    m.y_: np.eye(10, dtype=np.float32)[rng.randint(0, 10, 20)]}
            with sv.prepare_or_wait_for_session(server.target, config=config) as sess:
                sess.run(m.train_op, feed)
                md = tf.RunMetadata()
                sess.run(m.train_op, feed,
                         options=tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE),
                         run_metadata=md)
                sess.run(m.train_op, feed)          # the communicator is untouched afterwards
                out[task] = md
            sv.stop()
            server.stop()

    threads = [threading.Thread(target=job, args=(t,)) for t in range(world)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    assert set(out) == {0, 1}
    md = out[0]
    colls = [s for s in md.step_stats if s["cat"] == "collective"]
    assert [c["op"] for c in colls] == ["reduce_scatter", "all_gather"]
    arena = comms[0].world_size and colls[0]["args"]["bytes"]
    assert arena >= 79510 * 4 and colls[1]["args"]["bytes"] == arena
    assert all(c["end"] >= c["start"] for c in colls)
    summary = md.summary()
    assert summary["reduce_scatter"]["count"] == 1 and summary["reduce_scatter"]["bytes"] == arena
    trace = md.to_chrome_trace()
    tracks = {e["tid"] for e in trace["traceEvents"]}
    assert any("collectives" in t for t in tracks) and any("kernels" in t for t in tracks)
    assert "all_reduce" not in comms[0].__dict__          # instance wrappers were removed


def test_native_rendezvous_gathers_and_broadcasts_byte_strings_between_ranks():
    """The C++ control-plane rendezvous (``csrc/runtime/rendezvous.cc``) with four ranks as
    threads: all-gather of byte strings of different lengths, broadcast from a non-zero root,
    barrier; a rank that never shows up makes the others fail with CommError, not hang."""
    import threading
    from distributed_tensorflow_amd.parallel.comm import CommError
    from distributed_tensorflow_amd.parallel.rccl_native import Rendezvous
    from distributed_tensorflow_amd.train.server import _free_port
    port, world = _free_port(), 4
    out, errors = {}, []

    def run(rank):
        try:
            r = Rendezvous(rank, world, "127.0.0.1", port, timeout_s=20.0)
            table = r.all_gather_bytes(b"x" * (rank * 3) + bytes([rank]))
            root_says = r.broadcast_bytes(b"from two" if rank == 2 else None, src=2)
            r.barrier()
            big = r.all_gather_bytes(bytes([rank]) * 100000)
            out[rank] = (table, root_says, [len(b) for b in big], big[3][:2])
            r.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, e))

    threads = [threading.Thread(target=run, args=(r,)) for r in (3, 1, 0, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    assert not errors, errors
    want = [b"x" * (r * 3) + bytes([r]) for r in range(world)]
    for rank in range(world):
        assert out[rank] == (want, b"from two", [100000] * 4, bytes([3, 3]))
    # a missing peer: bounded failure on the ranks that did come
    port = _free_port()
    failed = []

    def lonely(rank):
        try:
            Rendezvous(rank, 3, "127.0.0.1", port, timeout_s=1.0)
        except CommError as e:
            failed.append((rank, str(e)))

    threads = [threading.Thread(target=lonely, args=(r,)) for r in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(15)
    assert len(failed) >= 1 and all("rendezvous" in m for _, m in failed)


def test_rccl_server_of_a_two_worker_cluster_never_becomes_a_world_of_one(monkeypatch):
    """A task of a declared two-worker cluster under ``protocol="rccl"`` either meets its peer or
    fails: it must not come up alone (the reference builds its server from the cluster alone,
    ``tf_distributed.py:9-18``; nothing of a launcher's environment is present here)."""
    for key in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DTF_COMM_BACKEND"):
        monkeypatch.delenv(key, raising=False)
    monkeypatch.setenv("DTF_RENDEZVOUS_TIMEOUT", "2")
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port(),
                                               "127.0.0.1:%d" % _free_port()]})
    t0 = time.monotonic()
    with pytest.raises(CommError):
        tf.train.Server(cluster, job_name="worker", task_index=1, protocol="rccl",
                        config=tf.ConfigProto(device="cpu"))
    assert time.monotonic() - t0 < 30
    # and both tasks together do form the world the cluster declares
    servers, errors = [None, None], []

    def task(i):
        try:
            servers[i] = tf.train.Server(cluster, job_name="worker", task_index=i,
                                         protocol="rccl", config=tf.ConfigProto(device="cpu"),
                                         comm=comms[i])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    comms = LoopbackComm.create_world(2)
    threads = [threading.Thread(target=task, args=(i,)) for i in (1, 0)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert [s.runtime.comm.world_size for s in servers] == [2, 2]
    for s in servers:
        s.stop()


def test_make_communicator_resolves_world_from_arguments_before_environment(monkeypatch):
    from distributed_tensorflow_amd.parallel.comm import make_communicator
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert isinstance(make_communicator(device="cpu"), SingleComm)
    assert isinstance(make_communicator(device="cpu", rank=0, world_size=1), SingleComm)
    with pytest.raises(ValueError):
        make_communicator(device="cpu", rank=2, world_size=2)
    with pytest.raises(ValueError):
        make_communicator(device="cpu", backend="mpi")
    with pytest.raises(CommError):           # a declared peer that never comes: an error, not a
        make_communicator(device="cpu", rank=1, world_size=2, master_addr="127.0.0.1",
                          master_port=_free_port(), timeout_s=1.5)     # world of one
    with pytest.raises(CommError):           # the native communicator needs a GPU device
        make_communicator(device="cpu", backend="rccl_native", rank=0, world_size=1)
