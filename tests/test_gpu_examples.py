"""The example scripts end to end on the GPU box: one worker per GPU through the launcher
(``--protocol rccl``; world size 1 on a one-GPU box), the whole user-facing flow -- flags,
ClusterSpec from the launcher environment, Server, replica_device_setter, Supervisor, HIP-graph
steps, device-resident data, event files."""
import os
import re
import subprocess
import sys

import pytest

from distributed_tensorflow_amd.summary import read_events

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def _launch(args, timeout=240):
    env = dict(os.environ, PYTHONPATH=ROOT)
    proc = subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.launch"] + args,
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          text=True, timeout=timeout)
    return proc.returncode, proc.stdout


@pytest.mark.parametrize("strategy", ["allreduce_sync", "reduce_scatter_sync"])
def test_mnist_example_on_gpu(strategy, tmp_path):
    """Without ``--data_seed`` the per-epoch shuffle is unseeded (as the reference's
    ``read_data_sets``), so the cost printed at step 100 is the loss of ANOTHER batch in every
    run and scatters by about +-1. With the seed the run is reproducible to the bit, and the
    captured run (HIP graph, feeds read through the step ring) must print exactly the costs of
    the run that issues every step kernel by kernel."""
    outs = {}
    for tag, extra, env in (("captured", [], {}),
                            ("eager", ["--hip_graphs", "false"], {"DTF_NO_RING_READS": "1"})):
        logs = str(tmp_path / ("logs_" + tag))
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            rc, out = _launch(["--num_workers", "1", "--protocol", "rccl", "--strategy", strategy,
                               "--timeout", "200", os.path.join(EX, "tf_distributed.py"), "--",
                               "--max_steps", "600", "--logs_path", logs, "--data_seed", "5",
                               "--data_dir", str(tmp_path / "nodata")] + extra)
        finally:
            for k, v in old.items():
                os.environ.pop(k) if v is None else os.environ.__setitem__(k, v)
        assert rc == 0, out
        assert "[worker:0] done" in out
        outs[tag] = re.findall(r"Cost: (\d+\.\d{4}),\s+AvgTime:\s*(\d+\.\d\d)ms", out)
    m = outs["captured"]
    assert len(m) >= 5
    costs = [float(c) for c, _ in m]
    assert costs == [float(c) for c, _ in outs["eager"]]         # every printed cost, to the digit
    assert costs[-1] < costs[0]                                   # and it learns
    assert min(float(t) for _, t in m) < 1.0                      # sub-millisecond steps
    acc = float(re.search(r"Test-Accuracy: (\d+\.\d\d)", out).group(1))
    assert 0.0 <= acc <= 1.0
    events = []
    for fname in os.listdir(logs):
        events += list(read_events(os.path.join(logs, fname)))
    steps = [e["step"] for e in events if e.get("summary")]
    assert len(steps) == 600 and steps[-1] == 600                 # one summary per step


def test_matmul_example_on_gpu(tmp_path):
    rc, out = _launch(["--num_workers", "1", "--protocol", "rccl", "--timeout", "200",
                       os.path.join(EX, "tf_distributed_1000Matrix.py"), "--",
                       "--max_steps", "301", "--logs_path", str(tmp_path / "logs")])
    assert rc == 0, out
    gflops = [float(g) for g in re.findall(r"GFLOPS: (\d+\.\d)", out)]
    assert gflops and max(gflops) > 5000.0                        # >5 TFLOP/s through sess.run
    assert "[worker:0] done" in out


def test_matmul_example_with_deferred_fetches_on_gpu(tmp_path):
    """``--deferred_fetches``: ``sess.run([checksum, summary_op])`` returns once the step is
    launched and the checksum is waited for when the reporter prints it; the printed value is the
    one a blocking run prints."""
    outs = []
    for extra in ([], ["--deferred_fetches"]):
        rc, out = _launch(["--num_workers", "1", "--protocol", "rccl", "--timeout", "200",
                           os.path.join(EX, "tf_distributed_1000Matrix.py"), "--",
                           "--max_steps", "301", "--logs_path", str(tmp_path / "logs")] + extra)
        assert rc == 0, out
        assert "[worker:0] done" in out
        outs.append(re.findall(r"Cost: (-?\d+\.\d+)", out))
    assert outs[0] and outs[0] == outs[1]


def test_linear_regression_example_on_gpu():
    rc, out = _launch(["--num_workers", "1", "--protocol", "rccl", "--strategy",
                       "reduce_scatter_sync", "--timeout", "200",
                       os.path.join(EX, "linear_regression.py"), "--", "--max_steps", "300",
                       "--features", "1024"])
    assert rc == 0, out
    err = float(re.search(r"Weight-Error: (\d+\.\d+)", out).group(1))
    assert err < 0.05


def test_device_resident_batches_match_host_batches():
    """``DataSet.to_device``: batches come from a per-epoch device-side shuffle (views of the
    permuted set) and must equal the host path's batches, including the one that straddles an
    epoch boundary."""
    import numpy as np
    import torch
    from distributed_tensorflow_amd.data.mnist import DataSet, synthetic_mnist
    images, labels = synthetic_mnist(230, seed=3)
    onehot = np.eye(10, dtype=np.float32)[labels]
    host = DataSet(images, onehot, seed=7)
    dev = DataSet(images, onehot, seed=7).to_device("cuda")
    for _ in range(12):                      # 230 = 2 * 100 + 30: boundaries fall inside batches
        hx, hy = host.next_batch(100)
        dx, dy = dev.next_batch(100)
        assert isinstance(dx, torch.Tensor) and dx.is_cuda
        assert np.array_equal(dx.cpu().numpy(), hx) and np.array_equal(dy.cpu().numpy(), hy)
    assert host.epochs_completed == dev.epochs_completed == 5


def test_captured_step_follows_variables_to_new_storage():
    """A HIP-graph capture holds raw addresses: when a variable is given new storage (here by
    adopting a fresh tensor, as a collective runtime does after re-initialisation) the next
    run must re-capture instead of updating the abandoned tensor."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    g = tf.Graph()
    with g.as_default():
        x = tf.placeholder(tf.float32, [None, 4])
        w = tf.Variable(tf.zeros([4, 1]), name="w")
        loss = tf.reduce_mean(tf.square(tf.matmul(x, w) - 1.0))
        train = tf.train.GradientDescentOptimizer(0.1).minimize(loss)
        init = tf.global_variables_initializer()
        with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
            sess.run(init)
            feed = {x: np.ones((8, 4), np.float32)}
            for _ in range(4):
                sess.run(train, feed)
            assert sess.stats["graph_replays"] >= 1
            captures = sess.stats["captures"]
            fresh = torch.zeros(4, 1, device="cuda")
            sess.variable_store.adopt(w.op.name, fresh)
            for _ in range(3):
                sess.run(train, feed)
            assert sess.stats["captures"] == captures + 1
            assert float(fresh.abs().sum()) > 0.0            # the adopted tensor is the one trained
            assert np.array_equal(sess.run(w), fresh.cpu().numpy())


@pytest.mark.gpu
def test_regression_batches_generated_on_device():
    """Device-resident batch stream: counter-based (two generators agree, consecutive batches
    differ), targets = x @ W* + b* + noise computed by the fp32 kernel."""
    import numpy as np
    import torch
    from distributed_tensorflow_amd.models.linear_regression import SyntheticRegressionData
    dev = torch.device("cuda")
    a = SyntheticRegressionData(64, 3, task=1).to_device(dev)
    b = SyntheticRegressionData(64, 3, task=1).to_device(dev)
    other = SyntheticRegressionData(64, 3, task=2).to_device(dev)
    x1, y1 = a.next_batch(128)
    x2, y2 = a.next_batch(128)
    bx1, by1 = b.next_batch(128)
    ox1, _ = other.next_batch(128)
    assert x1.is_cuda and y1.is_cuda and x1.shape == (128, 64) and y1.shape == (128, 3)
    assert torch.equal(x1, bx1) and torch.equal(y1, by1)
    assert not torch.equal(x1, x2) and not torch.equal(x1, ox1)
    want = x1.double().cpu().numpy() @ a.w.astype(np.float64) + a.b
    resid = y1.double().cpu().numpy() - want
    assert np.abs(resid).max() < 6 * a.noise and 0.5 * a.noise < resid.std() < 1.5 * a.noise
    assert abs(float(x1.mean())) < 0.05 and abs(float(x1.std()) - 1.0) < 0.05


def test_captured_step_gathers_device_feeds_through_the_address_table():
    """Inputs that already live in GPU memory reach a captured step through the pinned table of
    source addresses: different batch views (aligned and not) give the results of an eager
    session, also when a step fetches nothing and the next one follows immediately."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    g = tf.Graph()
    with g.as_default():
        x = tf.placeholder(tf.float32, [None, 6], name="x")
        y = tf.placeholder(tf.float32, [None, 3], name="y")
        w = tf.Variable(tf.zeros([6, 3]))
        acc = tf.Variable(tf.zeros([]))
        out = tf.reduce_sum(tf.matmul(x, w + 1.0) * y)
        bump = acc.assign_add(tf.reduce_sum(x))
        sess = tf.Session(config=tf.ConfigProto(device="cuda"))
        ref = tf.Session(config=tf.ConfigProto(device="cuda", hip_graphs=False))
        sess.run(tf.global_variables_initializer())
        ref.run(tf.global_variables_initializer())
        pool_x = torch.arange(40 * 6, dtype=torch.float32, device="cuda").reshape(40, 6) / 7
        pool_y = torch.arange(41 * 3, dtype=torch.float32, device="cuda").reshape(41, 3) / 5
        for i in range(12):
            bx = pool_x[(3 * i) % 30:(3 * i) % 30 + 10]
            by = pool_y[1 + i:11 + i]                      # rows of 12 bytes: mostly unaligned
            got = sess.run(out, {x: bx, y: by})
            want = ref.run(out, {x: bx.cpu().numpy(), y: by.cpu().numpy()})
            assert abs(got - want) <= 1e-4 * abs(want), (i, got, want)
        assert sess.stats["graph_replays"] >= 10
        cap = next(iter(next(p for p in sess._plans.values() if p.graphs).graphs.values()))
        assert cap.use_table and cap.ring.launched >= 10
        # steps without a fetch: back-to-back replays must each see their own batch
        total = 0.0
        for i in range(8):
            bx = pool_x[i:i + 10]
            sess.run(bump.op, {x: bx})
            total += float(bx.sum())
        assert abs(float(sess.run(acc)) - total) <= 1e-3 * total
        # small host data rides the same table (staged in pinned memory the kernel reads), also
        # mixed with a device tensor; changing values step to step must be seen
        for i in range(4):
            hx = (pool_x[i:i + 10] * (i + 1)).cpu().numpy()
            hy = pool_y[:10].cpu().numpy()
            got = sess.run(out, {x: hx, y: hy if i % 2 else pool_y[:10]})
            want = ref.run(out, {x: hx, y: hy})
            assert abs(got - want) <= 1e-4 * abs(want), i
        # a large host input (> 1 MiB) keeps the DMA staging path (its own capture)
        big_x = np.random.RandomState(0).rand(50000, 6).astype(np.float32)
        big_y = np.random.RandomState(1).rand(50000, 3).astype(np.float32)
        for _ in range(3):
            got = sess.run(out, {x: big_x, y: big_y})
        want = ref.run(out, {x: big_x, y: big_y})
        assert abs(got - want) <= 1e-3 * abs(want)
        kinds = {c.use_table for p in sess._plans.values() for c in p.graphs.values()}
        assert kinds == {True, False}


def test_captured_step_completion_polling_and_blocking_fallback(monkeypatch):
    """A captured step's results are read after polling the completion word; with a zero polling
    budget every step falls back to the stream synchronisation (and then skips polling for a
    while) -- the values are the same either way."""
    import numpy as np
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.client import session as S
    g = tf.Graph()
    with g.as_default():
        x = tf.placeholder(tf.float32, [None, 4], name="x")
        v = tf.Variable(tf.zeros([]))
        total = v.assign_add(tf.reduce_sum(x))
        sess = tf.Session(config=tf.ConfigProto(device="cuda"))
        sess.run(tf.global_variables_initializer())
        batch = np.ones((5, 4), np.float32)
        got = [float(sess.run(total, {x: batch * (i + 1)})) for i in range(6)]
        assert got == [20.0, 60.0, 120.0, 200.0, 300.0, 420.0]
        cap = next(iter(next(p for p in sess._plans.values() if p.graphs).graphs.values()))
        assert cap.layout and cap.ring.launched >= 4 and cap.ring.skip_spin == 0
        monkeypatch.setattr(S, "_SPIN_SECS", -1.0)       # no polling budget
        got = [float(sess.run(total, {x: batch})) for i in range(3)]
        assert got == [440.0, 460.0, 480.0]
        cap.ring.skip_spin = 2                             # as after a step that outlasted it
        got = [float(sess.run(total, {x: batch})) for i in range(2)]
        assert got == [500.0, 520.0] and cap.ring.skip_spin == 0     # two steps went unpolled
        assert float(sess.run(total, {x: batch})) == 540.0


def test_short_captured_steps_are_issued_call_by_call(monkeypatch):
    """A captured step of a few library calls (the reference's matmul step: product, mean, gather
    of the fetched scalar) is re-issued from its recorded call list instead of replayed as a HIP
    graph, after the list has been checked against the graph node for node; values equal those
    of the graph replay bit for bit, and a longer step stays a graph."""
    import numpy as np
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.client import session as S

    def run(max_calls):
        monkeypatch.setattr(S, "_DIRECT_MAX_CALLS", max_calls)
        g = tf.Graph()
        with g.as_default():
            tf.set_random_seed(1)
            a = tf.Variable(tf.random_normal([256, 256]))
            b = tf.Variable(tf.random_normal([256, 256]))
            x = tf.placeholder(tf.float32, [256, 256])
            checksum = tf.reduce_mean(tf.matmul(a + x, b))
            with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
                sess.run(tf.global_variables_initializer())
                feeds = [np.full((256, 256), float(i), np.float32) for i in range(8)]
                vals = [float(sess.run(checksum, {x: f})) for f in feeds]
                return vals, dict(sess.stats)

    direct_vals, direct_stats = run(8)
    graph_vals, graph_stats = run(0)
    assert direct_vals == graph_vals
    assert direct_stats.get("direct_steps", 0) >= 6 and direct_stats["torch_fallbacks"] == 0
    assert graph_stats.get("direct_steps", 0) == 0 and graph_stats["graph_replays"] >= 6
    # (graph_replays counts captured steps however they are launched)
    assert direct_stats["graph_replays"] == graph_stats["graph_replays"]


def test_fetch_ahead_of_the_optimizer_launch_matches_eager_values():
    """Captured training steps gather their small results before the optimizer launch (the step
    counter with its increment added). Loss, summaries and global_step must equal those of an
    eager session step for step; fetching a parameter the launch updates falls back to the usual
    order and returns the post-update value."""
    import numpy as np
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    rng = np.random.RandomState(3)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(6)]
    results = {}
    for label, graphs in (("graph", True), ("eager", False)):
        g = tf.Graph()
        with g.as_default():
            m = mnist_mlp.build_model(learning_rate=0.01, seed=1)
            sess = tf.Session(config=tf.ConfigProto(device="cuda", hip_graphs=graphs))
            sess.run(m.init_op)
            rows = []
            for bx, by in batches:
                _, loss, step = sess.run([m.train_op, m.loss, m.global_step],
                                         {m.x: bx, m.y_: by})
                rows.append((float(loss), float(step)))
            for bx, by in batches[:3]:
                _, loss, b2 = sess.run([m.train_op, m.loss, m.b2], {m.x: bx, m.y_: by})
                rows.append((float(loss), b2.tolist()))
            rows.append(sess.run(m.b2).tolist())
            rows.append(float(sess.run(m.global_step)))
            if graphs:
                assert sess.stats["graph_replays"] >= 6
            results[label] = rows
    ga, ea = results["graph"], results["eager"]
    assert [r[1] for r in ga[:6]] == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0] == [r[1] for r in ea[:6]]
    for a, b in zip(ga[:9], ea[:9]):
        assert abs(a[0] - b[0]) <= 1e-5 * max(1.0, abs(b[0]))
    for a, b in zip(ga[6:9], ea[6:9]):
        assert np.allclose(a[1], b[1], atol=1e-6)            # post-update b2 in both
    assert np.allclose(ga[9], ea[9], atol=1e-6) and ga[10] == ea[10] == 9.0
    assert np.allclose(ga[8][1], ga[9], atol=0)              # the fetched b2 was the final one


@pytest.mark.parametrize("shard", ["m", "k", "2d"])
def test_matmul_example_sharded_over_two_ranks_sharing_the_gpu(shard, tmp_path):
    """The reference's matmul graph (``tf_distributed_1000Matrix.py:33-48``) with the product
    sharded over 2 worker processes. The one-GPU box cannot host two RCCL ranks, so both share
    GPU 0 and the collectives of this rehearsal travel through gloo (staged host copies); the
    products run on the fp32 MFMA kernel. Row- and 2-D-sharded results must equal the single-rank
    product bit for bit."""
    import numpy as np

    def run(workers, mode, tag):
        dump = str(tmp_path / ("c_%s" % tag))
        env = dict(os.environ, PYTHONPATH=ROOT, DTF_COMM_BACKEND="gloo")
        proc = subprocess.run(
            [sys.executable, "-m", "distributed_tensorflow_amd.launch", "--num_workers",
             str(workers), "--protocol", "rccl", "--timeout", "200",
             os.path.join(EX, "tf_distributed_1000Matrix.py"), "--", "--max_steps", "101",
             "--shard", mode, "--logs_path", str(tmp_path / ("logs_" + tag)),
             "--dump_result", dump + "_{rank}.npy"],
            cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
            timeout=240)
        assert proc.returncode == 0, proc.stdout
        return proc.stdout, dump

    out1, dump1 = run(1, "none", "w1")
    out2, dump2 = run(2, shard, "w2")
    assert out2.count("Sharded variable") >= 2 and out2.count("] done") == 2
    assert re.search(r"AvgTime:\s*\d+\.\d\dms", out2)
    ref = np.load(dump1 + "_0.npy")
    for r in (0, 1):
        c = np.load(dump2 + "_%d.npy" % r)
        if shard == "k":
            assert np.abs(c - ref).max() < 1e-3 * np.abs(ref).max()
        else:
            assert np.array_equal(c, ref)


def test_example_graphs_select_no_torch_kernel_on_gpu():
    """Every op of the three example graphs (MNIST training + evaluation, the matrix product,
    linear regression) is planned onto a hand-written kernel or a pure view; a GPU session that
    meets anything else raises instead of running torch."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import linear_regression, mnist_mlp

    def check(sess, fetch_sets):
        for fetches, feed in fetch_sets:
            for _ in range(3):                      # eager warm-up, capture, replay
                sess.run(fetches, feed)
        assert sess.stats["torch_fallbacks"] == 0 and sess.fallback_log == []
        for plan in sess._plans.values():
            for st in plan.steps:
                from distributed_tensorflow_amd.framework.op_registry import get_opdef
                d = get_opdef(st.node.type)
                assert st.kernel is d.hip_kernel or st.node.type in (
                    "Const", "Identity", "StopGradient", "Reshape", "ReshapeLike", "NoOp",
                    "Placeholder", "VariableV2", "IsVariableInitialized"), st.node

    for fusion in (True, False):
        g = tf.Graph()
        with g.as_default():
            m = mnist_mlp.build_model()
            with tf.Session(config=tf.ConfigProto(device="cuda", graph_fusion=fusion)) as sess:
                sess.run(m.init_op)
                feed = {m.x: np.random.rand(100, 784).astype(np.float32),
                        m.y_: np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 100)]}
                check(sess, [([m.train_op, m.loss, m.summary_op, m.global_step], feed),
                             (m.accuracy, feed)])
    g = tf.Graph()
    with g.as_default():
        tf.set_random_seed(1)
        A = tf.Variable(tf.random_normal([1000, 1000]))
        B = tf.Variable(tf.random_normal([1000, 1000]))
        C = tf.matmul(A, B)
        chk = tf.reduce_mean(C)
        with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
            sess.run(tf.global_variables_initializer())
            check(sess, [(chk, None), (C, None)])
    g = tf.Graph()
    with g.as_default():
        m = linear_regression.build_model(64, 3, learning_rate=0.05)
        with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
            sess.run(m.init_op)
            feed = {m.x: np.random.rand(32, 64).astype(np.float32),
                    m.y_: np.random.rand(32, 3).astype(np.float32)}
            check(sess, [([m.train_op, m.loss, m.global_step], feed)])


def test_gpu_session_refuses_ops_without_a_kernel_unless_allowed():
    import numpy as np
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.ops.hip import HipKernelError
    g = tf.Graph()
    with g.as_default():
        x = tf.placeholder(tf.float32, [2, 3, 4])
        y = tf.transpose(x, [2, 0, 1])            # rank-3 permutation: no native kernel
        val = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
        with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
            with pytest.raises(HipKernelError):
                sess.run(y, {x: val})
        with tf.Session(config=tf.ConfigProto(device="cuda", allow_torch_fallback=True)) as sess:
            assert np.array_equal(sess.run(y, {x: val}), val.transpose(2, 0, 1))
            assert sess.stats["torch_fallbacks"] >= 1
            assert sess.fallback_log[0][0] == "Transpose"


def test_small_ops_run_native_kernels_through_the_session():
    """AssignAdd / AssignSub, Transpose, Sign, reduction and broadcast gradients, log-softmax
    gradient, truncated normal, integer casts and an fp8 ``tf.matmul``: all through
    ``Session.run`` on the GPU with fallbacks forbidden, against numpy."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    rng = np.random.RandomState(0)
    g = tf.Graph()
    with g.as_default():
        tf.set_random_seed(3)
        x = tf.placeholder(tf.float32, [6, 5])
        v = tf.Variable(tf.zeros([6, 5]))
        inc, dec = v.assign_add(x), v.assign_sub(tf.constant(0.5))
        xt = tf.transpose(x)
        sg = tf.sign(x)
        loss = tf.reduce_sum(tf.reduce_mean(tf.nn.log_softmax(x) * x, axis=1)) \
            + tf.reduce_mean(tf.abs(x)) + tf.reduce_sum(tf.reduce_sum(x, axis=0) * 2.0)
        gx, = tf.gradients(loss, [x])
        tn = tf.truncated_normal([4000], stddev=0.1)
        as_int = tf.cast(tf.cast(x * 10.0, tf.int32), tf.float32)
        a8 = tf.cast(tf.placeholder(tf.float32, [64, 128], name="a8") , tf.float8_e4m3fn)
        b8 = tf.cast(tf.placeholder(tf.float32, [128, 32], name="b8"), tf.float8_e4m3fn)
        c8 = tf.matmul(a8, b8)
        xv = rng.randn(6, 5).astype(np.float32)
        with tf.Session(config=tf.ConfigProto(device="cuda")) as sess:
            sess.run(tf.global_variables_initializer())
            sess.run(inc, {x: xv})
            sess.run(inc, {x: xv})
            sess.run(dec)
            assert np.allclose(sess.run(v), 2 * xv - 0.5)
            assert np.array_equal(sess.run(xt, {x: xv}), xv.T)
            assert np.array_equal(sess.run(sg, {x: xv}), np.sign(xv))
            xt_ = torch.tensor(xv, requires_grad=True)
            ref = (torch.log_softmax(xt_, -1) * xt_).mean(1).sum() + xt_.abs().mean() \
                + (xt_.sum(0) * 2.0).sum()
            ref.backward()
            assert np.allclose(sess.run(gx, {x: xv}), xt_.grad.numpy(), atol=1e-5)
            t = sess.run(tn)
            assert np.abs(t).max() <= 0.2 + 1e-6 and 0.07 < t.std() < 0.1
            assert np.array_equal(sess.run(as_int, {x: xv}), np.trunc(xv * 10.0))
            av = rng.randint(-2, 3, (64, 128)).astype(np.float32)
            bv = rng.randint(-2, 3, (128, 32)).astype(np.float32)
            c = sess.run(c8, {g.get_tensor_by_name("a8:0"): av, g.get_tensor_by_name("b8:0"): bv})
            assert np.array_equal(c, av @ bv)
            assert sess.stats["torch_fallbacks"] == 0


def test_traced_step_uses_hip_events_and_does_not_synchronise_between_steps(monkeypatch):
    """``RunOptions(trace_level=FULL_TRACE)`` on the GPU: every plan step carries a device-side
    duration from HIP events, the host's enqueue time is recorded separately, and the whole traced
    run synchronises the device twice (when the tracer takes its base time stamp and when it
    resolves the events) -- not once or twice per step."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    g = tf.Graph()
    with g.as_default():
        m = mnist_mlp.build_model()
        with tf.Session(config=tf.ConfigProto(device="cuda", step_fusion=False)) as sess:
            sess.run(m.init_op)
            feed = {m.x: np.random.rand(100, 784).astype(np.float32),
                    m.y_: np.eye(10, dtype=np.float32)[np.random.randint(0, 10, 100)]}
            fetches = [m.train_op, m.loss, m.global_step]
            for _ in range(3):
                sess.run(fetches, feed)
            calls = {"n": 0}
            real_sync = torch.cuda.synchronize

            def counting_sync(*a, **k):
                calls["n"] += 1
                return real_sync(*a, **k)
            monkeypatch.setattr(torch.cuda, "synchronize", counting_sync)
            md = tf.RunMetadata()
            sess.run(fetches, feed, options=tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE),
                     run_metadata=md)
            monkeypatch.setattr(torch.cuda, "synchronize", real_sync)
    stats = md.step_stats
    assert len(stats) >= 4 and calls["n"] == 2
    assert all(s["clock"] == "hip_event" for s in stats)
    kernels = [s for s in stats if s["op"] in ("FusedDense", "ClassifierHeadTrain", "MatMul",
                                               "MultiApplySGD")]
    assert len(kernels) >= 3
    assert all(0.0 < s["end"] - s["start"] < 5e-3 for s in kernels)
    # steps follow each other on the device in plan order, without a host round trip in between
    starts = [s["start"] for s in stats]
    assert starts == sorted(starts)
    trace = md.to_chrome_trace()
    assert {"host enqueue"} <= {e["tid"] for e in trace["traceEvents"]}


def _mnist_run(config, steps, device_feeds=True, look=True):
    """``steps`` training steps of the MNIST MLP on fixed synthetic batches; returns the fetched
    (loss, summary, global_step) per step, the final variables and the session's plans."""
    import numpy as np
    import torch
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    rng = np.random.RandomState(11)
    batches = [(rng.rand(100, 784).astype(np.float32),
                np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]) for _ in range(5)]
    if device_feeds:
        batches = [(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in batches]
    g = tf.Graph()
    with g.as_default():
        m = mnist_mlp.build_model(seed=1)
        with tf.Session(config=config) as sess:
            sess.run(m.init_op)
            hist = []
            for i in range(steps):
                bx, by = batches[i % len(batches)]
                out = sess.run([m.train_op, m.loss, m.summary_op, m.global_step],
                               feed_dict={m.x: bx, m.y_: by})[1:]
                hist.append([float(out[0]), bytes(out[1]), float(out[2])] if look else out)
            acc = float(sess.run(m.accuracy, feed_dict={m.x: batches[0][0], m.y_: batches[0][1]}))
            variables = sess.run([m.w1, m.b1, m.w2, m.b2])
            plans = [([st.node.type for st in p.steps if st.node.type != "VariableV2"],
                      [(c.use_table, bool(c.layout), c.ring.launched) for c in p.graphs.values()])
                     for p in sess._plans.values() if p.mutates_state and p.feed_slots]
            stats = dict(sess.stats)
    return hist, variables, acc, plans, stats


def test_mnist_step_fusion_is_one_node_and_bit_equal_to_the_kernel_chain_for_100_steps():
    """The MNIST training step through the facade with the whole-step node (three launches: layer
    1, head, update) and with the chain of fused kernels it replaces: loss, summary bytes, step
    counter, accuracy and all four variables are bit-equal after 100 steps, with feeds resident in
    GPU memory (read through the step ring, no gather launch) and with numpy feeds."""
    import numpy as np
    import distributed_tensorflow_amd as tf
    ref = _mnist_run(tf.ConfigProto(device="cuda", step_fusion=False), 100)
    for device_feeds in (True, False):
        new = _mnist_run(tf.ConfigProto(device="cuda"), 100, device_feeds=device_feeds)
        assert new[0] == ref[0]
        assert new[2] == ref[2]
        for p, q in zip(ref[1], new[1]):
            assert np.array_equal(p, q)
        types, caps = new[3][0]
        assert types == ["MLPTrainStep"]
        use_table, has_layout, launched = caps[0]
        assert has_layout and use_table and launched >= 98
        assert new[4]["torch_fallbacks"] == 0
    assert np.isfinite(ref[0][-1][0]) and ref[0][-1][2] == 100.0


def test_deferred_fetches_return_the_blocking_values_without_blocking():
    """``deferred_fetches``: ``run`` hands back DeferredValue objects and never waits; 50 steps
    are issued without looking at any result (more than three times round the 16-slot ring), and
    afterwards every value -- loss, merged summary, step counter -- equals the blocking run's."""
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.client.deferred import DeferredValue
    ref = _mnist_run(tf.ConfigProto(device="cuda"), 50)
    new = _mnist_run(tf.ConfigProto(device="cuda", deferred_fetches=True), 50, look=False)
    assert all(isinstance(v, DeferredValue) for v in new[0][-1])
    got = [[float(o[0]), bytes(o[1]), float(o[2])] for o in new[0]]
    assert got == ref[0]
    assert new[2] == ref[2]
    # the chain of fused kernels (generic gather launch) defers the same way
    chain = _mnist_run(tf.ConfigProto(device="cuda", deferred_fetches=True, step_fusion=False),
                       50, look=False)
    assert [[float(o[0]), bytes(o[1]), float(o[2])] for o in chain[0]] == ref[0]


def test_histogram_summary_of_the_weights_in_a_gpu_training_step(tmp_path):
    """A histogram of W1 (78 400 values) fetched next to the training op: the tensor reaches the
    host op with the step's other results, the step still trains, no torch kernel is planned."""
    import numpy as np
    import distributed_tensorflow_amd as tf
    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.summary import decode_histograms
    rng = np.random.RandomState(0)
    bx = rng.rand(100, 784).astype(np.float32)
    by = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]
    with tf.Graph().as_default():
        m = mnist_mlp.build_model(seed=1, with_summaries=False)
        tf.summary.histogram("w1", m.w1)
        tf.summary.scalar("cost", m.loss)
        merged = tf.summary.merge_all()
        with tf.Session(config=tf.ConfigProto(device="cuda:0")) as sess:
            sess.run(m.init_op)
            for _ in range(5):
                _, data, w1 = sess.run([m.train_op, merged, m.w1], {m.x: bx, m.y_: by})
            (tag, h), = decode_histograms(bytes(data))
            assert tag == "w1" and h["num"] == 78400
            assert np.isclose(h["sum"], float(np.asarray(w1, dtype=np.float64).sum()), rtol=1e-5)
            assert sess.stats["torch_fallbacks"] == 0 and not sess.fallback_log


def test_monitored_session_example_on_gpu(tmp_path):
    """``examples/mnist_monitored.py`` on the GPU under the real RCCL backend (world of one):
    dense layers + integer labels + Adam + SyncReplicasOptimizer + hooks, captured steps."""
    ckpt = str(tmp_path / "ckpt")
    rc, out = _launch(["--num_workers", "1", "--protocol", "rccl", "--timeout", "200",
                       os.path.join(EX, "mnist_monitored.py"), "--", "--last_step", "400",
                       "--optimizer", "adam", "--checkpoint_dir", ckpt,
                       "--data_dir", str(tmp_path / "nodata")])
    assert rc == 0, out
    assert "Global Step: 400  (local steps: 400)" in out, out
    acc = float(re.search(r"Test-Accuracy: (\d\.\d\d)", out).group(1))
    assert acc > 0.5, out
    files = os.listdir(ckpt)
    assert "model.ckpt-400.index" in files
    tags = {t for f in files if "tfevents" in f
            for e in read_events(os.path.join(ckpt, f)) for t, _ in e.get("summary", [])}
    assert {"cost", "accuracy"} <= tags
