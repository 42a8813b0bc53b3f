"""T0: graph construction, shape inference, Session.run contract, autodiff, fusion."""
import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.models import mnist_mlp

CPU = dict(device="cpu")


def _session(**kw):
    return tf.Session(config=tf.ConfigProto(device="cpu", **kw))


def test_reference_variable_names_and_shapes():
    m = mnist_mlp.build_model()
    names = [v.op.name for v in tf.global_variables()]
    assert names == ["global_step", "weights/Variable", "weights/Variable_1",
                     "biases/Variable", "biases/Variable_1"]
    assert [v.op.name for v in tf.trainable_variables()] == names[1:]
    assert m.global_step.dtype == tf.float32 and m.global_step.shape == ()
    assert m.x.shape == (None, 784) and m.y.shape == (None, 10) and m.loss.shape == ()
    assert m.x.op.name == "input/x-input" and m.y_.name == "input/y-input:0"
    assert mnist_mlp.NUM_PARAMS == 79510
    with pytest.raises(ValueError):
        tf.get_variable("global_step", [])


def test_name_scopes_unique_names_and_lookup():
    with tf.name_scope("a"):
        x = tf.constant(1.0, name="c")
        y = tf.constant(2.0, name="c")
        with tf.name_scope("b"):
            z = tf.add(x, y, name="sum")
    with tf.name_scope("a"):
        w = tf.constant(3.0, name="c")
    assert (x.op.name, y.op.name, z.op.name, w.op.name) == ("a/c", "a/c_1", "a/b/sum", "a_1/c")
    g = tf.get_default_graph()
    assert g.get_tensor_by_name("a/b/sum:0") is z
    assert g.get_operation_by_name("a/c") is x.op
    with pytest.raises(ValueError):
        tf.constant(1.0, name="bad name")


def test_shape_inference_and_type_errors():
    a = tf.placeholder(tf.float32, [None, 4])
    b = tf.placeholder(tf.float32, [4, 3])
    assert tf.matmul(a, b).shape == (None, 3)
    assert tf.matmul(a, a, transpose_a=True).shape == (4, 4)
    assert (a + tf.zeros([4])).shape == (None, 4)
    assert tf.reduce_sum(a, axis=[1]).shape == (None,)
    assert tf.reduce_mean(a).shape == () and tf.reduce_sum(a, 0, keepdims=True).shape == (1, 4)
    assert tf.argmax(a, 1).shape == (None,) and tf.argmax(a, 1).dtype == tf.int64
    assert tf.reshape(tf.zeros([6, 4]), [-1, 8]).shape == (3, 8)
    with pytest.raises(ValueError):
        tf.matmul(a, tf.zeros([5, 3]))
    with pytest.raises(TypeError):
        a + tf.placeholder(tf.int32, [4])
    with pytest.raises(ValueError):
        tf.zeros([3]) + tf.zeros([4])
    with pytest.raises(TypeError):
        bool(a)


def test_session_fetch_structures_and_feeds():
    x = tf.placeholder(tf.float32, [None, 2], name="x")
    y = x * 2.0
    op = tf.no_op()
    with _session() as sess:
        v = np.array([[1., 2.], [3., 4.]], np.float32)
        out = sess.run(y, {x: v})
        assert isinstance(out, np.ndarray) and np.allclose(out, v * 2)
        assert sess.run(op) is None
        res = sess.run([y, op, None, {"a": y, "b": (y, op)}], feed_dict={x: v})
        assert res[1] is None and res[2] is None
        assert np.allclose(res[3]["b"][0], v * 2) and res[3]["b"][1] is None
        assert np.allclose(sess.run("mul:0", {"x:0": v}), v * 2)       # by name
        assert np.allclose(sess.run(y, {y: v}), v)                       # feeding an intermediate
        assert sess.run(tf.reduce_sum(x), {x: v}).shape == ()
        with pytest.raises(RuntimeError, match="feed a value for placeholder"):
            sess.run(y)
        with pytest.raises(ValueError, match="Cannot feed value of shape"):
            sess.run(y, {x: np.zeros((2, 3), np.float32)})
        with pytest.raises(TypeError):
            sess.run(object())
    with pytest.raises(RuntimeError, match="closed"):
        sess.run(y, {x: v})


def test_uninitialized_variable_and_initializer():
    v = tf.Variable(tf.ones([3]))
    w = tf.Variable(tf.zeros([2]))
    ready = tf.report_uninitialized_variables()
    with _session() as sess:
        with pytest.raises(tf.FailedPreconditionError):
            sess.run(v)
        assert list(sess.run(ready)) == [b"Variable", b"Variable_1"]
        sess.run(v.initializer)
        assert list(sess.run(ready)) == [b"Variable_1"]
        assert not sess.run(tf.is_variable_initialized(w))
        sess.run(tf.global_variables_initializer())
        assert len(sess.run(ready)) == 0
        assert np.allclose(sess.run(v.assign_add(tf.ones([3]))), 2.0)
        assert np.allclose(sess.run(v.assign(tf.fill([3], 7.0))), 7.0)
        assert np.allclose(v.eval(session=sess), 7.0)


def test_pruning_eval_does_not_train():
    m = mnist_mlp.build_model()
    rng = np.random.RandomState(0)
    bx, by = rng.rand(20, 784).astype(np.float32), np.eye(10, dtype=np.float32)[rng.randint(0, 10, 20)]
    with _session() as sess:
        sess.run(m.init_op)
        w_before = sess.run(m.w1).copy()
        for _ in range(3):
            sess.run(m.accuracy, {m.x: bx, m.y_: by})
        assert sess.run(m.global_step) == 0 and np.array_equal(sess.run(m.w1), w_before)
        sess.run(m.train_op, {m.x: bx, m.y_: by})
        assert sess.run(m.global_step) == 1 and not np.array_equal(sess.run(m.w1), w_before)
        eval_plan = next(p for p in sess._plans.values()
                         if any(s.node.type == "ArgmaxAccuracy" for s in p.steps))
        assert not any(s.node.type in ("MultiApplySGD", "ApplyGradientDescent", "AssignAdd")
                       for s in eval_plan.steps)        # the eval plan holds no update op


def test_random_ops_are_seeded_and_counter_based():
    tf.set_random_seed(1)
    a = tf.random_normal([1000, 10])
    b = tf.random_normal([1000, 10])
    u = tf.random_uniform([5000], -2.0, 3.0)
    t = tf.truncated_normal([5000], stddev=2.0)
    with _session() as s1, _session() as s2:
        va, vb = s1.run([a, b])
        assert np.array_equal(va, s2.run(a))                   # same seed -> same values
        assert not np.allclose(va, vb)                          # different ops -> different streams
        assert abs(va.mean()) < 0.05 and abs(va.std() - 1) < 0.05
        vu, vt = s1.run([u, t])
        assert vu.min() >= -2 and vu.max() <= 3 and abs(vu.mean() - 0.5) < 0.1
        assert np.abs(vt).max() <= 4.0 + 1e-5 and abs(vt.std() - 2 * 0.88) < 0.1


def _torch_mlp_step(w1, w2, b1, b2, x, y, lr):
    ps = [torch.tensor(p, requires_grad=True) for p in (w1, w2, b1, b2)]
    a2 = torch.sigmoid(torch.tensor(x) @ ps[0] + ps[2])
    yhat = torch.softmax(a2 @ ps[1] + ps[3], dim=1)
    loss = (-(torch.tensor(y) * torch.log(yhat)).sum(1)).mean()
    loss.backward()
    return loss.item(), [(p - lr * p.grad).detach().numpy() for p in ps]


@pytest.mark.parametrize("fuse", [False, True])
def test_mlp_training_matches_torch_autograd(fuse):
    m = mnist_mlp.build_model(learning_rate=0.1)
    rng = np.random.RandomState(3)
    bx = (rng.rand(32, 784) * 0.1).astype(np.float32)
    by = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 32)]
    with _session(graph_fusion=fuse) as sess:
        sess.run(m.init_op)
        params = sess.run([m.w1, m.w2, m.b1, m.b2])
        for _ in range(3):
            ref_loss, params = _torch_mlp_step(*params, bx, by, 0.1)
            _, loss = sess.run([m.train_op, m.loss], {m.x: bx, m.y_: by})
            assert abs(loss - ref_loss) < 1e-4 * max(1, abs(ref_loss))
            for ours, ref in zip(sess.run([m.w1, m.w2, m.b1, m.b2]), params):
                assert np.allclose(ours, ref, atol=2e-5)
        types = {s.node.type for p in sess._plans.values() for s in p.steps}
        assert ("FusedDense" in types) == fuse and ("MultiApplySGD" in types) == fuse


def test_fused_and_unfused_plans_agree_bitwise_on_cpu():
    m = mnist_mlp.build_model(learning_rate=0.05)
    rng = np.random.RandomState(5)
    bx = (rng.rand(100, 784) * 0.1).astype(np.float32)
    by = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]
    outs = []
    for fuse in (False, True):
        with _session(graph_fusion=fuse) as sess:
            sess.run(m.init_op)
            hist = [sess.run([m.train_op, m.loss, m.global_step, m.accuracy],
                             {m.x: bx, m.y_: by})[1:] for _ in range(4)]
            outs.append((hist, sess.run(m.w2)))
            if fuse:
                # the whole step is four compute nodes: layer 1, the classifier head (layer 2 +
                # loss + every gradient but dW1 + accuracy), dW1, and one multi-tensor SGD
                plan = [p for p in sess._plans.values() if any(
                    st.node.type == "MultiApplySGD" for st in p.steps)][0]
                types = [st.node.type for st in plan.steps if st.node.type != "VariableV2"]
                assert types == ["FusedDense", "ClassifierHeadTrain", "MatMul", "MultiApplySGD"]
    for (l0, s0, a0), (l1, s1, a1) in zip(outs[0][0], outs[1][0]):
        assert abs(l0 - l1) < 1e-5 and s0 == s1 and a0 == a1
    assert np.allclose(outs[0][1], outs[1][1], atol=1e-6)
    # a fetched variable is materialised when the step has finished: it includes the increment
    assert [h[1] for h in outs[0][0]] == [1.0, 2.0, 3.0, 4.0]


def test_step_fusion_replaces_the_whole_mlp_step_and_changes_no_bit_on_cpu():
    """With ``step_fusion`` the training step of the two-layer classifier is ONE plan node; its
    results and parameters equal the four-node plan's bit for bit. The node is not used when
    something it would swallow is fetched (the first layer's weight gradient here)."""
    m = mnist_mlp.build_model(learning_rate=0.05)
    rng = np.random.RandomState(5)
    bx = (rng.rand(100, 784) * 0.1).astype(np.float32)
    by = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]
    outs = []
    for step_fusion in (False, True):
        with _session(step_fusion=step_fusion) as sess:
            sess.run(m.init_op)
            hist = [sess.run([m.train_op, m.loss, m.global_step, m.accuracy],
                             {m.x: bx, m.y_: by})[1:] for _ in range(5)]
            outs.append((hist, sess.run([m.w1, m.b1, m.w2, m.b2])))
            plan = [p for p in sess._plans.values() if p.mutates_state and p.feed_slots][0]
            types = [st.node.type for st in plan.steps if st.node.type != "VariableV2"]
            if step_fusion:
                assert types == ["MLPTrainStep"]
                dw1 = tf.gradients(m.loss, [m.w1])[0]
                sess.run([m.train_op, dw1], {m.x: bx, m.y_: by})
                other = [p for p in sess._plans.values() if p is not plan and p.mutates_state
                         and p.feed_slots][0]
                assert "MLPTrainStep" not in [st.node.type for st in other.steps]
            else:
                assert types == ["FusedDense", "ClassifierHeadTrain", "MatMul", "MultiApplySGD"]
    assert outs[0][0] == outs[1][0]
    for p, q in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(p, q)


def test_deferred_value_behaves_as_the_value_it_stands_for():
    from distributed_tensorflow_amd.client.deferred import DeferredValue, resolve
    calls = []

    def thunk():
        calls.append(1)
        return np.float32(2.5)

    v = DeferredValue(thunk)
    assert not v.ready and calls == []
    assert float(v) == 2.5 and v.ready and calls == [1]
    assert "%.2f" % v == "2.50" and "{:.1f}".format(v) == "2.5" and str(v) == "2.5"
    assert v + 1 == 3.5 and 2 * v == 5.0 and v > 2 and not v < 2 and int(v) == 2
    assert np.isfinite(v) and np.asarray(v).dtype == np.float32 and v.dtype == np.float32
    assert np.allclose([v, v], 2.5) and resolve(v) == np.float32(2.5) and resolve(7) == 7
    assert calls == [1]
    arr = DeferredValue(lambda: np.arange(6, dtype=np.float32).reshape(2, 3))
    assert arr.shape == (2, 3) and len(arr) == 2 and arr[1][2] == 5.0 and float(arr.sum()) == 15.0
    assert np.concatenate([arr, arr]).shape == (4, 3)
    blob = DeferredValue(lambda: b"abc")
    assert bytes(blob) == b"abc"


def test_gradients_of_assorted_ops_match_torch():
    x = tf.placeholder(tf.float32, [4, 3])
    w = tf.Variable(tf.random_normal([3, 5], seed=7))
    b = tf.Variable(tf.fill([5], 0.3))
    h = tf.tanh(tf.matmul(x, w) + b)
    r = tf.nn.relu(h) * tf.exp(-tf.square(h)) / (tf.sqrt(tf.abs(h) + 1.0))
    loss = tf.reduce_sum(tf.reduce_mean(r, axis=0)) + tf.reduce_mean(tf.nn.log_softmax(h))
    gw, gb, gx = tf.gradients(loss, [w, b, x])
    xv = np.random.RandomState(0).randn(4, 3).astype(np.float32)
    with _session() as sess:
        sess.run(tf.global_variables_initializer())
        wv, bv = sess.run([w, b])
        ours = sess.run([gw, gb, gx], {x: xv})
    tx, tw, tb = (torch.tensor(v, requires_grad=True) for v in (xv, wv, bv))
    th = torch.tanh(tx @ tw + tb)
    tr = torch.relu(th) * torch.exp(-th ** 2) / torch.sqrt(th.abs() + 1.0)
    (tr.mean(0).sum() + torch.log_softmax(th, 1).mean()).backward()
    for o, t in zip(ours, (tw.grad, tb.grad, tx.grad)):
        assert np.allclose(o, t.numpy(), atol=1e-5)


def test_softmax_cross_entropy_op_and_linear_regression():
    logits = tf.placeholder(tf.float32, [None, 4])
    labels = tf.placeholder(tf.float32, [None, 4])
    xe = tf.nn.softmax_cross_entropy_with_logits(labels=labels, logits=logits)
    (g,) = tf.gradients(tf.reduce_mean(xe), [logits])
    lv = np.random.RandomState(1).randn(6, 4).astype(np.float32)
    yv = np.eye(4, dtype=np.float32)[[0, 1, 2, 3, 0, 1]]
    with _session() as sess:
        xe_v, g_v = sess.run([xe, g], {logits: lv, labels: yv})
    tl = torch.tensor(lv, requires_grad=True)
    ref = torch.nn.functional.cross_entropy(tl, torch.tensor(yv).argmax(1), reduction="none")
    ref.mean().backward()
    assert np.allclose(xe_v, ref.detach().numpy(), atol=1e-5)
    assert np.allclose(g_v, tl.grad.numpy(), atol=1e-6)


def test_graph_def_serialisation_and_finalize():
    m = mnist_mlp.build_model()
    data = tf.get_default_graph().as_graph_def()
    assert b"weights/Variable" in data and b"MatMul" in data
    g = tf.get_default_graph()
    g.finalize()
    with pytest.raises(RuntimeError, match="finalized"):
        tf.constant(1.0)
    with _session() as sess:          # running a finalized graph is fine
        sess.run(m.init_op)
        assert sess.run(m.global_step) == 0


def test_run_metadata_trace():
    m = mnist_mlp.build_model()
    md = tf.RunMetadata()
    rng = np.random.RandomState(0)
    with _session() as sess:
        sess.run(m.init_op)
        sess.run(m.loss, {m.x: rng.rand(5, 784).astype(np.float32),
                          m.y_: np.eye(10, dtype=np.float32)[:5]},
                 options=tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE), run_metadata=md)
    trace = md.to_chrome_trace()
    assert any(e["cat"] in ("FusedDense", "MatMul") for e in trace["traceEvents"])
    assert all(e["dur"] >= 0 for e in trace["traceEvents"])


def test_make_callable_runs_the_same_plan_as_run():
    x = tf.placeholder(tf.float32, [None, 3])
    w = tf.Variable(tf.fill([3, 2], 0.5))
    y = tf.matmul(x, w)
    total = tf.reduce_sum(y)
    with _session() as sess:
        sess.run(tf.global_variables_initializer())
        f = sess.make_callable([y, total], feed_list=[x])
        bx = np.arange(6, dtype=np.float32).reshape(2, 3)
        got = f(bx)
        want = sess.run([y, total], {x: bx})
        assert isinstance(got, list) and np.array_equal(got[0], want[0]) and got[1] == want[1]
        assert len(sess._plans) == 2                      # init + the one training plan, shared
        g = sess.make_callable(total, [x], accept_options=True)
        md = tf.RunMetadata()
        assert g(bx, options=tf.RunOptions(trace_level=tf.RunOptions.FULL_TRACE),
                 run_metadata=md) == want[1] and md.step_stats
        with pytest.raises(TypeError):
            f(bx, bx)
        with pytest.raises(TypeError):
            f(bx, options=tf.RunOptions())


def test_one_hot_sparse_xent_l2_clip_and_dense_layer():
    """Facade compositions: integer-label cross-entropy equals the one-hot spelling, ``l2_loss``
    and ``clip_by_value`` differentiate, ``tf.layers.dense`` creates named variables and trains."""
    rng = np.random.RandomState(0)
    bx = rng.randn(12, 6).astype(np.float32)
    labels = rng.randint(0, 4, 12).astype(np.int64)
    with tf.Graph().as_default():
        tf.set_random_seed(2)
        x = tf.placeholder(tf.float32, [None, 6])
        y = tf.placeholder(tf.int64, [None])
        h = tf.layers.dense(x, 5, activation=tf.nn.sigmoid, name="hidden")
        logits = tf.layers.dense(h, 4, name="out")
        sparse = tf.nn.sparse_softmax_cross_entropy_with_logits(labels=y, logits=logits)
        dense = tf.nn.softmax_cross_entropy_with_logits(labels=tf.one_hot(y, 4), logits=logits)
        names = sorted(v.op.name for v in tf.trainable_variables())
        assert names == ["hidden/bias", "hidden/kernel", "out/bias", "out/kernel"]
        w = [v for v in tf.trainable_variables() if v.op.name == "out/kernel"][0]
        loss = tf.add(tf.reduce_mean(sparse), tf.scale(tf.nn.l2_loss(w), 0.01))
        clipped = tf.clip_by_value(logits, -0.05, 0.05)
        dclip = tf.gradients(tf.reduce_sum(clipped), [logits])[0]
        train = tf.train.GradientDescentOptimizer(0.5).minimize(loss)
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(tf.global_variables_initializer())
            a, b, oh = sess.run([sparse, dense, tf.one_hot(y, 4)], {x: bx, y: labels})
            assert np.allclose(a, b) and a.shape == (12,)
            assert np.array_equal(oh, np.eye(4, dtype=np.float32)[labels])
            assert np.array_equal(sess.run(tf.one_hot(tf.constant([1, 7, -1]), 3, 5.0, -1.0)),
                                  np.array([[-1, 5, -1], [-1, -1, -1], [-1, -1, -1]], np.float32))
            lg, cl, dc = sess.run([logits, clipped, dclip], {x: bx})
            assert np.allclose(cl, np.clip(lg, -0.05, 0.05))
            assert np.array_equal(dc, ((lg > -0.05) & (lg < 0.05)).astype(np.float32))
            wv = sess.run(w)
            assert np.isclose(sess.run(tf.nn.l2_loss(w)), 0.5 * (wv ** 2).sum(), rtol=1e-6)
            first = sess.run(loss, {x: bx, y: labels})
            for _ in range(60):
                sess.run(train, {x: bx, y: labels})
            assert sess.run(loss, {x: bx, y: labels}) < 0.6 * first


def test_gradient_clipping_by_global_norm_between_compute_and_apply():
    rng = np.random.RandomState(1)
    bx = (10 * rng.randn(8, 5)).astype(np.float32)
    with tf.Graph().as_default():
        tf.set_random_seed(7)
        x = tf.placeholder(tf.float32, [None, 5])
        w = tf.Variable(tf.random_normal([5, 3]), name="W")
        b = tf.Variable(tf.ones([3]), name="b")
        loss = tf.reduce_sum(tf.square(tf.add(tf.matmul(x, w), b)))
        opt = tf.train.GradientDescentOptimizer(0.1)
        gv = opt.compute_gradients(loss)
        clipped, norm = tf.clip_by_global_norm([g for g, _ in gv], 1.5)
        train = opt.apply_gradients(list(zip(clipped, [v for _, v in gv])))
        one = tf.clip_by_norm(gv[0][0], 0.5)
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(tf.global_variables_initializer())
            raw = sess.run([g for g, _ in gv], {x: bx})
            got, n, single = sess.run([clipped, norm, one], {x: bx})
            want_norm = np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in raw))
            assert np.isclose(n, want_norm, rtol=1e-5) and want_norm > 1.5
            for g, c in zip(raw, got):
                assert np.allclose(c, g * 1.5 / want_norm, rtol=1e-5)
            assert np.isclose(np.sqrt((single.astype(np.float64) ** 2).sum()), 0.5, rtol=1e-5)
            w0, b0 = sess.run([w, b])
            sess.run(train, {x: bx})
            w1, b1 = sess.run([w, b])
            step = np.sqrt(((w1 - w0) ** 2).sum() + ((b1 - b0) ** 2).sum())
            assert np.isclose(step, 0.1 * 1.5, rtol=1e-4)       # lr x clipped norm


def test_where_comparisons_expand_dims_and_squeeze():
    rng = np.random.RandomState(3)
    a = rng.randn(4, 5).astype(np.float32)
    b = rng.randn(4, 5).astype(np.float32)
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [None, 5])
        y = tf.placeholder(tf.float32, [None, 5])
        picked = tf.where(tf.greater_equal(x, y), x, y)                  # = maximum
        leaky = tf.where(tf.less_equal(x, 0.0), tf.scale(x, 0.1), x)     # scalar threshold
        gx, gy = tf.gradients(tf.reduce_sum(picked), [x, y])
        gl = tf.gradients(tf.reduce_sum(leaky), [x])[0]
        e = tf.expand_dims(x, 1)
        s = tf.squeeze(tf.expand_dims(tf.expand_dims(x, 0), -1))
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            p, l, dgx, dgy, dl, ev, sv = sess.run([picked, leaky, gx, gy, gl, e, s], {x: a, y: b})
    assert np.array_equal(p, np.maximum(a, b))
    assert np.allclose(l, np.where(a <= 0, 0.1 * a, a))
    assert np.array_equal(dgx, (a >= b).astype(np.float32))
    assert np.array_equal(dgy, (a < b).astype(np.float32))
    assert np.allclose(dl, np.where(a <= 0, 0.1, 1.0))
    assert ev.shape == (4, 1, 5) and sv.shape == (4, 5) and np.array_equal(sv, a)


def test_reduce_max_gradient_min_and_argmin():
    a = np.array([[1.0, 5.0, 5.0, -2.0], [0.5, -1.0, 0.25, 0.5], [3.0, 2.0, 1.0, 0.0]], np.float32)
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [None, 4])
        row_max = tf.reduce_max(x, axis=1)
        g_row = tf.gradients(tf.reduce_sum(tf.multiply(row_max, [1.0, 2.0, 3.0])), [x])[0]
        g_all = tf.gradients(tf.reduce_max(x), [x])[0]
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            rm, gr, ga, mn, am, cmin = sess.run(
                [row_max, g_row, g_all, tf.reduce_min(x, axis=1), tf.argmin(x, 1),
                 tf.reduce_min(x)], {x: a})
    assert np.array_equal(rm, a.max(1)) and np.array_equal(mn, a.min(1)) and cmin == a.min()
    assert np.array_equal(am, a.argmin(1))
    want = np.array([[0, 0.5, 0.5, 0], [1.0, 0, 0, 1.0], [3.0, 0, 0, 0]], np.float32)
    assert np.allclose(gr, want)                                  # ties share the gradient
    assert np.allclose(ga, (a == a.max()) / (a == a.max()).sum())


def _dropout_runs(device, runs=7, keep=0.6):
    a = np.random.RandomState(2).randn(50, 37).astype(np.float32)
    with tf.Graph().as_default():
        tf.set_random_seed(11)
        x = tf.placeholder(tf.float32, [None, 37])
        kp = tf.placeholder(tf.float32, [])
        y = tf.nn.dropout(x, keep)
        y_fed = tf.nn.dropout(x, kp)
        gx = tf.gradients(tf.reduce_sum(tf.multiply(y, 3.0)), [x])[0]
        with tf.Session(config=tf.ConfigProto(device=device)) as sess:
            outs = [sess.run([y, gx], {x: a}) for _ in range(runs)]
            ident = sess.run(y_fed, {x: a, kp: 1.0})
            half = sess.run(y_fed, {x: a, kp: 0.5})
            stats = dict(sess.stats)
    return a, outs, ident, half, stats


def test_dropout_draws_a_fresh_reproducible_mask_every_run():
    a, outs, ident, half, _ = _dropout_runs("cpu")
    masks = [np.where(a != 0, y / a, 0) for y, _ in outs]
    for (y, g), m in zip(outs, masks):
        assert np.all(np.isclose(m, 0.0) | np.isclose(m, 1 / 0.6, rtol=1e-5))
        assert abs((m > 0).mean() - 0.6) < 0.05
        assert np.allclose(g, 3.0 * m, rtol=1e-6)                 # the backward pass reuses the mask
    assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2])
    assert np.array_equal(ident, a)
    assert abs((half != 0).mean() - 0.5) < 0.06
    _, again, _, _, _ = _dropout_runs("cpu")
    assert all(np.array_equal(p[0], q[0]) for p, q in zip(outs, again))   # same seed, same rounds


def test_mean_squared_error_is_one_op_with_its_own_gradient():
    rng = np.random.RandomState(4)
    a, b = rng.randn(9, 3).astype(np.float32), rng.randn(9, 3).astype(np.float32)
    with tf.Graph().as_default() as g:
        x = tf.placeholder(tf.float32, [None, 3])
        t = tf.placeholder(tf.float32, [None, 3])
        w = tf.Variable(tf.ones([3, 3]), name="W")
        pred = tf.matmul(x, w)
        loss = tf.losses.mean_squared_error(t, pred)
        plain = tf.reduce_mean(tf.square(tf.subtract(pred, t)))
        gw, gt = tf.gradients(loss, [w, t])
        gw_plain = tf.gradients(plain, [w])[0]
        assert loss.op.type == "MeanSquaredError"
        row = tf.losses.mean_squared_error(tf.placeholder(tf.float32, [3]), pred)   # broadcast
        assert row.op.type != "MeanSquaredError"
        with tf.Session(config=tf.ConfigProto(device="cpu")) as sess:
            sess.run(tf.global_variables_initializer())
            l, lp, dw, dwp, dt = sess.run([loss, plain, gw, gw_plain, gt], {x: a, t: b})
    pred_v = a @ np.ones((3, 3), np.float32)
    assert np.isclose(l, ((pred_v - b) ** 2).mean(), rtol=1e-6) and np.isclose(l, lp, rtol=1e-6)
    assert np.allclose(dw, dwp, rtol=1e-5, atol=1e-6)
    assert np.allclose(dt, -2.0 * (pred_v - b) / b.size, rtol=1e-5, atol=1e-7)


def test_pow_gradient_for_base_and_exponent():
    """``tf.pow`` / ``**`` differentiate with respect to both operands (broadcast exponent
    included); where the base is not positive the exponent's gradient is 0, as in TensorFlow."""
    xv = np.abs(np.random.RandomState(0).randn(3, 4)).astype(np.float32) + 0.1
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [3, 4])
        y = tf.Variable(tf.constant(np.full((4,), 1.5, np.float32)))
        z = tf.reduce_sum(tf.pow(x, y)) + tf.reduce_sum(x ** 2.0)
        gx, gy = tf.gradients(z, [x, y])
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            a, b = sess.run([gx, gy], {x: xv})
            neg = sess.run(tf.gradients(tf.reduce_sum(tf.pow(x, 2.0 * tf.ones([4]))), [x])[0],
                           {x: -xv})
    xt = torch.tensor(xv, requires_grad=True)
    yt = torch.full((4,), 1.5, requires_grad=True)
    (torch.pow(xt, yt).sum() + (xt ** 2).sum()).backward()
    assert np.allclose(a, xt.grad.numpy(), rtol=1e-5) and np.allclose(b, yt.grad.numpy(), rtol=1e-5)
    assert np.allclose(neg, -2.0 * xv, rtol=1e-5)


def test_cos_and_sin_with_gradients():
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [5])
        y = tf.reduce_sum(tf.cos(x) * 2.0 + tf.sin(x))
        (dx,) = tf.gradients(y, [x])
        v = np.linspace(-3.0, 3.0, 5).astype(np.float32)
        with tf.Session() as sess:
            got_y, got_dx = sess.run([y, dx], {x: v})
    assert np.isclose(got_y, (2 * np.cos(v) + np.sin(v)).sum(), rtol=1e-5)
    assert np.allclose(got_dx, -2 * np.sin(v) + np.cos(v), rtol=1e-5, atol=1e-6)


def _logistic_run(device=None):
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(16, 6, generator=g) * 8).numpy()        # large values: no overflow
    targets = (torch.rand(16, 6, generator=g) > 0.5).float().numpy()
    wv = torch.randn(6, 3, generator=g).numpy()
    with tf.Graph().as_default():
        x = tf.placeholder(tf.float32, [16, 6])
        z = tf.placeholder(tf.float32, [16, 6])
        w = tf.Variable(tf.constant(wv))
        b = tf.Variable(tf.fill([3], 0.5))
        loss = tf.losses.sigmoid_cross_entropy(z, x)
        l1 = tf.losses.absolute_difference(z, tf.nn.relu6(x))
        dense = tf.nn.xw_plus_b(x, w, b)
        (dx,) = tf.gradients(loss, [x])
        cfg = tf.ConfigProto(device=device) if device else None
        with tf.Session(config=cfg) as sess:
            sess.run(tf.global_variables_initializer())
            for _ in range(3):                    # (a GPU session replays the captured step)
                out = sess.run([loss, dx, l1, dense, tf.nn.softplus(x)], {x: logits, z: targets})
            stats = dict(sess.stats)
    return logits, targets, wv, out, stats


def _check_logistic(logits, targets, wv, out):
    lt, zt = torch.from_numpy(logits).double().requires_grad_(), torch.from_numpy(targets).double()
    want = torch.nn.functional.binary_cross_entropy_with_logits(lt, zt)
    want.backward()
    assert np.isclose(out[0], want.item(), rtol=1e-5)
    assert np.allclose(out[1], lt.grad.numpy(), rtol=1e-4, atol=1e-7)
    assert np.isclose(out[2], np.abs(np.clip(logits, 0, 6) - targets).mean(), rtol=1e-5)
    assert np.allclose(out[3], logits @ wv + 0.5, rtol=1e-4, atol=1e-4)
    assert np.allclose(out[4], torch.nn.functional.softplus(lt.detach()).numpy(), rtol=1e-5, atol=1e-7)
    assert np.isfinite(out[4]).all()


def test_logistic_loss_softplus_relu6_and_xw_plus_b():
    logits, targets, wv, out, _ = _logistic_run()
    _check_logistic(logits, targets, wv, out)


@pytest.mark.gpu
def test_gpu_logistic_loss_runs_on_the_hip_kernels():
    logits, targets, wv, out, stats = _logistic_run("cuda:0")
    _check_logistic(logits, targets, wv, out)
    assert stats["torch_fallbacks"] == 0


def test_gc_pause_is_a_counted_process_wide_guard():
    """Thread A enters the pause (collection was on), thread B enters and leaves while A is still
    inside: collection must stay off until A leaves, and come back on afterwards."""
    import gc
    import threading
    from distributed_tensorflow_amd.client.session import _gc_paused
    assert gc.isenabled()
    a_in, b_done, a_go = threading.Event(), threading.Event(), threading.Event()
    seen = {}

    def thread_a():
        with _gc_paused():
            a_in.set()
            b_done.wait(10)
            seen["inside_a_after_b_left"] = gc.isenabled()
            a_go.wait(10)
        seen["after_a"] = gc.isenabled()

    def thread_b():
        a_in.wait(10)
        with _gc_paused():
            seen["inside_b"] = gc.isenabled()
        b_done.set()

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(), tb.start()
    tb.join(10)
    a_go.set()
    ta.join(10)
    assert seen == {"inside_b": False, "inside_a_after_b_left": False, "after_a": True}
    gc.disable()
    try:                                   # off before: stays off after
        with _gc_paused():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
