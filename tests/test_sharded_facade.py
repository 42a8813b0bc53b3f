"""Sharded products through the facade: ``tf.matmul`` of two variables under a collective
runtime is planned as a sharded GEMM (parallel/planner.py, parallel/sharded_ops.py).

T0: the planner's decisions on small graphs. T1: N worker sessions as threads of one process
over the loopback communicator -- results against a single-rank session, variable storage,
fetching and checkpointing sharded variables, and the initial-state exchange after a restore."""
import os
import threading

import numpy as np
import pytest
import torch

import distributed_tensorflow_amd as tf
from distributed_tensorflow_amd.framework.graph import Graph
from distributed_tensorflow_amd.parallel.comm import LoopbackComm
from distributed_tensorflow_amd.parallel.planner import (block_of, plan_matmul_sharding,
                                                         rank_grid)
from distributed_tensorflow_amd.train.server import _free_port


def _graph(m=64, k=32, n=48, extra_reader=False, const_init=False, dtype=tf.float32):
    g = Graph()
    with g.as_default():
        tf.set_random_seed(1)
        a0 = tf.fill([m, k], 0.5) if const_init else tf.random_normal([m, k], dtype=dtype)
        A = tf.Variable(a0)
        B = tf.Variable(tf.random_normal([k, n], dtype=dtype))
        C = tf.matmul(A, B)
        extra = tf.reduce_sum(A) if extra_reader else None
        init = tf.global_variables_initializer()
    return g, A, B, C, extra, init


# -- T0: planner -------------------------------------------------------------------------------
def test_rank_grid_and_blocks():
    assert [rank_grid(w) for w in (1, 2, 4, 6, 8)] == [(1, 1), (1, 2), (2, 2), (2, 3), (2, 4)]
    assert block_of(1000, 4, 3) == (750, 250)


def test_planner_modes_and_layouts():
    g, A, B, C, _, _ = _graph()
    assert not plan_matmul_sharding(g, 1, "m")
    assert not plan_matmul_sharding(g, 4, "none")
    with pytest.raises(ValueError):
        plan_matmul_sharding(g, 2, "rows")
    pm = plan_matmul_sharding(g, 4, "m")
    assert list(pm.matmuls) == [C.op] and set(pm.layouts) == {A.op.name}
    lay = pm.layouts[A.op.name]
    assert (lay.axis, lay.parts, lay.stored, lay.local_shape()) == (0, 4, True, (16, 32))
    assert lay.window(2, 4) == (32, 0, 16, 32)
    assert lay.init[0] is True and lay.init[3:] == (0.0, 1.0)          # N(0,1) window recipe
    assert pm.init_of[A.initializer] is lay
    pk = plan_matmul_sharding(g, 4, "k")
    assert pk.layouts[A.op.name].key() == (1, 4, "world")
    assert pk.layouts[B.op.name].key() == (0, 4, "world")
    p2 = plan_matmul_sharding(g, 8, "2d")                                # 2 x 4 grid
    assert p2.layouts[A.op.name].key() == (0, 2, "grid_row")
    assert p2.layouts[B.op.name].key() == (1, 4, "grid_col")
    assert p2.layouts[B.op.name].window(5, 8) == (0, 12, 32, 12)         # rank 5 = (1, 1)
    assert p2.layouts[A.op.name].window(5, 8) == (32, 0, 32, 32)


def test_planner_keeps_variables_with_other_readers_replicated():
    g, A, B, C, extra, _ = _graph(extra_reader=True)
    plan = plan_matmul_sharding(g, 2, "m")
    assert C.op in plan.matmuls                      # the product still runs sharded ...
    assert plan.layouts[A.op.name].stored is False   # ... on a view of the replica
    assert plan.stored(A.op.name) is None and not plan.init_of
    g2, A2, *_ = _graph()
    assert plan_matmul_sharding(g2, 2, "m", shard_variables=False).stored(A2.op.name) is None


def test_planner_skips_products_that_do_not_divide():
    g, A, B, C, _, _ = _graph(m=63)
    assert not plan_matmul_sharding(g, 2, "m")
    assert plan_matmul_sharding(g, 2, "k")           # K = 32 still divides


# -- T1: loopback worlds ----------------------------------------------------------------------
def _run_world(world, mode, build=_graph, body=None, shard_variables=True, **build_kw):
    """One session per rank (thread) over the loopback communicator; returns body's results."""
    comms = LoopbackComm.create_world(world, timeout_s=60.0) if world > 1 else [None]
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                               for _ in range(world)]})
    results, errors = [None] * world, []

    def work(r):
        try:
            g, A, B, C, extra, init = build(**build_kw)
            cfg = tf.ConfigProto(device="cpu", matmul_sharding=mode,
                                 shard_variables=shard_variables)
            server = tf.train.Server(cluster, job_name="worker", task_index=r, protocol="rccl",
                                     config=cfg, comm=comms[r])
            with g.as_default():
                sv = tf.train.Supervisor(graph=g, is_chief=(r == 0), init_op=init,
                                         global_step=None)
                with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
                    results[r] = body(sess, server, g, A, B, C, extra, r)
                sv.stop()
            server.stop()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            for c in comms:
                if c is not None:
                    c.kill()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    if errors:
        raise errors[0]
    return results


def _fetch_all(sess, server, g, A, B, C, extra, r):
    out = {"c": sess.run(C), "a": sess.run(A), "b": sess.run(B),
           "a_local": tuple(sess.variable_store.read(A.op.name).shape),
           "b_local": tuple(sess.variable_store.read(B.op.name).shape),
           "layouts": sorted(getattr(server.runtime, "var_layouts", {})),
           "gathered": getattr(server.runtime, "bytes_gathered", 0),
           "reduced": getattr(server.runtime, "bytes_reduced", 0)}
    if extra is not None:
        out["extra"] = sess.run(extra)
    return out


@pytest.mark.parametrize("world,mode", [(2, "m"), (4, "m"), (2, "k"), (4, "k"), (2, "2d"),
                                        (4, "2d"), (8, "2d")])
def test_sharded_matmul_matches_single_rank(world, mode):
    single = _run_world(1, "none", body=_fetch_all)[0]
    assert single["layouts"] == []
    res = _run_world(world, mode, body=_fetch_all)
    p, q = rank_grid(world)
    for r, out in enumerate(res):
        if mode == "k":
            np.testing.assert_allclose(out["c"], single["c"], rtol=1e-5, atol=1e-5)
        else:
            assert np.array_equal(out["c"], single["c"])         # same k-order per element
        # a fetched sharded variable is gathered back to the whole, identical everywhere
        assert np.array_equal(out["a"], single["a"]) and np.array_equal(out["b"], single["b"])
        want_a = {"m": (64 // world, 32), "k": (64, 32 // world),
                  "2d": (64 // p, 32) if p > 1 else (64, 32)}[mode]
        want_b = {"m": (32, 48), "k": (32 // world, 48), "2d": (32, 48 // q)}[mode]
        assert out["a_local"] == want_a and out["b_local"] == want_b
        if mode == "k":
            assert out["reduced"] > 0
        else:
            assert out["gathered"] > 0


def test_variable_with_another_reader_stays_whole_but_product_is_sharded():
    single = _run_world(1, "none", body=_fetch_all, extra_reader=True)[0]
    res = _run_world(2, "m", body=_fetch_all, extra_reader=True)
    for out in res:
        assert out["layouts"] == [] and out["a_local"] == (64, 32)
        assert out["gathered"] > 0                   # the all-gather of C still happened
        assert np.array_equal(out["c"], single["c"])
        assert out["extra"] == single["extra"]


def test_constant_initial_value_is_broadcast_whole_then_cut():
    res = _run_world(2, "m", body=_fetch_all, const_init=True)
    for out in res:
        assert out["layouts"] == ["Variable"] and out["a_local"] == (32, 32)
        assert np.all(out["a"] == 0.5) and out["a"].shape == (64, 32)


def test_shard_variables_off_multiplies_views_of_replicas():
    single = _run_world(1, "none", body=_fetch_all)[0]
    for out in _run_world(2, "k", body=_fetch_all, shard_variables=False):
        assert out["layouts"] == [] and out["a_local"] == (64, 32) and out["b_local"] == (32, 48)
        np.testing.assert_allclose(out["c"], single["c"], rtol=1e-5, atol=1e-5)


def test_checkpoint_of_sharded_variables_round_trips(tmp_path):
    """Saving gathers the blocks (every worker calls save), the checkpoint holds whole tensors
    under the usual keys, and a restore leaves each rank with its block again."""
    def body(sess, server, g, A, B, C, extra, r):
        with g.as_default():
            saver = tf.train.Saver()
        before = sess.run(C)
        prefix = saver.save(sess, os.path.join(str(tmp_path), "rank%d" % r, "model.ckpt"))
        from distributed_tensorflow_amd.train.saver import read_bundle
        saved = read_bundle(prefix)
        store = sess.variable_store
        store.read(A.op.name).zero_()                      # wreck the local block ...
        assert not np.array_equal(sess.run(C), before)
        saver.restore(sess, prefix)                        # ... and get it back
        return {"saved_a": tuple(saved[A.op.name].shape), "same": np.array_equal(sess.run(C), before),
                "a_local": tuple(store.read(A.op.name).shape)}

    for out in _run_world(2, "m", body=body):
        assert out == {"saved_a": (64, 32), "same": True, "a_local": (32, 32)}


def test_restore_on_chief_reaches_every_rank_as_blocks(tmp_path):
    """prepare_session restores on the chief only; the exchange that follows broadcasts the whole
    restored tensor once and every rank keeps its block."""
    first = _run_world(1, "none", body=lambda sess, server, g, A, B, C, e, r: (
        tf.train.Saver().save(sess, str(tmp_path / "ck" / "model.ckpt")), sess.run(C)))[0]
    prefix, c_ref = first

    comms = LoopbackComm.create_world(2, timeout_s=60.0)
    cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port() for _ in range(2)]})
    out, errors = [None, None], []

    def work(r):
        try:
            g, A, B, C, _, init = _graph()
            cfg = tf.ConfigProto(device="cpu", matmul_sharding="m")
            server = tf.train.Server(cluster, job_name="worker", task_index=r, protocol="rccl",
                                     config=cfg, comm=comms[r])
            with g.as_default():
                # poison the seeded initial values so that only a restore explains a match
                sv = tf.train.Supervisor(graph=g, is_chief=(r == 0), init_op=init,
                                         global_step=None, logdir=str(tmp_path / "ck"),
                                         save_model_secs=0)
                with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
                    out[r] = (sess.run(C), tuple(sess.variable_store.read(A.op.name).shape))
                sv.stop()
            server.stop()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            [c.kill() for c in comms]

    threads = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    if errors:
        raise errors[0]
    for c, a_local in out:
        assert a_local == (32, 32) and np.array_equal(c, c_ref)


def test_bf16_products_shard_too():
    single = _run_world(1, "none", body=_fetch_all, dtype=tf.bfloat16)[0]
    for mode in ("m", "k"):
        for out in _run_world(2, mode, body=_fetch_all, dtype=tf.bfloat16):
            np.testing.assert_allclose(out["c"], single["c"], rtol=2e-2, atol=2e-2)
            assert out["layouts"]


# -- planner: fp8, transposed right-hand variables, sizing ---------------------------------------
def test_planner_plans_fp8_2d_blocks_and_sizes_them_against_hbm():
    """BASELINE configuration 5 as a graph: ``tf.matmul`` of two 65536 x 65536 fp8 variables on
    8 ranks plans as 2-D blocks (2 x 4 grid); the planner's byte count per GPU is far inside
    288 GB, and nothing of the size is allocated to find that out."""
    g = Graph()
    with g.as_default():
        n = 65536
        A = tf.Variable(tf.random_normal([n, n], dtype=tf.float8_e4m3fn))
        B = tf.Variable(tf.random_normal([n, n], dtype=tf.float8_e4m3fn))
        C = tf.matmul(A, B)
    plan = plan_matmul_sharding(g, 8, "2d")
    spec = plan.matmuls[C.op]
    assert (spec.mode, spec.grid, spec.b_layout) == ("2d", (2, 4), "kn")
    assert plan.layouts[A.op.name].local_shape() == (32768, 65536)
    assert plan.layouts[B.op.name].local_shape() == (65536, 16384)
    by = spec.bytes_per_rank(8)
    gib = 1 << 30
    assert by["a"] == 2 * gib and by["b"] == 1 * gib and by["b_image"] == 1 * gib
    assert by["c"] == 8 * gib                       # bf16 result, whole on every rank
    assert by["total"] < 288e9 * 0.1
    one = plan_matmul_sharding(g, 8, "m").matmuls[C.op].bytes_per_rank(8)
    assert one["b"] == 4 * gib and one["a"] == gib // 2


def test_planner_takes_a_right_hand_variable_stored_transposed():
    g = Graph()
    with g.as_default():
        A = tf.Variable(tf.random_normal([64, 32]))
        Bt = tf.Variable(tf.random_normal([48, 32]))
        C = tf.matmul(A, Bt, transpose_b=True)
    pk = plan_matmul_sharding(g, 4, "k")
    assert pk.matmuls[C.op].b_layout == "nk" and (pk.matmuls[C.op].n, pk.matmuls[C.op].k) == (48, 32)
    assert pk.layouts[Bt.op.name].key() == (1, 4, "world")        # K blocks = column blocks
    p2 = plan_matmul_sharding(g, 8, "2d")
    assert p2.layouts[Bt.op.name].key() == (0, 4, "grid_col")     # N blocks = row blocks


def _graph_nk(m=64, k=32, n=48, dtype=tf.float32):
    g = Graph()
    with g.as_default():
        tf.set_random_seed(1)
        A = tf.Variable(tf.random_normal([m, k], dtype=dtype))
        B = tf.Variable(tf.random_normal([n, k], dtype=dtype))
        C = tf.matmul(A, B, transpose_b=True)
        init = tf.global_variables_initializer()
    return g, A, B, C, None, init


@pytest.mark.parametrize("world,mode", [(2, "m"), (4, "k"), (4, "2d"), (8, "2d")])
def test_sharded_matmul_with_transposed_right_hand_variable(world, mode):
    single = _run_world(1, "none", build=_graph_nk, body=_fetch_all)[0]
    for out in _run_world(world, mode, build=_graph_nk, body=_fetch_all):
        if mode == "k":
            np.testing.assert_allclose(out["c"], single["c"], rtol=1e-5, atol=1e-5)
        else:
            assert np.array_equal(out["c"], single["c"])
        assert np.array_equal(out["b"], single["b"])


def test_sharded_node_runs_on_the_benchmark_engine_with_panels():
    """The node is an adapter over ``ShardedMatmul``: with ``matmul_panels=2`` a 2-rank row-sharded
    product gathers two panels (panel-major) and hands out row-major C, bit-equal to one rank."""
    from distributed_tensorflow_amd.parallel.sharded_matmul import ShardedMatmul

    def body(sess, server, g, A, B, C, extra, r):
        c = sess.run(C)
        engines = [v for k, v in sess._ctx.buffers.items()
                   if isinstance(k, tuple) and len(k) == 2 and k[1] == "engine"]
        assert len(engines) == 1 and isinstance(engines[0], ShardedMatmul)
        return c, engines[0].strategy, engines[0].panels, sess.run(C)

    def run(world, mode, panels):
        comms = LoopbackComm.create_world(world, timeout_s=60.0) if world > 1 else [None]
        cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()
                                                   for _ in range(world)]})
        results, errors = [None] * world, []

        def work(r):
            try:
                g, A, B, C, extra, init = _graph(m=64, k=32, n=48)
                cfg = tf.ConfigProto(device="cpu", matmul_sharding=mode, matmul_panels=panels)
                server = tf.train.Server(cluster, job_name="worker", task_index=r,
                                         protocol="rccl", config=cfg, comm=comms[r])
                with g.as_default():
                    sv = tf.train.Supervisor(graph=g, is_chief=(r == 0), init_op=init,
                                             global_step=None)
                    with sv.prepare_or_wait_for_session(server.target, config=cfg) as sess:
                        results[r] = body(sess, server, g, A, B, C, extra, r)
                    sv.stop()
                server.stop()
            except BaseException as e:  # noqa: BLE001
                errors.append(e)
                [c.kill() for c in comms if c is not None]

        threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in threads]
        [t.join(120) for t in threads]
        if errors:
            raise errors[0]
        return results

    single = _run_world(1, "none", body=_fetch_all)[0]["c"]
    for mode, strategy in (("m", "m_shard_allgather"), ("2d", "block_2d_allgather")):
        for c, strat, panels, again in run(2, mode, 2):
            assert strat == strategy and panels == 2
            assert np.array_equal(c, single) and np.array_equal(again, single)


def test_fetching_the_product_op_runs_the_sharded_node():
    """``sess.run(C.op)`` (no value fetched) must run the planned sharded product, not the
    original op on the rank's blocks."""
    def body(sess, server, g, A, B, C, extra, r):
        before = server.runtime.bytes_gathered
        assert sess.run(C.op) is None
        return server.runtime.bytes_gathered - before

    for moved in _run_world(2, "m", body=body):
        assert moved == 64 * 48 * 4
