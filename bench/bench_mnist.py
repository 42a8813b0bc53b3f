#!/usr/bin/env python3
"""Single-GPU step-latency benchmarks of the reference's two workloads through the facade.

* ``mnist``: the MNIST MLP training step of ``tf_distributed.py:111`` -- ``sess.run([train_op,
  cross_entropy, summary_op, global_step], feed_dict=batch of 100)`` -- as the reference's
  AvgTime (ms per step) and steps/s, with HIP-graph replay on and off, the whole-step node on
  and off, blocking and deferred fetches, plus the 10 000-image evaluation pass of ``:126``.
* ``matmul``: ``sess.run`` of the 1000 x 1000 fp32 product of ``tf_distributed_1000Matrix.py:48``
  (operands resident in HBM) as ms per step and GFLOP/s.

Synthetic data of the reference's shapes; weights random-initialised with the reference's seed.
One JSON line per measurement (and ``--out`` file).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402


def _sync(device: str) -> None:
    if device.startswith("cuda"):
        torch.cuda.synchronize()


def bench_mnist(device: str, steps: int, warmup: int, hip_graphs: bool, fusion: bool,
                batch: int = 100, device_data: bool = False, step_fusion: bool = True,
                deferred: bool = False, optimizer: str = "sgd") -> dict:
    g = tf.Graph()
    with g.as_default():
        opt = {"sgd": None, "momentum": lambda: tf.train.MomentumOptimizer(0.0005, 0.9),
               "adam": lambda: tf.train.AdamOptimizer(0.001)}[optimizer]
        model = mnist_mlp.build_model(seed=1, optimizer=opt() if opt else None)
        rng = np.random.RandomState(0)
        batches = [(rng.rand(batch, 784).astype(np.float32),
                    np.eye(10, dtype=np.float32)[rng.randint(0, 10, batch)]) for _ in range(16)]
        test_x = rng.rand(10000, 784).astype(np.float32)
        test_y = np.eye(10, dtype=np.float32)[rng.randint(0, 10, 10000)]
        if device_data:      # data set resident in HBM: feeds are device tensors
            batches = [(torch.from_numpy(bx).to(device), torch.from_numpy(by).to(device))
                       for bx, by in batches]
            test_x, test_y = torch.from_numpy(test_x).to(device), torch.from_numpy(test_y).to(device)
        cfg = tf.ConfigProto(device=device, hip_graphs=hip_graphs, graph_fusion=fusion,
                             step_fusion=step_fusion and device.startswith("cuda"),
                             deferred_fetches=deferred)
        with tf.Session(config=cfg) as sess:
            sess.run(model.init_op)
            fetches = [model.train_op, model.loss, model.summary_op, model.global_step]

            def one(i):
                bx, by = batches[i % len(batches)]
                return sess.run(fetches, feed_dict={model.x: bx, model.y_: by})

            for i in range(warmup):
                one(i)
            _sync(device)
            t0 = time.perf_counter()
            for i in range(steps):
                out = one(i)
            final_cost = float(out[1])       # (deferred fetches: waits for the last step)
            _sync(device)
            train_ms = (time.perf_counter() - t0) * 1e3 / steps
            fd = {model.x: test_x, model.y_: test_y}
            for _ in range(3):
                sess.run(model.accuracy, feed_dict=fd)
            _sync(device)
            t0 = time.perf_counter()
            for _ in range(10):
                acc = sess.run(model.accuracy, feed_dict=fd)
            _sync(device)
            eval_ms = (time.perf_counter() - t0) * 1e3 / 10
            stats = dict(getattr(sess, "stats", {}))
    return {"bench": "mnist_mlp_train_step", "device": device, "batch": batch,
            "hip_graphs": hip_graphs, "fusion": fusion, "device_data": device_data,
            "step_fusion": bool(step_fusion and fusion and device.startswith("cuda")),
            "deferred_fetches": deferred, "optimizer": optimizer,
            "ms_per_step": round(train_ms, 4),
            "steps_per_s": round(1e3 / train_ms, 1), "eval_10k_ms": round(eval_ms, 3),
            "final_cost": final_cost, "global_step": float(out[3]), "accuracy": float(acc),
            "session_stats": {k: v for k, v in stats.items() if isinstance(v, (int, float))}}


def bench_matmul(device: str, steps: int, warmup: int, n: int, hip_graphs: bool,
                 deferred: bool = False) -> dict:
    g = tf.Graph()
    with g.as_default():
        tf.set_random_seed(1)
        a = tf.Variable(tf.random_normal([n, n]))
        b = tf.Variable(tf.random_normal([n, n]))
        c = tf.matmul(a, b)
        checksum = tf.reduce_mean(c)
        init = tf.global_variables_initializer()
        cfg = tf.ConfigProto(device=device, hip_graphs=hip_graphs, deferred_fetches=deferred)
        with tf.Session(config=cfg) as sess:
            sess.run(init)
            for _ in range(warmup):
                sess.run(checksum)
            _sync(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                v = sess.run(checksum)
            v = float(v)                     # (deferred fetches: waits for the last step)
            _sync(device)
            ms = (time.perf_counter() - t0) * 1e3 / steps
            # device-side op only (no fetch): what the GEMM itself costs inside the step
            t0 = time.perf_counter()
            for _ in range(steps):
                sess.run(c.op)
            _sync(device)
            ms_nofetch = (time.perf_counter() - t0) * 1e3 / steps
    return {"bench": "matmul_%d_f32_step" % n, "device": device, "hip_graphs": hip_graphs,
            "deferred_fetches": deferred, "ms_per_step": round(ms, 4), "gflops": round(2.0 * n ** 3 / ms / 1e6, 1),
            "ms_per_step_no_fetch": round(ms_nofetch, 4),
            "gflops_no_fetch": round(2.0 * n ** 3 / ms_nofetch / 1e6, 1), "checksum": float(v)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--which", default="mnist,matmul")
    ap.add_argument("--out", default="")
    ap.add_argument("--optimizers", default="sgd",
                    help="comma-separated: sgd (the reference's), momentum, adam -- the rules "
                         "with slots run on the captured rows (HIP graphs + device data) only")
    ap.add_argument("--rows", default="", help="comma-separated indices of the mnist "
                    "configurations to run (default: all)")
    args = ap.parse_args()
    rows = []
    if "mnist" in args.which:
        # (HIP graphs, fusion, data resident in HBM, whole-step node, deferred fetches)
        only = {int(v) for v in args.rows.split(",") if v != ""}
        for idx, (graphs, fusion, dev_data, step, deferred) in enumerate((
                (True, True, True, True, True), (True, True, True, True, False),
                (True, True, True, False, False), (True, True, False, True, False),
                (False, True, False, True, False), (False, False, False, False, False))):
            if (graphs or dev_data) and not args.device.startswith("cuda"):
                continue
            if only and idx not in only:
                continue
            for opt in args.optimizers.split(","):
                if opt != "sgd" and not (graphs and dev_data and step):
                    continue
                rows.append(bench_mnist(args.device, args.steps, args.warmup, graphs, fusion,
                                        device_data=dev_data, step_fusion=step,
                                        deferred=deferred, optimizer=opt))
                print(json.dumps(rows[-1]), flush=True)
    if "matmul" in args.which:
        cases = ((True, True), (True, False), (False, False)) if args.device.startswith("cuda") \
            else ((False, False),)
        for graphs, deferred in cases:
            rows.append(bench_matmul(args.device, args.steps, args.warmup, 1000, graphs, deferred))
            print(json.dumps(rows[-1]), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
