#!/usr/bin/env python3
"""Where the MNIST example's per-step time goes beyond Session.run: the same captured training
step timed with (a) fixed device batches, (b) DataSet.next_batch on the resident set, (c) plus
FileWriter.add_summary, (d) plus the step-time reporter."""
import json, os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.data.mnist import DataSet, synthetic_mnist  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402
from distributed_tensorflow_amd.utils.step_timer import StepTimeReporter  # noqa: E402

images, labels = synthetic_mnist(55000, seed=0)
onehot = np.eye(10, dtype=np.float32)[labels]
ds = DataSet(images, onehot, seed=1).to_device("cuda")
model = mnist_mlp.build_model(seed=1)
sess = tf.Session(config=tf.ConfigProto(device="cuda"))
sess.run(model.init_op)
fetches = [model.train_op, model.loss, model.summary_op, model.global_step]
fixed = [ds.next_batch(100) for _ in range(16)]
writer = tf.summary.FileWriter(tempfile.mkdtemp(), graph=None)
reporter = StepTimeReporter(100, printer=lambda line: None)


def run(n, batches, use_writer, use_reporter):
    for i in range(50):
        bx, by = fixed[i % 16]
        sess.run(fetches, {model.x: bx, model.y_: by})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        bx, by = ds.next_batch(100) if batches else fixed[i % 16]
        _, cost, summ, step = sess.run(fetches, {model.x: bx, model.y_: by})
        if use_writer:
            writer.add_summary(summ, step)
        if use_reporter and reporter is not None:
            reporter.step(step, 0, i, n, cost)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, args in (("fixed_batches", (False, False, False)), ("next_batch", (True, False, False)),
                   ("next_batch+writer", (True, True, False)),
                   ("next_batch+writer+reporter", (True, True, True))):
    print(json.dumps({"loop": name, "us_per_step": round(run(3000, *args), 1)}), flush=True)
writer.close()
