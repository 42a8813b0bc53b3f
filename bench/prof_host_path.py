#!/usr/bin/env python3
"""cProfile of the host side of Session.run on the captured MNIST step (deferred fetches, data
resident in HBM): where the microseconds of a step go once the GPU is no longer the bound."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402

g = tf.Graph()
with g.as_default():
    m = mnist_mlp.build_model(seed=1)
    rng = np.random.RandomState(0)
    batches = [(torch.from_numpy(rng.rand(100, 784).astype(np.float32)).cuda(),
                torch.from_numpy(np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]).cuda())
               for _ in range(16)]
    with tf.Session(config=tf.ConfigProto(device="cuda", deferred_fetches=bool(int(os.environ.get("DEFERRED", "1"))))) as sess:
        sess.run(m.init_op)
        fetches = [m.train_op, m.loss, m.summary_op, m.global_step]
        for i in range(50):
            sess.run(fetches, feed_dict={m.x: batches[i % 16][0], m.y_: batches[i % 16][1]})
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3000):
            out = sess.run(fetches, feed_dict={m.x: batches[i % 16][0], m.y_: batches[i % 16][1]})
        float(out[1])
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
