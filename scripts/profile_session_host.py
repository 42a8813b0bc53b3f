"""cProfile of the host side of a HIP-graph MNIST training step (where do the non-GPU
microseconds of Session.run go?)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402

model = mnist_mlp.build_model(seed=1)
rng = np.random.RandomState(0)
bx = torch.from_numpy(rng.rand(100, 784).astype(np.float32)).cuda()
by = torch.from_numpy(np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]).cuda()
sess = tf.Session(config=tf.ConfigProto(device="cuda"))
sess.run(model.init_op)
f = [model.train_op, model.loss, model.summary_op, model.global_step]
fd = {model.x: bx, model.y_: by}
for _ in range(50):
    sess.run(f, fd)
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    sess.run(f, fd)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
