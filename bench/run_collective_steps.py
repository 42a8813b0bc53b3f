#!/usr/bin/env python3
"""N training steps of the MNIST MLP under the collective runtime (RCCL, a world of one rank on
one GPU) with a chosen strategy -- the program rocprofv3 is pointed at to list every kernel of the
steady-state step (profiles/collective_step_kernels.txt: no torch kernel among them)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import distributed_tensorflow_amd as tf  # noqa: E402
from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402
from distributed_tensorflow_amd.parallel.comm import TorchDistComm  # noqa: E402
from distributed_tensorflow_amd.train.server import _free_port  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--strategy", default="reduce_scatter_sync")
ap.add_argument("--eager", action="store_true")
ap.add_argument("--optimizer", default="sgd", choices=("sgd", "adam", "momentum"))
ap.add_argument("--chain", action="store_true",
                help="the chain of general kernels instead of the small-step kernels in gradient "
                     "mode (ConfigProto(step_fusion=False))")
args = ap.parse_args()

comm = TorchDistComm(rank=0, world_size=1, device=torch.device("cuda", 0), backend="nccl",
                     master_addr="127.0.0.1", master_port=_free_port())
rng = np.random.RandomState(0)
batches = [(torch.from_numpy(rng.rand(100, 784).astype(np.float32)).cuda(),
            torch.from_numpy(np.eye(10, dtype=np.float32)[rng.randint(0, 10, 100)]).cuda())
           for _ in range(8)]
cfg = tf.ConfigProto(device="cuda:0", strategy=args.strategy, hip_graphs=not args.eager,
                     step_fusion=False if args.chain else None)
cluster = tf.train.ClusterSpec({"worker": ["127.0.0.1:%d" % _free_port()]})
server = tf.train.Server(cluster, job_name="worker", task_index=0, protocol="rccl", config=cfg,
                         comm=comm)
with tf.device(tf.train.replica_device_setter(cluster=cluster)):
    opt = {"sgd": None, "adam": tf.train.AdamOptimizer(0.01),
           "momentum": tf.train.MomentumOptimizer(0.01, 0.9)}[args.optimizer]
    model = mnist_mlp.build_model(learning_rate=0.01, seed=1, optimizer=opt)
sv = tf.train.Supervisor(is_chief=True, global_step=model.global_step, init_op=model.init_op)
sess = sv.prepare_or_wait_for_session(server.target, config=cfg)
for i in range(args.steps):
    bx, by = batches[i % 8]
    _, loss, step = sess.run([model.train_op, model.loss, model.global_step],
                             {model.x: bx, model.y_: by})
rt = server.runtime
print(json.dumps({"steps": args.steps, "strategy": args.strategy, "final_loss": float(loss),
                  "global_step": float(step), "graph_replays": sess.stats["graph_replays"],
                  "eager_runs": sess.stats["eager_runs"], "sharded_steps": rt.sharded_steps,
                  "grads_in_place": rt.grads_in_place, "grads_copied": rt.grads_copied}))
sess.close()
sv.stop()
comm.close()
