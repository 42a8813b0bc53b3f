#!/usr/bin/env python3
"""Device time of one MNIST-MLP training step (batch 100, 784-100-10, tf_distributed.py:58-76)
as a replayed HIP graph: the three-launch step kernels against the chain of general kernels.
No session, no feeds, no fetches -- the kernels alone. One JSON line per variant."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def main() -> None:
    b, d, h, c = 100, 784, 100, 10
    dev = torch.device("cuda")
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    x = torch.rand(b, d, device=dev)
    labels = torch.eye(c, device=dev)[torch.randint(0, c, (b,), device=dev)]
    params = [torch.randn(d, h, device=dev), torch.zeros(h, device=dev),
              torch.randn(h, c, device=dev), torch.zeros(c, device=dev)]
    step = torch.zeros((), device=dev)
    dloss = torch.ones((), device=dev)
    scratch = hip.mlp_step_scratch(b, d, h, c, dev)
    packed = torch.zeros(8, device=dev)

    def chain():
        a = hip.gemm_f32(x, params[0], bias=params[1], activation=hip.ACT_SIGMOID)
        loss, y, dw2, db2, dz1, db1, acc = hip.mlp_head_train(a, params[2], params[3], labels,
                                                              dloss, True, True)
        dw1 = hip.gemm_f32(x, dz1, trans_a=True)
        hip.multi_sgd(params, [dw1, db1, dw2, db2], [0.001] * 4, [step], [1.0])
        hip.pack_f32([loss, step, acc], packed)

    def fused():
        hip.mlp_step_train(x, labels, *params, [0.001] * 4, scratch, (b, d, h, c),
                           counters=[step], deltas=[1.0],
                           pack=[scratch["loss"], step, scratch["accuracy"]], pack_out=packed)

    for name, fn in (("kernel_chain", chain), ("three_launch_step", fused)):
        fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        for _ in range(50):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / steps
        print(json.dumps({"bench": "mlp_step_device", "variant": name, "us_per_step": round(us, 2),
                          "steps": steps}), flush=True)


if __name__ == "__main__":
    main()
