#!/usr/bin/env python3
"""Host cost of getting small kernels onto the GPU: HIP-graph replay against direct launches from
one C call, as a function of the number of kernels. Prints one JSON line per case."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

dev = torch.device("cuda")
x = torch.zeros(64, device=dev)
L = hip.lib()
N = 3000


def tiny():
    L.dtf_fill_f32(x.data_ptr(), 1.0, 64, torch.cuda.current_stream().cuda_stream)


def timed(fn, n=N):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e6 / n, (t2 - t0) * 1e6 / n


for nodes in (1, 3, 8, 24):
    g = torch.cuda.CUDAGraph()
    tiny()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(nodes):
            tiny()
    host, total = timed(g.replay)
    print(json.dumps({"case": "graph_replay", "kernels": nodes, "host_us_per_launch": round(host, 2),
                      "wall_us_per_launch": round(total, 2)}), flush=True)

stream = torch.cuda.current_stream().cuda_stream
for nodes in (1, 3, 8):
    def direct():
        for _ in range(nodes):
            L.dtf_fill_f32(x.data_ptr(), 1.0, 64, stream)
    host, total = timed(direct)
    print(json.dumps({"case": "direct_ctypes", "kernels": nodes, "host_us_per_launch": round(host, 2),
                      "wall_us_per_launch": round(total, 2)}), flush=True)

# the three-launch MLP step from ONE C call with a prepared argument block
b, d, h, c = 100, 784, 100, 10
xs = torch.rand(b, d, device=dev)
labels = torch.eye(c, device=dev)[torch.randint(0, c, (b,), device=dev)]
params = [torch.randn(d, h, device=dev), torch.zeros(h, device=dev),
          torch.randn(h, c, device=dev), torch.zeros(c, device=dev)]
scratch = hip.mlp_step_scratch(b, d, h, c, dev)


def step_py():
    hip.mlp_step_train(xs, labels, *params, [0.001] * 4, scratch, (b, d, h, c))


host, total = timed(step_py)
print(json.dumps({"case": "mlp_step_python_wrapper", "kernels": 3,
                  "host_us_per_launch": round(host, 2), "wall_us_per_launch": round(total, 2)}),
      flush=True)
