#!/usr/bin/env python3
"""Kernel-only timing of the fp32 GEMM on the reference's shapes, across split-K factors."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from bench_gemm import timeit  # noqa: E402

shapes = [(1000, 1000, 1000, False, False), (100, 100, 784, False, False),
          (784, 100, 100, True, False), (10000, 100, 784, False, False), (2048, 2048, 2048, False, False),
          (1536, 1536, 1536, False, False)]
for m, n, k, ta, tb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    out = torch.empty(m, n, device="cuda")
    # vendor reference (rocBLAS/hipBLASLt through torch.matmul), same process, same graph replay
    torch.backends.cuda.matmul.allow_tf32 = False
    ra = a.t() if ta else a
    rb = b.t() if tb else b
    g = torch.cuda.CUDAGraph()
    torch.matmul(ra, rb, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            torch.matmul(ra, rb, out=out)
    ms = timeit(g.replay, 20) / 20
    print(json.dumps({"m": m, "n": n, "k": k, "ta": ta, "tb": tb, "impl": "torch.matmul",
                      "us": round(ms * 1e3, 2), "tflops": round(2.0 * m * n * k / ms / 1e9, 2)}),
          flush=True)
    for tile, splits in ((0, None), (64, 1), (64, 4), (128, 1)):
        g = torch.cuda.CUDAGraph()
        hip.gemm_f32(a, b, ta, tb, out=out, splits=splits, tile=tile)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20):
                hip.gemm_f32(a, b, ta, tb, out=out, splits=splits, tile=tile)
        ms = timeit(g.replay, 20) / 20
        print(json.dumps({"m": m, "n": n, "k": k, "ta": ta, "tb": tb, "tile": tile, "splits": splits,
                          "us": round(ms * 1e3, 2), "tflops": round(2.0 * m * n * k / ms / 1e9, 2)}),
              flush=True)
