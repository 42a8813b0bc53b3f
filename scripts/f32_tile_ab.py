#!/usr/bin/env python3
"""fp32 GEMM: 64x64, 128x128 and 256x256 block tiles against torch.matmul on square sizes,
with a check of the result against the 64x64 tile's."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from f32_kscan import timeit  # noqa: E402

if __name__ == "__main__":
    for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2048,3072,4096,8192").split(",")]:
        a = torch.randn(n, n, device="cuda")
        b = torch.randn(n, n, device="cuda")
        out = torch.empty(n, n, device="cuda")
        want = hip.gemm_f32(a, b, splits=1, tile=64)
        us = timeit(lambda: torch.matmul(a, b, out=out), 10)
        print(json.dumps({"n": n, "kernel": "torch", "us": round(us, 1),
                          "tflops": round(2.0 * n ** 3 / us / 1e6, 1)}), flush=True)
        for tile in (0, 64, 129, 128, 256):
            us = timeit(lambda: hip.gemm_f32(a, b, out=out, splits=1, tile=tile), 10)
            err = float((out - want).abs().max())
            print(json.dumps({"n": n, "tile": tile, "us": round(us, 1),
                              "tflops": round(2.0 * n ** 3 / us / 1e6, 1),
                              "max_abs_diff_vs_64": err}), flush=True)
