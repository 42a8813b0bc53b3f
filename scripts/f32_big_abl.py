#!/usr/bin/env python3
"""Timing ablations of the 128x128 fp32 GEMM loop at a large size (DTF_GEMM_EXPERIMENTS build).

ABL bits: 1 = no global loads in the K loop, 2 = no LDS stash, 4 = no barrier, 8 = no fragment
reads. Results of an ablated kernel are wrong; only its time means anything.
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ks = [int(x) for x in os.environ.get("KS", str(n)).split(",")]
    abls = [int(x) for x in os.environ.get("ABLS", "0,1,2,4,8,6,7,14,15,0").split(",")]
    out = torch.empty(n, n, device="cuda")
    for k in ks:
        a = torch.randn(n, k, device="cuda")
        b = torch.randn(k, n, device="cuda")
        flop = 2.0 * n * n * k
        ref = timeit(lambda: torch.matmul(a, b, out=out))
        print(json.dumps({"n": n, "k": k, "kernel": "torch", "us": round(ref, 1),
                          "tflops": round(flop / ref / 1e6, 1)}), flush=True)
        for abl in abls:
            os.environ["DTF_F32_ABL"] = str(abl)
            us = timeit(lambda: hip.gemm_f32(a, b, out=out, splits=1, tile=tile))
            print(json.dumps({"n": n, "k": k, "tile": tile, "abl": abl, "us": round(us, 1),
                              "tflops": round(flop / us / 1e6, 1)}), flush=True)
