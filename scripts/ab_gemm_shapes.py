#!/usr/bin/env python3
"""A/B timing of bf16 NT GEMM variants over rectangular shapes (one JSON line per shape/variant)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="2048x16384x16384,4096x16384x16384,8192x16384x16384,16384x16384x2048,8192x8192x8192")
    ap.add_argument("--variants", default="0,260")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    for shape in args.shapes.split(","):
        m, n, k = (int(x) for x in shape.split("x"))
        a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        bt = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(args.rounds):
            for v in (int(x) for x in args.variants.split(",")):
                ms = timeit(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=v), args.iters)
                print(json.dumps({"shape": shape, "variant": v, "ms": round(ms, 4),
                                  "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
