#!/usr/bin/env python3
"""fp32 GEMM kernel time against K at fixed M = N (fixed cost vs per-K-tile cost)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(10):
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
    mn = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    for k in [int(x) for x in os.environ.get("KS", "16,128,256,512,1000,2000,4000").split(",")]:
        a = torch.randn(mn, k, device="cuda")
        b = torch.randn(k, mn, device="cuda")
        out = torch.empty(mn, mn, device="cuda")
        ours = timeit(lambda: hip.gemm_f32(a, b, out=out, splits=1, tile=64))
        ref = timeit(lambda: torch.matmul(a, b, out=out))
        print(json.dumps({"mn": mn, "k": k, "ours_us": round(ours, 2), "torch_us": round(ref, 2)}), flush=True)
