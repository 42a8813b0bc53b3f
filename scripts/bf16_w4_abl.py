#!/usr/bin/env python3
"""Timing ablations of the four-wave bf16 kernel (variant 3; DTF_GEMM_EXPERIMENTS build).
Mode bits: 1 = no LDS-DMA in the K loop, 2 = every load an L2 hit, 4 = no barrier, 8 = no wait in
front of the barrier. Results of an ablated kernel are wrong; only its time means anything."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
bt = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)


def rate(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n ** 3 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12


print(json.dumps({"kernel": "torch", "tflops": round(rate(lambda: torch.matmul(a, bt.t(), out=out)), 1)}), flush=True)
print(json.dumps({"kernel": "default", "tflops": round(rate(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)), 1)}), flush=True)
for mode in (0, 2, 4, 6, 14, 1, 5, 13):
    print(json.dumps({"kernel": "w4", "mode": mode,
                      "tflops": round(rate(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=3 | (mode << 4))), 1)}), flush=True)
