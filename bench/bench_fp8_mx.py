#!/usr/bin/env python3
"""fp8 GEMM throughput, 16384^3 (and 8192^3): block-scaled (MX, E8M0 per 32 k) against the
per-tensor-scaled kernel, plus the cost of quantising an operand. One JSON line per case."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for n in (8192, 16384):
    a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    qa, qb = hip.quantize_mx_fp8(a, 0), hip.quantize_mx_fp8(b, 1)
    pa, ia = hip.quantize_fp8(a)
    pb, ib = hip.quantize_fp8(b)
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    iters = 20 if n == 16384 else 50
    rows = []
    for rep in range(3):
        t_mx = timed(lambda: hip.gemm_fp8_mx_nt(qa, qb, torch.bfloat16, out=out), iters)
        t_pt = timed(lambda: hip.gemm_fp8_nt(pa, pb, torch.bfloat16, alpha=ia * ib, out=out), iters)
        rows.append((t_mx, t_pt))
    t_mx, t_pt = min(r[0] for r in rows), min(r[1] for r in rows)
    t_q = timed(lambda: hip.quantize_mx_fp8(a, 0), 10)
    print(json.dumps({"n": n, "mx_tflops": round(2 * n ** 3 / t_mx / 1e12, 1),
                      "per_tensor_tflops": round(2 * n ** 3 / t_pt / 1e12, 1),
                      "mx_over_per_tensor": round(t_pt / t_mx, 4),
                      "quantize_mx_ms": round(t_q * 1e3, 3),
                      "quantize_gb_per_s": round(n * n * 3 / t_q / 1e9, 1)}), flush=True)
    del a, b, qa, qb, pa, pb, out
