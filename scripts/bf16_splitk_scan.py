#!/usr/bin/env python3
"""Mid-size bf16 products: the default plan against 256x256 tiles with K split over work items
(fp32 planes + reduction), per split count, and torch.matmul."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from f32_kscan import timeit  # noqa: E402

shapes = [(2048, 2048, 8192), (2048, 2048, 2048), (1024, 4096, 4096), (2048, 4096, 4096),
          (1536, 1536, 1536), (3072, 3072, 3072), (2048, 2048, 4096), (2560, 2560, 2560),
          (4096, 4096, 1024), (2048, 4096, 16384)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]
for (m, n, k) in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    row = {"mnk": [m, n, k], "tiles256": ((m + 255) // 256) * ((n + 255) // 256),
           "plan_splits": hip.lib().dtf_gemm_bf16_splits(m, n, k),
           "torch_us": round(timeit(lambda: torch.matmul(a, bt.t(), out=out), 30), 1),
           "default_us": round(timeit(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out), 30), 1)}
    for sp in (2, 3, 4, 6, 8):
        if k // sp < 512:
            continue
        row["split%d_us" % sp] = round(timeit(
            lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=sp), 30), 1)
    print(json.dumps(row), flush=True)
