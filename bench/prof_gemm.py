#!/usr/bin/env python3
"""Tiny driver for rocprofv3: run one GEMM kernel variant a few times and exit."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=0, help="K of the bf16 product (default: n)")
ap.add_argument("--pad", type=int, default=0, help="bf16: operands are views of buffers this many elements wider than K")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--impl", default="ours", choices=("ours", "torch"),
                help="torch: the vendor library (hipBLASLt through torch.matmul) on the same operands")
args = ap.parse_args()
n = args.n
if args.dtype == "bf16":
    k = args.k or n
    a = (torch.rand(n, k + args.pad, device="cuda") * 2 - 1).bfloat16()[:, :k]
    bt = (torch.rand(n, k + args.pad, device="cuda") * 2 - 1).bfloat16()[:, :k]
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(args.iters):
        if args.impl == "torch":
            torch.matmul(a, bt.t(), out=out)
        else:
            hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=args.variant)
elif args.dtype == "fp8":
    a = hip.philox_window(n, n, 1, 1, 0, 0, n, torch.float8_e4m3fn, "cuda", False, -1.0, 2.0)
    bt = hip.philox_window(n, n, 1, 2, 0, 0, n, torch.float8_e4m3fn, "cuda", False, -1.0, 2.0)
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(args.iters):
        hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out)
else:
    a = torch.rand(n, n, device="cuda") * 2 - 1
    b = torch.rand(n, n, device="cuda") * 2 - 1
    out = torch.empty(n, n, device="cuda")
    for _ in range(args.iters):
        if args.impl == "torch":
            torch.matmul(a, b, out=out)
        else:
            hip.gemm_f32(a, b, out=out)
torch.cuda.synchronize()
print("done")
