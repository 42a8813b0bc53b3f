#!/usr/bin/env python3
"""Run the 1000^3 fp32 product through the hand-written kernel and torch.matmul (for kernel traces)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
a = torch.randn(n, n, device="cuda")
b = torch.randn(n, n, device="cuda")
out = torch.empty(n, n, device="cuda")
for _ in range(30):
    hip.gemm_f32(a, b, out=out, splits=1, tile=64)
torch.cuda.synchronize()
for _ in range(30):
    hip.gemm_f32(a, b, out=out)
torch.cuda.synchronize()
for _ in range(30):
    torch.matmul(a, b, out=out)
torch.cuda.synchronize()
