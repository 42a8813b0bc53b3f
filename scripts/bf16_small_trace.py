"""GPU-side time (run under rocprofv3 --kernel-trace) of small bf16 NT products: default plan, the 128 x 128-tile
kernel (variant 5), an eight-way K split, and torch.matmul; 20 calls per case, cases separated by a fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    cases = [lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out),
             lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=5),
             lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=8 if k % 512 == 0 else 1),
             lambda: torch.matmul(a, bt.t(), out=out)]
    for i, fn in enumerate(cases):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); torch.zeros(8 + i, device="cuda")
torch.cuda.synchronize()
