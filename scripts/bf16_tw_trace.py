"""GPU-side time (run under rocprofv3 --kernel-trace) of mid-size bf16 NT products: the default plan, the 128 x 256-tile
kernel with request-issuing waves (variant 13) and torch.matmul; 20 calls per case, cases separated by a float64 fill."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    torch.matmul(a, bt.t(), out=out)
    torch.cuda.synchronize(); torch.zeros(7, device="cuda", dtype=torch.float64)
    for i, fn in enumerate((lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out),
                            lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=13),
                            lambda: torch.matmul(a, bt.t(), out=out))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); torch.zeros(8 + i, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
