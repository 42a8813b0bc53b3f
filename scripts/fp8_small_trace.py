"""GPU-side time (run under rocprofv3 --kernel-trace) of small and mid fp8 NT products: this library's default launch
against torch._scaled_mm; 20 calls per case, cases separated by a float64 fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
FP8 = torch.float8_e4m3fn
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
one = torch.ones((), device="cuda")
for m, n, k in shapes:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(FP8); bt = (torch.rand(n, k, device="cuda") * 2 - 1).to(FP8)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); torch.zeros(7, device="cuda", dtype=torch.float64)
    for _ in range(20):
        hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out)
    torch.cuda.synchronize(); torch.zeros(8, device="cuda", dtype=torch.float64)
    for _ in range(20):
        torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); torch.zeros(9, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
