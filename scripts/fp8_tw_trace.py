"""GPU-side time (run under rocprofv3 --kernel-trace) of mid-size fp8 NT products with (default) and without
(DTF_FP8_NO_TW=1, honoured by the experiments build of the library only: DTF_GEMM_EXPERIMENTS=1 python
build_native.py) the 128 x 256-tile kernel with request-issuing waves, and torch._scaled_mm."""
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
    for i in range(3):
        if i == 0: os.environ["DTF_FP8_NO_TW"] = "1"
        else: os.environ.pop("DTF_FP8_NO_TW", None)
        for _ in range(20):
            if i < 2: hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out)
            else: torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        torch.cuda.synchronize(); torch.zeros(8 + i, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
