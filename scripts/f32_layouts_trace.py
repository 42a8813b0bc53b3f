"""GPU-side time (run under rocprofv3 --kernel-trace) of fp32 products in the four storage layouts, ours and
torch.matmul; 10 calls per case, cases separated by a fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, k, n in shapes:
    for ta in (False, True):
        for tb in (False, True):
            a = torch.randn((k, m) if ta else (m, k), device="cuda"); b = torch.randn((n, k) if tb else (k, n), device="cuda")
            out = torch.empty(m, n, device="cuda")
            for _ in range(10):
                hip.gemm_f32(a, b, out=out, trans_a=ta, trans_b=tb)
            torch.cuda.synchronize(); torch.zeros(8, device="cuda")
            aa, bb = (a.t() if ta else a), (b.t() if tb else b)
            for _ in range(10):
                torch.matmul(aa, bb, out=out)
            torch.cuda.synchronize(); torch.zeros(9, device="cuda")
torch.cuda.synchronize()
