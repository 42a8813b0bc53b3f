"""GPU-side time (run under rocprofv3 --kernel-trace) of fp32 products under the default plan, on the 32 x 32-tile
small-product kernel and in torch.matmul; 20 calls per case, cases separated by a fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); out = torch.empty(m, n, device="cuda")
    for tile in (0, 32):
        for _ in range(20):
            hip.gemm_f32(a, b, out=out, tile=tile)
        torch.cuda.synchronize()
        torch.zeros(m + tile, device="cuda")
    for _ in range(20):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    torch.zeros(m + 1, device="cuda")
torch.cuda.synchronize()
