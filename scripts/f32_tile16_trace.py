import sys, torch
sys.path.insert(0, "/root/repo")
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); out = torch.empty(m, n, device="cuda")
    for tile in (0, 16):
        for _ in range(20):
            hip.gemm_f32(a, b, out=out, tile=tile)
        torch.cuda.synchronize()
        torch.zeros(m + tile, device="cuda")
torch.cuda.synchronize()
