"""GPU-side time (run under rocprofv3 --kernel-trace) of K-split bf16 products with the split work items on 256 x 256
tiles (ping-pong kernel) and on 128 x 128 tiles, for 2, 4, 8 and 16 splits; exactness on integer data checked first."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, n, k in shapes:
    ai = torch.randint(-2, 3, (m, k)).float(); bi = torch.randint(-2, 3, (n, k)).float()
    ref = (ai.cuda() @ bi.cuda().t())
    for t128 in ("0", "1"):
        os.environ["DTF_BF16_SPLIT_T128"] = t128
        for s in (2, 4, 8, 16):
            if k % (s * 64): continue
            got = hip.gemm_bf16_nt(ai.bfloat16().cuda(), bi.bfloat16().cuda(), torch.float32, splits=s)
            assert torch.equal(got, ref), (m, n, k, t128, s)
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize(); torch.zeros(7, device="cuda")
    for t128 in ("0", "1"):
        os.environ["DTF_BF16_SPLIT_T128"] = t128
        for s in (2, 4, 8, 16):
            if k % (s * 64) or k // s < 256:
                s = 1
            for _ in range(20):
                hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=s)
            torch.cuda.synchronize(); torch.zeros(8 + s, device="cuda")
torch.cuda.synchronize()
print("exact")
