"""GPU-side time (run under rocprofv3 --kernel-trace) of small bf16 NT products after the 128 x 128-tile kernel got its
request-issuing waves: unsplit on that kernel (variant 11), unsplit on its 64 x 64-tile sibling (variant 12), K split 2 / 4 / 8 / 16 ways with the work items on it, the
default plan, and torch.matmul; 20 calls per case, cases separated by a fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
os.environ["DTF_BF16_SPLIT_T128"] = "1"
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize(); torch.zeros(7, device="cuda", dtype=torch.float64)
    cases = [lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=11),
             lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=12)]
    for s in (2, 4, 8, 16):
        ok = k % (s * 64) == 0 and k // s >= 256
        cases.append((lambda s=s: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=s)) if ok else None)
    for i, fn in enumerate(cases):
        if fn is not None:
            for _ in range(20):
                fn()
        torch.cuda.synchronize(); torch.zeros(8 + i, device="cuda", dtype=torch.float64)
    del os.environ["DTF_BF16_SPLIT_T128"]
    for _ in range(20):
        hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)
    torch.cuda.synchronize(); torch.zeros(20, device="cuda", dtype=torch.float64)
    os.environ["DTF_BF16_SPLIT_T128"] = "1"
    for _ in range(20):
        torch.matmul(a, bt.t(), out=out)
    torch.cuda.synchronize(); torch.zeros(21, device="cuda", dtype=torch.float64)
torch.cuda.synchronize()
