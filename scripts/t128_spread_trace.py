"""GPU-side time (run under rocprofv3 --kernel-trace) of bf16 NT products on the 128 x 128-tile kernel in its four-wave
form (variant 5: every wave issues its own LDS-DMA requests, two K-tile buffers) and with four request-issuing waves
(variant 11: three buffers), and torch.matmul; 20 calls per case, cases separated by a fill launch. Also checks that the
two forms agree bit for bit. (The other tables of profiles/gemm_bf16_t128_request_waves.txt came from builds of an
experiment slot in the four-wave form that the tree no longer has.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = []
    for i, mode in enumerate((5, 11, "torch")):
        if mode != "torch":
            fn = lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=mode)
        else:
            fn = lambda: torch.matmul(a, bt.t(), out=out)
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); res.append(out.clone()); torch.zeros(8 + i, device="cuda")
    ref = (a.float() @ bt.float().t())
    print("%dx%dx%d" % (m, n, k), "spread == run:", torch.equal(res[0], res[1]),
          "max err vs fp32:", float((res[1].float() - ref).abs().max()), "torch err:",
          float((res[2].float() - ref).abs().max()), file=sys.stderr)
torch.cuda.synchronize()
