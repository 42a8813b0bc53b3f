"""Kernel plans for small and mid-size bf16 products side by side: torch.matmul, the default plan, explicit
K splits, and the 128 x 128-tile kernels (variant 5: two buffers, 11: four-buffer ring, 9: 128-row ping-pong)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
def timed(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for m, n, k in [(1024,1024,1024),(1536,1536,1536),(1024,1024,4096),(2048,2048,1024),(1024,2048,2048),(2048,2048,2048),(2048,2048,4096),(2560,2560,2560),(512,512,4096),(768,768,768)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {"torch": timed(lambda: torch.matmul(a, bt.t(), out=out)),
           "default": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out))}
    for s in (1, 2, 4, 8):
        try: res["splits%d" % s] = timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=s))
        except Exception as e: res["splits%d" % s] = None
    for v in (5, 11, 9):
        try: res["variant%d" % v] = timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=v))
        except Exception as e: res["variant%d" % v] = None
    print(json.dumps({"shape": [m, n, k], "us": {k2: (round(v, 1) if v else None) for k2, v in res.items()}}), flush=True)
