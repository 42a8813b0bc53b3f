"""Small fp8 products (at most 32 tiles of 256 x 256): this library's launch against torch._scaled_mm."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
FP8 = torch.float8_e4m3fn
def timed(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
one = torch.ones((), device="cuda")
for m, n, k in [(512,512,1024),(1024,1024,1024),(1024,1024,4096),(2048,1024,1024),(1024,2048,2048),(1280,1280,1280),(2048,2048,2048),(256,1024,2048),(768,768,4096)]:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(FP8); bt = (torch.rand(n, k, device="cuda") * 2 - 1).to(FP8)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {"ours": timed(lambda: hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out))}
    try:
        res["torch"] = timed(lambda: torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
    except Exception:
        res["torch"] = None
    print(json.dumps({"shape": [m, n, k], "us": {k2: (round(v, 1) if v else None) for k2, v in res.items()}}), flush=True)
