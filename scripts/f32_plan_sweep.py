"""fp32 products: every (block tile, K split) plan of gemm_f32 against its default and torch.matmul."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
def timed(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
    [(1024,1024,1024),(256,1024,4096),(1536,1536,1536),(1024,4096,512),(4096,256,1024),(2048,512,2048),(512,512,512),(768,768,768),(1280,1280,1280)]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda")
    out = torch.empty(m, n, device="cuda")
    res = {"torch": timed(lambda: torch.matmul(a, b, out=out)), "default": timed(lambda: hip.gemm_f32(a, b, out=out))}
    L = hip.lib()
    plan = (L.dtf_gemm_f32_tile(m, n, k), L.dtf_gemm_f32_splits(m, n, k, None))
    for tile in (64, 128, 256):
        for s in (1, 2, 4, 8):
            if k % (s * 32): continue
            try:
                res["t%d_s%d" % (tile, s)] = timed(lambda: hip.gemm_f32(a, b, out=out, tile=tile, splits=s))
            except Exception:
                pass
    res["default_again"] = timed(lambda: hip.gemm_f32(a, b, out=out))
    best = min((v, k2) for k2, v in res.items() if k2.startswith("t") and k2 != "torch")
    print(json.dumps({"m_k_n": [m, k, n], "default_plan": plan, "best": best[1], "default_over_best": round(res["default"] / best[0], 3),
                      "us": {k2: round(v, 1) for k2, v in res.items()}}), flush=True)
