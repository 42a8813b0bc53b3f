"""Default plan against each block-tile form (variant 4: 256 rows, 6: 192, 9: 128, 5: 128 x 128 tiles) and
torch.matmul on mid-size bf16 products, to check the launch rule's choices."""
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
sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2304, 2560, 2816, 3328, 3584, 3840, 4608]
ks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1024, 4096]
for n in sizes:
    for k in ks:
        a = torch.randn(n, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
        out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
        res = {"torch": timed(lambda: torch.matmul(a, bt.t(), out=out)),
               "default": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out))}
        for v in (4, 6, 9, 5):
            res["variant%d" % v] = timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=v))
        best = min((v, k2) for k2, v in res.items() if k2.startswith("variant"))
        print(json.dumps({"shape": [n, n, k], "tiles256": ((n + 255) // 256) ** 2,
                          "us": {k2: round(v, 1) for k2, v in res.items()}, "best": best[1],
                          "default_vs_best": round(res["default"] / best[0], 3)}), flush=True)
