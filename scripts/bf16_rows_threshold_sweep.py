import json, os, sys, torch
sys.path.insert(0, "/root/repo")
from distributed_tensorflow_amd.ops import hip
def timed(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for m, n in [(2304,2304),(2560,2048),(2176,2304),(2048,2304),(2048,2560),(2816,2048),(2048,2048),(1920,2304),(2432,2048),(1664,2560)]:
    for k in (1024, 4096):
        a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        res = {"default": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out))}
        for v in (9, 5, 6):
            res["v%d" % v] = timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=v))
        print(json.dumps({"shape": [m, n, k], "wg128rows": ((m + 127) // 128) * ((n + 255) // 256), "wg192rows": ((m + 191) // 192) * ((n + 255) // 256),
                          "us": {k2: round(v, 1) for k2, v in res.items()}}), flush=True)
