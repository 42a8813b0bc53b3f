import json, os, sys, torch
sys.path.insert(0, "/root/repo")
from distributed_tensorflow_amd.ops import hip
def timed(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
shapes=[(256,256,2048),(256,256,512),(512,512,512),(512,512,1024),(512,512,2048),(512,1024,1024),(1024,1024,512),(1024,1024,2048),(1024,1024,3072),(768,768,2048),(1280,1280,1280),(1024,1536,1536),(2048,1024,1024),(2048,2048,512),(128,4096,1024),(1000,1000,1024)]
for m,n,k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res = {"default": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)),
           "splits1": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, splits=1)),
           "variant5": timed(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=5))}
    print(json.dumps({"shape":[m,n,k],"tiles256":((m+255)//256)*((n+255)//256),"us":{k2:round(v,1) for k2,v in res.items()}}), flush=True)
