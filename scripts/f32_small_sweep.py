"""fp32 products of assorted small and mid sizes: this library's gemm_f32 against torch.matmul (us per call)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
torch.backends.cuda.matmul.allow_tf32 = False
def timed(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
shapes = [(100,784,100),(100,100,10),(512,512,512),(1000,1000,1000),(1024,1024,1024),(2048,512,2048),(256,1024,4096),(4096,256,1024),
          (1536,1536,1536),(2048,2048,2048),(3072,1024,3072),(2560,2560,2560),(4096,4096,1024),(784,100,100),(2000,2000,2000),(1024,4096,512)]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda")
    out = torch.empty(m, n, device="cuda")
    t = timed(lambda: torch.matmul(a, b, out=out)); o = timed(lambda: hip.gemm_f32(a, b, out=out))
    print(json.dumps({"m_k_n": [m, k, n], "torch_us": round(t, 1), "ours_us": round(o, 1), "ours_over_torch": round(o / t, 3)}), flush=True)
