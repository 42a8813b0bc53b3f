import sys, json, torch
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from distributed_tensorflow_amd.ops import hip
from f32_kscan import timeit
for n in (8192, 4096, 2048):
    for k in (64, 128, 256, 384, 512):
        a = torch.randn(n, k, device="cuda"); b = torch.randn(k, n, device="cuda")
        out = torch.empty(n, n, device="cuda")
        row = {"n": n, "k": k, "torch_us": round(timeit(lambda: torch.matmul(a, b, out=out), 20), 1)}
        for t in (64, 128):
            row["tile%d_us" % t] = round(timeit(lambda: hip.gemm_f32(a, b, out=out, splits=1, tile=t), 20), 1)
        row["auto_us"] = round(timeit(lambda: hip.gemm_f32(a, b, out=out), 20), 1)
        row["auto_tile"] = hip.lib().dtf_gemm_f32_tile(n, n, k)
        print(json.dumps(row), flush=True)
