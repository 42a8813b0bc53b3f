import sys, json, torch
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
from distributed_tensorflow_amd.ops import hip
from f32_kscan import timeit
for (m, n, k) in [(1000, 1000, 1000), (4096, 4096, 1000), (8192, 8192, 1000), (100, 100, 784)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out))
    code_us = timeit(lambda: hip.lib().dtf_gemm_bf16_generic(a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k, a.stride(0), 1, 1, bt.stride(0), out.stride(0), 1.0, 0.0, 0, None))
    ust = timeit(lambda: torch.matmul(a, bt.t(), out=out))
    fl = 2.0 * m * n * k / 1e6
    print(json.dumps({"mnk": [m, n, k], "ours_us": round(us, 1), "generic_only_us": round(code_us, 1), "torch_us": round(ust, 1), "ours_tflops": round(fl / us, 1), "torch_tflops": round(fl / ust, 1)}), flush=True)
