#!/usr/bin/env python3
"""bf16 products off the LDS-DMA kernel's grid (K not a multiple of 64, odd K, odd N): padded
images + fast kernel (what gemm_bf16_nt does) against the predicated generic kernel alone and
torch.matmul."""
import sys, json, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from f32_kscan import timeit  # noqa: E402

for (m, n, k) in [(1000, 1000, 1000), (4096, 4096, 1000), (8192, 8192, 1000), (100, 100, 784),
                  (4097, 4099, 4101), (8192, 8191, 8192)]:
    a = torch.randn(m, k, device="cuda").bfloat16()
    bt = torch.randn(n, k, device="cuda").bfloat16()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out), 20)
    gen = timeit(lambda: hip.lib().dtf_gemm_bf16_generic(
        a.data_ptr(), bt.data_ptr(), out.data_ptr(), m, n, k, a.stride(0), 1, 1, bt.stride(0),
        out.stride(0), 1.0, 0.0, 0, None), 5)
    ust = timeit(lambda: torch.matmul(a, bt.t(), out=out), 20)
    fl = 2.0 * m * n * k / 1e6
    print(json.dumps({"mnk": [m, n, k], "ours_us": round(us, 1), "generic_only_us": round(gen, 1),
                      "torch_us": round(ust, 1), "ours_tflops": round(fl / us, 1),
                      "torch_tflops": round(fl / ust, 1)}), flush=True)
