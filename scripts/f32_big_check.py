#!/usr/bin/env python3
"""Slot-scheduled fp32 GEMM (128x128 and 256x256 tiles): results against the general 128x128
kernel's (bitwise: all sum k in the same order) and timing against torch.matmul and the general
kernel, for the four operand layouts and some ragged shapes."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from f32_kscan import timeit  # noqa: E402

shapes = [(8192, 8192, 8192), (4096, 4096, 4096), (4100, 5000, 4104), (2048, 16384, 1000),
          (2048, 16384, 16384), (300, 260, 64), (6144, 6144, 6144), (2048, 2048, 2048),
          (8192, 8192, 512), (1000, 1000, 1000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]
for (m, n, k) in shapes:
    for ta in (False, True):
        for tb in (False, True):
            a = torch.randn((k, m) if ta else (m, k), device="cuda")
            b = torch.randn((n, k) if tb else (k, n), device="cuda")
            want = hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, splits=1, tile=129)
            same = all(bool(torch.equal(hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, splits=1, tile=t), want))
                       for t in (128, 256))
            out = torch.empty(m, n, device="cuda")
            opa, opb = (a.t() if ta else a), (b.t() if tb else b)
            us_t = timeit(lambda: torch.matmul(opa, opb, out=out), 5)
            us_gen = timeit(lambda: hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, out=out, splits=1, tile=129), 5)
            us_128 = timeit(lambda: hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, out=out, splits=1, tile=128), 5)
            us_auto = timeit(lambda: hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, out=out), 5)
            us_256 = timeit(lambda: hip.gemm_f32(a, b, trans_a=ta, trans_b=tb, out=out, splits=1, tile=256), 5)
            fl = 2.0 * m * n * k / 1e6
            print(json.dumps({"mnk": [m, n, k], "trans": [ta, tb], "bitwise_equal_general": same,
                              "torch_tflops": round(fl / us_t, 1),
                              "general128_tflops": round(fl / us_gen, 1),
                              "slot128_tflops": round(fl / us_128, 1),
                              "slot256_tflops": round(fl / us_256, 1),
                              "auto_tflops": round(fl / us_auto, 1),
                              "auto_tile": hip.lib().dtf_gemm_f32_tile(m, n, k)}), flush=True)
