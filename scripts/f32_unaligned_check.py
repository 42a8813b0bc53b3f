#!/usr/bin/env python3
"""fp32 GEMM on operands the branch-free vector loads cannot take (odd K, odd N, odd M): exact on
integer data in the four layouts, and kernel time against torch.matmul."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402
from f32_kscan import timeit  # noqa: E402

shapes = [(999, 1001, 1003), (1000, 1001, 1000), (1001, 1000, 1000), (1000, 1000, 1002),
          (1000, 1000, 1000), (2049, 2047, 2051), (4097, 4099, 4101), (130, 70, 9)]
for (m, n, k) in shapes:
    for ta, tb in ((False, False), (False, True), (True, False), (True, True)):
        g = torch.Generator().manual_seed(m + k)
        ai = torch.randint(-3, 4, (m, k), generator=g).float().cuda()
        bi = torch.randint(-3, 4, (k, n), generator=g).float().cuda()
        a_in = ai.t().contiguous() if ta else ai
        b_in = bi.t().contiguous() if tb else bi
        exact = all(bool(torch.equal(hip.gemm_f32(a_in, b_in, ta, tb, tile=t), ai @ bi))
                    for t in (0, 64, 128))
        out = torch.empty(m, n, device="cuda")
        opa, opb = (a_in.t() if ta else a_in), (b_in.t() if tb else b_in)
        us_t = timeit(lambda: torch.matmul(opa, opb, out=out), 20)
        us = timeit(lambda: hip.gemm_f32(a_in, b_in, ta, tb, out=out), 20)
        print(json.dumps({"mnk": [m, n, k], "trans": [ta, tb], "exact_on_integers": exact,
                          "ours_us": round(us, 1), "torch_us": round(us_t, 1),
                          "tile": hip.lib().dtf_gemm_f32_tile(m, n, k)}), flush=True)
