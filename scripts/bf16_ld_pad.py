#!/usr/bin/env python3
"""Does the power-of-two row pitch of the operands cost L2 set conflicts? The same bf16 product with
operands stored contiguously (pitch K) and as views of wider buffers (pitch K + pad), ours and
torch.matmul, interleaved rounds."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pads = [int(p) for p in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 64, 192]
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
rounds = 5


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
for pad in pads:
    a = ((torch.rand(n, n + pad, device="cuda") * 2 - 1).to(torch.bfloat16))[:, :n]
    bt = ((torch.rand(n, n + pad, device="cuda") * 2 - 1).to(torch.bfloat16))[:, :n]
    kernels = {"torch": lambda: torch.matmul(a, bt.t(), out=out)}
    for v in variants:
        kernels["variant%d" % v] = (lambda v=v: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out,
                                                                  variant=v))
    times = {k: [] for k in kernels}
    for _ in range(rounds):
        for k, fn in kernels.items():
            times[k].append(timed(fn))
    for k in kernels:
        ts = sorted(times[k])
        print(json.dumps({"n": n, "pitch_pad_elements": pad, "kernel": k,
                          "median_tflops": round(2.0 * n ** 3 / ts[len(ts) // 2] / 1e12, 1),
                          "best_tflops": round(2.0 * n ** 3 / ts[0] / 1e12, 1)}), flush=True)
