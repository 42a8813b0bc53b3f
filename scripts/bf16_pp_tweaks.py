#!/usr/bin/env python3
"""Schedule tweaks of the bf16 ping-pong kernel against the shipped schedule and torch.matmul
(DTF_GEMM_EXPERIMENTS build). Tweak bits (variant = 4 | bits << 4): 64 = LDS reads waited for in
front of the phase's first barrier, 128 / 512 / 640 = matrix pipe handed over 2 / 4 / 8 MFMAs
early, 256 = no s_setprio. Unlike the ablations these kernels are correct: every result is
compared bit for bit with the 256 x 256-tile kernel's (variant 4). Interleaved rounds, best and median per kernel."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

sizes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1].split(",") if len(sys.argv) > 1
                                                       else ["16384x16384x16384"])]
# entries: schedule bits of the ping-pong kernel, or "v<variant>" for a whole launcher variant
tweaks = sys.argv[2].split(",") if len(sys.argv) > 2 else ["256", "2048", "2304", "4096"]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for m, n, k in sizes:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    bt = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ref = hip.gemm_bf16_nt(a, bt, torch.bfloat16, variant=4).clone()
    kernels = {"torch": lambda: torch.matmul(a, bt.t(), out=out),
               "shipped": lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)}
    for t in tweaks:
        v = int(t[1:]) if t.startswith("v") else 4 | (int(t) << 4)
        kernels[("variant" if t.startswith("v") else "tweak") + t.lstrip("v")] = (
            lambda v=v: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=v))
    equal = {}
    for name, fn in kernels.items():
        out.zero_()
        fn()
        torch.cuda.synchronize()
        equal[name] = bool(torch.equal(out, ref)) if name != "torch" else None
    iters = max(3, int(2e-2 / (2.0 * m * n * k / 1.4e15)))
    times = {name: [] for name in kernels}
    for _ in range(rounds):
        for name, fn in kernels.items():
            times[name].append(timed(fn, iters))
    for name in kernels:
        ts = sorted(times[name])
        tf = [2.0 * m * n * k / t / 1e12 for t in ts]
        print(json.dumps({"shape": [m, n, k], "kernel": name, "bit_equal_to_256_tiles": equal[name],
                          "best_tflops": round(tf[0], 1), "median_tflops": round(tf[len(tf) // 2], 1)}),
              flush=True)
