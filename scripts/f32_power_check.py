"""Is the large fp32 GEMM limited by power (clock) or by cycles? A power-limited kernel runs
faster on all-zero operands (no toggling in the multipliers); a cycle-bound one does not care.
Prints TFLOP/s of the hand-written kernel and of torch.matmul for random and zero operands."""
import json
import sys

import torch

sys.path.insert(0, ".")
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda")


def rate(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return 2.0 * n ** 3 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e12


out = torch.empty((n, n), device=dev)
for kind in ("random", "zeros", "random"):
    a = torch.randn(n, n, device=dev) if kind == "random" else torch.zeros(n, n, device=dev)
    b = torch.randn(n, n, device=dev) if kind == "random" else torch.zeros(n, n, device=dev)
    print(json.dumps({"n": n, "operands": kind,
                      "kernel_tflops": round(rate(lambda: hip.gemm_f32(a, b, out=out)), 1),
                      "torch_tflops": round(rate(lambda: torch.matmul(a, b, out=out)), 1)}),
          flush=True)
