"""Effective bandwidth of the full reduction (the Mean over a matmul result in the 1000x1000
example and its larger variants)."""
import json, sys
import torch
sys.path.insert(0, ".")
from distributed_tensorflow_amd.ops import hip  # noqa: E402
for n in (1000 * 1000, 4096 * 4096, 16384 * 16384):
    x = torch.randn(n, device="cuda")
    hip.reduce_all(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.reduce_all(x)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(json.dumps({"n": n, "us": round(us, 2), "GBps": round(4 * n / us / 1e3, 1),
                      "torch_sum_us": None}), flush=True)
