#!/usr/bin/env python3
"""Bandwidth of the optimizer update kernels on large tensors (bytes moved per second against
what a plain device-to-device copy of the same size reaches on the same box)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters * 1e-3


def main():
    rows = []
    for n in (1 << 20, 1 << 24, 1 << 26, 1 << 28):
        w, g, m, v = (torch.rand(n, device="cuda") + 0.1 for _ in range(4))
        p1, p2 = torch.tensor(0.9).cuda(), torch.tensor(0.999).cuda()
        cases = {
            "copy": (lambda: w.copy_(g), 8),
            "sgd": (lambda: hip.multi_sgd([w], [g], [1e-3]), 12),
            "momentum": (lambda: hip.multi_optim("momentum", [w], [g], [m], (), (1e-3, 0.9)), 20),
            "adam": (lambda: hip.multi_optim("adam", [w], [g], [m], [v], (1e-3, 0.9, 0.999, 1e-8),
                                             [p1, p2]), 28),
        }
        for name, (fn, bytes_per) in cases.items():
            secs = timed(fn)
            rows.append({"kernel": name, "elements": n, "us": round(secs * 1e6, 1),
                         "TB_per_s": round(n * bytes_per / secs / 1e12, 2)})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
