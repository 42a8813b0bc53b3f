#!/usr/bin/env python3
"""Single-GPU GEMM kernel benchmark: TFLOP/s of the hand-written kernels on random data,
with torch.matmul (hipBLASLt) measured in the same process for reference only."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def timeit(fn, iters: int, warmup: int = 10) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters   # ms


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,2048,4096,8192,16384")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-torch", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rows = []
    for size in args.sizes.split(","):
        # "4096" = a cube, "2048x16384x16384" = M x N x K (a rank's share of a row-sharded product)
        m, n, k = (int(v) for v in size.split("x")) if "x" in size else (int(size),) * 3
        flops = 2.0 * m * n * k
        if args.dtype == "bf16":
            a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
            bt = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
            out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
            ours = timeit(lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=args.variant), args.iters)
            ref = None if args.no_torch else timeit(lambda: torch.matmul(a, bt.t()), args.iters)
            if max(m, n, k) <= 4096:
                err = (out.float() - a.float() @ bt.float().t()).abs().max().item()
            else:
                err = (out[:256].float() - a[:256].float() @ bt.float().t()).abs().max().item()
        elif args.dtype == "fp8":
            a = hip.philox_window(m, k, 1, 1, 0, 0, k, torch.float8_e4m3fn, "cuda", False, -1.0, 2.0)
            bt = hip.philox_window(n, k, 1, 2, 0, 0, k, torch.float8_e4m3fn, "cuda", False, -1.0, 2.0)
            out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
            ours = timeit(lambda: hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out), args.iters)
            ref = None
            if not args.no_torch and hasattr(torch, "_scaled_mm"):
                one = torch.ones((), device="cuda")
                try:
                    ref = timeit(lambda: torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one,
                                                          out_dtype=torch.bfloat16), args.iters)
                except Exception as e:  # noqa: BLE001 -- reference only
                    print("torch._scaled_mm unavailable: %s" % str(e)[:100], file=sys.stderr)
            err = (out[:256].float() - a[:256].float() @ bt.float().t()).abs().max().item()
        else:
            a = torch.rand(m, k, device="cuda") * 2 - 1
            b = torch.rand(k, n, device="cuda") * 2 - 1
            out = torch.empty(m, n, device="cuda")
            ours = timeit(lambda: hip.gemm_f32(a, b, out=out), args.iters)
            ref = None if args.no_torch else timeit(lambda: torch.matmul(a, b), args.iters)
            err = (out[:128].double() - a[:128].double() @ b.double()).abs().max().item()
        row = {"dtype": args.dtype, "variant": args.variant, "m": m, "n": n, "k": k,
               "ms": round(ours, 4),
               "tflops": round(flops / ours / 1e9, 1), "max_err": err}
        if ref is not None:
            row["torch_ms"] = round(ref, 4)
            row["torch_tflops"] = round(flops / ref / 1e9, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
