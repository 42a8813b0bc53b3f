#!/usr/bin/env python3
"""Per-segment cycle breakdown of the bf16 ping-pong GEMM's phases (timing experiment).

Needs the kernel library built with ``DTF_GEMM_EXPERIMENTS=1 python build_native.py``: the
instrumented instantiation accumulates, per wave and per phase, the shader-clock cycles spent in
[issue LDS reads + DMA | counted vmcnt wait | barrier | 32 MFMAs | barrier] and overwrites the head
of each output tile with the sums. Prints the mean over tiles for one wave of each group.
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--modes", default="8,10,9")
    args = ap.parse_args()
    n = args.n
    a = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    bt = torch.randn(n, n, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    names = {8: "shipped", 10: "all_l2_hit", 9: "no_dma"}
    for mode in [int(m) for m in args.modes.split(",")]:
        for _ in range(2):
            hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out, variant=4 | (mode << 4))
        torch.cuda.synchronize()
        o32 = out.view(torch.int32)
        tiles = n // 256
        acc = torch.zeros(8, 4, 6, dtype=torch.float64)
        cnt = 0
        for tm in range(0, tiles, 7):
            for tn in range(0, tiles, 5):
                blk = o32[tm * 256: tm * 256 + 8, tn * 128: tn * 128 + 24].cpu().to(torch.float64)
                acc += blk.view(8, 4, 6)
                cnt += 1
        acc /= cnt
        for wave in (0, 4):
            for p in range(4):
                row = acc[wave, p]
                k = max(1.0, float(row[5]))
                print(json.dumps({"mode": names.get(mode, mode), "wave": wave, "phase": p,
                                  "issue": round(float(row[0]) / k, 1),
                                  "vmwait": round(float(row[1]) / k, 1),
                                  "bar1": round(float(row[2]) / k, 1),
                                  "mfma": round(float(row[3]) / k, 1),
                                  "bar2": round(float(row[4]) / k, 1),
                                  "total": round(float(row[:5].sum()) / k, 1), "count": k}),
                      flush=True)


if __name__ == "__main__":
    main()
