#!/usr/bin/env python3
"""Shader-clock cycles per k-pair of the fp32 GEMM's interior K-tiles (DTF_GEMM_EXPERIMENTS build,
instrumented instantiation DTF_F32_ABL=16). A k-pair of a 128x128 tile is 4 MFMAs of 64 cycles
per wave (256 cycles when the wave has the SIMD to itself, 512 with two waves per SIMD); of a
256x256 tile 16 MFMAs (1024 cycles, one wave per SIMD)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
os.environ["DTF_F32_ABL"] = "16"
L = hip.lib()
a = torch.randn(n, n, device="cuda")
b = torch.randn(n, n, device="cuda")
out = torch.empty(n, n, device="cuda")
tiles = [int(t) for t in sys.argv[2].split(',')] if len(sys.argv) > 2 else [128, 256]
for tile in tiles:
    wgs = ((n + tile - 1) // tile) ** 2
    ws = torch.zeros(wgs * 4 * 16, device="cuda")
    for _ in range(2):
        rc = L.dtf_gemm_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, n, n, n, n, n, n, 0, 0,
                            0, ws.data_ptr(), 1, tile, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    t = ws.view(wgs, 4, 16).cpu().double()
    mean = t.mean(dim=(0, 1))
    worst = t[:, :, :8].sum(dim=2).max()
    best = t[:, :, :8].sum(dim=2).min()
    print(json.dumps({"n": n, "tile": tile, "kpair_cycles": [round(float(v), 1) for v in mean[:8]],
                      "tile_cycles_mean": round(float(mean[:8].sum()), 1),
                      "tile_cycles_min_wave": round(float(best), 1),
                      "tile_cycles_max_wave": round(float(worst), 1),
                      "tiles_counted": float(mean[15])}), flush=True)
