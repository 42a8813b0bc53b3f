#!/usr/bin/env python3
"""fp8 GEMM on shapes whose 256-row tiling fills the chip badly: the default launch (192- or
128-row block tiles where `dtf_pp_rows` picks them) against 256-row tiles forced
(DTF_FP8_FORCE_256, DTF_GEMM_EXPERIMENTS build) and torch._scaled_mm; exact on integer data."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

FP8 = torch.float8_e4m3fn
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")]
one = torch.ones((), device="cuda")


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for m, n, k in shapes:
    g = torch.Generator().manual_seed(m + n)
    ai = torch.randint(-2, 3, (m, k), generator=g).float()
    bi = torch.randint(-2, 3, (n, k), generator=g).float()
    got = hip.gemm_fp8_nt(ai.to(FP8).cuda(), bi.to(FP8).cuda(), torch.float32)
    exact = bool(torch.equal(got, ai.cuda() @ bi.cuda().t()))
    a = (torch.rand(m, k, device="cuda") * 2 - 1).to(FP8)
    bt = (torch.rand(n, k, device="cuda") * 2 - 1).to(FP8)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)

    def ours(force):
        def run():
            if force:
                os.environ["DTF_FP8_FORCE_256"] = "1"
            else:
                os.environ.pop("DTF_FP8_FORCE_256", None)
            hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out)
        return run
    kernels = {"torch._scaled_mm": lambda: torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one,
                                                            out_dtype=torch.bfloat16),
               "default": ours(False), "rows256": ours(True)}
    iters = max(3, int(1e-2 / (2.0 * m * n * k / 2.8e15)))
    res = {}
    for _ in range(3):
        for name, fn in kernels.items():
            res.setdefault(name, []).append(2.0 * m * n * k / timed(fn, iters) / 1e12)
    print(json.dumps({"shape": [m, n, k], "exact_on_integers": exact,
                      **{name: round(sorted(v)[1], 1) for name, v in res.items()}}), flush=True)
