#!/usr/bin/env python3
"""Schedule variants of the fp8 ping-pong kernel (DTF_GEMM_EXPERIMENTS build; DTF_FP8_TWEAK bits:
256 = s_setprio around the MFMA segment, 2048 = LDS reads issued ahead of the phase's DMA -- the
two choices the shipped schedule dropped) against the shipped schedule
and torch._scaled_mm, interleaved rounds, bit-equality with the shipped kernel checked."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
FP8 = torch.float8_e4m3fn
a = (torch.rand(n, n, device="cuda") * 2 - 1).to(FP8)
bt = (torch.rand(n, n, device="cuda") * 2 - 1).to(FP8)
out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
one = torch.ones((), device="cuda")


def ours(bits):
    def run():
        if bits:
            os.environ["DTF_FP8_TWEAK"] = str(bits)
        else:
            os.environ.pop("DTF_FP8_TWEAK", None)
        hip.gemm_fp8_nt(a, bt, torch.bfloat16, out=out)
    return run


kernels = {"torch._scaled_mm": lambda: torch._scaled_mm(a, bt.t(), scale_a=one, scale_b=one,
                                                        out_dtype=torch.bfloat16),
           "shipped": ours(0)}
for t in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["256", "2048", "2304"]):
    kernels["tweak" + t] = ours(int(t))
ours(0)()
ref = out.clone()
equal = {}
for name, fn in kernels.items():
    if name.startswith("torch"):
        continue
    out.zero_()
    fn()
    torch.cuda.synchronize()
    equal[name] = bool(torch.equal(out, ref))
times = {k: [] for k in kernels}
for _ in range(rounds):
    for name, fn in kernels.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e-3 / 5)
for name in kernels:
    ts = sorted(times[name])
    print(json.dumps({"n": n, "kernel": name, "bit_equal_to_shipped": equal.get(name),
                      "best_tflops": round(2.0 * n ** 3 / ts[0] / 1e12, 1),
                      "median_tflops": round(2.0 * n ** 3 / ts[len(ts) // 2] / 1e12, 1)}), flush=True)
