#!/usr/bin/env python3
"""Socket power and shader clock while one GEMM kernel runs back to back for a few seconds:
torch.matmul (hipBLASLt) against this library's bf16 kernel and its schedule variants. Power is
hwmon power1_input, the clock the starred level of pp_dpm_sclk, sampled every 20 ms by a host
thread; the first half second of each run is dropped."""
import glob, json, os, re, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
dev = "/sys/class/drm/card0/device"
power_file = (glob.glob(dev + "/hwmon/hwmon*/power1_input") + [None])[0]


def sample():
    p = c = None
    try:
        p = int(open(power_file).read()) / 1e6
    except Exception:
        pass
    try:
        m = re.search(r"(\d+)Mhz \*", open(dev + "/pp_dpm_sclk").read())
        c = int(m.group(1)) if m else None
    except Exception:
        pass
    return p, c


a = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
bt = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
kernels = {"torch": lambda: torch.matmul(a, bt.t(), out=out),
           "shipped": lambda: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out)}
for v in variants:
    kernels["variant%d" % v] = (lambda v=v: hip.gemm_bf16_nt(a, bt, torch.bfloat16, out=out,
                                                              variant=v))
for name, fn in kernels.items():
    fn()
    torch.cuda.synchronize()
    time.sleep(1.0)
    samples, stop = [], threading.Event()

    def watch():
        t0 = time.time()
        while not stop.is_set():
            samples.append((time.time() - t0,) + sample())
            time.sleep(0.02)
    th = threading.Thread(target=watch)
    th.start()
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    count = 0
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(8):
            fn()
        count += 8
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / count
    kept = [s for s in samples if s[0] > 0.5]
    pw = [s[1] for s in kept if s[1] is not None]
    ck = [s[2] for s in kept if s[2] is not None]
    print(json.dumps({"n": n, "kernel": name, "tflops": round(2.0 * n ** 3 / ms / 1e9, 1),
                      "watts_mean": round(sum(pw) / len(pw), 1) if pw else None,
                      "watts_max": max(pw) if pw else None,
                      "sclk_mhz_mean": round(sum(ck) / len(ck), 1) if ck else None,
                      "samples": len(kept)}), flush=True)
