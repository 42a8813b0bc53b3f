"""Is the default plan's launch the same as the explicit (tile, splits) launch it names? Alternating timings."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
def timed(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
m, k, n = 1536, 4096, 1536
a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); out = torch.empty(m, n, device="cuda")
L = hip.lib()
print("plan", L.dtf_gemm_f32_tile(m, n, k), L.dtf_gemm_f32_splits(m, n, k, None))
for r in range(3):
    print("default %.1f  explicit(64,4) %.1f  explicit(64,1) %.1f" % (
        timed(lambda: hip.gemm_f32(a, b, out=out)), timed(lambda: hip.gemm_f32(a, b, out=out, tile=64, splits=4)),
        timed(lambda: hip.gemm_f32(a, b, out=out, tile=64, splits=1))))
c1 = hip.gemm_f32(a, b); c2 = hip.gemm_f32(a, b, tile=64, splits=4)
print("bit equal", bool(torch.equal(c1, c2)))
