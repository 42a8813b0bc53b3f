"""GPU-side time (run under rocprofv3 --kernel-trace) of fp32 products under the default K-split plan, unsplit and
split in two: 20 calls per case, cases separated by a fill launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
shapes = [tuple(int(v) for v in t.split('x')) for t in sys.argv[1].split(',')] if len(sys.argv) > 1 else [(512,512,512),(768,768,768),(1280,1280,1280),(1024,1024,1024),(640,640,640),(896,896,896)]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); out = torch.empty(m, n, device="cuda")
    for s in (None, 1, 2):
        # marker launch: a fill whose size encodes the case
        for _ in range(20):
            hip.gemm_f32(a, b, out=out, tile=64, splits=s)
        torch.cuda.synchronize()
        torch.zeros(m + (0 if s is None else s), device="cuda")
torch.cuda.synchronize()
