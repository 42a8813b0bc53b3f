import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import hip
torch.manual_seed(0)
bad = 0
for m, k, n in [(32,8,32),(33,9,31),(100,784,100),(512,512,512),(384,384,384),(500,300,700),(768,768,768),(64,4096,64),(1,1,1),(257,1000,129)]:
    for ta in (False, True):
        for tb in (False, True):
            a = torch.randint(-3, 4, (k, m) if ta else (m, k)).float().cuda()
            b = torch.randint(-3, 4, (n, k) if tb else (k, n)).float().cuda()
            bias = torch.randint(-2, 3, (n,)).float().cuda()
            ref = (a.t() if ta else a) @ (b.t() if tb else b) + bias
            got = hip.gemm_f32(a, b, bias=bias, trans_a=ta, trans_b=tb, tile=32)
            if not torch.equal(got, ref):
                bad += 1; print("MISMATCH", m, k, n, ta, tb, float((got - ref).abs().max()))
            # strided / offset views
            if not ta and not tb and m > 4 and k > 8:
                big = torch.randint(-3, 4, (m, k + 5)).float().cuda()
                av = big[:, 3:3 + k]
                got = hip.gemm_f32(av, b, tile=32)
                if not torch.equal(got, av @ b):
                    bad += 1; print("MISMATCH view", m, k, n)
print("bad", bad)
