"""GPU-side time of torch.matmul (fp32) for a list of MxKxN shapes: run under rocprofv3 --kernel-trace; cases are
separated by a fill launch."""
import sys, torch
shapes = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for m, k, n in shapes:
    a = torch.randn(m, k, device="cuda"); b = torch.randn(k, n, device="cuda"); out = torch.empty(m, n, device="cuda")
    for _ in range(20):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    torch.zeros(m, device="cuda")
torch.cuda.synchronize()
