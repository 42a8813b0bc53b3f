"""Which hipBLASLt kernel does torch.matmul run for these bf16 NT shapes? Run under
`rocprofv3 --kernel-trace --stats` and read the kernel names."""
import sys, torch
shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
    [(2048, 2048, 2048), (2048, 2048, 4096), (2560, 2560, 2560), (3072, 2048, 4096), (2048, 4096, 4096)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    bt = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, bt.t(), out=out)
    torch.cuda.synchronize()
