"""Run torch.matmul (hipBLASLt) on the headline shape so a kernel trace shows which macro-tile /
workgroup / LDS configuration the vendor library picks (reference information only)."""
import torch
n = 16384
a = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
bt = (torch.rand(n, n, device="cuda") * 2 - 1).bfloat16()
for _ in range(5):
    c = torch.matmul(a, bt.t())
torch.cuda.synchronize()
