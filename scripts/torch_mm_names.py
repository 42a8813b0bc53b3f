import sys, torch
shapes = [(2048,2048,2048),(3072,2048,4096),(2048,4096,4096),(4096,4096,4096),(1536,4096,8192)]
for m,n,k in shapes:
    a = torch.randn(m,k,device="cuda",dtype=torch.bfloat16); bt = torch.randn(n,k,device="cuda",dtype=torch.bfloat16)
    out = torch.empty(m,n,device="cuda",dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, bt.t(), out=out)
    torch.cuda.synchronize()
