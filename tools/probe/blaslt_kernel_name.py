"""Probe: which hipBLASLt kernel torch.matmul picks for a shape (run under rocprofv3 --kernel-trace; the name carries the macro tile)."""
import sys
import torch
M, N, K = (int(v) for v in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
for _ in range(5):
    torch.matmul(A, W.t())
torch.cuda.synchronize()
