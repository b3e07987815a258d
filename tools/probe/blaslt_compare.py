"""Probe: torch.matmul / F.linear (hipBLASLt / rocBLAS under PyTorch-ROCm) on the step's plain GEMM shapes next to the product's
NT kernel -- a reference point for how far the hand-written kernel is from the vendor library on this box.
usage: python tools/probe/blaslt_compare.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402

L = lib()
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (M, N, K) in [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (50432, 768, 2304), (65536, 768, 768),
                  (65536, 3072, 768), (65536, 768, 3072), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g)
    bb = b.to(torch.bfloat16)
    t_lib = timeit(lambda: torch.nn.functional.linear(A, W, bb))
    t_mm = timeit(lambda: torch.matmul(A, W.t()))
    t_own = timeit(lambda: ops.gemm_nt(A, W, L.PVRL_EPI_BF16, bias=b)) if ops.OP16 == torch.bfloat16 else float("nan")
    fl = 2.0 * M * N * K / 1e6
    print(f"M {M:6d} N {N:5d} K {K:5d}:  F.linear {t_lib:7.1f} us ({fl / t_lib:5.0f} TF/s)   matmul {t_mm:7.1f} us ({fl / t_mm:5.0f})   "
          f"pvrl_gemm_nt_bf16 {t_own:7.1f} us ({fl / t_own:5.0f})", flush=True)
