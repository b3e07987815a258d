"""Probe: torch.matmul(P.t(), Q) (hipBLASLt under PyTorch-ROCm) on the step's weight-gradient shapes next to pvrl_gemm_tn_bf16 and
the grouped launch of one block's seven gradients.  usage: python tools/probe/blaslt_compare_tn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from procedurevrl_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, reps=10):
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


M = 50432
shapes = [(768, 768), (2304, 768), (3072, 768), (768, 3072)]
tot_lib = tot_own = 0.0
probs = []
for (N, K) in shapes:
    P = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    Q = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    t_lib = timeit(lambda: torch.matmul(P.t(), Q))
    t_own = timeit(lambda: ops.gemm_tn(P, Q, dW, dbias=db))
    fl = 2.0 * M * N * K / 1e6
    print(f"dW {N:5d} x {K:5d} (M {M}):  matmul(P.t(), Q) {t_lib:7.1f} us ({fl / t_lib:5.0f} TF/s, bf16 result, no bias gradient)   "
          f"pvrl_gemm_tn_bf16 {t_own:7.1f} us ({fl / t_own:5.0f}; fp32 result + column sums)", flush=True)
    probs.append((P, Q, dW, db, fl, t_lib))
# one block's seven gradients: qkv x2 (temporal + spatial), proj-like x3 (768 x 768), fc1, fc2
blk = [1, 1, 0, 0, 0, 2, 3]
t_lib7 = sum(probs[i][5] for i in blk)
fl7 = sum(probs[i][4] for i in blk)
group = [(probs[i][0], probs[i][1], probs[i][2].clone(), probs[i][3].clone(), 0.0) for i in blk]
try:
    t_grp = timeit(lambda: ops.gemm_tn_grouped(group))
    print(f"one block's seven gradients: library, seven calls {t_lib7:7.1f} us ({fl7 / t_lib7:5.0f} TF/s)   grouped launch + reduces {t_grp:7.1f} us ({fl7 / t_grp:5.0f})")
except Exception as e:  # noqa
    print("grouped call failed:", repr(e))
