"""Probe: what does the bias gradient (column sums of dY, computed by the wk == 0 waves of the tk == 0 tiles) cost the grouped
weight-gradient launch?  Times a block's seven weight gradients with and without dbias."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from procedurevrl_amd import ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
R, M, BT = 50176, 50208, 256
shapes = [(M, 768, 3072), (M, 3072, 768), (R + BT, 768, 768), (M, 2304, 768), (R, 768, 768), (R, 2304, 768), (R, 768, 768)]
P = [(torch.randn(m, N, device=DEV, generator=g) * 0.05).to(ops.OP16) for m, N, K in shapes]
Q = [torch.randn(m, K, device=DEV, generator=g).to(ops.OP16) for m, N, K in shapes]
dW = [torch.empty(N, K, device=DEV) for m, N, K in shapes]
db = [torch.empty(N, device=DEV) for m, N, K in shapes]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = sum(2.0 * m * N * K for m, N, K in shapes)
for rep in range(2):
    for with_bias in (True, False):
        probs = [(P[i], Q[i], dW[i], db[i] if with_bias else None, 0.0) for i in range(7)]
        us = timeit(lambda: ops.gemm_tn_grouped(probs))
        print(f"grouped TN, dbias {'on ' if with_bias else 'off'}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
