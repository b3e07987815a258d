"""Probe: the weight-gradient (TN) GEMM shapes of an MViTv2-S step: product dispatch vs the 128x128 kernel (probe tile 1)
and the register-transposed 256x256 kernel (probe tile 8), with their own slice counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from procedurevrl_amd import ops  # noqa: E402
import probe_lib as pl  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
OP = ops.OP16


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B = 32
for (tok, dim) in ((25088, 96), (6272, 192), (1568, 384), (392, 768)):
    M = B * tok + B
    C = (dim + 127) // 128 * 128
    C3, C4 = (3 * dim + 127) // 128 * 128, 4 * dim
    for name, N, K in (("qkv", C3, C), ("proj", C, C), ("fc1", C4, C), ("fc2", C, C4)):
        P = rnd(M, N).to(OP); Q = rnd(M, K).to(OP)
        dW = torch.empty(N, K, device=DEV); db = torch.empty(N, device=DEV)
        row = [f"{name:5s} M {M:7d} N {N:5d} K {K:5d}: product {timeit(lambda: ops.gemm_tn(P, Q, dW, db, beta=0.0)):7.1f} (splits {ops.tn_splits(M, N, K)})"]
        for t in (1, 8):
            try:
                sp = pl.tn_splits(t, M, N, K)
                row.append(f"  t{t} {timeit(lambda: pl.gemm_tn(t, P, Q, dW, db, beta=0.0)):7.1f} (splits {sp})")
                if t == 1:
                    for sp2 in (sp // 2, sp * 2):
                        if sp2 >= 8 and sp2 % 8 == 0:
                            row.append(f" [{sp2}: {timeit(lambda: pl.gemm_tn(t, P, Q, dW, db, beta=0.0, splits=sp2)):7.1f}]")
            except Exception as e:  # noqa
                row.append(f"  t{t} err {e}")
        print("".join(row), flush=True)
        del P, Q
