"""Probe: the distinct NT GEMM shapes of an MViTv2-S step under every tile variant of the probe library
(1 = 128x128, 2 = 256x128, 3 = 256x256 [product dispatch picks among these], 7/8/9 = multi-workgroup-per-CU ring variants)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402
import probe_lib as pl  # noqa: E402

DEV = "cuda:0"
L = lib()
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
shapes = []
for (tok, dim) in ((25088, 96), (6272, 192), (1568, 384), (392, 768)):
    M = B * tok + B
    C = (dim + 127) // 128 * 128
    C3, C4 = (3 * dim + 127) // 128 * 128, 4 * dim
    shapes += [("qkv", M, C3, C, L.PVRL_EPI_BF16), ("proj", M, C, C, L.PVRL_EPI_RESID_F32), ("fc1", M, C4, C, L.PVRL_EPI_GELU),
               ("fc2", M, C, C4, L.PVRL_EPI_RESID_F32), ("dfc2", M, C4, C, L.PVRL_EPI_DGELU), ("dfc1", M, C, C4, L.PVRL_EPI_BF16),
               ("dqkv", M, C, C3, L.PVRL_EPI_F32)]
for name, M, N, K, epi in shapes:
    A = rnd(M, K).to(OP); W = (rnd(N, K) * 0.02).to(OP)
    kw = {}
    if epi == L.PVRL_EPI_RESID_F32:
        kw["aux"] = rnd(M, N); kw["bias"] = rnd(N)
    elif epi == L.PVRL_EPI_DGELU:
        kw["aux"] = rnd(M, N).to(OP)
    else:
        kw["bias"] = rnd(N)
    row = [f"{name:5s} M {M:7d} N {N:5d} K {K:5d} epi {epi}: product {timeit(lambda: ops.gemm_nt(A, W, epi, **kw)):7.1f}"]
    for t in (1, 2, 3, 7, 8, 15, 16):
        if t == 3 and N % 256:
            row.append("  t3    -  ")
            continue
        try:
            row.append(f"  t{t} {timeit(lambda: pl.gemm_nt(t, A, W, epi, **kw)):7.1f}")
        except Exception as e:  # noqa
            row.append(f"  t{t}  err  ")
    print("".join(row), flush=True)
    del A, W, kw
