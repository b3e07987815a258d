"""Probe: main-loop quality of every NT variant in tools/probe/libpvrl_probe.so on full-round shapes (M = 65,536: no ragged last round)
and the K-heavy shapes of the step, next to the product kernel and the vendor library.  usage: python tools/probe/nt_mainloop_sweep.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import probe_lib as pl  # noqa: E402
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402

L = lib()
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


for (M, N, K) in [(65536, 768, 3072), (65536, 3072, 768), (65536, 768, 768), (50432, 768, 2304), (50432, 768, 3072)]:
    A = torch.randn(M, K, device=DEV, generator=g).to(ops.OP16)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.02).to(ops.OP16)
    b = torch.randn(N, device=DEV, generator=g)
    fl = 2.0 * M * N * K / 1e6
    row = [f"M {M} N {N} K {K}:"]
    t = timeit(lambda: ops.gemm_nt(A, W, L.PVRL_EPI_BF16, bias=b)); row.append(f"product {t:.0f} us ({fl / t:.0f})")
    t = timeit(lambda: torch.matmul(A, W.t())); row.append(f"library {t:.0f} ({fl / t:.0f})")
    for tile in (3, 4, 5, 6, 10, 11, 12, 13):
        try:
            t = timeit(lambda: pl.gemm_nt(tile, A, W, L.PVRL_EPI_BF16, bias=b))
            row.append(f"t{tile} {t:.0f} ({fl / t:.0f})")
        except Exception as e:  # noqa
            row.append(f"t{tile} ERR")
    print("  ".join(row), flush=True)
