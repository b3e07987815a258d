"""Results check of every GEMM variant in tools/probe/libpvrl_probe.so against CPU fp32 math on bf16-rounded operands
(run on the GPU box: `python tools/probe/check_variants.py`).  These kernels are benchmark probes, not product code."""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import probe_lib as pl  # noqa: E402
from kernel_checks import BF, TOL_BF16, bf, dev, rel  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402


def check_nt():
    L = lib()
    g = torch.Generator().manual_seed(41)
    out = []
    for (M, N, K) in [(700, 512, 768), (300, 256, 192), (260, 256, 64)]:
        A = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) * 0.05
        bias = torch.randn(N, generator=g); resid = torch.randn(M, N, generator=g)
        ref = bf(A) @ bf(W).t()
        Ad, Wd = A.to(dev(), BF), W.to(dev(), BF)
        for knob in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14):
            o = pl.gemm_nt(knob, Ad, Wd, L.PVRL_EPI_BF16, bias=bias.to(dev()))
            out.append((f"gemm_nt tile{knob} bf16 {M}x{N}x{K}", rel(o, ref + bias), TOL_BF16))
            o = pl.gemm_nt(knob, Ad, Wd, L.PVRL_EPI_RESID_F32, bias=bias.to(dev()), aux=resid.to(dev()))
            out.append((f"gemm_nt tile{knob} resid {M}x{N}x{K}", rel(o, resid + ref + bias), 1e-4))
            u, gl = pl.gemm_nt(knob, Ad, Wd, L.PVRL_EPI_GELU, bias=bias.to(dev()))
            out.append((f"gemm_nt tile{knob} gelu {M}x{N}x{K}", rel(gl, F.gelu(ref + bias)), TOL_BF16))
    return out


def check_tn():
    g = torch.Generator().manual_seed(31)
    out = []
    M, N, K = 1111, 512, 256
    P = torch.randn(M, N, generator=g); Q = torch.randn(M, K, generator=g)
    ref = bf(P).t() @ bf(Q)
    for knob, name in ((1, "128x128"), (2, "lds-dma"), (3, "256x256"), (4, "256x256 w128")):
        dW = torch.zeros(N, K, device=dev()); db = torch.zeros(N, device=dev())
        pl.gemm_tn(knob, P.to(dev(), BF), Q.to(dev(), BF), dW, db, splits=16)
        out.append((f"gemm_tn[{name}] dW", rel(dW, ref), 1e-4))
        out.append((f"gemm_tn[{name}] dbias", rel(db, bf(P).sum(0)), 1e-4))
    # the 4-wave 256x256 kernels need M % 64 == 0; slices of 2, 4, ... stages and empty slices
    for knob, name in ((5, "ring"), (6, "rt"), (7, "rt32"), (8, "rt8")):
        for (M2, N2, K2, sp) in [(1152, 512, 256, 8), (4160, 256, 768, 16), (128, 256, 256, 8), (6400, 768, 768, 32),
                                 (3200, 768, 256, 9 if knob >= 6 else 8)] + ([(1111, 512, 256, 5), (1569, 256, 256, 3), (40, 256, 512, 4)] if knob >= 6 else []):
            P2 = torch.randn(M2, N2, generator=g); Q2 = torch.randn(M2, K2, generator=g)
            ref2 = bf(P2).t() @ bf(Q2)
            dW = torch.zeros(N2, K2, device=dev()); db = torch.zeros(N2, device=dev())
            pl.gemm_tn(knob, P2.to(dev(), BF), Q2.to(dev(), BF), dW, db, splits=sp)
            out.append((f"gemm_tn[{name}] dW {M2}x{N2}x{K2} s={sp}", rel(dW, ref2), 1e-4))
            out.append((f"gemm_tn[{name}] dbias {M2}x{N2}x{K2}", rel(db, bf(P2).sum(0)), 1e-4))
    return out


if __name__ == "__main__":
    bad = 0
    for fn in (check_nt, check_tn):
        for label, err, tol in fn():
            ok = err <= tol
            bad += not ok
            if not ok:
                print(f"FAIL {label}: {err:.3e} > {tol:g}")
    print("variants:", "all ok" if not bad else f"{bad} FAILED")
    sys.exit(1 if bad else 0)
