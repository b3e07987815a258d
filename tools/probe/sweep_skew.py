"""Probe: start-up phase skew of the product NT kernel (gemm_nt_core.h: skew_n, skew_len) on the epilogue-bound shapes."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import probe_lib as pl  # noqa: E402
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402
from sweep_epilogue_shapes import cases, rnd, timeit, OP  # noqa: E402

L = lib()
for name, M_, N_, K_, epi in cases:
    A = rnd(M_, K_).to(OP); W = (rnd(N_, K_) * 0.02).to(OP)
    args = dict(bias=rnd(N_))
    if epi == L.PVRL_EPI_RESID_F32:
        args["aux"] = rnd(M_, N_)
    if epi == L.PVRL_EPI_DGELU:
        args["aux"] = rnd(M_, N_).to(OP); args.pop("bias")
    line = [f"{name:22s}"]
    for sk in ("0,0", "2,4", "2,8", "2,12", "3,4", "3,8", "4,2", "4,4", "4,6", "8,1", "8,2", "8,3", "16,1"):
        os.environ["PVRL_PROBE_SKEW"] = sk
        us = timeit(lambda: pl.gemm_nt(3, A, W, epi, **args))
        line.append(f"[{sk}]:{us:.0f}")
    print("  ".join(line), flush=True)
