"""Probe: every NT variant (tools/probe/libpvrl_probe.so) on the epilogue-bound shapes of the 32-clip step -- fp32-residual
K=768, GELU pair, dGELU -- next to the product kernel.  Usage: python tools/probe/sweep_epilogue_shapes.py"""
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
OP = ops.OP16
g = torch.Generator(device=DEV).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
M = 50208
R = 50176


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cases = [("resid K=768 N=768", R, 768, 768, L.PVRL_EPI_RESID_F32), ("bf16  K=768 N=768", R, 768, 768, L.PVRL_EPI_BF16),
         ("gelu  K=768 N=3072", M, 3072, 768, L.PVRL_EPI_GELU), ("dgelu K=768 N=3072", M, 3072, 768, L.PVRL_EPI_DGELU),
         ("resid K=3072 N=768", M, 768, 3072, L.PVRL_EPI_RESID_F32), ("bf16 K=768 N=2304", M, 2304, 768, L.PVRL_EPI_BF16)]
def main():
  for name, M_, N_, K_, epi in cases:
      A = rnd(M_, K_).to(OP); W = (rnd(N_, K_) * 0.02).to(OP)
      args = dict(bias=rnd(N_))
      if epi == L.PVRL_EPI_RESID_F32:
          args["aux"] = rnd(M_, N_)
      if epi == L.PVRL_EPI_DGELU:
          args["aux"] = rnd(M_, N_).to(OP); args.pop("bias")
      us = timeit(lambda: ops.gemm_nt(A, W, epi, **args))
      line = [f"{name:22s} product {us:7.1f} us ({2.0 * M_ * N_ * K_ / us / 1e6:6.0f} TF)"]
      for tile in (3, 14, 3, 14):
          try:
              us = timeit(lambda: pl.gemm_nt(tile, A, W, epi, **args))
              line.append(f"t{tile}:{us:.0f}")
          except Exception as e:  # noqa
              line.append(f"t{tile}:ERR")
      print("  ".join(line), flush=True)


if __name__ == "__main__":
  main()
