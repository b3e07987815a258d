"""Probe: every GEMM of one MViTv2-S training step (32 clips of 16x224^2) at its padded shape -- time, the HBM floor of
its operand + result traffic at 8 TB/s, and the MFMA rate.  Says which of the shared GEMM kernels' launches are far
from either bound at MViT's narrow widths (96 .. 768)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from procedurevrl_amd import ops, ops_mvit as om  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402
from procedurevrl_amd.config import get_cfg  # noqa: E402
from procedurevrl_amd.mvit import mvit_plan  # noqa: E402

cfg = get_cfg()
cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE = 16, 224
mv = cfg.MVIT
mv.DEPTH, mv.NUM_HEADS, mv.EMBED_DIM = 16, 1, 96
mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING = [3, 7, 7], [2, 4, 4], [1, 3, 3]
mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
mv.POOL_KVQ_KERNEL, mv.POOL_KV_STRIDE_ADAPTIVE = [3, 3, 3], [1, 8, 8]
mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
thw0, plan = mvit_plan(cfg)
B, DEV = 32, "cuda:0"
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


seen = {}
tot = {}


def nt(tag, M, N, K, epi, out_bytes, extra_bytes=0):
    """C[M,N] = A[M,K] W[N,K]^T; bytes = A + W + outputs (+ aux inputs)"""
    key = ("nt", M, N, K, epi)
    if key not in seen:
        A = rnd(M, K).to(OP); W = (rnd(N, K) * 0.02).to(OP)
        kw = {}
        if epi == L.PVRL_EPI_RESID_F32:
            kw["aux"] = rnd(M, N); kw["bias"] = rnd(N)
        elif epi == L.PVRL_EPI_DGELU:
            kw["aux"] = rnd(M, N).to(OP)
        else:
            kw["bias"] = rnd(N)
        seen[key] = timeit(lambda: ops.gemm_nt(A, W, epi, **kw))
        del A, W, kw
    us = seen[key]
    byt = M * K * 2 + N * K * 2 + out_bytes + extra_bytes
    floor = byt / 8e12 * 1e6
    print(f"  {tag:14s} NT M {M:7d} N {N:5d} K {K:5d} epi {epi}: {us:7.1f} us  HBM floor {floor:6.1f} us ({floor / us:4.0%})  {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
    tot[tag.split()[0]] = tot.get(tag.split()[0], 0.0) + us
    return us


def tn(tag, M, N, K):
    key = ("tn", M, N, K)
    if key not in seen:
        P = rnd(M, N).to(OP); Q = rnd(M, K).to(OP)
        dW = torch.empty(N, K, device=DEV); db = torch.empty(N, device=DEV)
        seen[key] = timeit(lambda: ops.gemm_tn(P, Q, dW, db, beta=0.0))
        del P, Q
    us = seen[key]
    floor = (M * N * 2 + M * K * 2) / 8e12 * 1e6
    print(f"  {tag:14s} TN M {M:7d} N {N:5d} K {K:5d}       : {us:7.1f} us  HBM floor {floor:6.1f} us ({floor / us:4.0%})  {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
    tot[tag.split()[0]] = tot.get(tag.split()[0], 0.0) + us
    return us


p128 = om.pad128
step = 0.0
for i, pl in enumerate(plan):
    dim, dout, thw, sq = pl["dim"], pl["dim_out"], tuple(pl["in_thw"]), tuple(pl["stride_q"])
    Lin = thw[0] * thw[1] * thw[2]
    q_thw = om.pool_out_thw(thw, sq)
    Lout = q_thw[0] * q_thw[1] * q_thw[2]
    Mi, Mo = B * Lin + B, B * Lout + B
    Ci, Co, C3, C4 = p128(dim), p128(dout), p128(3 * dout), p128(4 * dout)
    print(f"blk {i:2d} dim {dim} -> {dout}, tokens {Lin} -> {Lout}")
    t = 0.0
    t += nt("fwd qkv", Mi, C3, Ci, L.PVRL_EPI_BF16, Mi * C3 * 2)
    if dim != dout:
        t += nt("fwd skip", Mi, Co, Ci, L.PVRL_EPI_F32, Mi * Co * 4)
    t += nt("fwd proj", Mo, Co, Co, L.PVRL_EPI_RESID_F32, Mo * Co * 4, Mo * Co * 4)
    t += nt("fwd fc1", Mo, C4, Co, L.PVRL_EPI_GELU, 2 * Mo * C4 * 2)
    t += nt("fwd fc2", Mo, Co, C4, L.PVRL_EPI_RESID_F32, Mo * Co * 4, Mo * Co * 4)
    t += nt("bwd fc2", Mo, C4, Co, L.PVRL_EPI_DGELU, Mo * C4 * 2, Mo * C4 * 2)
    t += tn("wg fc2", Mo, Co, C4)
    t += nt("bwd fc1", Mo, Co, C4, L.PVRL_EPI_BF16, Mo * Co * 2)
    t += tn("wg fc1", Mo, C4, Co)
    t += nt("bwd proj", Mo, Co, Co, L.PVRL_EPI_BF16, Mo * Co * 2)
    t += tn("wg proj", Mo, Co, Co)
    t += nt("bwd qkv", Mi, Ci, C3, L.PVRL_EPI_F32, Mi * Ci * 4)
    t += tn("wg qkv", Mi, C3, Ci)
    if dim != dout:
        t += nt("bwd skip", Mi, Ci, Co, L.PVRL_EPI_RESID_F32, Mi * Ci * 4, Mi * Ci * 4)
        t += tn("wg skip", Mi, Co, Ci)
    print(f"  block total {t / 1e3:.2f} ms")
    step += t
print(f"per step: {step / 1e3:.2f} ms;  " + "  ".join(f"{k} {v / 1e3:.2f}" for k, v in tot.items()))
