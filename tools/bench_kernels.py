"""Micro-benchmark of individual kernels at the BASELINE config-2 shapes (32 clips): event-timed, L2-flushed between
reps by cycling through several operand sets.  Usage: python tools/bench_kernels.py [filter ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe"))
import probe_lib as pl  # noqa: E402   (measured-and-rejected variants: tools/probe/libpvrl_probe.so)

DEV = "cuda:0"
BF = torch.bfloat16


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    L = lib()
    flt = sys.argv[1:]
    want = lambda n: not flt or any(f in n for f in flt)
    B, N, T, C = 32, 196, 8, 768
    R = B * N * T
    M = R + B
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    rows = []

    sel = {"tile": 0, "gm": 0}     # 0 = the product library; otherwise the probe library's explicit selectors

    def gemm_case(name, M_, N_, K_, epi, **kw):
        if not want(name):
            return
        A = rnd(M_, K_).to(BF); W = (rnd(N_, K_) * 0.02).to(BF)
        bias = rnd(N_)
        args = dict(bias=bias)
        if epi == L.PVRL_EPI_RESID_F32:
            args["aux"] = rnd(M_, N_)
        if epi in (L.PVRL_EPI_DGELU,):
            args["aux"] = rnd(M_, N_).to(BF); args.pop("bias")
        if sel["tile"] or sel["gm"]:
            us = timeit(lambda: pl.gemm_nt(sel["tile"], A, W, epi, gm=sel["gm"], **args))
        else:
            us = timeit(lambda: ops.gemm_nt(A, W, epi, **args))
        rows.append((name, us, 2.0 * M_ * N_ * K_ / us / 1e6))

    for tile in (1, 2, 3):
        sel["tile"] = tile
        tg = {1: "128x128", 2: "256x128", 3: "256x256"}[tile]
        gemm_case(f"nt[{tg}] qkv   bf16  M x2304x768", M, 2304, 768, L.PVRL_EPI_BF16)
        gemm_case(f"nt[{tg}] proj  bf16  M x768x768", R, 768, 768, L.PVRL_EPI_BF16)
        gemm_case(f"nt[{tg}] fc    resid M x768x768", R, 768, 768, L.PVRL_EPI_RESID_F32)
        gemm_case(f"nt[{tg}] fc1   gelu  M x3072x768", M, 3072, 768, L.PVRL_EPI_GELU)
        gemm_case(f"nt[{tg}] fc2   resid M x768x3072", M, 768, 3072, L.PVRL_EPI_RESID_F32)
        gemm_case(f"nt[{tg}] dfc2  dgelu M x3072x768", M, 3072, 768, L.PVRL_EPI_DGELU)
        gemm_case(f"nt[{tg}] dfc1  bf16  M x768x3072", M, 768, 3072, L.PVRL_EPI_BF16)
        gemm_case(f"nt[{tg}] dqkv  bf16  M x768x2304", M, 768, 2304, L.PVRL_EPI_BF16)
    for rep in range(2):
        for tile, tg in ((3, "256 16 waves"), (13, "4 waves, register-staged, 32x32x16")):
            sel["tile"] = tile
            gemm_case(f"ab[{tg}] qkv   bf16  M x2304x768", M, 2304, 768, L.PVRL_EPI_BF16)
            gemm_case(f"ab[{tg}] fc1   gelu  M x3072x768", M, 3072, 768, L.PVRL_EPI_GELU)
            gemm_case(f"ab[{tg}] proj  bf16  M x768x768", R, 768, 768, L.PVRL_EPI_BF16)
            gemm_case(f"ab[{tg}] fc    resid M x768x768", R, 768, 768, L.PVRL_EPI_RESID_F32)
            gemm_case(f"ab[{tg}] fc2   resid M x768x3072", M, 768, 3072, L.PVRL_EPI_RESID_F32)
            gemm_case(f"ab[{tg}] dfc1  bf16  M x768x3072", M, 768, 3072, L.PVRL_EPI_BF16)
            gemm_case(f"ab[{tg}] dqkv  bf16  M x768x2304", M, 768, 2304, L.PVRL_EPI_BF16)
            gemm_case(f"ab[{tg}] dfc2  dgelu M x3072x768", M, 3072, 768, L.PVRL_EPI_DGELU)
    sel["tile"] = 0
    for rep in range(2):
        for gm in (1, 2, 3, 4):
            sel["gm"] = gm
            gemm_case(f"gm[{gm}] qkv   bf16  M x2304x768", M, 2304, 768, L.PVRL_EPI_BF16)
            gemm_case(f"gm[{gm}] dfc1  bf16  M x768x3072", M, 768, 3072, L.PVRL_EPI_BF16)
            gemm_case(f"gm[{gm}] fc1   gelu  M x3072x768", M, 3072, 768, L.PVRL_EPI_GELU)
            gemm_case(f"gm[{gm}] proj  bf16  M x768x768", R, 768, 768, L.PVRL_EPI_BF16)
    sel["gm"] = 0
    gemm_case("nt-auto qkv      bf16  M x2304x768", M, 2304, 768, L.PVRL_EPI_BF16)
    gemm_case("nt proj     bf16  M x768x768", R, 768, 768, L.PVRL_EPI_BF16)
    gemm_case("nt fc/projs resid M x768x768", R, 768, 768, L.PVRL_EPI_RESID_F32)
    gemm_case("nt fc1      gelu  M x3072x768", M, 3072, 768, L.PVRL_EPI_GELU)
    gemm_case("nt fc2      resid M x768x3072", M, 768, 3072, L.PVRL_EPI_RESID_F32)
    gemm_case("nt dfc2     dgelu M x3072x768", M, 3072, 768, L.PVRL_EPI_DGELU)
    gemm_case("nt dfc1     bf16  M x768x3072", M, 768, 3072, L.PVRL_EPI_BF16)
    gemm_case("nt dqkv     bf16  M x768x2304", M, 768, 2304, L.PVRL_EPI_BF16)

    tn_sel = {"tile": 0}

    def tn_case(name, M_, N_, K_, splits=None):
        if not want(name):
            return
        P = rnd(M_, N_).to(BF); Q = rnd(M_, K_).to(BF)
        dW = torch.zeros(N_, K_, device=DEV); db = torch.zeros(N_, device=DEV)
        if tn_sel["tile"]:
            if splits is None:
                splits = pl.tn_splits(tn_sel["tile"], M_, N_, K_)
            us = timeit(lambda: pl.gemm_tn(tn_sel["tile"], P, Q, dW, db, splits=splits))
        else:
            if splits is None:
                splits = ops.tn_splits(M_, N_, K_)
            us = timeit(lambda: ops.gemm_tn(P, Q, dW, db, splits=splits))
        name = name + f" s={splits}"
        rows.append((name, us, 2.0 * M_ * N_ * K_ / us / 1e6))

    for tile in (7, 8, 7, 8):
        tn_sel["tile"] = tile
        tg = {1: "128x128 tr-read", 0: "default", 6: "rt 16x16x32", 7: "rt 32x32x16", 8: "rt 8 waves"}[tile]
        tn_case(f"tn[{tg}] wqkv  2304x768", M, 2304, 768)
        tn_case(f"tn[{tg}] wproj 768x768", R, 768, 768)
        tn_case(f"tn[{tg}] wfc1  3072x768", M, 3072, 768)
        tn_case(f"tn[{tg}] wfc2  768x3072", M, 768, 3072)
    tn_sel["tile"] = 0

    if want("f32"):
        a = rnd(32, 512); lab = rnd(9871, 512); labt = lab.t().contiguous(); dyl = rnd(32, 9871)
        us = timeit(lambda: ops.gemm_nt_f32(a, lab, alpha=50.0)); rows.append(("f32 logits fwd 32x9871x512", us, 2.0 * 32 * 9871 * 512 / us / 1e6))
        us = timeit(lambda: ops.gemm_nt_f32(dyl, labt, alpha=50.0)); rows.append(("f32 logits bwd 32x512x9871", us, 2.0 * 32 * 9871 * 512 / us / 1e6))
        f = rnd(32, 768); wh = rnd(512, 768)
        us = timeit(lambda: ops.gemm_nt_f32(f, wh)); rows.append(("f32 head 32x512x768", us, 2.0 * 32 * 512 * 768 / us / 1e6))
    if want("attn"):
        H = 12
        qkv = rnd(M, 3 * C).to(BF)
        o = torch.empty(R + B * T, C, device=DEV, dtype=BF)
        lse = torch.empty(B * T, H, N + 1, device=DEV)
        f = lambda: ops.attn_fwd(qkv, B * T, N + 1, H, 0.125, mode=1, T=T, cls_base=R, o=o[:R], o_cls=o[R:], lse=lse)
        us = timeit(f)
        fl = 4.0 * B * T * H * (N + 1) ** 2 * 64
        rows.append(("attn spatial fwd", us, fl / us / 1e6))
        do = rnd(R + B * T, C).to(BF)
        dq = torch.empty(M + B * T, 3 * C, device=DEV, dtype=BF)
        fb = lambda: ops.attn_bwd(qkv, o[:R], o[R:], do[:R], do[R:], lse, B * T, N + 1, H, 0.125, mode=1, T=T, cls_base=R,
                                  dqkv=dq[:M], dqkv_cls=dq[M:])
        us = timeit(fb)
        rows.append(("attn spatial bwd (q + kv kernels)", us, 2.5 * fl / us / 1e6))
        qt = rnd(R, 3 * C).to(BF)
        us = timeit(lambda: ops.attn_t8_fwd(qt, B * N, H, 0.125))
        rows.append(("attn t8 fwd [GB/s]", us, (R * 4 * C * 2) / us / 1e3))
        dot = rnd(R, C).to(BF)
        us = timeit(lambda: ops.attn_t8_bwd(qt, dot, B * N, H, 0.125))
        rows.append(("attn t8 bwd [GB/s]", us, (R * 8 * C * 2) / us / 1e3))

    if want("ln"):
        x = rnd(M, C); gam = rnd(C); bet = rnd(C)
        us = timeit(lambda: ops.layernorm_fwd(x, gam, bet, 1e-6))
        rows.append(("ln fwd [GB/s]", us, M * C * 6 / us / 1e3))
        y, mean, rstd = ops.layernorm_fwd(x, gam, bet, 1e-6)
        dy = rnd(M, C).to(BF); dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV); dx = rnd(M, C)
        us = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, gam, dg, db, dx_in=dx, dx_out=dx))
        rows.append(("ln bwd (+reduce) [GB/s]", us, M * C * 14 / us / 1e3))
        us = timeit(lambda: ops.cast_scale(x, None))
        rows.append(("cast_scale [GB/s]", us, M * C * 6 / us / 1e3))

    for name, us, rate in rows:
        print(f"{name:42s} {us:9.1f} us   {rate:9.1f} {'TFLOP/s' if 'GB/s' not in name else 'GB/s'}")


if __name__ == "__main__":
    main()
