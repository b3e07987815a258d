"""Round 6 probe: the kernels the split residual stream touches, fp32 stream against split stream, at the 32-clip shapes
(M = 50,176 patch rows + 32 cls rows, C = 768).  Event-timed over rotating operand sets (nothing stays L2-resident).
usage: [PVRL_LN_BWD_BLOCKS=n] python tools/probe/resid16_bench.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
OP = ops.OP16


def timeit(fns, reps=24, warm=4):
    n = len(fns)
    for i in range(warm):
        fns[i % n]()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fns[i % n]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    L = lib()
    R, B, C = 50176, 32, 768
    M = R + B
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    NS = 4
    gam, bet = rnd(C), rnd(C)
    sets = []
    for _ in range(NS):
        x = rnd(M, C)
        sets.append(dict(x=x, xs=ops.SplitRows(x[:R].to(OP), x[R:].contiguous()), dy=rnd(M, C).to(OP), dxi=rnd(M, C),
                         dxo=torch.empty(M, C, device=DEV), dxs=torch.empty(R, C, device=DEV, dtype=OP), sc=torch.rand(M, device=DEV)))
        s = sets[-1]
        s["dxis"] = ops.SplitRows(s["dxi"][:R].to(OP), s["dxi"][R:].contiguous())
        s["dxos"] = ops.SplitRows(torch.empty(R, C, device=DEV, dtype=OP), torch.empty(B, C, device=DEV))
        _, s["mean"], s["rstd"] = ops.layernorm_fwd(x, gam, bet, 1e-6)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    print(f"PVRL_LN_BWD_BLOCKS={os.environ.get('PVRL_LN_BWD_BLOCKS', '(512)')}")
    t = timeit([lambda s=s: ops.layernorm_fwd(s["x"], gam, bet, 1e-6) for s in sets])
    print(f"ln_fwd fp32 rows  {t:7.1f} us  {(M * C * 6) / t / 1e6:5.2f} TB/s")
    t = timeit([lambda s=s: ops.layernorm_fwd(s["xs"], gam, bet, 1e-6) for s in sets])
    print(f"ln_fwd split rows {t:7.1f} us  {(M * C * 4) / t / 1e6:5.2f} TB/s")
    t = timeit([lambda s=s: ops.layernorm_bwd(s["dy"], s["x"], s["mean"], s["rstd"], gam, dg, db, dx_in=s["dxi"], dx_out=s["dxo"], dxs=s["dxs"],
                                              dxs_scale=s["sc"]) for s in sets])
    print(f"ln_bwd fp32 rows  {t:7.1f} us  {(M * C * 16) / t / 1e6:5.2f} TB/s")
    t = timeit([lambda s=s: ops.layernorm_bwd(s["dy"], s["xs"], s["mean"], s["rstd"], gam, dg, db, dx_in=s["dxis"], dx_out=s["dxos"], dxs=s["dxs"],
                                              dxs_scale=s["sc"]) for s in sets])
    print(f"ln_bwd split rows {t:7.1f} us  {(M * C * 10) / t / 1e6:5.2f} TB/s")
    if os.environ.get("PVRL_LN_BWD_BLOCKS"):
        return
    for (K, b2) in ((768, True), (768, False), (3072, False)):
        gs = []
        for _ in range(3):
            gs.append(dict(A=rnd(R, K).to(OP), W=(rnd(C, K) * 0.02).to(OP), bias=rnd(C), b2=rnd(C) if b2 else None, rs=torch.rand(R, device=DEV),
                           aux=rnd(R, C), o32=torch.empty(R, C, device=DEV), o16=torch.empty(R, C, device=DEV, dtype=OP)))
            gs[-1]["aux16"] = gs[-1]["aux"].to(OP)
        t3 = timeit([lambda s=s: ops.gemm_nt(s["A"], s["W"], L.PVRL_EPI_RESID_F32, bias=s["bias"], rowscale=s["rs"], bias2=s["b2"], aux=s["aux"],
                                             out0=s["o32"]) for s in gs])
        t7 = timeit([lambda s=s: ops.gemm_nt(s["A"], s["W"], L.PVRL_EPI_RESID_16, bias=s["bias"], rowscale=s["rs"], bias2=s["b2"], aux=s["aux16"],
                                             out0=s["o16"]) for s in gs])
        t0 = timeit([lambda s=s: ops.gemm_nt(s["A"], s["W"], L.PVRL_EPI_BF16, bias=s["bias"], rowscale=s["rs"], out0=s["o16"]) for s in gs])
        fl = 2.0 * R * C * K
        print(f"gemm 50176x768x{K} bias2={b2}: resid_f32 {t3:6.1f} us ({fl / t3 / 1e6:5.0f} TF)  resid_16 {t7:6.1f} us ({fl / t7 / 1e6:5.0f} TF)  "
              f"plain 16-bit out {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF)")


if __name__ == "__main__":
    main()
