"""Spatial attention backward, fused (one kernel) vs two-pass, at the benchmark geometry (32 clips x 8 frames x 12 heads, S = 197).

  python tools/probe/attn_bwd_ab.py check        parity of both forms against fp32 math (tests/kernel_checks.py cases)
  python tools/probe/attn_bwd_ab.py time         event-timed forward / backward, each arm in its own process (the switch is read once)
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timeit(fn, reps=30, warm=5):
    import torch
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


def arm():
    import torch
    from procedurevrl_amd import ops
    from procedurevrl_amd.ops import OP16
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    for (B, T, N, H) in [(32, 8, 196, 12)]:
        HD = H * 64; S = N + 1; R = B * N * T; nseq = B * T
        sets = []
        for _ in range(3):      # cycle operand sets so L2 / MALL do not serve the re-reads
            qkv = torch.randn(R + B, 3 * HD, device=dev, generator=g).to(OP16)
            obuf = torch.zeros(R + nseq, HD, device=dev, dtype=OP16)
            do = torch.randn(R + nseq, HD, device=dev, generator=g).to(OP16)
            dbuf = torch.zeros(R + B + nseq, 3 * HD, device=dev, dtype=OP16)
            _, _, lse = ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=1, T=T, cls_base=R, o=obuf[:R], o_cls=obuf[R:])
            sets.append((qkv, obuf, do, dbuf, lse))
        k = [0]

        def fwd():
            qkv, obuf, do, dbuf, lse = sets[k[0] % 3]; k[0] += 1
            ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=1, T=T, cls_base=R, o=obuf[:R], o_cls=obuf[R:], lse=lse)

        def bwd():
            qkv, obuf, do, dbuf, lse = sets[k[0] % 3]; k[0] += 1
            ops.attn_bwd(qkv, obuf[:R], obuf[R:], do[:R], do[R:], lse, nseq, S, H, 0.125, mode=1, T=T, cls_base=R,
                         dqkv=dbuf[:R + B], dqkv_cls=dbuf[R + B:])
        tf = [timeit(fwd) for _ in range(3)]
        tb = [timeit(bwd) for _ in range(3)]
        print(f"  B={B} T={T} S={S} H={H}: fwd {min(tf):7.1f} us   bwd {min(tb):7.1f} us   (runs: fwd {tf}, bwd {tb})", flush=True)


def arm32():
    """temporal attention at 32 frames (BASELINE configs[3]): 8 clips x 196 patches = 1,568 contiguous sequences of 32 tokens, 12 heads"""
    import torch
    from procedurevrl_amd import ops
    from procedurevrl_amd.ops import OP16
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    nseq, S, H = 8 * 196, 32, 12
    HD = H * 64
    sets = []
    for _ in range(3):
        qkv = torch.randn(nseq * S, 3 * HD, device=dev, generator=g).to(OP16)
        do = torch.randn(nseq * S, HD, device=dev, generator=g).to(OP16)
        o, _, lse = ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=0)
        sets.append((qkv, o, do, torch.empty_like(qkv), lse))
    k = [0]

    def fwd():
        qkv, o, do, dq, lse = sets[k[0] % 3]; k[0] += 1
        ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=0, o=o, lse=lse)

    def bwd():
        qkv, o, do, dq, lse = sets[k[0] % 3]; k[0] += 1
        ops.attn_bwd(qkv, o, None, do, None, lse, nseq, S, H, 0.125, mode=0, dqkv=dq)
    tf = [timeit(fwd) for _ in range(3)]
    tb = [timeit(bwd) for _ in range(3)]
    print(f"  nseq={nseq} S={S} H={H}: fwd {min(tf):7.1f} us   bwd {min(tb):7.1f} us", flush=True)


def race(n=30):
    """the fused / one-wave backward kernels launched n times next to an uneven load: a race on the streamed LDS images (or a counted
    wait that is one short) shows as a run-to-run difference; every result is compared bit for bit with the first"""
    import torch
    from procedurevrl_amd import ops
    from procedurevrl_amd.ops import OP16
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(3)
    bad = 0
    burn = torch.randn(8192, 8192, device=dev)
    side = torch.cuda.Stream()
    for (B, T, N, H, mode) in [(32, 8, 196, 12, 1), (5, 4, 120, 3, 1), (1568, 1, 31, 12, 0)]:
        HD = H * 64
        if mode == 1:
            S = N + 1; R = B * N * T; nseq = B * T
            qkv = torch.randn(R + B, 3 * HD, device=dev, generator=g).to(OP16)
            obuf = torch.zeros(R + nseq, HD, device=dev, dtype=OP16)
            do = torch.randn(R + nseq, HD, device=dev, generator=g).to(OP16)
            _, _, lse = ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=1, T=T, cls_base=R, o=obuf[:R], o_cls=obuf[R:])
            run = lambda: ops.attn_bwd(qkv, obuf[:R], obuf[R:], do[:R], do[R:], lse, nseq, S, H, 0.125, mode=1, T=T, cls_base=R)
        else:
            nseq, S = B, N + 1
            qkv = torch.randn(nseq * S, 3 * HD, device=dev, generator=g).to(OP16)
            do = torch.randn(nseq * S, HD, device=dev, generator=g).to(OP16)
            o, _, lse = ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=0)
            run = lambda: ops.attn_bwd(qkv, o, None, do, None, lse, nseq, S, H, 0.125, mode=0)
        ref = [x.clone() if x is not None else None for x in run()]
        for it in range(n):
            if it % 3 != 0:
                with torch.cuda.stream(side):
                    burn @ burn                     # uneven load on another stream next to the launch
            out = run()
            for a, b in zip(ref, out):
                if a is not None and not torch.equal(a.view(torch.int16), b.view(torch.int16)):
                    bad += 1
        torch.cuda.synchronize()
        print(f"race B={B} T={T} S={N + 1} H={H} mode {mode}: {n} launches, mismatching so far {bad}", flush=True)
    print("RACE", "FAILED" if bad else "OK", flush=True)


def trace():
    """PVRL_LIB_PATH = a -DPVRL_FB_TRACE=1 build: cycle stamps of every wave of workgroup 8 (dumped through the dvec argument)."""
    import torch
    from procedurevrl_amd import ops
    from procedurevrl_amd.ops import OP16, _ptr, _ld, _stream
    from procedurevrl_amd._lib import lib
    L = lib()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    B, T, N, H = 32, 8, 196, 12
    HD = H * 64; S = N + 1; R = B * N * T; nseq = B * T
    qkv = torch.randn(R + B, 3 * HD, device=dev, generator=g).to(OP16)
    obuf = torch.zeros(R + nseq, HD, device=dev, dtype=OP16)
    do = torch.randn(R + nseq, HD, device=dev, generator=g).to(OP16)
    dbuf = torch.zeros(R + B + nseq, 3 * HD, device=dev, dtype=OP16)
    _, _, lse = ops.attn_fwd(qkv, nseq, S, H, 0.125, mode=1, T=T, cls_base=R, o=obuf[:R], o_cls=obuf[R:])
    dvec = torch.zeros_like(lse)
    for _ in range(3):
        L.call("pvrl_attn_bwd", _ptr(qkv), _ld(qkv), nseq, S, H, 1, T, R, 0.125, 0, None, _ptr(obuf[:R]), _ptr(obuf[R:]),
               _ptr(do[:R]), _ptr(do[R:]), _ld(obuf), _ptr(lse), _ptr(dvec), _ptr(dbuf[:R + B]), _ptr(dbuf[R + B:]), _ld(dbuf), _stream())
    torch.cuda.synchronize()
    st = dvec.view(-1)[:8 * 64].view(torch.int32).view(8, 8, 8).cpu().to(torch.int64) & 0xffffffff
    t0 = int(st[:, 7, 0].min())
    print("second item of workgroup 8; cycles since the first wave left the item seam; per block: key waves [top | S,dP issued | P,dS done | dK,dV + dS stores issued | barrier passed], "
          "dQ wave [at barrier | passed | start values of a streamed slot written | block done]")
    for w in range(8):
        k = st[w, 7] - t0
        print(f"wave {w} ({'dQ wave' if w == 7 else 'key wave'}): seam left {int(k[0])}  item done {int(k[4])}")
        for jb in range(7):
            r = st[w, jb] - t0
            n = 4 if w == 7 else 5
            print(f"    jb {jb}: " + " ".join(f"{int(r[i]):7d}" for i in range(n)) + "   d: " + " ".join(f"{int(r[i + 1] - r[i]):5d}" for i in range(n - 1)))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    if mode == "arm":
        return arm()
    if mode == "arm32":
        return arm32()
    if mode == "race":
        return race()
    if mode == "time32":
        for tag, val in (("two-pass", "0"), ("one wave per item", "1"), ("two-pass", "0"), ("one wave per item", "1")):
            env = dict(os.environ, PVRL_ATTN_BWD_S32=val)
            print(f"--- {tag} (PVRL_ATTN_BWD_S32={val})", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "arm32"], env=env, check=False)
        return
    if mode == "trace":
        return trace()
    if mode == "check":
        import kernel_checks as kc
        bad = 0
        for fn in (kc.check_attn_mfma_contig, kc.check_attn_mfma_spatial):
            for name, err, tol in fn():
                flag = "ok " if err <= tol else "BAD"
                bad += err > tol
                print(f"  {flag} {name:50s} err {err:.3e} tol {tol:.1e}")
        print("CHECK", "OK" if not bad else f"FAILED ({bad})")
        return
    for tag, val in (("two-pass", "0"), ("fused", "1"), ("two-pass", "0"), ("fused", "1")):
        env = dict(os.environ, PVRL_ATTN_BWD_FUSED=val)
        print(f"--- {tag} (PVRL_ATTN_BWD_FUSED={val})", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=env, check=False)


if __name__ == "__main__":
    main()
