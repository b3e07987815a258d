"""Cache-policy / rasterisation sweep of the NT GEMM epilogues.

Builds csrc/gemm_nt.hip several times with different -DPVRL_NT_ST_AUX / -DPVRL_NT_LD_AUX / -DPVRL_NT_GM (aux bits of the
raw buffer instructions on gfx950: 1 = sc0, 2 = nt, 16 = sc1) into tools/probe/cpol/libnt_<tag>.so and times every variant
on the encoder's 50k-row shapes, rotating over operand sets larger than the 256 MiB Infinity Cache.

    python tools/probe/nt_cache_policy.py build            (here: hipcc cross-compiles)
    python tools/probe/nt_cache_policy.py run [tag ...]    (GPU box) -> table on stdout
    NT_ONE=<tag> rocprofv3 --pmc FETCH_SIZE ... python tools/probe/nt_cache_policy.py run <tag>   (one variant per PMC pass)
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "cpol")
#            tag            ST  LD  GM
VARIANTS = [("pers0", ["-DPVRL_NT_PERSIST_BUILD=1", "-DPVRL_NT_PERSIST_DEFAULT=0"]),
            ("pers_all", ["-DPVRL_NT_PERSIST_BUILD=1", "-DPVRL_NT_PERSIST_DEFAULT=0x7f"])]
#  (tag, extra compiler switches); the round-3 policy / rasterisation / tails lists: profiles/r3_nt_cache_policy.txt, r3_nt_tail_subtiles.txt


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "procedurevrl_amd", "csrc", "gemm_nt.hip")
    procs = []
    for tag, flags in VARIANTS:
        so = os.path.join(OUT, f"libnt_{tag}.so")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-result"] + flags + [src, "-o", so]
        procs.append((tag, subprocess.Popen(cmd)))
        if len(procs) % 4 == 0:
            for t, p in procs[-4:]:
                assert p.wait() == 0, t
    for t, p in procs:
        assert p.wait() == 0, t
    print("built", len(VARIANTS), "variants in", OUT)


def run(tags):
    import torch
    from procedurevrl_amd._lib import parse_header, _CTYPES, _RET, header_constants
    proto = parse_header()["pvrl_gemm_nt_bf16"]
    K = header_constants()
    dev = "cuda:0"
    BF = torch.bfloat16
    B, N, T = 32, 196, 8
    R = B * N * T
    M = R + B
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    NSET = 3
    shapes = [("fc1  gelu  Mx3072x768", M, 3072, 768, K["PVRL_EPI_GELU"]), ("dfc2 dgelu Mx3072x768", M, 3072, 768, K["PVRL_EPI_DGELU"]),
              ("qkv  bf16  Mx2304x768", M, 2304, 768, K["PVRL_EPI_BF16"]), ("dfc1 bf16  Mx768x3072", M, 768, 3072, K["PVRL_EPI_BF16"]),
              ("proj resid Rx768x768", R, 768, 768, K["PVRL_EPI_RESID_F32"]), ("fc2  resid Mx768x3072", M, 768, 3072, K["PVRL_EPI_RESID_F32"])]
    sets = {}
    for name, M_, N_, K_, epi in shapes:
        ss = []
        for _ in range(NSET):
            d = dict(A=rnd(M_, K_).to(BF), W=(rnd(N_, K_) * 0.02).to(BF), bias=rnd(N_))
            f32 = epi in (K["PVRL_EPI_RESID_F32"],)
            d["out0"] = torch.empty(M_, N_, device=dev, dtype=torch.float32 if f32 else BF)
            d["out1"] = torch.empty(M_, N_, device=dev, dtype=BF) if epi == K["PVRL_EPI_GELU"] else None
            d["aux"] = rnd(M_, N_) if f32 else (rnd(M_, N_).to(BF) if epi == K["PVRL_EPI_DGELU"] else None)
            ss.append(d)
        sets[name] = ss
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    import random
    want = [v for v in VARIANTS if not tags or v[0] in tags]
    fns = {}
    for tag, *_ in want:
        so = ctypes.CDLL(os.path.join(OUT, f"libnt_{tag}.so"))
        fn = so.pvrl_gemm_nt_bf16
        fn.restype = _RET[proto[0]]
        fn.argtypes = [_CTYPES[t] for t, _ in proto[1]]
        fns[tag] = fn
    samples = {}
    rng = random.Random(0)
    ROUNDS = 5 if len(want) > 1 else 1
    for name, M_, N_, K_, epi in shapes:
        ss = sets[name]

        def call(fn, d):
            rc = fn(vp(d["A"]), K_, vp(d["W"]), K_, M_, N_, K_, epi, vp(d["bias"]), None, vp(d["aux"]), N_ if d["aux"] is not None else 0, 0,
                    vp(d["out0"]), N_, vp(d["out1"]), N_ if d["out1"] is not None else 0, None, stream)
            assert rc == 0, rc
        ref = None
        for tag, *_ in want:                      # every variant computes the same bits
            call(fns[tag], ss[0])
            torch.cuda.synchronize()
            cur = (ss[0]["out0"].clone(), ss[0]["out1"].clone() if ss[0]["out1"] is not None else None)
            if ref is None:
                ref = cur
            else:
                assert torch.equal(ref[0], cur[0]) and (ref[1] is None or torch.equal(ref[1], cur[1])), (tag, name)
        if os.environ.get("NT_STRESS"):           # races show as run-to-run differences: 12 more launches per variant, every bit compared
            for tag, *_ in want:
                for r in range(12):
                    d = ss[r % NSET]
                    d["out0"].fill_(0)
                    call(fns[tag], d)
                    call(fns[want[0][0]], dict(d, out0=d.setdefault("chk0", torch.empty_like(d["out0"])),
                                               out1=(d.setdefault("chk1", torch.empty_like(d["out1"])) if d["out1"] is not None else None)))
                    torch.cuda.synchronize()
                    assert torch.equal(d["out0"], d["chk0"]), (tag, name, r)
                    assert d["out1"] is None or torch.equal(d["out1"], d["chk1"]), (tag, name, r)
        for rnd_i in range(ROUNDS):               # variants interleaved in a fresh random order every round: no first-runner bias
            order = [t for t, *_ in want]
            rng.shuffle(order)
            for tag in order:
                fn = fns[tag]
                call(fn, ss[0])
                reps = 9
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for r in range(reps):
                    call(fn, ss[(r + 1) % NSET])
                e1.record()
                torch.cuda.synchronize()
                samples.setdefault((tag, name), []).append(e0.elapsed_time(e1) / reps * 1e3)
    results = {k: sorted(v)[len(v) // 2] for k, v in samples.items()}
    names = [s[0] for s in shapes]
    print(f"{'variant':16s} " + " ".join(f"{n.split()[0] + ' ' + n.split()[1]:>12s}" for n in names) + "      sum")
    for tag, *_ in want:
        row = [results[(tag, n)] for n in names]
        print(f"{tag:16s} " + " ".join(f"{u:12.1f}" for u in row) + f" {sum(row):8.1f}")


def msweep(tag):
    """time vs M at fixed (N, K): how much of a launch is the ragged last round of 256x256 tiles on 256 CUs"""
    import torch
    from procedurevrl_amd._lib import parse_header, _CTYPES, _RET, header_constants
    proto = parse_header()["pvrl_gemm_nt_bf16"]
    Kc = header_constants()
    so = ctypes.CDLL(os.path.join(OUT, f"libnt_{tag}.so"))
    fn = so.pvrl_gemm_nt_bf16
    fn.restype = _RET[proto[0]]
    fn.argtypes = [_CTYPES[t] for t, _ in proto[1]]
    dev = "cuda:0"
    BF = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    for N_, K_ in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
        tn = N_ // 256
        print(f"N={N_} K={K_} (tiles_n={tn}); M: us  [tiles, rounds of 256, us per full round-equivalent]")
        for M_ in (256 * (512 // tn), 256 * (512 // tn) + 2048, 43520, 46080, 50208, 256 * (768 // tn), 65536):
            sets = [dict(A=torch.randn(M_, K_, device=dev, generator=g).to(BF), W=(torch.randn(N_, K_, device=dev, generator=g) * 0.02).to(BF),
                         out=torch.empty(M_, N_, device=dev, dtype=BF)) for _ in range(3)]

            def call(d):
                rc = fn(vp(d["A"]), K_, vp(d["W"]), K_, M_, N_, K_, Kc["PVRL_EPI_BF16"], None, None, None, 0, 0, vp(d["out"]), N_, None, 0, None, stream)
                assert rc == 0
            for d in sets:
                call(d)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(12):
                call(sets[r % 3])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 12 * 1e3
            tiles = -(-M_ // 256) * tn
            print(f"   M={M_:6d}: {us:7.1f} us  [{tiles} tiles, {tiles / 256:.2f} rounds, {2.0 * M_ * N_ * K_ / us / 1e6:6.0f} TFLOP/s]")
            del sets


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "msweep":
        msweep(sys.argv[2] if len(sys.argv) > 2 else "st0_ld0_gm2")
    else:
        run(sys.argv[2:])
