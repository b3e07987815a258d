"""A/B of the persistent 8-wave NT GEMM (csrc/gemm_nt8_core.h) against the 16-wave one-tile kernel it replaces, in ONE process.

The product library reads PVRL_NT8 / PVRL_NT_TAILS once per loaded image, so the script loads COPIES of libpvrl_hip.so under different
environments:  old = PVRL_NT8=0,  new = PVRL_NT8=1,  new_nt = PVRL_NT8=1 PVRL_NT_TAILS=0 (no half items in the last round).
  check   every epilogue x a set of shapes (ragged last panel, sub-round, K = 128 ...): outputs must be BIT-IDENTICAL (same MFMA order)
  time    the step's 50k-row shapes and full-round shapes, variants interleaved, median of rounds (us and TFLOP/s)
usage: python tools/probe/nt8_ab.py [check] [time] [race N]
"""
import ctypes
import os
import shutil
import statistics
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from procedurevrl_amd import _lib  # noqa: E402

DEV = "cuda:0"
OP16 = _lib.operand_torch_dtype()
C = _lib.header_constants()
EPI = {k[len("PVRL_EPI_"):]: v for k, v in C.items() if k.startswith("PVRL_EPI_")}
_tmp = tempfile.mkdtemp(prefix="nt8ab_")


def load(tag, env, src=None):
    """a private copy of the library (or of the variant build `src`), its once-read switches fixed to `env`"""
    path = os.path.join(_tmp, f"libpvrl_{tag}.so")
    shutil.copy(src or _lib.LIB_PATH, path)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    dll = ctypes.CDLL(path)
    ret, args = _lib.parse_header()["pvrl_gemm_nt_bf16"]
    dll.pvrl_gemm_nt_bf16.restype = ctypes.c_int
    dll.pvrl_gemm_nt_bf16.argtypes = [_lib._CTYPES[t] for t, _ in args]
    # first call fixes the switches of this image
    a = torch.zeros(4096, 128, device=DEV, dtype=OP16)
    w = torch.zeros(256, 128, device=DEV, dtype=OP16)
    run(dll, a, w, EPI["BF16"])
    torch.cuda.synchronize()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return dll


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def run(dll, A, W, epi, bias=None, rowscale=None, aux=None, aux_rowmod=0, bias2=None, outs=None):
    M, K = A.shape
    N = W.shape[0]
    f32 = epi in (EPI["RESID_F32"], EPI["F32"])
    two = epi in (EPI["GELU"], EPI["QGELU"])
    if outs is None:
        out0 = torch.empty((M, N), device=A.device, dtype=torch.float32 if f32 else OP16)
        out1 = torch.empty((M, N), device=A.device, dtype=OP16) if two else None
    else:
        out0, out1 = outs
    rc = dll.pvrl_gemm_nt_bf16(_p(A), A.stride(0), _p(W), W.stride(0), M, N, K, epi, _p(bias), _p(rowscale), _p(aux),
                               aux.stride(0) if aux is not None else 0, aux_rowmod, _p(out0), out0.stride(0), _p(out1),
                               out1.stride(0) if out1 is not None else 0, _p(bias2),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"pvrl_gemm_nt_bf16 -> {rc}")
    return out0, out1


def operands(M, N, K, epi, g, lda_pad=0):
    A = torch.randn(M, K + lda_pad, device=DEV, generator=g).to(OP16)[:, :K]
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).to(OP16)
    kw = dict(bias=torch.randn(N, device=DEV, generator=g))
    if epi == EPI["RESID_F32"]:
        kw["aux"] = torch.randn(M, N, device=DEV, generator=g)
    if epi in (EPI["DGELU"], EPI["DQGELU"]):
        kw["aux"] = torch.randn(M, N, device=DEV, generator=g).to(OP16)
    return A, W, kw


def check(libs):
    g = torch.Generator(device=DEV).manual_seed(1)
    bad = 0
    shapes = [(50208, 768, 768), (50208, 2304, 768), (4096 + 17, 256, 128), (8192, 512, 192), (12000, 768, 3072), (6273 * 8, 768, 768),
              (65536, 768, 256), (300 * 256 + 1, 256, 128), (50208, 3072, 768)]
    for (M, N, K) in shapes:
        for name, epi in EPI.items():
            variants = [dict()]
            if name == "RESID_F32":
                variants = [dict(), dict(rs=True), dict(rs=True, b2=True), dict(b2=True), dict(tab=True)]
            elif name in ("BF16", "DGELU"):
                variants = [dict(), dict(rs=True)]
            for v in variants:
                A, W, kw = operands(M, N, K, epi, g, lda_pad=64 if K == 192 else 0)
                if v.get("rs"):
                    kw["rowscale"] = (torch.rand(M, device=DEV, generator=g) > 0.1).float() / 0.9
                if v.get("b2"):
                    kw["bias2"] = torch.randn(N, device=DEV, generator=g)
                if v.get("tab"):
                    kw["aux"] = torch.randn(197, N, device=DEV, generator=g)
                    kw["aux_rowmod"] = 197
                ref = run(libs["old"], A, W, epi, **kw)
                for tag in [t for t in libs if t != "old"]:
                    out = run(libs[tag], A, W, epi, **kw)
                    for a, b in zip(ref, out):
                        if a is None:
                            continue
                        same = torch.equal(a.view(torch.int16 if a.dtype != torch.float32 else torch.int32),
                                           b.view(torch.int16 if b.dtype != torch.float32 else torch.int32))
                        if not same:
                            bad += 1
                            d = (a.float() - b.float()).abs()
                            rows = (d.amax(1) > 0).nonzero().flatten()
                            print(f"MISMATCH M {M} N {N} K {K} {name} {v} {tag}: max {d.max().item():.3e}, {rows.numel()} rows, first {rows[:6].tolist()} "
                                  f"last {rows[-3:].tolist()}, nan {torch.isnan(b.float()).sum().item()}", flush=True)
        print(f"checked M {M} N {N} K {K}: mismatches so far {bad}", flush=True)
    # one reference against fp32 math, so that "identical" is not "identically wrong"
    A, W, kw = operands(8192, 512, 256, EPI["F32"], g)
    out, _ = run(libs["new"], A, W, EPI["F32"], **kw)
    ref = A.float() @ W.float().t() + kw["bias"]
    err = ((out - ref).abs().max() / ref.abs().max()).item()
    print(f"new vs fp32 matmul: rel err {err:.2e}", flush=True)
    if err > 1e-5:
        bad += 1
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


def race(libs, n):
    """the same launch n times: a race on the LDS ring shows as a run-to-run difference"""
    g = torch.Generator(device=DEV).manual_seed(2)
    bad = 0
    for (M, N, K, name) in [(50208, 768, 768, "RESID_F32"), (50208, 3072, 768, "GELU"), (50208, 768, 3072, "BF16"), (65536, 2304, 768, "BF16")]:
        A, W, kw = operands(M, N, K, EPI[name], g)
        ref = run(libs["old"], A, W, EPI[name], **kw)
        burn = torch.randn(8192, 8192, device=DEV)
        for it in range(n):
            if it % 3 == 1:
                burn @ burn                   # uneven load next to the launch
            out = run(libs["new"], A, W, EPI[name], **kw)
            for a, b in zip(ref, out):
                if a is not None and not torch.equal(a.view(torch.int32 if a.dtype == torch.float32 else torch.int16),
                                                     b.view(torch.int32 if b.dtype == torch.float32 else torch.int16)):
                    bad += 1
        print(f"race M {M} N {N} K {K} {name}: {n} launches, mismatching so far {bad}", flush=True)
    print("RACE", "FAILED" if bad else "OK", flush=True)
    return bad


def timeit(fn, reps):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def time_all(libs, rounds=7, reps=10):
    g = torch.Generator(device=DEV).manual_seed(3)
    M = 50208
    cases = [("qkv bf16", M, 2304, 768, "BF16"), ("proj resid", M, 768, 768, "RESID_F32"), ("fc1 gelu", M, 3072, 768, "GELU"),
             ("fc2 resid", M, 768, 3072, "RESID_F32"), ("dqkv bf16", M, 768, 2304, "BF16"), ("dfc2 dgelu", M, 3072, 768, "DGELU"),
             ("dfc1 bf16", M, 768, 3072, "BF16"), ("dproj bf16", M, 768, 768, "BF16"),
             ("full 64k x768 x3072", 65536, 768, 3072, "BF16"), ("full 64k x3072 x768", 65536, 3072, 768, "BF16"),
             ("full 64k x768 x768", 65536, 768, 768, "BF16"), ("8192^3", 8192, 8192, 8192, "BF16")]
    tags = list(libs)
    for (label, M_, N, K, name) in cases:
        A, W, kw = operands(M_, N, K, EPI[name], g)
        outs = run(libs["old"], A, W, EPI[name], **kw)
        fns = {t: (lambda t=t: run(libs[t], A, W, EPI[name], outs=outs, **kw)) for t in tags}
        for t in tags:
            timeit(fns[t], 3)
        res = {t: [] for t in tags}
        for _ in range(rounds):
            for t in tags:
                res[t].append(timeit(fns[t], reps))
        fl = 2.0 * M_ * N * K / 1e6
        line = f"{label:22s} M {M_} N {N} K {K} {name:10s}"
        for t in tags:
            med = statistics.median(res[t])
            line += f" | {t} {med:7.1f} us {fl / med:6.0f} TF (min {min(res[t]):.1f})"
        base = statistics.median(res["old"])
        line += " | new/old %.3f" % (statistics.median(res["new"]) / base)
        print(line, flush=True)


def trace(shapes=((65536, 768, 3072), (65536, 768, 768), (8192, 8192, 8192))):
    """per-phase timeline of waves 0 and 4 of workgroup 8 (variant build -DPVRL_NT8_TRACE=1: tools/build_variant.py nt8trace ...)"""
    src = os.path.join(HERE, "..", "..", "procedurevrl_amd", "csrc", "variants", "libpvrl_hip_nt8trace.so")
    dll = load("trace", {"PVRL_NT8": "1"}, src)
    g = torch.Generator(device=DEV).manual_seed(4)
    KT = 10
    for (M, N, K) in shapes:
        A, W, kw = operands(M, N, K, EPI["RESID_F32"], g)
        tb = torch.zeros(16384, device=DEV, dtype=torch.float32)
        kw["bias2"] = tb
        for _ in range(3):
            run(dll, A, W, EPI["RESID_F32"], **kw)
        torch.cuda.synchronize()
        t = tb.view(torch.int64)[:2 * KT * 16].cpu().view(2, KT, 4, 4)        # [group][kt][phase][stamp]
        t0 = int(t[0, 0, 0, 0])
        print(f"--- trace M {M} N {N} K {K}: cycles since G0's first stamp; per phase: mem = work of the memory segment, wait1 = at its barrier, "
              f"cmp = 16 MFMAs issued, wait2 = at the compute barrier", flush=True)
        for gi in range(2):
            for kt in range(min(KT, K // 64)):
                row = []
                for ph in range(4):
                    s0, s1, s2, s3 = (int(x) for x in t[gi, kt, ph])
                    nxt = int(t[gi, kt, ph + 1, 0]) if ph < 3 else (int(t[gi, kt + 1, 0, 0]) if kt + 1 < min(KT, K // 64) else s3)
                    row.append(f"[{s0 - t0:6d} mem {s1 - s0:4d} wait1 {s2 - s1:4d} cmp {s3 - s2:4d} wait2 {nxt - s3:4d}]")
                print(f"G{gi} kt {kt}: " + " ".join(row), flush=True)
        per = (int(t[0, min(KT, K // 64) - 1, 0, 0]) - int(t[0, 1, 0, 0])) / (min(KT, K // 64) - 2)
        print(f"cycles per K-tile (G0, kt 1..): {per:.0f}  (16 MFMA clusters x 8 = 2048 at the issue rate)", flush=True)
        clk = tb.view(torch.int64)[2 * KT * 16:2 * KT * 16 + 4].cpu().tolist()
        for gi in range(2):
            cyc, ticks = clk[2 * gi], clk[2 * gi + 1]
            print(f"G{gi}: K loop of the traced tile = {cyc} shader cycles in {ticks} ticks of 100 MHz -> {cyc / max(ticks, 1) * 0.1:.3f} GHz, "
                  f"{cyc / (K // 64):.0f} cycles per K-tile incl. {min(KT, K // 64)} traced ones", flush=True)


def ablate(rounds=5, reps=10):
    """time of the main loop with one ingredient removed (variant builds -DPVRL_NT8_ABLATE=<bits>; outputs are garbage)"""
    vdir = os.path.join(HERE, "..", "..", "procedurevrl_amd", "csrc", "variants")
    names = {0: "full", 1: "no DMA", 2: "no frag reads", 4: "no MFMA", 8: "no barrier behind compute", 16: "no setprio", 3: "no DMA, no reads (MFMA + barriers)",
             6: "no reads, no MFMA (DMA + barriers)", 5: "no DMA, no MFMA (reads + barriers)", 7: "barriers only",
             38: "DMA + barriers, every K-tile the same 64 KB per tile", 102: "DMA + barriers, ONE 64 KB image for the chip",
             32: "full, every K-tile the same 64 KB per tile"}
    libs = {}
    for bits, nm in names.items():
        src = os.path.join(vdir, f"libpvrl_hip_nt8abl{bits}.so") if bits else None
        if src is None or os.path.exists(src):
            libs[nm] = load(f"abl{bits}", {"PVRL_NT8": "1"}, src)
    g = torch.Generator(device=DEV).manual_seed(5)
    for (M, N, K) in [(65536, 768, 3072), (8192, 8192, 8192), (65536, 768, 768)]:
        A, W, kw = operands(M, N, K, EPI["BF16"], g)
        outs = run(libs["full"], A, W, EPI["BF16"], **kw)
        res = {t: [] for t in libs}
        for t in libs:
            timeit(lambda: run(libs[t], A, W, EPI["BF16"], outs=outs, **kw), 3)
        for _ in range(rounds):
            for t in libs:
                res[t].append(timeit(lambda: run(libs[t], A, W, EPI["BF16"], outs=outs, **kw), reps))
        fl = 2.0 * M * N * K / 1e6
        print(f"--- ablation M {M} N {N} K {K}", flush=True)
        for t in libs:
            med = statistics.median(res[t])
            print(f"  {t:40s} {med:8.1f} us  ({fl / med:6.0f} TF-equivalent)", flush=True)


def gelu_ab(rounds=7, reps=10):
    """the GELU / dGELU GEMMs of the MLP with the one-transcendental GELU forms (product build) next to variant builds:
    gelu0 = -DPVRL_GELU_FORM=0 (Abramowitz-Stegun, both directions), dgelu0 = -DPVRL_DGELU_FORM=0 (derivative only)"""
    vdir = os.path.join(HERE, "..", "..", "procedurevrl_amd", "csrc", "variants")
    libs = {"new": load("g_new", {"PVRL_NT8": "1"})}
    for tag in ("gelu0", "dgelu0"):
        src = os.path.join(vdir, f"libpvrl_hip_{tag}.so")
        if os.path.exists(src):
            libs[tag] = load("g_" + tag, {"PVRL_NT8": "1"}, src)
    g = torch.Generator(device=DEV).manual_seed(3)
    M = 50208
    for (label, M_, N, K, name) in [("fc1 gelu", M, 3072, 768, "GELU"), ("dfc2 dgelu", M, 3072, 768, "DGELU"), ("plain bf16", M, 3072, 768, "BF16")]:
        A, W, kw = operands(M_, N, K, EPI[name], g)
        outs = run(libs["new"], A, W, EPI[name], **kw)
        outs_l = [o for o in outs if o is not None]
        ref = [o.clone() for o in outs_l]
        fns = {t: (lambda t=t: run(libs[t], A, W, EPI[name], outs=outs, **kw)) for t in libs}
        line = f"{label:12s} M {M_} N {N} K {K}"
        res = {t: [] for t in libs}
        for t in libs:
            timeit(fns[t], 3)
        for _ in range(rounds):
            for t in libs:
                res[t].append(timeit(fns[t], reps))
        for t in libs:
            fns[t]()
            torch.cuda.synchronize()
            diff = max(float((o.float() - r.float()).abs().max()) for o, r in zip(outs_l, ref))
            nd = sum(int((o != r).sum()) for o, r in zip(outs_l, ref))
            line += f" | {t} {statistics.median(res[t]):7.1f} us (min {min(res[t]):.1f}; vs new: max abs diff {diff:.2e}, {nd} elements differ)"
        print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    if what == ["window"]:
        store_window()
        sys.exit(0)
    if what == ["epi"]:
        epi_ablate()
        sys.exit(0)
    if what == ["gelu"]:
        gelu_ab()
        sys.exit(0)
    libs = {"old": load("old", {"PVRL_NT8": "0"}), "new": load("new", {"PVRL_NT8": "1"}),
            "new_nt": load("new_nt", {"PVRL_NT8": "1", "PVRL_NT_TAILS": "0"})}
    for tag in filter(None, os.environ.get("NT8_EXTRA", "").split(",")):       # variant builds (tools/build_variant.py <tag> ...) next to the product
        libs[tag] = load(tag, {"PVRL_NT8": "1"}, os.path.join(HERE, "..", "..", "procedurevrl_amd", "csrc", "variants", f"libpvrl_hip_{tag}.so"))
    rc = 0
    if "trace" in what:
        trace()
    if "ablate" in what:
        ablate()
    if "check" in what:
        rc |= check(libs)
    if "race" in what:
        rc |= race(libs, int(what[what.index("race") + 1]) if what.index("race") + 1 < len(what) and what[what.index("race") + 1].isdigit() else 20)
    if "time" in what:
        time_all(libs)
    sys.exit(1 if rc else 0)
