"""Static checks of the persistent 8-wave NT GEMM's gfx950 code (procedurevrl_amd/csrc/gemm_nt8_core.h), on the build host.

The kernel keeps LDS-DMA in flight across barriers with COUNTED `s_waitcnt vmcnt(N)`; two of its counts depend on what the compiler
emits, so they are pinned here instead of trusted:
  * the wait of the first K-tile behind a tile seam is vmcnt(8 + NST), NST = the epilogue's store instructions per wave -- a larger
    assumed NST than the real one would let the wait pass before the prefetched K-tile has landed;
  * no scratch (a spill reload is followed by vmcnt(0): it would drain the prefetch) and no waterfall loop around a buffer
    instruction (a descriptor the compiler cannot prove uniform), in any epilogue flavour.
hipcc cross-compiles without a GPU (~10 s)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "procedurevrl_amd", "csrc", "gemm_nt.hip")
NST = {0: 16, 1: 32, 2: 32, 3: 32, 4: 32, 5: 16, 6: 16, 7: 16}      # PVRL_EPI_* -> stores per wave of one whole tile


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "gemm_nt.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Wno-unused-result", SRC,
                    "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


def _kernel(asm, epi):
    name = f"_ZN12_GLOBAL__N_115gemm_nt8_kernelILi{epi}EEEvNS_6GemmNTE"
    a = asm.index(name + ":")
    body = asm[a:asm.index(".Lfunc_end", a)]
    meta = asm[asm.index(".name:", asm.index("amdhsa.kernels")):]
    m = re.search(r"\.name:\s+" + name + r"\n(.*?)(?=\n\s+- \.a|\Z)", asm[asm.index("amdhsa.kernels"):], re.S)
    return body, (m.group(0) if m else meta)


@pytest.mark.parametrize("epi", sorted(NST))
def test_nt8_kernel_code(asm, epi):
    body, _ = _kernel(asm, epi)
    # whole-tile epilogue = two 64-row blocks, the half item's epilogue = one more: 3 blocks of NST / 2 stores
    assert body.count("buffer_store_dwordx4") == 3 * NST[epi] // 2
    assert body.count("buffer_store_") == body.count("buffer_store_dwordx4")
    assert body.count(f"s_waitcnt vmcnt({8 + NST[epi]})") == 2        # phases 1 and 3 of the first K-tile behind a seam
    assert "scratch_" not in body and "buffer_load_dword v" not in body          # no spill traffic
    assert "s_and_saveexec" not in body                                           # no waterfall loops
    # LDS-DMA: 14 (cold start) + 8 (K-tile body) + 10 + 6 (half item); the compiler may peel the K loop's first iteration (+ 8)
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", body)) in (38, 46)
    # every MFMA sits in a cluster between s_setprio 1 / 0 (a phase's compute segment): 32 in the whole tiles' two phases per K-tile,
    # 16 in the half items' quadrant phases -- and nothing else does
    assert body.count("s_setprio 1") == body.count("s_setprio 0")
    n_mfma = 0
    for seg in body.split("s_setprio 1")[1:]:
        cluster = seg[:seg.index("s_setprio 0")]
        ops = [l.split()[0] for l in cluster.splitlines() if l.strip() and not l.strip().startswith(";")]
        assert [o for o in ops if not o.startswith("v_mfma") and o != "s_waitcnt" and o != "s_nop"] == [], ops
        k = sum(o.startswith("v_mfma") for o in ops)
        assert k in (16, 32), k
        n_mfma += k
    assert body.count("v_mfma_f32_16x16x32") == n_mfma


def test_nt8_kernel_resources(asm):
    for epi in NST:
        name = f"_ZN12_GLOBAL__N_115gemm_nt8_kernelILi{epi}EEEvNS_6GemmNTE"
        m = re.search(r"\.amdhsa_kernel " + name + r"\n(.*?)\.end_amdhsa_kernel", asm, re.S)
        assert m, name
        d = m.group(1)
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", d).group(1))
        assert vg <= 256, (epi, vg)
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", d).group(1)) == 0
        assert int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", d).group(1)) == 131072
