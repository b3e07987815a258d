"""Static checks of the ping-pong weight-gradient kernel's gfx950 code (procedurevrl_amd/csrc/gemm_tn8_core.h), on the build host.

Like the persistent NT kernel it keeps LDS-DMA in flight across barriers with COUNTED `s_waitcnt vmcnt(8)`: the count is the number of
copies a wave issues per stage, so the instruction counts are pinned here; a spill (scratch reload = vmcnt(0)) or an array the compiler
indexes dynamically (the column-sum duty once put the fragments in scratch) would break the schedule silently.  ~10 s of hipcc."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "procedurevrl_amd", "csrc", "gemm_tn.hip")
KERNELS = ["_ZN12_GLOBAL__N_115gemm_tn8_kernelENS_6GemmTNE", "_ZN12_GLOBAL__N_123gemm_tn8_grouped_kernelENS_7TnGroupE"]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "gemm_tn.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Wno-unused-result", SRC,
                    "-o", str(out)], check=True, capture_output=True)
    return out.read_text()


@pytest.mark.parametrize("name", KERNELS)
def test_tn8_kernel_code(asm, name):
    a = asm.index(name + ":")
    body = asm[a:asm.index(".Lfunc_end", a)]
    assert "scratch_" not in body                                                # no spills, no dynamically indexed register arrays
    assert "s_and_saveexec" not in body.split("s_barrier")[1]                     # no waterfall loop around a buffer instruction in the loop
    # LDS-DMA: 14 (cold start) + 8 (one stage); the compiler may peel the loop's first iteration (+ 8)
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", body)) in (22, 30)
    assert body.count("s_waitcnt vmcnt(8)") in (2, 4) and body.count("s_waitcnt vmcnt(6)") == 1
    # every MFMA of the loop sits in a 32-instruction cluster between s_setprio 1 / 0 (two phases per stage); only the bias-gradient dot
    # products share it
    assert body.count("s_setprio 1") == body.count("s_setprio 0")
    assert body.count("v_mfma_f32_16x16x32") == 32 * body.count("s_setprio 1")
    # fragments come through the transposing read: 48 per stage and wave
    assert body.count("ds_read_b64_tr_b16") % 48 == 0 and body.count("ds_read_b64_tr_b16") > 0
    assert "ds_write" not in body.split("s_barrier")[1].split("s_barrier")[0]       # nothing is staged through registers


def test_tn8_kernel_resources(asm):
    for name in KERNELS:
        m = re.search(r"\.amdhsa_kernel " + name + r"\n(.*?)\.end_amdhsa_kernel", asm, re.S)
        assert m, name
        d = m.group(1)
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", d).group(1)) <= 256
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", d).group(1)) == 0
        assert int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", d).group(1)) == 131072
