"""Build tools/probe/libpvrl_probe.so: the measured-and-rejected GEMM variants (gemm_variants.hip).  Not product code."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpvrl_probe.so")


def build(force=False):
    src = os.path.join(HERE, "gemm_variants.hip")
    csrc = os.path.join(HERE, "..", "..", "procedurevrl_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("gemm_nt_core.h", "gemm_tn_core.h", "common.h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
        return LIB
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-unused-result",
           src, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
