"""Compile the gfx950 HIP kernels into one C-ABI shared library, in-tree.

`python -m procedurevrl_amd.csrc.build_ext` (or `__graft_entry__.build()`) runs
`hipcc --offload-arch=gfx950 -O3 -shared -fPIC` over every `*.hip` next to this file and
writes `libpvrl_hip.so` here.  hipcc cross-compiles without a GPU.  Objects are cached by
source mtime so an edit to one kernel recompiles one file.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpvrl_hip.so")
# (library, object directory, extra flags): the same sources built for each 16-bit operand type (common.h)
FLAVOURS = {"bf16": (LIB, "build", []), "f16": (os.path.join(HERE, "libpvrl_hip_f16.so"), "build_f16", ["-DPVRL_OPERAND_F16"])}
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "..", "include", "pvrl.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=True, flavours=("bf16", "f16")):
    """compile every flavour's library; returns the default (bf16) library path"""
    for fl in flavours:
        _build_one(fl, force, verbose)
    return LIB


def _build_one(flavour, force, verbose):
    LIB, objname, extra = FLAVOURS[flavour]
    objdir = os.path.join(HERE, objname)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for src in sources():
        sp = os.path.join(HERE, src)
        op = os.path.join(objdir, src[:-4] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_m):
            jobs.append((sp, op))

    def compile_one(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + extra + ["-c", sp, "-o", op]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stderr}")
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[pvrl build] compiled {os.path.basename(done)}", file=sys.stderr)
    need_link = bool(jobs) or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[pvrl build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
