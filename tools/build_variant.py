"""Build a variant of the bf16 library with extra compiler switches, for same-box A/B runs of bench.py:
    python tools/build_variant.py <tag> [--only a.hip,b.hip] -DPVRL_NT_GM=4 ...   ->  procedurevrl_amd/csrc/variants/libpvrl_hip_<tag>.so
    PVRL_LIB_PATH=procedurevrl_amd/csrc/variants/libpvrl_hip_<tag>.so python bench.py ...
--only: recompile just these sources with the switches and link the product build's objects (csrc/build/) for the rest."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from procedurevrl_amd.csrc import build_ext as be  # noqa: E402


def main():
    tag, extra = sys.argv[1], sys.argv[2:]
    only = None
    if extra and extra[0] == "--only":
        only, extra = set(extra[1].split(",")), extra[2:]
        be.build(flavours=("bf16",), verbose=False)         # the other objects come from the (up-to-date) product build
    out = os.path.join(be.HERE, "variants")
    obj = os.path.join(out, "build_" + tag)
    os.makedirs(obj, exist_ok=True)
    hipcc = be._hipcc()

    def one(src):
        if only is not None and src not in only:
            return os.path.join(be.HERE, "build", src[:-4] + ".o")
        o = os.path.join(obj, src[:-4] + ".o")
        r = subprocess.run([hipcc] + be.FLAGS + extra + ["-c", os.path.join(be.HERE, src), "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return o
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(one, be.sources()))
    lib = os.path.join(out, f"libpvrl_hip_{tag}.so")
    r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={be.ARCH}", "-o", lib] + objs, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    print(lib)


if __name__ == "__main__":
    main()
