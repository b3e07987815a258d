"""Run every per-kernel parity check on cuda:0 and print / dump a table (debug helper;
the same checks run under pytest -m gpu)."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    import torch
    import kernel_checks as kc
    import e2e_checks as ec
    import mvit_checks as mc
    names = sys.argv[1:]
    rows = []
    ok = True
    for chk in kc.ALL_CHECKS + ec.ALL_CHECKS + mc.ALL_CHECKS:
        if names and not any(n in chk.__name__ for n in names):
            continue
        t0 = time.time()
        try:
            res = chk()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            traceback.print_exc()
            rows.append((chk.__name__, "EXCEPTION " + repr(e)[:200], 0.0, False))
            ok = False
            continue
        for label, err, tol in res:
            good = err <= tol
            ok &= good
            rows.append((label, err, tol, good))
        print(f"# {chk.__name__}: {time.time() - t0:.1f}s", flush=True)
    for label, err, tol, good in rows:
        e = err if isinstance(err, str) else f"{err:.3e}"
        print(f"{'OK  ' if good else 'FAIL'} {label:55s} err={e} tol={tol:g}")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump([list(map(str, r)) for r in rows], open("gpurun_out/kernel_checks.json", "w"), indent=1)
    print("ALL OK" if ok else "SOME FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
