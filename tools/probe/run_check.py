import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import kernel_checks as kc
for name in sys.argv[1:]:
    for n, err, tol in getattr(kc, name)():
        print(("ok  " if err <= tol else "BAD ") + f"{n:60s} err {err:.3e} tol {tol:.1e}", flush=True)
