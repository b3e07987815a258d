"""A block's grouped weight-gradient launch (gemm_tn_rt8_grouped_kernel + reduce) under variant builds of the library.
  python tools/probe/tn_ab.py arm            one process: check against fp32 matmul, then time (the library is PVRL_LIB_PATH or the product build)
  python tools/probe/tn_ab.py tw0 tw8 ...    the product build and procedurevrl_amd/csrc/variants/libpvrl_hip_<tag>.so, interleaved, twice
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def arm():
    import torch
    from procedurevrl_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    R, M, BT = 50176, 50208, 256
    shapes = [(M, 768, 3072), (M, 3072, 768), (R + BT, 768, 768), (M, 2304, 768), (R, 768, 768), (R, 2304, 768), (R, 768, 768)]
    sets = []
    for _ in range(2):          # two operand sets: the re-reads do not come from L2 / MALL
        P = [(torch.randn(m, N, device=dev, generator=g) * 0.05).to(ops.OP16) for m, N, K in shapes]
        Q = [torch.randn(m, K, device=dev, generator=g).to(ops.OP16) for m, N, K in shapes]
        sets.append((P, Q))
    dW = [torch.empty(N, K, device=dev) for m, N, K in shapes]
    db = [torch.empty(N, device=dev) for m, N, K in shapes]
    P, Q = sets[0]
    ops.gemm_tn_grouped([(P[i], Q[i], dW[i], db[i], 0.0) for i in range(7)])
    sig = sum(int(dW[i].view(torch.int32).to(torch.int64).sum()) for i in range(7)) & 0xffffffffffff      # bit signature of the seven dW
    worst = 0.0
    for i in (0, 1, 3, 6):
        ref = P[i].float().t() @ Q[i].float()
        worst = max(worst, float((dW[i] - ref).abs().max() / ref.abs().max()))
        worst = max(worst, float((db[i] - P[i].float().sum(0)).abs().max() / P[i].float().sum(0).abs().max()))
    k = [0]

    nodb = bool(os.environ.get("TN_NO_DBIAS"))

    def run():
        P, Q = sets[k[0] & 1]; k[0] += 1
        ops.gemm_tn_grouped([(P[i], Q[i], dW[i], None if nodb else db[i], 0.0) for i in range(7)])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = sum(2.0 * m * N * K for m, N, K in shapes)
    ts.sort()
    print(f"  dW bits {sig:012x} | worst rel err vs fp32 matmul {worst:.2e} {'OK' if worst < 2e-5 else 'BAD (expected for an ablation build)'} | grouped launch + reduce: median {ts[2]:7.1f} us (min {ts[0]:.1f}) = {fl / ts[2] / 1e6:6.0f} TFLOP/s", flush=True)


def main():
    if sys.argv[1:] == ["arm"]:
        return arm()
    tags = ["product"] + sys.argv[1:]
    for rep in range(2):
        for t in tags:
            env = dict(os.environ)
            if t.endswith(":rt8"):              # the same build with the register-transposed kernel (PVRL_TN8=0)
                env["PVRL_TN8"] = "0"
                t = t[:-4]
            if t.endswith(":nodb"):             # the same build without the bias gradients (column sums of dY)
                env["TN_NO_DBIAS"] = "1"
                t = t[:-5]
            if t != "product":
                env["PVRL_LIB_PATH"] = os.path.join(ROOT, "procedurevrl_amd", "csrc", "variants", f"libpvrl_hip_{t}.so")
            print(f"--- {t}{' without dbias' if env.get('TN_NO_DBIAS') else ''}{' PVRL_TN8=0' if env.get('PVRL_TN8') else ''}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=env, check=False)


if __name__ == "__main__":
    main()
