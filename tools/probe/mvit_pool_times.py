"""Probe: per-block time of MViTv2-S's conv-pool kernels (forward, backward) at 32 clips of 16x224^2."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from procedurevrl_amd import ops, ops_mvit as om  # noqa: E402
from procedurevrl_amd.config import get_cfg  # noqa: E402
from procedurevrl_amd.mvit import mvit_plan  # noqa: E402

cfg = get_cfg()
cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE = 16, 224
mv = cfg.MVIT
mv.DEPTH, mv.NUM_HEADS, mv.EMBED_DIM = 16, 1, 96
mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING = [3, 7, 7], [2, 4, 4], [1, 3, 3]
mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
mv.POOL_KVQ_KERNEL, mv.POOL_KV_STRIDE_ADAPTIVE = [3, 3, 3], [1, 8, 8]
mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
thw0, plan = mvit_plan(cfg)
B, DEV = 32, "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot_f = tot_b = 0.0
ONLY = [int(v) for v in os.environ.get("PVRL_POOL_BLOCKS", "").split(",") if v]      # e.g. PVRL_POOL_BLOCKS=4 under rocprofv3
for i, pl in enumerate(plan):
    if ONLY and i not in ONLY:
        continue
    H, dout, thw = pl["heads"], pl["dim_out"], tuple(pl["in_thw"])
    L = thw[0] * thw[1] * thw[2]
    ld = om.pad128(3 * dout)
    qkv = torch.randn(B * L + B, ld, device=DEV, generator=g).to(ops.OP16)
    w = torch.randn(96, 27, device=DEV, generator=g) * 0.1
    gam = torch.ones(96, device=DEV); bet = torch.zeros(96, device=DEV)
    row = [f"blk {i:2d} thw {thw} H {H}"]
    for name, col0, st in (("q", 0, tuple(pl["stride_q"])), ("kv", dout, tuple(pl["stride_kv"]))):
        y, c = om.pool_fwd(qkv, col0, B, H, thw, st, w, gam, bet, 1e-6)
        tf = timeit(lambda: om.pool_fwd(qkv, col0, B, H, thw, st, w, gam, bet, 1e-6))
        dy = torch.randn_like(y)
        dqkv = torch.zeros_like(qkv)
        dw = torch.zeros(96, 27, device=DEV); dg = torch.zeros(96, device=DEV); db = torch.zeros(96, device=DEV)
        tb = timeit(lambda: om.pool_bwd(dy, c, qkv, dqkv, col0, B, H, thw, st, w, gam, 1e-6, dw, dg, db))
        mult = 1 if name == "q" else 2
        tot_f += mult * tf; tot_b += mult * tb
        row.append(f"{name} s{st}: fwd {tf:7.1f} bwd {tb:7.1f}")
    print("  ".join(row), flush=True)
print(f"per step: fwd {tot_f / 1e3:.2f} ms, bwd {tot_b / 1e3:.2f} ms")
