"""Probe: per-block time of MViTv2-S's relative-position and pooling-attention kernels at 32 clips of 16x224^2."""
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



from procedurevrl_amd.mvit import rel_index  # noqa: E402

tot = dict(rel_f=0.0, rel_b=0.0, att_f=0.0, att_b=0.0)
for i, pl in enumerate(plan):
    H, dout, thw = pl["heads"], pl["dim_out"], tuple(pl["in_thw"])
    q_thw, k_thw = om.pool_out_thw(thw, pl["stride_q"]), om.pool_out_thw(thw, pl["stride_kv"])
    Lq, Lk = q_thw[0] * q_thw[1] * q_thw[2], k_thw[0] * k_thw[1] * k_thw[2]
    BH = B * H
    mk = lambda L: (torch.randn(BH, L + 1, 96, device=DEV, generator=g) * 0.5).to(ops.OP16)
    q, k, v = mk(Lq), mk(Lk), mk(Lk)
    tabs = [torch.randn(2 * max(a, b) - 1, 96, device=DEV, generator=g) * 0.1 for a, b in
            ((q_thw[1], k_thw[1]), (q_thw[2], k_thw[2]), (q_thw[0], k_thw[0]))]
    idx = [rel_index(a, b).to(DEV, torch.int32).contiguous() for a, b in
           ((q_thw[1], k_thw[1]), (q_thw[2], k_thw[2]), (q_thw[0], k_thw[0]))]
    rel = om.rel_fwd(q, BH, q_thw, k_thw, *tabs, *idx, out_scale=96 ** 0.5)
    t_rf = timeit(lambda: om.rel_fwd(q, BH, q_thw, k_thw, *tabs, *idx, out_scale=96 ** 0.5))
    ldo = om.pad128(dout)
    o, lse = om.attn_fwd(q, k, v, rel, B, H, Lq, k_thw, 96 ** -0.5, ldo)
    t_af = timeit(lambda: om.attn_fwd(q, k, v, rel, B, H, Lq, k_thw, 96 ** -0.5, ldo))
    d_o = (torch.randn(o.shape, device=DEV, generator=g) * 0.1).to(ops.OP16)
    dq, dk, dv, drel = om.attn_bwd(q, k, v, rel, B, H, Lq, k_thw, 96 ** -0.5, o, d_o, lse)
    t_ab = timeit(lambda: om.attn_bwd(q, k, v, rel, B, H, Lq, k_thw, 96 ** -0.5, o, d_o, lse))
    dts = [torch.zeros_like(t) for t in tabs]
    t_rb = timeit(lambda: om.rel_bwd(drel, q, dq, BH, q_thw, k_thw, *tabs, *idx, *dts))
    tot["rel_f"] += t_rf; tot["rel_b"] += t_rb; tot["att_f"] += t_af; tot["att_b"] += t_ab
    print(f"blk {i:2d} H {H} q {q_thw} k {k_thw}: rel fwd {t_rf:7.1f} bwd {t_rb:7.1f}   attn fwd {t_af:7.1f} bwd {t_ab:7.1f}", flush=True)
print("per step (ms): " + "  ".join(f"{k} {v / 1e3:.2f}" for k, v in tot.items()))
