"""Probe: magnitude statistics of the 16-bit GEMM operands of one training step at the bench configuration (forward
activations and backward gradient operands), to size the loss scale an fp16-operand datapath would need.
Prints per call site: absmax, rms, fraction below fp16's smallest normal (2^-14) and below its smallest subnormal (2^-24)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["PVRL_HIP_GRAPHS"] = "0"
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd.build import build_model  # noqa: E402
from procedurevrl_amd.config import get_cfg  # noqa: E402
from procedurevrl_amd.datasets import synthetic_label_emb  # noqa: E402
from procedurevrl_amd.functional import kl_topk_loss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = get_cfg()
cfg.MODEL.MODEL_NAME, cfg.MODEL.ARCH, cfg.MODEL.NUM_CLASSES, cfg.MODEL.PRETRAINED = "vit_base_patch16_224_develop", "vit", 9871, False
cfg.MODEL.LOSS_FUNC, cfg.MODEL.DROP_PATH, cfg.DEV.MATCH_LANG_EMB, cfg.NUM_GPUS = "kldiv", 0.1, True, 1
cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0).train()
vt = model.model
with torch.no_grad():
    for blk in vt.blocks:
        torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, 3, 8, 224, 224, device="cuda", generator=g)
teacher = torch.randn(B, 9871, device="cuda", generator=g) * 4

stats = {}
phase = ["fwd"]
orig = ops.gemm_nt


def rec(tag, t):
    t = t.float()
    a = t.abs()
    s = stats.setdefault(tag, [0.0, 0.0, 0, 0.0, 0.0, 0])
    s[0] = max(s[0], float(a.max()))
    s[1] += float((t * t).sum())
    s[2] += t.numel()
    nz = a > 0
    s[3] += float((nz & (a < 2.0 ** -14)).sum())
    s[4] += float((nz & (a < 2.0 ** -24)).sum())
    s[5] += 1


def gemm_nt(A, W, epi, *a, **k):
    rec(f"{phase[0]} nt epi{epi} A[{A.shape[0]}x{A.shape[1]}]->N{W.shape[0]}", A)
    out = orig(A, W, epi, *a, **k)
    o0 = out[0] if isinstance(out, tuple) else out
    if o0.dtype != torch.float32:
        rec(f"{phase[0]} nt epi{epi} OUT N{W.shape[0]}", o0)
    return out


ops.gemm_nt = gemm_nt
import procedurevrl_amd.engine as eng  # noqa: E402
eng.ops.gemm_nt = gemm_nt
pred = model(x)
loss = kl_topk_loss(pred, teacher, 5)
phase[0] = "bwd"
loss.backward()
torch.cuda.synchronize()
print(f"loss {float(loss):.4f}  clips {B}")
for k, s in sorted(stats.items()):
    print(f"{k:52s} calls {s[5]:3d} absmax {s[0]:.3e} rms {(s[1] / s[2]) ** 0.5:.3e}  <2^-14 {s[3] / s[2]:.4f}  <2^-24 {s[4] / s[2]:.4f}")
