"""MViTv2-S at real size: HIP-graph replay vs eager launches, several consecutive replays, per-parameter gradient
differences (the MViT backward has fp32 atomics: expect ~1e-6 relative noise, nothing larger)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.argv = ["bench.py", "--arch", "mvit"]
B = 4
from procedurevrl_amd.build import build_model
from procedurevrl_amd.config import get_cfg
from procedurevrl_amd.datasets import synthetic_label_emb

cfg = get_cfg()
cfg.MODEL.MODEL_NAME = "MViT"; cfg.MODEL.ARCH = "mvit"; cfg.MODEL.NUM_CLASSES = 9871; cfg.MODEL.PRETRAINED = False
cfg.DEV.MATCH_LANG_EMB = True
cfg.DATA.NUM_FRAMES = 16; cfg.DATA.INPUT_CHANNEL_NUM = [3]; cfg.DATA.TRAIN_CROP_SIZE = cfg.DATA.TEST_CROP_SIZE = 224
mv = cfg.MVIT
mv.ZERO_DECAY_POS_CLS, mv.USE_ABS_POS, mv.REL_POS_SPATIAL, mv.REL_POS_TEMPORAL = False, False, True, True
mv.DEPTH, mv.NUM_HEADS, mv.EMBED_DIM = 16, 1, 96
mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING = [3, 7, 7], [2, 4, 4], [1, 3, 3]
mv.DROPPATH_RATE, mv.MODE, mv.CLS_EMBED_ON = 0.0, "conv", True
mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]; mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
mv.POOL_KVQ_KERNEL, mv.POOL_KV_STRIDE_ADAPTIVE = [3, 3, 3], [1, 8, 8]
mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
mv.DIM_MUL_IN_ATT, mv.RESIDUAL_POOLING = True, True
cfg.NUM_GPUS = 1
cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
torch.manual_seed(0)
model = build_model(cfg, gpu_id=0).train()
vt = model.model
eng = vt.engine
g = torch.Generator(device="cuda").manual_seed(1)
frames = [torch.randn(B, 3, 16, 224, 224, device="cuda", generator=g) for _ in range(2)]
C = vt.embed_dim
dfeat = [torch.randn(B, C, device="cuda", generator=g) for _ in range(2)]
params = eng._enc_params()
names = [n for n, q in vt.named_parameters() if any(q is r for r in params)]


def run(i):
    for p in vt.parameters():
        p.grad = None
    feat = eng.forward(frames[i], True).clone()
    eng.backward(dfeat[i])
    return feat, [p.grad.clone() if p.grad is not None else None for p in params]


def cmp(a, b, tag):
    worst = []
    for n, x, y in zip(names, a[1], b[1]):
        if x is None or y is None:
            if (x is None) != (y is None):
                worst.append((float("inf"), n))
            continue
        d = float((x - y).norm() / y.norm().clamp_min(1e-30))
        worst.append((d, n))
    worst.sort(reverse=True)
    print(tag, "feat equal", torch.equal(a[0], b[0]), "worst grads:", [(f"{d:.2e}", n) for d, n in worst[:4]])
FILL = os.environ.get("FILL")
if FILL:
    _empty, _empty_like = torch.empty, torch.empty_like
    val = float("nan") if FILL == "nan" else 0.0
    def empty(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point() and t.is_cuda:
            t.fill_(val)
        return t
    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        if t.is_floating_point() and t.is_cuda:
            t.fill_(val)
        return t
eng.use_graphs = False
ref = [run(0), run(1)]
if FILL:
    torch.empty, torch.empty_like = empty, empty_like
    cmp(run(0), ref[0], f"eager with torch.empty filled with {FILL}")
    cmp(run(1), ref[1], f"eager with torch.empty filled with {FILL}")
ref2 = run(0)
cmp(ref2, ref[0], "eager vs eager (atomics noise)")
eng.use_graphs = True
for _ in range(eng.GRAPH_WARMUP + 1):
    run(0)
def bwd_only(i):
    for p in vt.parameters():
        p.grad = None
    key = list(eng._graphs.keys())[0]
    eng._gkey = key
    eng.saved = eng._graphs[key]["saved"]
    eng.backward(dfeat[i])
    return None, [p.grad.clone() if p.grad is not None else None for p in params]
def cmpg(a, b, tag):
    worst = []
    for n, x, y in zip(names, a[1], b[1]):
        d = float((x - y).norm() / y.norm().clamp_min(1e-30))
        worst.append((d if d == d else float("inf"), n))
    worst.sort(reverse=True)
    bad = [w for w in worst if w[0] > 1e-2 and "norm_k.bias" not in w[1]]
    print(tag, "bad grads:", len(bad), [(f"{d:.1e}", n.replace("video_encoder.", "")) for d, n in bad[:5]])
r = run(0); cmpg(r, ref[0], "A fwd+bwd replay (input 0)")
r = bwd_only(0); cmpg(r, ref[0], "B bwd replay again, no fwd")
r = bwd_only(0); cmpg(r, ref[0], "C bwd replay again, no fwd")
r = run(0); cmpg(r, ref[0], "D fwd+bwd replay")
r = run(0); cmpg(r, ref[0], "E fwd+bwd replay")
eng.use_graphs = False
r = run(0); cmpg(r, ref[0], "F eager after graphs")
