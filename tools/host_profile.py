"""Host-side cost of one encoder forward + backward (no autograd thread: engine.forward / engine.backward are called
directly so cProfile sees both).  usage: python tools/host_profile.py [clips]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from procedurevrl_amd.build import build_model
from procedurevrl_amd.config import get_cfg
from procedurevrl_amd.datasets import synthetic_label_emb

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = get_cfg()
cfg.MODEL.MODEL_NAME = "vit_base_patch16_224_develop"
cfg.MODEL.NUM_CLASSES = 9871
cfg.MODEL.PRETRAINED = False
cfg.MODEL.DROP_PATH = 0.1
cfg.DEV.MATCH_LANG_EMB = True
cfg.NUM_GPUS = 1
cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
model = build_model(cfg, gpu_id=0).train()
eng = model.model.engine
frames = torch.randn(B, 3, 8, 224, 224, device="cuda")
dfeat = torch.randn(B, 768, device="cuda")


def step():
    eng.forward(frames, True)
    eng.backward(dfeat)


for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    eng.forward(frames, True)
    t1 = time.perf_counter()
    eng.backward(dfeat)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"host enqueue: forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms; step incl. GPU {1e3 * (t3 - t0):.1f} ms")
prof = cProfile.Profile()
prof.enable()
for _ in range(3):
    step()
prof.disable()
torch.cuda.synchronize()
pstats.Stats(prof, stream=sys.stdout).sort_stats("tottime").print_stats(28)
