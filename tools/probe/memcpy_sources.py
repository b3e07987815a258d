"""Round 6 probe: which host calls issue the ~70 device-to-device copies of a training step (rocprof: __amd_rocclr_copyBuffer)?
One eager step of the bench model under torch.profiler with stacks; prints every op that launched a Memcpy DtoD with its call site."""
import collections
import os
import sys

os.environ["PVRL_HIP_GRAPHS"] = "0"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import e2e_checks as ec  # noqa: E402
from procedurevrl_amd.datasets import synthetic_label_emb  # noqa: E402
from procedurevrl_amd.functional import kl_topk_loss  # noqa: E402
from procedurevrl_amd.optimizer import construct_optimizer, set_lr  # noqa: E402

cfg = ec.make_cfg(12, 224, 9871, drop_path=0.1)
cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
model = ec.build(cfg, synthetic_label_emb(9871, 512, seed=0)).to("cuda:0").train()
opt = construct_optimizer(model, cfg)
set_lr(opt, 5e-5)
x = torch.randn(8, 3, 8, 224, 224, device="cuda:0")
t = torch.randn(8, 9871, device="cuda:0")


def step():
    opt.zero_grad(set_to_none=True)
    loss = kl_topk_loss(model(x), t, 5)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == "CPU" and any("Memcpy" in k.name or "copyBuffer" in k.name for k in ev.kernels):
        site = next((s for s in ev.stack if "procedurevrl_amd" in s or "bench" in s or "e2e_checks" in s), ev.stack[0] if ev.stack else "?")
        cnt[(ev.name, site)] += 1
for (name, site), n in cnt.most_common(40):
    print(f"{n:4d}  {name:28s} {site}")
print("total", sum(cnt.values()))
