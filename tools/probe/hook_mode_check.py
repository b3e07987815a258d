import os, sys, socket
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch, torch.distributed as dist
import e2e_checks as ec
from procedurevrl_amd import distributed as du
from procedurevrl_amd.datasets import synthetic_label_emb
from procedurevrl_amd.functional import kl_topk_loss
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group(backend="gloo", rank=0, world_size=1)
dev = "cuda:0"
torch.manual_seed(0)
cfg = ec.make_cfg(2, 32, 64)
model = ec.build(cfg, synthetic_label_emb(64, 512, seed=1)).to(dev).train()
vt = model.model
with torch.no_grad():
    for blk in vt.blocks:
        torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
g = torch.Generator(device=dev).manual_seed(100)
x = torch.randn(4, 3, 8, 32, 32, device=dev, generator=g)
teacher = torch.randn(4, 64, device=dev, generator=g) * 3
def step(reducer):
    model.zero_grad(set_to_none=True)
    kl_topk_loss(model(x), teacher, 5).backward()
    if reducer is not None:
        reducer.finish()
    gs = vt.adopt_grads()
    return gs.flat[:gs.end].clone()
own = step(None)
own2 = step(None)
print("own repeat equal:", torch.equal(own, own2))
red = du.GradReducer(vt, enabled=True)
gs = vt.grad_store()
for k in range(5):
    r = step(red)
    bad = []
    for n, o, p in zip(gs.names, gs.offsets, gs.params):
        a, b = r[o:o + p.numel()], own[o:o + p.numel()]
        e = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)
        if e > 1e-6: bad.append((n, f"{e:.2e}"))
    print("step", k, "mismatching params:", bad[:12], len(bad))
dist.destroy_process_group()
