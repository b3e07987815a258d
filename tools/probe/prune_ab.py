"""Last-block pruning (EncoderEngine.prune_last) on / off in one process: same weights, inputs and DropPath draws ->
loss and every parameter gradient of one step compared, then the loss of six optimiser steps from the same start.
usage: python tools/probe/prune_ab.py [clips]"""
import copy
import os
import sys

os.environ.setdefault("PVRL_HIP_GRAPHS", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402


def main():
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.functional import kl_topk_loss, l2norm
    from procedurevrl_amd.losses import MILNCELoss
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME, cfg.MODEL.ARCH, cfg.MODEL.NUM_CLASSES = "vit_base_patch16_224_develop", "vit", 9871
    cfg.MODEL.PRETRAINED, cfg.MODEL.LOSS_FUNC, cfg.MODEL.DROP_PATH = False, "kldiv", 0.1
    cfg.DEV.MATCH_LANG_EMB = True
    cfg.DATA.NUM_FRAMES = 8
    cfg.NUM_GPUS = 1
    cfg.SOLVER.OPTIMIZING_METHOD, cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY = "adamw", 5e-5, 1e-4
    torch.manual_seed(0)
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
    model = build_model(cfg, gpu_id=0)
    vt = model.model
    with torch.no_grad():
        for blk in vt.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
        torch.nn.init.trunc_normal_(vt.time_embed, std=0.02)
    model.train()
    start = copy.deepcopy(model.state_dict())
    g = torch.Generator(device=dev).manual_seed(1234)
    frames = torch.randn(B, 3, 8, 224, 224, device=dev, generator=g)
    teacher = torch.randn(B, 9871, device=dev, generator=g) * 4.0
    text_emb = l2norm(torch.randn(B, 512, device=dev, generator=g))
    nce = MILNCELoss()

    def loss_of():
        pred = model(frames)
        loss = kl_topk_loss(pred, teacher, 5)
        v = vt.last_video_emb
        return loss + nce(v * (1.0 / 0.07 ** 0.5), text_emb * (1.0 / 0.07 ** 0.5))

    res = {}
    for prune in (False, True):
        vt.engine.prune_last = prune
        model.load_state_dict(start)
        opt = construct_optimizer(model, cfg)
        set_lr(opt, cfg.SOLVER.BASE_LR)
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(7)
        loss = loss_of()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        traj = []
        for it in range(6):
            opt.zero_grad(set_to_none=True)
            torch.manual_seed(100 + it)
            l = loss_of()
            l.backward()
            opt.step()
            traj.append(float(l))
        res[prune] = (float(loss), grads, traj)
        del opt
    (l0, g0, t0), (l1, g1, t1) = res[False], res[True]
    print(f"clips {B}: loss of the first step  all rows {l0:.7f}   cls rows only {l1:.7f}   rel diff {abs(l0 - l1) / abs(l0):.2e}")
    print("parameters with a gradient:", len(g0), len(g1), "same set" if set(g0) == set(g1) else "DIFFERENT SETS")
    worst = []
    for k in g0:
        a, b = g1[k].float(), g0[k].float()
        worst.append(((a - b).norm().item() / max(b.norm().item(), 1e-30), k, b.norm().item()))
    worst.sort(reverse=True)
    for e, k, n in worst[:8]:
        print(f"   grad rel diff {e:.2e}  |g| {n:.3e}  {k}")
    print("loss per optimiser step, all rows     :", " ".join(f"{x:.5f}" for x in t0))
    print("loss per optimiser step, cls rows only:", " ".join(f"{x:.5f}" for x in t1))


if __name__ == "__main__":
    main()
