"""Side measurement (not the bench.py contract): the reference's FULL pre-training step at real size -- b videos x 9 clips
through the encoder, the frozen 12-layer CLIP text tower, the order/diffusion transformer, KL(top-5) + MSE, backward,
fused AdamW ("exercised in a separate 36-clip = 4 x 9 full-step run").
usage: python tools/bench_full_step.py [--arch vit|mvit] [--videos 4] [--steps 8]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="vit", choices=["vit", "mvit"])
    ap.add_argument("--videos", type=int, default=4)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=6)   # graphs are captured on the 3rd step; the next one is slow once
    ap.add_argument("--host-profile", action="store_true", help="cProfile the host side of the timed steps (stderr)")
    args = ap.parse_args()
    import torch
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import SyntheticHowTo100M, synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    from procedurevrl_amd.vit import pretrain_loss
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.PRETRAINED", "False", "MODEL.NUM_CLASSES", "9871", "MODEL.TEXT_MODEL", "clip_vit_b_16",
                         "MODEL.LOSS_FUNC", "kldiv", "MODEL.DROP_PATH", "0.1", "DEV.MATCH_LANG_EMB", "True",
                         "DEV.ORDER_PRETRAIN_ENABLED", "True", "NUM_GPUS", "1", "SOLVER.OPTIMIZING_METHOD", "adamw"])
    frames = 8
    if args.arch == "mvit":
        frames = 16
        cfg.MODEL.MODEL_NAME, cfg.MODEL.ARCH = "MViT", "mvit"
        cfg.DATA.INPUT_CHANNEL_NUM = [3]
        mv = cfg.MVIT
        mv.ZERO_DECAY_POS_CLS, mv.USE_ABS_POS, mv.REL_POS_SPATIAL, mv.REL_POS_TEMPORAL = False, False, True, True
        mv.DEPTH, mv.NUM_HEADS, mv.EMBED_DIM = 16, 1, 96
        mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING = [3, 7, 7], [2, 4, 4], [1, 3, 3]
        mv.DROPPATH_RATE, mv.MODE, mv.CLS_EMBED_ON = 0.0, "conv", True
        mv.DIM_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
        mv.HEAD_MUL = [[1, 2.0], [3, 2.0], [14, 2.0]]
        mv.POOL_KVQ_KERNEL, mv.POOL_KV_STRIDE_ADAPTIVE = [3, 3, 3], [1, 8, 8]
        mv.POOL_Q_STRIDE = [[i, 1, 2, 2] if i in (1, 3, 14) else [i, 1, 1, 1] for i in range(16)]
        mv.DIM_MUL_IN_ATT, mv.RESIDUAL_POOLING = True, True
    else:
        cfg.MODEL.MODEL_NAME = "vit_base_patch16_224_develop"
    cfg.DATA.NUM_FRAMES = frames
    cfg.DATA.TRAIN_CROP_SIZE = cfg.DATA.TEST_CROP_SIZE = 224
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(9871, 512, seed=0)
    torch.manual_seed(0)
    model = build_model(cfg, gpu_id=0).train()
    vt = model.model
    vt.text_model.eval()
    opt = construct_optimizer(model, cfg)
    set_lr(opt, 5e-5)
    dev = torch.device("cuda", 0)
    ds = SyntheticHowTo100M(cfg, num_videos=args.videos, seed=1)
    items = [ds[i] for i in range(args.videos)]
    inputs = torch.stack([it[0] for it in items]).to(dev)
    meta = {k: torch.stack([it[3][k] for it in items]).to(dev) for k in ("clip_text_ids", "clip_vis_feat")}
    meta = {k: v.view(-1, v.shape[-1]) for k, v in meta.items()}

    def step():
        pred, teacher, mse = model([inputs, meta])
        loss, l1, l2 = pretrain_loss(pred, teacher, mse, cfg)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        vt.adopt_grads()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    prof = None
    if args.host_profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(args.steps):
        loss = step()
        evs[k + 1].record()
    t_enq = (time.perf_counter() - t0) / args.steps
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(35)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    per_step = [round(evs[k].elapsed_time(evs[k + 1]), 1) for k in range(args.steps)]
    clips = args.videos * 9
    from procedurevrl_amd._lib import OPERAND
    print(json.dumps({"metric": f"training clips/sec ({frames}f x 224^2, {'ViT-B TimeSformer' if args.arch == 'vit' else 'MViTv2-S'}), "
                                "FULL pre-training step", "value": round(clips / dt, 3), "unit": "clips/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 3), "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": OPERAND, "data": "synthetic",
                      "config": {"workload": f"full pre-training step (reference cfg shape): {args.videos} videos x 9 clips of "
                                             f"{frames}x224^2, frozen CLIP-text teacher (12 layers, ctx 77) + order / diffusion "
                                             "transformer + top-5 KL + MSE, fwd+bwd+AdamW (SURVEY 8d's separate 36-clip run)",
                                 "clips_per_gpu": clips, "global_batch": clips, "parallelism": "dp1"},
                      "per_step_ms": per_step, "host_enqueue_ms_per_step": round(1e3 * t_enq, 3), "loss": float(loss)}))


if __name__ == "__main__":
    main()
