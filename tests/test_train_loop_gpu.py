"""The launcher surface on a GPU: run_net-style config -> train() on synthetic data (full pre-training forward with the
CLIP-text teacher and the order transformer), `.pyth` checkpoint written with the reference's schema, auto-resume."""
import os

import pytest
import torch


def _cfg(tmp):
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.MODEL_NAME", "vit_base_patch16_224_develop", "MODEL.PRETRAINED", "False",
                         "MODEL.NUM_CLASSES", "64", "MODEL.TEXT_MODEL", "clip_vit_b_16", "MODEL.LOSS_FUNC", "kldiv",
                         "MODEL.DROP_PATH", "0.1", "TIMESFORMER.DEPTH", "2", "DATA.TRAIN_CROP_SIZE", "32",
                         "DEV.MATCH_LANG_EMB", "True", "DEV.ORDER_PRETRAIN_ENABLED", "True", "TRAIN.BATCH_SIZE", "2",
                         "TRAIN.TEXT", "synthetic", "NUM_GPUS", "1", "GLOBAL_BATCH_SIZE", "2", "SOLVER.MAX_EPOCH", "2",
                         "SOLVER.BASE_LR", "1e-4", "SOLVER.OPTIMIZING_METHOD", "adamw", "SOLVER.LR_POLICY",
                         "steps_with_relative_lrs", "SOLVER.STEPS", "[0, 1]", "SOLVER.LRS", "[1, 0.1]", "LOG_PERIOD", "2",
                         "TRAIN.CHECKPOINT_PERIOD", "1", "SYNTHETIC.ENABLE", "True", "SYNTHETIC.NUM_VIDEOS", "4",
                         "SYNTHETIC.TEXT_LAYERS", "2", "OUTPUT_DIR", str(tmp)])
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(64)
    return cfg


@pytest.mark.gpu
def test_train_checkpoint_resume(tmp_path):
    from procedurevrl_amd import checkpoint as cu
    from procedurevrl_amd.train_net import train
    cfg = _cfg(tmp_path)
    cfg.SOLVER.MAX_EPOCH = 1
    model, opt = train(cfg)
    path = cu.get_path_to_checkpoint(str(tmp_path), 1)
    assert os.path.exists(path) and path.endswith("checkpoints/checkpoint_epoch_00001.pyth")
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck.keys()) == {"epoch", "model_state", "optimizer_state", "cfg"} and ck["epoch"] == 0
    assert "model.blocks.0.temporal_attn.qkv.weight" in ck["model_state"]
    assert "model.order_tfm.temporalModelling.resblocks.0.attn.in_proj_weight" in ck["model_state"]
    assert isinstance(ck["cfg"], str) and "ORDER_PRETRAIN_ENABLED: true" in ck["cfg"]
    # auto-resume: a second run continues at epoch 1 with identical weights and optimiser moments
    cfg2 = _cfg(tmp_path)
    cfg2.SOLVER.MAX_EPOCH = 1
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.optimizer import construct_optimizer
    m2 = build_model(cfg2)
    o2 = construct_optimizer(m2, cfg2)
    start = cu.load_train_checkpoint(cfg2, m2, o2)
    assert start == 1
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    s1, s2 = opt.state_dict(), o2.state_dict()
    assert s1["fused"]["steps"] == s2["fused"]["steps"] > 0
    k = max(s1["state"].keys())
    assert torch.equal(s1["state"][k]["exp_avg"].cpu(), s2["state"][k]["exp_avg"].cpu())


@pytest.mark.gpu
def test_loss_decreases_on_fixed_batch():
    """Optimisation sanity on the HIP path: 12 AdamW steps on one fixed synthetic batch reduce the KL loss."""
    import e2e_checks as ec
    from procedurevrl_amd.functional import kl_topk_loss
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    from procedurevrl_amd.datasets import synthetic_label_emb
    cfg = ec.make_cfg(2, 32, 64)
    cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
    model = ec.build(cfg, synthetic_label_emb(64))
    with torch.no_grad():
        for blk in model.model.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    model.train()
    opt = construct_optimizer(model, cfg)
    set_lr(opt, 3e-4)
    g = torch.Generator(device="cuda:0").manual_seed(0)
    x = torch.randn(6, 3, 8, 32, 32, device="cuda:0", generator=g)
    teacher = torch.randn(6, 64, device="cuda:0", generator=g) * 4
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = kl_topk_loss(model(x), teacher, 5)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses


@pytest.mark.gpu
def test_device_prefetcher_yields_the_loaders_batches_on_the_device(tmp_path):
    """datasets.DevicePrefetcher (pinned staging + copy stream): same batches, same order, on the GPU, nested
    meta dict included; the copy of batch i+1 is issued before batch i is consumed."""
    from procedurevrl_amd.datasets import DevicePrefetcher, construct_loader
    cfg = _cfg(tmp_path)
    cfg.SYNTHETIC.NUM_VIDEOS = 5
    cfg.TRAIN.BATCH_SIZE = 2
    torch.manual_seed(3)
    plain = list(construct_loader(cfg, "train"))
    torch.manual_seed(3)
    pf = DevicePrefetcher(construct_loader(cfg, "train"), "cuda:0")
    assert len(pf) == len(plain) == 2 and pf.stream is not None
    got = []
    for inputs, labels, index, meta in pf:
        assert inputs.is_cuda and labels.is_cuda and meta["clip_text_ids"].is_cuda and meta["clip_vis_feat"].is_cuda
        (inputs * 2).sum().item()                       # consume on the current stream
        got.append((inputs.cpu(), labels.cpu(), index.cpu(), {k: v.cpu() for k, v in meta.items()}))
    for a, b in zip(got, plain):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for k in b[3]:
            assert torch.equal(a[3][k], b[3][k]), k


@pytest.mark.gpu
def test_eval_epoch_runs_every_eval_period_and_matches_recomputation(tmp_path, capsys):
    """tools/train_net.py:251-350, 516-518: train() evaluates every TRAIN.EVAL_PERIOD epochs (and after the last one); the
    logged `val_epoch` top-1 / top-5 errors equal a recomputation from the model's own eval forward on the val split."""
    import json
    from procedurevrl_amd import train_net as tn
    from procedurevrl_amd.datasets import construct_loader
    cfg = _cfg(tmp_path)
    cfg.SOLVER.MAX_EPOCH = 3
    cfg.TRAIN.EVAL_PERIOD = 2
    cfg.TRAIN.CHECKPOINT_PERIOD = 10
    model, _ = tn.train(cfg, max_iters=1)
    lines = [json.loads(l.split("json_stats: ", 1)[1]) for l in capsys.readouterr().out.splitlines() if "json_stats: " in l]
    val = [l for l in lines if l["_type"] == "val_epoch"]
    assert [l["epoch"] for l in val] == ["2/3", "3/3"], val          # epoch 2 (period) and epoch 3 (last)
    # recomputation on the final weights: per-batch errors weighted by batch size, as the meter does
    model.eval()
    mis1 = mis5 = n = 0.0
    with torch.no_grad():
        for inputs, labels, _, _ in construct_loader(cfg, "val"):
            p = model(inputs.cuda())
            lab = labels.cuda().reshape(-1)
            top = p.topk(5, dim=1).indices
            e1 = 100.0 * (1.0 - float((top[:, 0] == lab).float().mean()))
            e5 = 100.0 * (1.0 - float((top == lab[:, None]).any(1).float().mean()))
            mis1 += e1 * inputs.size(0); mis5 += e5 * inputs.size(0); n += inputs.size(0)
    assert abs(val[-1]["top1_err"] - mis1 / n) < 1e-3 and abs(val[-1]["top5_err"] - mis5 / n) < 1e-3, (val[-1], mis1 / n, mis5 / n)
    assert val[-1]["min_top1_err"] <= val[0]["top1_err"] + 1e-9
    assert tn.train.last_val_stats["_type"] == "val_epoch"


def _finetune_cfg(tmp, dataset):
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    cfg = get_cfg()
    cfg.merge_from_list(["MODEL.MODEL_NAME", "vit_base_patch16_224_develop", "MODEL.PRETRAINED", "False",
                         "MODEL.NUM_CLASSES", "10", "MODEL.LOSS_FUNC", "cross_entropy", "MODEL.DROP_PATH", "0.0",
                         "TIMESFORMER.DEPTH", "2", "DATA.TRAIN_CROP_SIZE", "32", "DEV.MATCH_LANG_EMB", "False",
                         "DEV.ORDER_PRETRAIN_ENABLED", "False", "TRAIN.DATASET", dataset, "TRAIN.BATCH_SIZE", "4", "NUM_GPUS", "1",
                         "GLOBAL_BATCH_SIZE", "4", "SOLVER.MAX_EPOCH", "3", "SOLVER.BASE_LR", "3e-4", "SOLVER.OPTIMIZING_METHOD",
                         "adamw", "SOLVER.LR_POLICY", "steps_with_relative_lrs", "SOLVER.STEPS", "[0]", "SOLVER.LRS", "[1]",
                         "LOG_PERIOD", "2", "TRAIN.CHECKPOINT_PERIOD", "10", "TRAIN.EVAL_PERIOD", "10", "SYNTHETIC.ENABLE", "True",
                         "SYNTHETIC.NUM_VIDEOS", "8", "OUTPUT_DIR", str(tmp)])
    cfg.DEV.TEST_LANG_EMB = synthetic_label_emb(16, seed=3)      # frozen projection into the language space (vit.py:246-255)
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["kinetics", "Epickitchens"])
def test_finetune_branch_trains(tmp_path, dataset, capsys):
    """tools/train_net.py:149-150,163-169,195-231: the fine-tuning call model(inputs) -> cross entropy (EPIC-Kitchens: verb
    + noun heads, action accuracy) through the HIP encoder; three epochs on eight labelled synthetic clips reduce the loss, the
    frozen language projection does not move, the logged line carries the reference meter's columns."""
    import json
    from procedurevrl_amd.train_net import train
    cfg = _finetune_cfg(tmp_path, dataset)
    if dataset == "Epickitchens":
        cfg.TRAIN.EVAL_PERIOD = 100
        cfg.SOLVER.MAX_EPOCH = 3
    torch.manual_seed(0)
    model, opt = train(cfg)                                      # (ends with the evaluation of the last epoch, train_net.py:516-518)
    out = capsys.readouterr().out
    lines = [json.loads(l.split("json_stats: ", 1)[1]) for l in out.splitlines() if "json_stats: " in l and '"train_iter"' in l]
    assert len(lines) >= 3
    losses = [l["loss"] for l in lines]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    if dataset == "Epickitchens":
        for key in ("verb_loss", "noun_loss", "verb_top1_acc", "noun_top5_acc", "top1_acc", "top5_acc"):
            assert key in lines[-1]
        assert abs(lines[-1]["loss"] - 0.5 * (lines[-1]["verb_loss"] + lines[-1]["noun_loss"])) < 5e-2
        assert model.model.head_v.weight.grad is not None
        val = [json.loads(l.split("json_stats: ", 1)[1]) for l in out.splitlines() if "json_stats: " in l and '"val_epoch"' in l]
        assert val, "the EPIC run must end with a val_epoch line (EPICValMeter, lib/utils/meters.py:926-960)"
        for key in ("verb_top1_acc", "noun_top1_acc", "top1_acc", "top5_acc", "max_top1_acc"):
            assert key in val[-1] and 0.0 <= val[-1][key] <= 100.0
        assert val[-1]["top5_acc"] <= min(val[-1]["verb_top5_acc"], val[-1]["noun_top5_acc"]) + 1e-6   # action = verb AND noun
    else:
        assert "top1_err" in lines[-1] and 0.0 <= lines[-1]["top1_err"] <= 100.0
        assert model.model.head_cls.weight.grad is not None
    assert not any(p.requires_grad for p in model.model.head.parameters())


@pytest.mark.gpu
def test_non_finite_loss_never_reaches_the_weights_and_raises_at_the_log_point(tmp_path):
    """tools/train_net.py:174: `misc.check_nan_losses(loss)` raises before the optimiser step of the bad iteration.  The loop here
    syncs with the device once per LOG_PERIOD, so the bad step is dropped ON the device (optimizer.skip_flag) and the same error is
    raised at the next log point: the weights are those of the last good iteration."""
    from procedurevrl_amd import train_net as tn
    cfg = _cfg(tmp_path)
    cfg.SOLVER.MAX_EPOCH = 1
    cfg.LOG_PERIOD = 4
    cfg.SYNTHETIC.NUM_VIDEOS = 8                                   # 4 iterations: good, BAD, bad, bad -> raise at the log point
    real = tn.pretrain_loss
    calls = {"n": 0}
    snaps = {}

    def loss_fn(pred, teacher, mse, c):
        loss, l1, l2 = real(pred, teacher, mse, c)
        calls["n"] += 1
        if calls["n"] == 2:                                        # weights after the one good step
            snaps["w"] = {k: v.detach().clone() for k, v in snaps["model"].state_dict().items() if v.is_floating_point()}
        if calls["n"] >= 2:
            loss = loss * float("inf")
        return loss, l1, l2

    real_build = tn.build_model

    def build(c, *a, **k):
        snaps["model"] = real_build(c, *a, **k)
        return snaps["model"]

    tn.pretrain_loss, tn.build_model = loss_fn, build
    try:
        with pytest.raises(RuntimeError, match="Got NaN losses"):
            tn.train(cfg)
    finally:
        tn.pretrain_loss, tn.build_model = real, real_build
    assert calls["n"] == 4
    now = snaps["model"].state_dict()
    for k, v in snaps["w"].items():
        assert torch.equal(now[k], v), f"{k} was changed by an iteration whose loss was not finite"


@pytest.mark.gpu
def test_linear_probing_runs_the_encoder_in_eval_mode(tmp_path):
    """tools/train_net.py:72-85 (`TRAIN.LINEAR`, set by three of the four shipped COIN fine-tune configs): the wrapper trains, but
    `pos_drop` and `blocks` are put in eval mode -- DropPath off in the frozen encoder -- and only head / order parameters are
    optimised (lib/models/optimizer.py:21-34).  A LINEAR training step's features equal the eval-mode features (with DropPath 0.3
    they would not), no encoder parameter receives a gradient or moves, the classification head does."""
    from procedurevrl_amd import train_net as tn
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.datasets import construct_loader
    from procedurevrl_amd.distributed import GradReducer
    from procedurevrl_amd.optimizer import construct_optimizer
    cfg = _finetune_cfg(tmp_path, "kinetics")
    cfg.TRAIN.LINEAR = True
    cfg.MODEL.DROP_PATH = 0.3
    torch.manual_seed(0)
    model = build_model(cfg)
    vt = model.model
    with torch.no_grad():                                        # (the reference zero-initialises temporal_fc: wake that branch up)
        for blk in vt.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    opt = construct_optimizer(model, cfg)
    assert not any(p.requires_grad for n, p in model.named_parameters() if "head" not in n and "order" not in n)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    feats = []
    real = vt.forward_features

    def spy(x, *a, **k):
        f = real(x, *a, **k)
        feats.append((x.detach().clone(), f.detach().clone(), vt.training, vt.blocks.training))
        return f

    vt.forward_features = spy
    try:
        loader = construct_loader(cfg, "train")
        tn.train_epoch(loader, model, opt, GradReducer(vt, enabled=False), 0, cfg, max_iters=2)
    finally:
        vt.forward_features = real
    assert len(feats) == 2 and all(t and not bt for _, _, t, bt in feats), "wrapper in train mode, blocks in eval mode"
    model.eval()
    with torch.no_grad():
        for x, f, _, _ in feats:
            assert torch.equal(real(x), f), "a LINEAR step's features are the eval-mode features (DropPath off)"
    model.train()                                                # for contrast: with the blocks in train mode DropPath 0.3 changes them
    with torch.no_grad():
        assert not torch.equal(real(feats[0][0]), feats[0][1])
    for n, p in model.named_parameters():
        moved = not torch.equal(p.detach(), before[n])
        if "head" not in n and "order" not in n:
            assert p.grad is None and not moved, n
    assert not torch.equal(vt.head_cls.weight.detach(), before["model.head_cls.weight"])
