"""Host-side logic that needs no GPU: config surface, registry, state_dict keys, LR schedule, drop-path expansion,
optimizer grouping, gradient-store layout."""
import glob
import os

import pytest
import torch

from procedurevrl_amd.config import CfgNode, get_cfg

REF_CFG = "/root/reference/configs"


def test_cfg_defaults_and_overrides():
    cfg = get_cfg()
    assert cfg.DEV.TEMP == 0.02 and cfg.TIMESFORMER.DEPTH == 12 and cfg.TRAIN.TOPK == 5
    cfg.merge_from_list(["SOLVER.BASE_LR", "1e-4", "NUM_GPUS", "0", "MVIT.PATCH_KERNEL", "(3, 7, 7)"])
    assert cfg.SOLVER.BASE_LR == 1e-4 and cfg.NUM_GPUS == 0 and cfg.MVIT.PATCH_KERNEL == [3, 7, 7]
    with pytest.raises(KeyError):
        cfg.merge_from_list(["NO.SUCH_KEY", "1"])
    with pytest.raises(ValueError):
        cfg.merge_from_list(["NUM_GPUS", "'eight'"])
    again = CfgNode.load_cfg(cfg.dump())
    assert again.SOLVER.BASE_LR == 1e-4 and again.DEV.ORDER_PRETRAIN_MAX_LEN == 9
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.NUM_GPUS = 3


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference configs not present")
def test_reference_yaml_files_load_unchanged():
    files = sorted(glob.glob(os.path.join(REF_CFG, "*", "*.yaml")))
    assert len(files) == 8
    for f in files:
        cfg = get_cfg()
        cfg.merge_from_file(f)
        assert cfg.TIMESFORMER.ATTENTION_TYPE == "divided_space_time"
        assert isinstance(cfg.SOLVER.BASE_LR, float) and isinstance(cfg.SOLVER.WEIGHT_DECAY, float)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF_CFG, "HowTo100M", "procedurevrl_mvitv2_adamw.yaml"))
    assert cfg.MVIT.PATCH_KERNEL == [3, 7, 7] and cfg.MVIT.PATCH_STRIDE == [2, 4, 4]   # "(3, 7, 7)" strings literal_eval'ed


def _small_model(text=True):
    from procedurevrl_amd.build import MODEL_REGISTRY
    from procedurevrl_amd import vit  # noqa: F401
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "vit_base_patch16_224_develop"
    cfg.MODEL.PRETRAINED = False
    cfg.MODEL.NUM_CLASSES = 64
    cfg.MODEL.TEXT_MODEL = "clip_vit_b_16" if text else ""
    cfg.TIMESFORMER.DEPTH = 2
    cfg.DATA.TRAIN_CROP_SIZE = 32
    cfg.DEV.MATCH_LANG_EMB = True
    cfg.DEV.ORDER_PRETRAIN_ENABLED = True
    cfg.SYNTHETIC.TEXT_LAYERS = 2
    cfg.NUM_GPUS = 0
    cfg.TRAIN.LABEL_EMB = torch.randn(64, 512)
    return cfg, MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)


def test_state_dict_keys_equal_the_reference_model():
    f = torch.load(os.path.join(os.path.dirname(__file__), "golden", "e2e.pt"), weights_only=False)
    _, model = _small_model()
    assert sorted(model.state_dict().keys()) == f["state_keys"]
    # constructor facts of the reference: temporal_fc of every block and time_embed start at zero
    for blk in model.model.blocks:
        assert float(blk.temporal_fc.weight.abs().sum()) == 0.0
    assert float(model.model.time_embed.abs().sum()) == 0.0


def test_cpu_tensors_are_rejected_loudly():
    _, model = _small_model(text=False)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 9, 3, 8, 32, 32))


def test_grad_store_layout_and_optimizer_groups():
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    cfg, model = _small_model()
    vt = model.model
    gs = vt.grad_store()
    assert gs.flat.numel() >= sum(p.numel() for p in vt.parameters() if p.requires_grad)
    assert all(o % 64 == 0 for o in gs.offsets)
    assert not any(n.startswith("text_model") for n in gs.names)           # frozen teacher is not in the flat buffer
    tgt, beta = gs.target(vt.norm.weight)
    assert beta == 0.0 and vt.norm.weight.grad.data_ptr() == tgt.data_ptr()
    assert gs.target(vt.norm.weight)[1] == 1.0                               # second touch accumulates
    cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
    opt = construct_optimizer(model, cfg)
    assert len(opt.param_groups) == 2 and len(opt.param_groups[0]["params"]) == 0       # [bn (empty), non-bn + text]
    n_all = len(list(model.parameters()))
    assert len(opt.param_groups[1]["params"]) == n_all
    set_lr(opt, 0.25)
    assert all(g["lr"] == 0.25 * g["lr_mult"] for g in opt.param_groups)


def test_droppath_expansion_matches_reference_granularity():
    from procedurevrl_amd.engine import EncoderEngine
    B, N, T = 2, 3, 4
    s1 = torch.arange(B * N, dtype=torch.float32)
    s2 = torch.arange(B * T, dtype=torch.float32) + 100
    s3 = torch.arange(B, dtype=torch.float32) + 1000
    d = EncoderEngine.expand_droppath(s1, s2, s3, B, N, T)
    for b in range(B):
        for n in range(N):
            for t in range(T):
                r = (b * N + n) * T + t
                assert d["s1_tok"][r] == s1[b * N + n]          # temporal branch: per (b h w) row  (vit.py:132)
                assert d["s2_tok"][r] == s2[b * T + t]          # spatial branch: per (b t) row     (vit.py:144)
                assert d["s3_all"][r] == s3[b]                  # mlp: per b                        (vit.py:157)
        assert d["s3_all"][B * N * T + b] == s3[b]


def test_order_transformer_draws_follow_reference_ranges():
    _, model = _small_model()
    ot = model.model.order_tfm
    torch.manual_seed(0)
    for _ in range(50):
        d = ot.draw(16, "cpu")
        m, p = d["mask_inds"], d["pad_start"]
        assert int(m.min()) >= 0 and int(m.max()) <= 8
        last = m + 1 == 9
        assert torch.all(p[last] == 9)                           # mask at the last position: nothing padded
        assert torch.all((p[~last] > m[~last]) & (p[~last] <= 8))  # randint(mask + 1, max_len)


def test_test_meter_multi_view_ensemble():
    """TestMeter semantics of the reference (lib/utils/meters.py:90-128,164-203): clip predictions of a video are summed
    (or max-ed), labels recorded per video, top-k accuracy on the video level."""
    from procedurevrl_amd.test_net import TestMeter
    g = torch.Generator().manual_seed(0)
    num_videos, num_clips, K = 6, 4, 10
    preds = torch.rand(num_videos * num_clips, K, generator=g)
    labels_v = torch.randint(0, K, (num_videos,), generator=g)
    clip_ids = torch.randperm(num_videos * num_clips, generator=g)
    labels = labels_v[clip_ids // num_clips]
    for method in ("sum", "max"):
        m = TestMeter(num_videos, num_clips, K, ensemble_method=method)
        for lo in range(0, len(clip_ids), 5):                      # ragged batches
            sl = slice(lo, lo + 5)
            m.update_stats(preds[clip_ids][sl], labels[sl], clip_ids[sl])
        ref = torch.zeros(num_videos, K)
        for ind in range(len(clip_ids)):                           # the reference's per-clip loop
            v = int(clip_ids[ind]) // num_clips
            p = preds[clip_ids][ind]
            ref[v] = ref[v] + p if method == "sum" else torch.max(ref[v], p)
        assert torch.allclose(m.video_preds, ref, atol=1e-6)
        assert torch.equal(m.video_labels, labels_v) and torch.all(m.clip_count == num_clips)
        stats = m.finalize_metrics(ks=(1, 5))
        top1 = (ref.argmax(1) == labels_v).float().mean().item() * 100
        assert stats["top1_acc"] == "{:.2f}".format(top1)


def _mvit_cfg_from_golden(name, frames, crop):
    import os
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", name + ".pt"), weights_only=False)
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "MViT"
    cfg.MODEL.PRETRAINED = False
    cfg.DATA.NUM_FRAMES = frames
    cfg.DATA.TRAIN_CROP_SIZE = cfg.DATA.TEST_CROP_SIZE = crop
    cfg.DATA.INPUT_CHANNEL_NUM = [3]
    cfg.DEV.MATCH_LANG_EMB = True
    cfg.NUM_GPUS = 0
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(16, 512, seed=1)
    for k, v in g["mvit"].items():
        setattr(cfg.MVIT, k, v)
    return cfg, g


def test_mvit_module_tree_matches_reference_shapes():
    """MViTv2-S (16 x 224^2): every parameter name and shape of the reference MViT_encoder (tests/golden/mvit_s.pt),
    under the reference wrapper's prefix `model.video_encoder.`; geometry of the four stages."""
    from procedurevrl_amd.build import MODEL_REGISTRY
    from procedurevrl_amd import mvit  # noqa: F401
    cfg, g = _mvit_cfg_from_golden("mvit_s", 16, 224)
    model = MODEL_REGISTRY.get("MViT")(cfg)
    sd = model.state_dict()
    got = {k[len("model.video_encoder."):]: tuple(v.shape) for k, v in sd.items() if k.startswith("model.video_encoder.")}
    assert got == {k: tuple(v) for k, v in g["shapes"].items()}
    assert {"model.head.weight", "model.head.bias"} <= set(sd)
    plan = model.model.video_encoder.plan
    assert [(b["dim"], b["dim_out"], b["heads"]) for b in plan][:4] == [(96, 96, 1), (96, 192, 2), (192, 192, 2), (192, 384, 4)]
    assert plan[0]["in_thw"] == [8, 56, 56] and plan[0]["stride_kv"] == [1, 8, 8]
    assert plan[1]["stride_q"] == [1, 2, 2] and plan[1]["stride_kv"] == [1, 4, 4]
    assert plan[14]["in_thw"] == [8, 14, 14] and plan[15]["in_thw"] == [8, 7, 7] and plan[15]["dim_out"] == 768
    import math
    assert sum(p.numel() for p in model.model.video_encoder.parameters()) == sum(math.prod(v) for v in g["shapes"].values()) == 34230144   # 34.23 M


def test_mvit_unsupported_settings_raise():
    from procedurevrl_amd import mvit
    cfg, _ = _mvit_cfg_from_golden("mvit_small", 4, 64)
    cfg.MVIT.MODE = "max"
    with pytest.raises(NotImplementedError):
        mvit.MViT(cfg)
    cfg, _ = _mvit_cfg_from_golden("mvit_small", 4, 64)
    cfg.MVIT.RESIDUAL_POOLING = False
    with pytest.raises(NotImplementedError):
        mvit.MViT(cfg)


def test_load_pretrained_matches_reference_loader(tmp_path):
    """checkpoint.load_pretrained vs the reference's lib/models/helpers.py:load_pretrained (tests/golden/pretrained.pt: the
    reference loader run on the reference model with an ImageNet-ViT-shaped checkpoint): the same 24 tensors change and
    end up with the same contents -- classifier dropped (1000 != 512 rows), pos_embed resized 196 -> 49 patches
    (nearest), attn / norm1 cloned into temporal_attn / temporal_norm1."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import imagenet_vit_shapes, tensor_stats
    from oracle import timesformer_oracle as orc
    from procedurevrl_amd.build import MODEL_REGISTRY
    from procedurevrl_amd import vit  # noqa: F401
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.datasets import synthetic_label_emb
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "pretrained.pt"), weights_only=False)
    ck = tmp_path / "jx_vit_base_p16_224.pth"
    torch.save(orc.seeded_state(imagenet_vit_shapes(g["depth"]), g["seed"]), ck)
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "vit_base_patch16_224_develop"
    cfg.MODEL.PRETRAINED = False
    cfg.MODEL.NUM_CLASSES = g["K"]
    cfg.TIMESFORMER.DEPTH = g["depth"]
    cfg.DATA.TRAIN_CROP_SIZE = g["crop"]
    cfg.DEV.MATCH_LANG_EMB = True
    cfg.NUM_GPUS = 0
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(g["K"], 512, seed=1)
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    inner = model.model
    before = {k: v.clone() for k, v in inner.state_dict().items()}
    cfg.TIMESFORMER.PRETRAINED_MODEL = str(ck)
    from procedurevrl_amd.checkpoint import load_pretrained
    load_pretrained(inner, cfg)
    after = inner.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    assert changed == g["changed"]
    for k in changed:
        got, ref = tensor_stats(after[k]), g["stats"][k]
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, ref)), (k, got, ref)


def test_mvit_image_checkpoint_conversion_matches_reference_loader(tmp_path):
    """checkpoint.load_pretrained_mvit vs the reference's helpers.load_pretrained on its MViT wrapper with an image-MViTv2
    shaped checkpoint (tests/golden/mvit_pretrained.pt): conv weights repeated over time, rel-pos tables interpolated,
    `video_encoder.` prefix -- the same 101 tensors change and end up with the same contents."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import mvit_image_state_shapes, tensor_stats
    from oracle import mvit_oracle as mo
    from oracle import timesformer_oracle as orc
    from procedurevrl_amd import mvit
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "mvit_pretrained.pt"), weights_only=False)
    c = g["cfg"]
    cfg, _ = _mvit_cfg_from_golden("mvit_small", c["frames"], c["crop"])
    model = mvit.MViT(cfg)
    inner = model.model
    fake = orc.seeded_state(mvit_image_state_shapes(mo.encoder_shapes(g["mvit"], c["frames"], c["crop"])), g["seed"])
    ck = tmp_path / "MViTv2_S_in1k.pyth"
    torch.save({"model_state": fake}, ck)
    before = {k: v.clone() for k, v in inner.state_dict().items()}
    cfg.TIMESFORMER.PRETRAINED_MODEL = str(ck)
    from procedurevrl_amd.checkpoint import load_pretrained_mvit
    load_pretrained_mvit(inner, cfg)
    after = inner.state_dict()
    changed = sorted(k for k in after if not torch.equal(after[k], before[k]))
    assert changed == g["changed"]
    for k in changed:
        got, ref = tensor_stats(after[k]), g["stats"][k]
        assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(got, ref)), (k, got, ref)


def test_spatial_sampling_params_properties():
    """The host mirror of utils.spatial_sampling: for arbitrary frame sizes the crop window lies inside the rescaled
    frame, the short side equals the drawn size, test-mode crops are the three uniform positions."""
    import numpy as np
    from hypothesis import given, settings, strategies as st
    from procedurevrl_amd.transform import spatial_sampling_params

    @settings(max_examples=200, deadline=None)
    @given(st.integers(120, 720), st.integers(120, 720), st.integers(0, 2 ** 31 - 1), st.sampled_from([-1, 0, 1, 2]), st.booleans())
    def prop(h, w, seed, sidx, inv):
        np.random.seed(seed)
        mn, mx, crop = (256, 320, 224) if sidx == -1 else (224, 224, 224)
        nh, nw, yo, xo, flip = spatial_sampling_params(h, w, sidx, mn, mx, crop, True, inv)
        assert min(nh, nw) >= crop and mn <= min(nh, nw) <= mx
        assert 0 <= yo <= nh - crop and 0 <= xo <= nw - crop
        assert flip in (0, 1) and (sidx == -1 or flip == 0)
        if sidx != -1:
            if nh > nw:
                assert yo == {0: 0, 1: int(np.ceil((nh - crop) / 2)), 2: nh - crop}[sidx]
            else:
                assert xo == {0: 0, 1: int(np.ceil((nw - crop) / 2)), 2: nw - crop}[sidx]
    prop()


def test_tn_split_plan_properties():
    """pvrl_gemm_tn_plan_splits (a pure host function of the C ABI, callable without a GPU): one round of <= 256 workgroups for
    the 256x256 kernel (N * K >= 256 * 256), a multiple of 8 slices for the 128x128 kernel, never more slices than 64-row blocks."""
    from hypothesis import given, settings, strategies as st
    from procedurevrl_amd._lib import lib
    L = lib()

    @settings(max_examples=300, deadline=None)
    @given(st.integers(1, 200000), st.integers(1, 24), st.integers(1, 24))
    def prop(M, n, k):
        N, K = 128 * n, 128 * k
        s = L.call("pvrl_gemm_tn_plan_splits", M, N, K)
        assert s >= 1
        if N * K >= 256 * 256:                       # 256x256 8-wave kernel; half tiles count as tiles
            tiles = -(-N // 256) * -(-K // 256)
            assert s == 1 or (s * tiles <= 256 and s <= max(1, M // 64))
        else:
            assert s % 8 == 0 and (s == 8 or M // s >= 256)
        assert L.call("pvrl_gemm_tn_workspace_bytes", N, K, s) == s * (N * K + N) * 4 + 256
    prop()


def test_val_meter_epoch_stats_and_eval_schedule():
    """ValMeter (lib/utils/meters.py:420-580) and is_eval_epoch (lib/utils/misc.py:189-210)"""
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd.train_net import ValMeter, is_eval_epoch
    cfg = get_cfg()
    cfg.LOG_PERIOD, cfg.SOLVER.MAX_EPOCH, cfg.TRAIN.EVAL_PERIOD = 2, 7, 3
    m = ValMeter(3, cfg)
    m.update_stats(50.0, 10.0, 4)
    m.update_stats(100.0, 20.0, 2)
    s = m.log_epoch_stats(0)
    assert abs(s["top1_err"] - (50.0 * 4 + 100.0 * 2) / 6) < 1e-9 and abs(s["top5_err"] - (10.0 * 4 + 20.0 * 2) / 6) < 1e-9
    assert s["min_top1_err"] == s["top1_err"] and s["_type"] == "val_epoch" and s["epoch"] == "1/7"
    m.reset()
    m.update_stats(80.0, 5.0, 1)
    s2 = m.log_epoch_stats(1)
    assert s2["min_top1_err"] == s["top1_err"] and s2["min_top5_err"] == 5.0      # minima persist across reset()
    assert [e for e in range(7) if is_eval_epoch(cfg, e)] == [2, 5, 6]


def test_finetune_metrics_and_losses_match_their_definitions():
    """The fine-tuning branch of the train loop (tools/train_net.py:126-136,163-169,195-222): multi-task (EPIC action) top-k
    accuracy = both labels within their task's top k, by brute force; `smooth` = timm's label-smoothing cross entropy;
    the EPIC loss is the mean of the verb and the noun cross entropy."""
    import torch.nn.functional as F
    from procedurevrl_amd import train_net as tn
    g = torch.Generator().manual_seed(3)
    pv, pn = torch.randn(33, 97, generator=g), torch.randn(33, 300, generator=g)
    lv, ln = torch.randint(0, 97, (33,), generator=g), torch.randint(0, 300, (33,), generator=g)
    pv[torch.arange(0, 33, 3), lv[::3]] += 6.0                       # make a share of the samples right
    pn[torch.arange(0, 33, 2), ln[::2]] += 6.0
    a1, a5 = tn.multitask_topk_accuracies((pv, pn), (lv, ln), (1, 5))
    for k, got in ((1, a1), (5, a5)):
        hit = 0
        for b in range(33):
            hit += int(lv[b] in pv[b].topk(k).indices and ln[b] in pn[b].topk(k).indices)
        assert abs(float(got) - 100.0 * hit / 33) < 1e-4
    v1, = tn.topk_accuracies(pv, lv, (1,))
    assert abs(float(v1) - 100.0 * float((pv.argmax(1) == lv).float().mean())) < 1e-4
    x, t = torch.randn(9, 11, generator=g), torch.randint(0, 11, (9,), generator=g)
    ref = F.cross_entropy(x, t, label_smoothing=0.0) * 0.8 + 0.2 * (-F.log_softmax(x, -1).mean(-1)).mean()
    assert torch.allclose(tn.LabelSmoothingCrossEntropy(0.2)(x, t), ref, atol=1e-6)
    cfg = get_cfg()
    cfg.TRAIN.DATASET = "Epickitchens"
    loss, (l_v, l_n) = tn.finetune_loss((pv, pn), {"verb": lv, "noun": ln}, cfg)
    assert torch.allclose(loss, 0.5 * (F.cross_entropy(pv, lv) + F.cross_entropy(pn, ln)), atol=1e-6)
    cfg.TRAIN.DATASET = "kinetics"
    loss, none = tn.finetune_loss(pv, lv, cfg)
    assert none is None and torch.allclose(loss, F.cross_entropy(pv, lv), atol=1e-6)
    assert not tn.is_pretraining(cfg)
    cfg.MIXUP.ENABLED = True
    with pytest.raises(NotImplementedError):
        tn.finetune_loss(pv, lv, cfg)


def test_hook_groups_cover_every_block_once_and_split_the_tail():
    """engine.GraphReplay._group_of: the data-parallel gradient hook runs per group of `hook_group` blocks (one merged collective each),
    and per block for the backward's LAST group (blocks 0 .. hook_group - 1: nothing is left to hide its collective under).  Whatever
    the depth and group size, walking the blocks from the last to the first must visit every block exactly once, in order."""
    from procedurevrl_amd.engine import GraphReplay

    class E(GraphReplay):
        pass

    for nb in (1, 2, 3, 12, 16):
        for grp in (1, 2, 3, 5):
            for split in (True, False):
                e = E()
                e.hook_group, e.hook_tail_split = grp, split
                seen, i, groups = [], nb - 1, []
                while i >= 0:
                    lo, hi = e._group_of(i, nb)
                    assert lo <= i <= hi < nb and e._group_of(lo, nb) == (lo, hi)
                    if i == hi:
                        groups.append((lo, hi))
                    seen.append(i)
                    i -= 1
                assert seen == list(range(nb - 1, -1, -1))
                covered = [b for lo, hi in groups for b in range(hi, lo - 1, -1)]
                assert covered == list(range(nb - 1, -1, -1)), (nb, grp, split, groups)
                if split and nb > grp:
                    assert groups[-min(grp, nb):] == [(b, b) for b in range(min(grp, nb) - 1, -1, -1)]
                if not split and grp > 1 and nb >= grp:
                    assert groups[-1] == (0, grp - 1)
