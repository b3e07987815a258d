"""TimeSformer ViT-B video encoder + step-matching head, MI355X-native.

Drop-in for the reference `lib/models/vit.py`: same class names (Mlp :44, Attention :62, Block :94,
PatchEmbed :160, VisionTransformer :183, vit_base_patch16_224_develop :473), same constructor
arguments read from `cfg`, same `state_dict()` keys, same call signatures and outputs
(train: `model([inputs, meta]) -> (pred, teacher_pred, [mse_target, mse_pred])`, vit.py:325-352; eval:
softmax probabilities, vit.py:355-356).  The sub-modules own parameters only; the arithmetic is the
kernel schedule of `engine.EncoderEngine` / `tfm_engine.StackEngine` over libpvrl_hip.so.  There is
no PyTorch fallback: calling the model on CPU tensors raises.
"""
import math
import os
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .build import MODEL_REGISTRY
from .engine import EncoderEngine, GradStore
from .functional import EncoderFn, kl_topk_loss, l2norm, linear_f32, mse_loss, step_logits
from .head_engine import PretrainHeadEngine, PretrainHeadFn
from .tfm_model import ClipTextModel, DiffusionTransformer as OrderTransformer


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """Truncated normal init (lib/models/vit_utils.py:59-80 semantics: cut at absolute [a, b])."""
    with torch.no_grad():
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_path=0.1,
                 norm_layer=nn.LayerNorm, attention_type="divided_space_time"):
        super().__init__()
        if attention_type != "divided_space_time":
            raise NotImplementedError(
                f"TIMESFORMER.ATTENTION_TYPE={attention_type!r}: only 'divided_space_time' (every shipped config, "
                "vit.py:124-127) is built on the HIP path")
        self.attention_type = attention_type
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.temporal_norm1 = norm_layer(dim)
        self.temporal_attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.temporal_fc = nn.Linear(dim, dim)
        self.drop_path_rate = drop_path
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.1, norm_layer=nn.LayerNorm, num_frames=8, attention_type="divided_space_time",
                 label_emb="", mlp=0, text_model="", lp=False, num_seg=0, extra_tr="order", drope=0.0, cfg=None):
        super().__init__()
        assert patch_size == 16 and in_chans == 3, "patchify kernel is built for 16x16 RGB patches"
        assert drop_rate == 0.0 and attn_drop_rate == 0.0, "dropout is 0 in every shipped config (vit.py:489)"
        self.cfg = cfg
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.temp = cfg.DEV.TEMP
        self.order_pretrain = cfg.DEV.ORDER_PRETRAIN_ENABLED
        self.order_max_len = cfg.DEV.ORDER_PRETRAIN_MAX_LEN
        self.order_tfm_layers = cfg.DEV.ORDER_TFM_LAYERS
        self.order_recog_batch = cfg.DEV.ORDER_RECOG_BATCH
        self.attention_type = attention_type
        self.depth = depth
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.time_embed = nn.Parameter(torch.zeros(1, num_frames, embed_dim))
        # vit.py:209,214: identity at p = 0 (asserted above); kept as attributes because the linear-probing loop reaches for them
        # (`model.model.pos_drop.eval()`, tools/train_net.py:72-85)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.time_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]   # vit.py:220
        self.drop_path_rates = dpr
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop_path=dpr[i], norm_layer=norm_layer, attention_type=attention_type) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.ln_eps = self.norm.eps

        self._init_heads(embed_dim, label_emb, mlp, text_model, num_seg, num_classes, cfg)
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        # vit.py:272-281 zeroes temporal_fc of EVERY block (the ModuleList itself consumes i == 0)
        if depth != 0 and attention_type == "divided_space_time":
            for blk in self.blocks:
                nn.init.constant_(blk.temporal_fc.weight, 0)
                nn.init.constant_(blk.temporal_fc.bias, 0)

        self.engine = EncoderEngine(self)
        self.weight_cache = self.engine._weight
        self._grad_store = None
        self._label_cache = None
        if hasattr(self, "order_tfm"):
            self.order_tfm.bind(self)
        if hasattr(self, "text_model"):
            self.text_model.bind(self)

    head_engine = None      # PretrainHeadEngine, created on the first pre-training forward

    def _init_heads(self, embed_dim, label_emb, mlp, text_model, num_seg, num_classes, cfg):
        """Projection head, step embeddings, order transformer and frozen text tower (vit.py:228-267; the MViT wrapper
        lib/models/mvit.py:72-107 is line-for-line the same block)."""
        self.mlp = mlp
        self.label = label_emb
        if not (isinstance(label_emb, str) and label_emb == ""):   # pre-training (vit.py:231-237)
            self.label_emb = torch.load(label_emb) if isinstance(label_emb, str) else label_emb
            self.head = nn.Linear(embed_dim, self.label_emb.shape[1])
            self.order_tfm = OrderTransformer(num_seg=self.order_max_len - 1, tfm_layers=self.order_tfm_layers,
                                              dropout=cfg.MODEL.DROP_E, hidden_size=self.head.weight.shape[0], cfg=cfg)
        else:                 # fine-tuning / zero-shot (vit.py:238-255)
            emb = torch.load(cfg.DEV.TEST_LANG_EMB) if isinstance(cfg.DEV.TEST_LANG_EMB, str) else cfg.DEV.TEST_LANG_EMB
            if cfg.DEV.MATCH_LANG_EMB:
                self.label_emb = emb
                self.head = nn.Linear(embed_dim, emb.shape[1])
                for p in self.head.parameters():
                    p.requires_grad = False
            else:
                self.label_emb = False
                self.test_lang_emb = emb
                self.head = nn.Linear(embed_dim, emb.shape[1])
                for p in self.head.parameters():
                    p.requires_grad = False
                if cfg.TRAIN.DATASET == "Epickitchens":
                    self.head_n = nn.Linear(emb.shape[1], 300)
                    self.head_v = nn.Linear(emb.shape[1], 97)
                else:
                    self.head_cls = nn.Linear(emb.shape[1], num_classes)
            self.apply(self._init_weights)

        self.text = text_model
        if text_model == "clip_vit_b_16":
            layers = int(getattr(getattr(cfg, "SYNTHETIC", None), "TEXT_LAYERS", 12)) if cfg is not None else 12
            self.text_model = ClipTextModel(layers=layers).float()
            for p in self.text_model.parameters():
                p.requires_grad = False
        if num_seg > 0:
            self.num_seg = num_seg
            self.order_tfm = OrderTransformer(num_seg=num_seg, tfm_layers=self.order_tfm_layers, dropout=cfg.MODEL.DROP_E,
                                              hidden_size=self.head.weight.shape[0], cfg=cfg)

    # ----------------------------------------------------------------------------- plumbing
    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "time_embed"}

    def encoder_named_parameters(self):
        return list(self.named_parameters())

    def grad_store(self):
        dev = self.cls_token.device
        if self._grad_store is None or self._grad_store.flat.device != dev:
            named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
            self._grad_store = GradStore(named, dev)
            self.engine._grads = self._grad_store
        return self._grad_store

    def grad_target(self, p):
        return self.grad_store().target(p)

    def anchor(self):
        return self.cls_token

    def adopt_grads(self, keep_none=False, zero_unused=True):
        """Move gradients that autograd allocated itself (head, small embeddings) into the flat buffer so
        that every trainable parameter's .grad is a view of one allocation (all-reduce / fused optimiser).
        A parameter without a gradient gets a zeroed view; with `keep_none` its .grad stays None (the optimiser then
        skips it, as torch.optim does) and only the buffer is zeroed -- unless `zero_unused` is off (the optimiser's own
        call: it never reads the slot of a parameter it skips, and one fill kernel per unused parameter -- 54 for the order
        transformer in the contrastive-only step -- is 0.3 ms of 5-us launches per step)."""
        gs = self.grad_store()
        for p, v in zip(gs.params, gs.views):
            if p.grad is None:
                if zero_unused:
                    v.zero_()
                if not keep_none:
                    p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        return gs

    def _labels(self, device):
        """check_device_norm (vit.py:435-440): the step embeddings are moved to the model's device and
        row-normalised on first use; a transposed copy is kept for the logits backward."""
        c = self._label_cache
        if c is None or c[0].device != device:
            le = self.label_emb.to(device=device, dtype=torch.float32)
            le = le / le.norm(dim=1, keepdim=True)
            c = (le.contiguous(), le.t().contiguous())
            self._label_cache = c
        return c

    # ----------------------------------------------------------------------------- forward
    def forward_features(self, x, cls=True, droppath=None):
        """x fp32 [B, 3, T, H, W] -> [B, embed_dim] (vit.py:365-423 with cls=True)."""
        if not x.is_cuda:
            raise RuntimeError("procedurevrl_amd runs on the HIP path only: move the model and inputs to the GPU "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        if not cls:
            raise NotImplementedError("forward_features(cls=False) has no caller on the hot path")
        if self.training and torch.is_grad_enabled():
            self.grad_store()
        return EncoderFn.apply(self.cls_token, x.float(), self, droppath)

    def get_pseudo_labels(self, device, text):
        """vit.py:425-433"""
        vis_emb = text["clip_vis_feat"]
        ids = text["clip_text_ids"]
        if ids.dim() == 3:
            ids = ids.squeeze(dim=1)
        text_emb = self.text_model.encode_text(ids)
        text_emb = (text_emb + vis_emb.float()) / 2.0
        le, le_t = self._labels(device)
        text_emb = l2norm(text_emb.contiguous())
        return step_logits(text_emb, le, le_t, self.temp)

    def get_mask_samples(self, all_samples, mask_inds):
        L = self.order_max_len
        s = all_samples.view(all_samples.shape[0] // L, L, -1)
        return s[torch.arange(s.shape[0], device=s.device), mask_inds, :]

    def _pretrain_head(self, feat, teacher_x, rng, batch_size):
        """Everything between the encoder's features and the loss in the pre-training forward (vit.py:298-352):
        projection head, L2 norm, step logits, order / diffusion transformer, output assembly.
        -> (pred [13b, K], teacher [13b, K], [mse_target, mse_pred])"""
        dev = feat.device
        le, le_t = self._labels(dev)
        x = linear_f32(feat, self.head.weight, self.head.bias)
        video_emb = l2norm(x)
        self.last_video_emb = video_emb
        x = step_logits(video_emb, le, le_t, self.temp)
        pred_video_emb, mask_inds, mse, intermediate = self.order_tfm(video_emb, is_pretrain=True,
                                                                     rng=(rng or {}).get("order"))
        # (vit.py:331-334 also computes `mask_pred` from pred_video_emb; it is never used and is skipped)
        masked_teacher_x = self.get_mask_samples(teacher_x, mask_inds)
        intermediate = l2norm(intermediate.contiguous())
        intermediate_pred = step_logits(intermediate, le, le_t, self.temp)
        lv = self.order_tfm.level_batch
        intermediate_teacher_x = masked_teacher_x.unsqueeze(0).expand(lv, -1, -1).reshape(-1, masked_teacher_x.size(-1))
        n_keep = batch_size * self.order_recog_batch
        rand_inds = (rng or {}).get("rand_inds")
        if rand_inds is None:       # vit.py:345 torch.randperm; argsort of uniforms is the same distribution, sync-free and
            rand_inds = torch.rand(x.shape[0], device=dev).argsort()        # capturable in a HIP graph
        rand_inds = rand_inds.to(dev)[:n_keep]
        x = torch.cat((x.index_select(0, rand_inds), intermediate_pred), dim=0)      # (index_select: see tfm_model.forward_pretrain)
        teacher_x = torch.cat((teacher_x.index_select(0, rand_inds), intermediate_teacher_x), dim=0)
        return x, teacher_x, mse

    _text_side = None      # HIP stream of the frozen text tower

    @staticmethod
    def _text_overlap():
        v = os.environ.get("PVRL_TEXT_OVERLAP", "head")
        return "head" if v == "1" else v

    def _teacher_begin(self, text, dev):
        """The frozen CLIP-text teacher (vit.py:425-433) depends on the narrations only: its ~110 small launches (one HIP-graph replay)
        are issued on a side stream and the main stream joins where the teacher logits are first read (the output assembly behind the
        head, `_pretrain_forward`).  Serially they sat between the encoder and the head on one queue (~1 ms per step with the GPU
        nearly idle).  Where the side stream starts (PVRL_TEXT_OVERLAP): "head" (default) = behind the encoder forward, so the tower
        runs under the pre-training head's ~300 launch-sized kernels, which leave the chip almost empty; "encoder" = before the
        encoder forward (round 5's first form: the tower then shares the chip with the persistent GEMMs for 4.9 ms and stretches
        them by ~14 %, profiles/r5_timeline_full.txt); "0": no side stream."""
        if dev.type != "cuda" or self._text_overlap() == "0":
            return None
        if self._text_side is None or self._text_side.device != dev:
            self._text_side = torch.cuda.Stream(device=dev)
        side = self._text_side
        side.wait_stream(torch.cuda.current_stream())       # the ids are ready -- and everything that read last step's teacher logits
        with torch.cuda.stream(side):                        # is done before their memory (this stream's pool) is written again
            teacher_x = self.get_pseudo_labels(dev, text)
        return teacher_x, side.record_event()

    def _pretrain_forward(self, feat, text, rng, batch_size, teacher=None):
        engine = os.environ.get("PVRL_HEAD_ENGINE", "1") == "1"
        if teacher is None and feat.is_cuda and self._text_overlap() == "head" and engine:
            teacher = self._teacher_begin(text, feat.device)                 # side stream, from here: under the head's forward

        def teacher_logits():
            if teacher is not None:
                teacher_x, ev = teacher
                torch.cuda.current_stream().wait_event(ev)
                return teacher_x
            return self.get_pseudo_labels(feat.device, text)                 # frozen text tower: its own HIP graph
        if not engine:
            return self._pretrain_head(feat, teacher_logits(), rng, batch_size)     # the same head wired through torch.autograd (eager)
        # One autograd node with a hand-written backward (head_engine.PretrainHeadEngine): its ~1,000 small launches are
        # replayed from two HIP graphs.  (Capturing the autograd-wired head crashes hipStreamEndCapture on ROCm 7.x: an
        # AccumulateGrad node is bound to the default stream -- round 2)
        if self.head_engine is None:
            self.head_engine = PretrainHeadEngine(self)
        dr = self.head_engine.draws(rng, batch_size, feat.shape[0], feat.device)
        pred, x0_rep, inter, rows, ri = PretrainHeadFn.apply(self.anchor(), feat, self, dr)
        teacher_out = self.head_engine.assemble_teacher(teacher_logits(), rows, ri)
        return pred, teacher_out, [x0_rep, inter]

    def forward(self, x, rng=None):
        """`rng` (optional) pins the random draws of the pre-training forward:
        dict(order=<DiffusionTransformer.draw()>, rand_inds=<permutation>, droppath=<per-block dicts>)."""
        text = None
        if len(self.text) > 0 and self.training:
            x, text = x
        batch_size = x.shape[0]
        if self.order_pretrain:
            x = x.reshape((-1,) + tuple(x.shape[2:]))                     # 'b m c t h w -> (b m) c t h w'
        elif hasattr(self, "num_seg") and self.num_seg > 0:
            b, c, mt, h, w = x.shape
            t = mt // self.num_seg
            x = x.view(b, c, self.num_seg, t, h, w).permute(0, 2, 1, 3, 4, 5).reshape(b * self.num_seg, c, t, h, w)
        pretrain = isinstance(self.label_emb, torch.Tensor) and len(self.text) > 0 and self.training
        if pretrain and (not self.cfg.DEV.MATCH_LANG_EMB or (hasattr(self, "num_seg") and self.num_seg > 0)):
            raise NotImplementedError("pre-training forward (vit.py:325-352) is built for DEV.MATCH_LANG_EMB True and "
                                      "MODEL.NUM_SEG 0, the setting of every shipped pre-training config")
        teacher = None
        if pretrain and isinstance(x, torch.Tensor) and (self._text_overlap() == "encoder" or os.environ.get("PVRL_HEAD_ENGINE", "1") != "1"):
            teacher = self._teacher_begin(text, x.device)
        x = feat = self.forward_features(x.contiguous(), droppath=(rng or {}).get("droppath"))
        dev = x.device
        if pretrain:
            return self._pretrain_forward(feat, text, rng, batch_size, teacher)

        if self.cfg.DEV.MATCH_LANG_EMB:
            le, le_t = self._labels(dev)
            x = linear_f32(x, self.head.weight, self.head.bias)
            x = l2norm(x)
            video_emb = x
            self.last_video_emb = video_emb        # exposed for the all-gather contrastive loss (MILNCELoss)
            if hasattr(self, "num_seg") and self.num_seg > 0:
                x = self.order_tfm(video_emb)
                x = l2norm(x.contiguous())
            x = step_logits(x, le, le_t, self.temp)
        else:
            if hasattr(self, "num_seg") and self.num_seg > 0:
                x = linear_f32(x, self.head.weight, self.head.bias)
                video_emb = l2norm(x)
                x = self.order_tfm(video_emb)
                x = linear_f32(x.contiguous(), self.head_cls.weight, self.head_cls.bias)
            else:
                x = linear_f32(x, self.head.weight, self.head.bias)
                x = l2norm(x)
                if hasattr(self, "head_n"):
                    v = linear_f32(x, self.head_v.weight, self.head_v.bias) / self.temp
                    n = linear_f32(x, self.head_n.weight, self.head_n.bias) / self.temp
                    return (v, n)
                x = linear_f32(x, self.head_cls.weight, self.head_cls.bias) / self.temp

        if not self.training:
            x = ops.softmax_rows(x.contiguous())
        return x


@MODEL_REGISTRY.register()
class vit_base_patch16_224_develop(nn.Module):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.pretrained = cfg.MODEL.PRETRAINED
        self.model = VisionTransformer(
            img_size=cfg.DATA.TRAIN_CROP_SIZE, num_classes=cfg.MODEL.NUM_CLASSES, patch_size=16, embed_dim=768,
            depth=cfg.TIMESFORMER.DEPTH, num_heads=12, mlp_ratio=4, qkv_bias=True,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_rate=0.0, attn_drop_rate=0.0,
            drop_path_rate=cfg.MODEL.DROP_PATH, num_frames=cfg.DATA.NUM_FRAMES,
            attention_type=cfg.TIMESFORMER.ATTENTION_TYPE, label_emb=cfg.TRAIN.LABEL_EMB, mlp=cfg.MODEL.MLP,
            text_model=cfg.MODEL.TEXT_MODEL, lp=cfg.MODEL.TEXT_LP, num_seg=cfg.MODEL.NUM_SEG, extra_tr=cfg.MODEL.EXTRA_TR,
            drope=cfg.MODEL.DROP_E, cfg=cfg, **kwargs)
        self.attention_type = cfg.TIMESFORMER.ATTENTION_TYPE
        self.num_patches = (cfg.DATA.TRAIN_CROP_SIZE // 16) ** 2
        if self.pretrained:
            from .checkpoint import load_pretrained
            load_pretrained(self.model, cfg)
        else:
            print("not loading any pretrained weights!")

    def forward(self, x, rng=None):
        return self.model(x) if rng is None else self.model(x, rng=rng)


def pretrain_loss(pred, teacher_pred, mse, cfg):
    """The loss block of tools/train_net.py:152-162 for MODEL.LOSS_FUNC == 'kldiv'."""
    loss1 = kl_topk_loss(pred, teacher_pred, int(cfg.TRAIN.TOPK))
    loss2 = mse_loss(mse[0].contiguous(), mse[1].contiguous())
    return loss1 + loss2, loss1, loss2
