"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain eager fp32 PyTorch restatement of the reference's video-narration pre-training hot path
(facebookresearch/ProcedureVRL), op for op in the reference's own tensor layout, each function citing
the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module; nothing under procedurevrl_amd/ does.

Parity pinning: the reference itself is importable in the build container (package-init bypass + stubs,
see tests/golden/make_golden.py).  That script runs the UNMODIFIED reference modules
(lib/models/vit.py, lib/models/tfm_model.py, lib/utils/distributed.py) and commits their inputs/outputs as
fixtures under tests/golden/; tests/test_oracle_golden.py checks every function here against them.
Two parts cannot be pinned that way and are restated from the cited lines only:
  * the CLIP text tower lives in third-party openai/CLIP (un-vendored, unpinned in the reference,
    `clip.load("ViT-B/16")`, vit.py:258): its published architecture is restated in `clip_encode_text`
    and pinned against the reference's own CLIP-derived blocks (tfm_model.py:18-67) -- "parity unpinned"
    w.r.t. openai/CLIP itself;
  * tools/train_net.py is un-importable as shipped (lib/models/optimizer.py:40-41 is a SyntaxError):
    the loss block :152-162 is restated in `pretrain_loss` and pinned by a golden vector that executes
    exactly those source lines.

State dicts use the reference's key names (e.g. `blocks.0.temporal_attn.qkv.weight`).
"""
import math
import time

import torch
import torch.nn.functional as F
from einops import rearrange

LN_EPS_VIT = 1e-6   # vit.py:488  partial(nn.LayerNorm, eps=1e-6)
LN_EPS_TFM = 1e-5   # tfm_model.py:18-24 (nn.LayerNorm default)


# ------------------------------------------------------------------------------------------
# encoder (lib/models/vit.py)
# ------------------------------------------------------------------------------------------
def mlp(sd, pre, x):
    """Mlp.forward, vit.py:54-60 (exact-erf GELU, dropout p=0)."""
    x = F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    x = F.gelu(x)
    return F.linear(x, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def attention(sd, pre, x, num_heads=12):
    """Attention.forward, vit.py:75-92: scale applied AFTER q @ k^T (:84)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, num_heads, C // num_heads)
    qkv = qkv.permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // num_heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def drop_path_apply(x, keep_mask_scaled):
    """drop_path, lib/models/vit_utils.py:140-155.  `keep_mask_scaled` = floor(keep + U) / keep per dim-0 row,
    or None (eval / rate 0).  The draw itself is an explicit input."""
    if keep_mask_scaled is None:
        return x
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    return x * keep_mask_scaled.view(shape)


def block(sd, pre, x, B, T, W, num_heads=12, dp=None):
    """Block.forward (divided_space_time), vit.py:119-158.  dp = (s1 [B*H*W], s2 [B*T], s3 [B]) or None."""
    num_spatial_tokens = (x.size(1) - 1) // T
    H = num_spatial_tokens // W
    s1, s2, s3 = dp if dp is not None else (None, None, None)
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[pre + n + ".weight"], sd[pre + n + ".bias"], LN_EPS_VIT)
    # temporal (:129-135)
    xt = x[:, 1:, :]
    xt = rearrange(xt, "b (h w t) m -> (b h w) t m", b=B, h=H, w=W, t=T)
    res_temporal = drop_path_apply(attention(sd, pre + "temporal_attn.", ln(xt, "temporal_norm1"), num_heads), s1)
    res_temporal = rearrange(res_temporal, "(b h w) t m -> b (h w t) m", b=B, h=H, w=W, t=T)
    res_temporal = F.linear(res_temporal, sd[pre + "temporal_fc.weight"], sd[pre + "temporal_fc.bias"])
    xt = x[:, 1:, :] + res_temporal
    # spatial (:137-151)
    init_cls_token = x[:, 0, :].unsqueeze(1)
    cls_token = init_cls_token.repeat(1, T, 1)
    cls_token = rearrange(cls_token, "b t m -> (b t) m", b=B, t=T).unsqueeze(1)
    xs = rearrange(xt, "b (h w t) m -> (b t) (h w) m", b=B, h=H, w=W, t=T)
    xs = torch.cat((cls_token, xs), 1)
    res_spatial = drop_path_apply(attention(sd, pre + "attn.", ln(xs, "norm1"), num_heads), s2)
    cls_token = res_spatial[:, 0, :]
    cls_token = rearrange(cls_token, "(b t) m -> b t m", b=B, t=T)
    cls_token = torch.mean(cls_token, 1, True)
    res_spatial = res_spatial[:, 1:, :]
    res_spatial = rearrange(res_spatial, "(b t) (h w) m -> b (h w t) m", b=B, h=H, w=W, t=T)
    # merge + MLP (:155-157)
    x = torch.cat((init_cls_token, xt), 1) + torch.cat((cls_token, res_spatial), 1)
    x = x + drop_path_apply(mlp(sd, pre + "mlp.", ln(x, "norm2")), s3)
    return x


def patch_embed(sd, x):
    """PatchEmbed.forward, vit.py:174-180."""
    B, C, T, H, W = x.shape
    x = rearrange(x, "b c t h w -> (b t) c h w")
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    Wp = x.size(-1)
    x = x.flatten(2).transpose(1, 2)
    return x, T, Wp


def forward_features(sd, x, depth, num_heads=12, droppath=None):
    """VisionTransformer.forward_features (cls=True), vit.py:365-423.
    droppath: list over blocks of (s1, s2, s3) or None."""
    B = x.shape[0]
    x, T, W = patch_embed(sd, x)
    cls_tokens = sd["cls_token"].expand(x.size(0), -1, -1)
    x = torch.cat((cls_tokens, x), dim=1)
    pos_embed = sd["pos_embed"]
    if x.size(1) != pos_embed.size(1):   # :374-386 nearest resize
        cls_pos = pos_embed[0, 0, :].unsqueeze(0).unsqueeze(1)
        other = pos_embed[0, 1:, :].unsqueeze(0).transpose(1, 2)
        P = int(other.size(2) ** 0.5)
        H = x.size(1) // W
        other = other.reshape(1, x.size(2), P, P)
        new = F.interpolate(other, size=(H, W), mode="nearest").flatten(2).transpose(1, 2)
        x = x + torch.cat((cls_pos, new), 1)
    else:
        x = x + pos_embed
    cls_tokens = x[:B, 0, :].unsqueeze(1)
    x = x[:, 1:]
    x = rearrange(x, "(b t) n m -> (b n) t m", b=B, t=T)
    time_embed = sd["time_embed"]
    if T != time_embed.size(1):          # :398-402
        te = F.interpolate(time_embed.transpose(1, 2), size=(T), mode="nearest").transpose(1, 2)
        x = x + te
    else:
        x = x + time_embed
    x = rearrange(x, "(b n) t m -> b (n t) m", b=B, t=T)
    x = torch.cat((cls_tokens, x), dim=1)
    for i in range(depth):
        x = block(sd, f"blocks.{i}.", x, B, T, W, num_heads, None if droppath is None else droppath[i])
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], LN_EPS_VIT)
    return x[:, 0]


def input_pipeline(frames_u8, params, mean, std, crop):
    """CPU-worker input chain for ONE clip given its draws: tensor_normalize (lib/datasets/utils.py:309-326), 'T H W C ->
    C T H W' (howto100m.py:439), bilinear short-side rescale with align_corners=False (transform.py:52-61), crop
    (transform.py:109-111 / 184-186), horizontal flip (transform.py:142-143).
    frames_u8 [T, H0, W0, 3] uint8, params = (new_h, new_w, y_off, x_off, flip) -> fp32 [3, T, crop, crop]."""
    new_h, new_w, y_off, x_off, flip = [int(v) for v in params]
    x = frames_u8.float() / 255.0
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    x = x.permute(3, 0, 1, 2)
    if (new_h, new_w) != tuple(x.shape[2:]):
        x = F.interpolate(x, size=(new_h, new_w), mode="bilinear", align_corners=False)
    x = x[:, :, y_off:y_off + crop, x_off:x_off + crop]
    if flip:
        x = x.flip((-1))
    return x.contiguous()


def patch_rows(x):
    """fp32 clip(s) [B, 3, T, H, W] -> im2col rows ordered (b, n, t) x (c, py, px): the operand of PatchEmbed's
    16x16/16 conv (vit.py:172-180) in the token order of vit.py:396."""
    B, C, T, H, W = x.shape
    p = x.reshape(B, C, T, H // 16, 16, W // 16, 16).permute(0, 3, 5, 2, 1, 4, 6)
    return p.reshape(B * (H // 16) * (W // 16) * T, C * 256)


def l2n(x):
    return x / x.norm(dim=1, keepdim=True)


def head_logits(sd, feat, label_emb, temp):
    """vit.py:298-307: head, L2 norm, `x @ label_emb.t() / temp`.  label_emb must already be row-normalised
    (check_device_norm, vit.py:435-440, normalises on the first device move)."""
    x = F.linear(feat, sd["head.weight"], sd["head.bias"])
    x = l2n(x)
    return x, x @ label_emb.t() / temp


# ------------------------------------------------------------------------------------------
# CLIP-style residual attention stack (lib/models/tfm_model.py:18-67)
# ------------------------------------------------------------------------------------------
def resblock(sd, pre, x, n_head, attn_mask=None, pad_mask=None):
    """ResidualAttentionBlock.forward, tfm_model.py:43-53; x is sequence-first [t, b, c]."""
    ln = lambda t, n: F.layer_norm(t.float(), (t.shape[-1],), sd[pre + n + ".weight"], sd[pre + n + ".bias"], LN_EPS_TFM)
    h = ln(x, "ln_1")
    a = F.multi_head_attention_forward(
        h, h, h, h.shape[-1], n_head, sd[pre + "attn.in_proj_weight"], sd[pre + "attn.in_proj_bias"], None, None, False,
        0.0, sd[pre + "attn.out_proj.weight"], sd[pre + "attn.out_proj.bias"], training=False,
        key_padding_mask=pad_mask, need_weights=False, attn_mask=attn_mask)[0]
    x = x + a
    h = F.linear(ln(x, "ln_2"), sd[pre + "mlp.c_fc.weight"], sd[pre + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)                      # QuickGELU, tfm_model.py:27-29
    return x + F.linear(h, sd[pre + "mlp.c_proj.weight"], sd[pre + "mlp.c_proj.bias"])


def stack(sd, pre, x, layers, n_head, attn_mask=None, pad_mask=None):
    """TemporalModelling.forward, tfm_model.py:63-67."""
    for i in range(layers):
        x = resblock(sd, f"{pre}resblocks.{i}.", x, n_head, attn_mask, pad_mask)
    return x


def clip_encode_text(sd, pre, text, layers, n_head=8):
    """openai/CLIP `CLIP.encode_text` (third-party; published algorithm): token + positional embedding,
    causal transformer, ln_final, take the features at the EOT token (argmax id), @ text_projection."""
    x = sd[pre + "token_embedding.weight"][text] + sd[pre + "positional_embedding"]
    S = text.shape[1]
    mask = torch.full((S, S), float("-inf")).triu_(1)      # tfm_model.py:265-270 build_attention_mask
    x = stack(sd, pre + "transformer.", x.permute(1, 0, 2), layers, n_head, attn_mask=mask).permute(1, 0, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd[pre + "ln_final.weight"], sd[pre + "ln_final.bias"], LN_EPS_TFM)
    return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ sd[pre + "text_projection"]


def pseudo_labels(sd, text_ids, clip_vis_feat, label_emb, temp, text_layers):
    """get_pseudo_labels, vit.py:425-433."""
    text_emb = clip_encode_text(sd, "text_model.", text_ids, text_layers)
    text_emb = (text_emb + clip_vis_feat) / 2.0
    text_emb = l2n(text_emb)
    return text_emb @ label_emb.t() / temp


# ------------------------------------------------------------------------------------------
# order / diffusion transformer (lib/models/tfm_model.py:70-302)
# ------------------------------------------------------------------------------------------
def sinusoidal(time, dim):
    """SinusoidalPositionEmbeddings, lib/models/diffusion_model.py:34-47."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = time[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def diffusion_coefs(levels):
    """configure_diffusion, tfm_model.py:106-127 with linear_beta_schedule (diffusion_model.py:328-331)."""
    betas = torch.linspace(0.0001, 0.02, levels)
    ac = torch.cumprod(1.0 - betas, 0)
    return torch.sqrt(ac), torch.sqrt(1.0 - ac)


def order_tfm_pretrain(sd, pre, x, max_len, layers, n_head, mask_inds, pad_start, noises):
    """DiffusionTransformer.forward(is_pretrain=True): tfm_model.py:129-156 -> pad_sequence :272-289 ->
    diffusion_signal_training :165-204.  RNG draws (mask_inds :145, pad_start :283, noise :180) are inputs."""
    hidden = x.shape[1]
    clip_feats = rearrange(x, "(b t) c -> t b c", t=max_len).clone()
    bsz = clip_feats.size(1)
    temp_emb = sd[pre + "temporalEmbedding.weight"][torch.arange(max_len)][:, None, :].expand(-1, bsz, -1)
    bs_inds = torch.arange(bsz)
    x0 = clip_feats[mask_inds, bs_inds]
    pad_mask = torch.zeros(bsz, max_len, dtype=torch.bool)
    for i in range(bsz):                                          # pad_sequence
        ps = int(pad_start[i])
        if ps < max_len:
            clip_feats[ps:, i] = sd[pre + "pad_embedding.weight"]
        pad_mask[i, ps:] = True
    sa, sb = diffusion_coefs(layers)
    orig = clip_feats
    inter = []
    denoised = None
    for time_i in range(layers):
        cf = orig.clone()
        t_index = layers - 1 - time_i
        t = torch.full((bsz,), t_index, dtype=torch.long)
        src = x0.clone().detach() if time_i == 0 else denoised.clone().detach()
        noisy = sa[t_index] * src + sb[t_index] * noises[time_i]            # ennoise, :291-302
        cf[mask_inds, bs_inds] = noisy
        type_emb = sd[pre + "type_embedding.weight"][torch.zeros(max_len, bsz, dtype=torch.long)].clone()
        type_emb[mask_inds, bs_inds] = sd[pre + "type_embedding.weight"][torch.ones(bsz, dtype=torch.long)]
        h = cf + type_emb + temp_emb
        tm = sinusoidal(t, hidden // 4)
        tm = F.linear(tm, sd[pre + "time_mlp.1.weight"], sd[pre + "time_mlp.1.bias"])
        tm = F.linear(F.gelu(tm), sd[pre + "time_mlp.3.weight"], sd[pre + "time_mlp.3.bias"])
        h = h + tm[None, :, :]
        out = stack(sd, pre + "temporalModelling.", h, layers, n_head, pad_mask=pad_mask)
        denoised = out[mask_inds, bs_inds]
        inter.append(denoised)
    x0_rep = x0.unsqueeze(0).expand(layers, -1, -1).reshape(-1, x0.size(-1))
    inter = torch.cat(inter)
    return denoised, mask_inds, [x0_rep, inter], inter


def order_tfm_forecast(sd, pre, x, num_seg, max_len, layers, n_head):
    """DiffusionTransformer.diffusion_signal_forecast, tfm_model.py:206-249 (the appended 'noise' token is all-zero)."""
    hidden = x.shape[1]
    clip_feats = rearrange(x, "(b t) c -> t b c", t=num_seg)
    bsz = clip_feats.size(1)
    temp_emb = sd[pre + "temporalEmbedding.weight"][torch.arange(max_len)][:, None, :].expand(-1, bsz, -1)
    bs_inds = torch.arange(bsz)
    mask_inds = torch.full((bsz,), max_len - 1, dtype=torch.long)
    noise = torch.zeros((1, bsz, hidden))
    orig = torch.cat((clip_feats, noise), dim=0)
    cf = orig.clone()
    sa, sb = diffusion_coefs(layers)
    denoised = None
    for time_i in range(layers):
        t_index = layers - 1 - time_i
        t = torch.full((bsz,), t_index, dtype=torch.long)
        if time_i != 0:
            cf[mask_inds, bs_inds] = sa[t_index] * denoised.clone().detach() + sb[t_index] * noise[0]
        type_emb = sd[pre + "type_embedding.weight"][torch.zeros(max_len, bsz, dtype=torch.long)].clone()
        type_emb[mask_inds, bs_inds] = sd[pre + "type_embedding.weight"][torch.ones(bsz, dtype=torch.long)]
        h = cf + type_emb + temp_emb
        tm = sinusoidal(t, hidden // 4)
        tm = F.linear(tm, sd[pre + "time_mlp.1.weight"], sd[pre + "time_mlp.1.bias"])
        tm = F.linear(F.gelu(tm), sd[pre + "time_mlp.3.weight"], sd[pre + "time_mlp.3.bias"])
        h = h + tm[None, :, :]
        out = stack(sd, pre + "temporalModelling.", h, layers, n_head)
        denoised = out[mask_inds, bs_inds]
        cf = orig.clone()
        cf[mask_inds, bs_inds] = denoised
    return cf[mask_inds, bs_inds]


def vit_forward_forecast_eval(sd, x, label_emb, temp, depth, num_seg, max_len=9, order_layers=4):
    """VisionTransformer.forward in eval mode with MODEL.NUM_SEG > 0 and MATCH_LANG_EMB (vit.py:292-307, 355-356)."""
    x = rearrange(x, "b c (m t) h w -> (b m) c t h w", m=num_seg, t=x.shape[2] // num_seg)
    feat = forward_features(sd, x, depth)
    emb = l2n(F.linear(feat, sd["head.weight"], sd["head.bias"]))
    z = l2n(order_tfm_forecast(sd, "order_tfm.", emb, num_seg, max_len, order_layers, 8))
    return torch.softmax(z @ label_emb.t() / temp, dim=1)


def vit_forward_train(sd, inputs, meta, label_emb, temp, depth, max_len, order_layers, text_layers, rng,
                      order_recog_batch=9, droppath=None, encoder=None):
    """VisionTransformer.forward in pre-training mode, vit.py:283-352 (ORDER_PRETRAIN_ENABLED, MATCH_LANG_EMB,
    text model present, training).  `encoder(x) -> features` replaces the TimeSformer encoder for the MViT wrapper
    (lib/models/mvit.py:109-229 is the same code around `self.video_encoder`)."""
    batch_size = inputs.shape[0]
    x = rearrange(inputs, "b m c t h w -> (b m) c t h w", m=max_len)
    feat = encoder(x) if encoder is not None else forward_features(sd, x, depth, droppath=droppath)
    video_emb, logits = head_logits(sd, feat, label_emb, temp)
    teacher_x = pseudo_labels(sd, meta["clip_text_ids"], meta["clip_vis_feat"], label_emb, temp, text_layers)
    pred_emb, mask_inds, mse, inter = order_tfm_pretrain(sd, "order_tfm.", video_emb, max_len, order_layers, 8,
                                                         rng["mask_inds"], rng["pad_start"], rng["noises"])
    ts = rearrange(teacher_x, "(b m) c -> b m c", m=max_len)
    masked_teacher = ts[torch.arange(ts.shape[0]), mask_inds, :]
    inter = l2n(inter)
    inter_pred = inter @ label_emb.t() / temp
    inter_teacher = masked_teacher.unsqueeze(0).expand(order_layers, -1, -1).reshape(-1, masked_teacher.size(-1))
    rand_inds = rng["rand_inds"][:batch_size * order_recog_batch]
    pred = torch.cat((logits[rand_inds], inter_pred), dim=0)
    teacher = torch.cat((teacher_x[rand_inds], inter_teacher), dim=0)
    return pred, teacher, mse


# ------------------------------------------------------------------------------------------
# loss head (tools/train_net.py:152-162) and the contrastive operator API
# ------------------------------------------------------------------------------------------
def pretrain_loss(pred, teacher_pred, mse, topk=5):
    with torch.no_grad():
        teacher_pred = F.softmax(teacher_pred, 1)
        if topk != 0:
            teacher_pred = (teacher_pred.unsqueeze(1) * (teacher_pred.unsqueeze(1) == teacher_pred.topk(k=topk, dim=1)[0].unsqueeze(2)).float()).sum(1)
            teacher_pred = teacher_pred / teacher_pred.sum(1, keepdim=True)
    loss1 = torch.nn.KLDivLoss(reduction="batchmean")(F.log_softmax(pred, dim=1), teacher_pred)
    loss2 = torch.nn.MSELoss(reduction="mean")(mse[0], mse[1]) if mse is not None else torch.zeros(())
    return loss1 + loss2, loss1, loss2


def milnce(video_embd, text_embd):
    """MILNCELoss.forward, lib/models/losses.py:15-23 (th.eye kept on the input's device instead of .cuda())."""
    x = torch.matmul(video_embd, text_embd.t())
    x = x.view(video_embd.shape[0], video_embd.shape[0], -1)
    nominator = x * torch.eye(x.shape[0])[:, :, None]
    nominator = nominator.sum(dim=1)
    nominator = torch.logsumexp(nominator, dim=1)
    denominator = torch.cat((x, x.permute(1, 0, 2)), dim=1).view(x.shape[0], -1)
    denominator = torch.logsumexp(denominator, dim=1)
    return torch.mean(denominator - nominator)


def allgather_forward_backward(local_tensors, grad_output):
    """Semantics of du.AllGather (lib/utils/distributed.py:13-29) simulated for a list of per-rank tensors:
    forward = concatenation over ranks; backward on rank r = rows [b*r, b*(r+1)) of ITS grad_output, no sum."""
    out = torch.cat(local_tensors, 0)
    b = local_tensors[0].shape[0]
    return out, [g[b * r: b * (r + 1)] for r, g in enumerate(grad_output)]


def lr_at_epoch(cfg, cur_epoch):
    """lib/utils/lr_policy.py:8-87 (steps_with_relative_lrs / cosine + warm-up)."""
    s = cfg.SOLVER
    def f(ep):
        if s.LR_POLICY == "cosine":
            return s.COSINE_END_LR + (s.BASE_LR - s.COSINE_END_LR) * (math.cos(math.pi * ep / s.MAX_EPOCH) + 1.0) * 0.5
        steps = list(s.STEPS) + [s.MAX_EPOCH]
        ind = 0
        for ind, st in enumerate(steps):
            if ep < st:
                break
        return s.LRS[ind - 1] * s.BASE_LR
    lr = f(cur_epoch)
    if cur_epoch < s.WARMUP_EPOCHS:
        alpha = (f(s.WARMUP_EPOCHS) - s.WARMUP_START_LR) / s.WARMUP_EPOCHS
        lr = cur_epoch * alpha + s.WARMUP_START_LR
    return lr


# ------------------------------------------------------------------------------------------
# weights + CPU-baseline timing helpers
# ------------------------------------------------------------------------------------------
def encoder_shapes(depth, num_frames=8, num_patches=196, dim=768, head_dim=512):
    sh = {"cls_token": (1, 1, dim), "pos_embed": (1, num_patches + 1, dim), "time_embed": (1, num_frames, dim),
          "patch_embed.proj.weight": (dim, 3, 16, 16), "patch_embed.proj.bias": (dim,),
          "norm.weight": (dim,), "norm.bias": (dim,), "head.weight": (head_dim, dim), "head.bias": (head_dim,)}
    for i in range(depth):
        p = f"blocks.{i}."
        for n in ("norm1", "temporal_norm1", "norm2"):
            sh[p + n + ".weight"] = (dim,); sh[p + n + ".bias"] = (dim,)
        for a in ("attn", "temporal_attn"):
            sh[p + a + ".qkv.weight"] = (3 * dim, dim); sh[p + a + ".qkv.bias"] = (3 * dim,)
            sh[p + a + ".proj.weight"] = (dim, dim); sh[p + a + ".proj.bias"] = (dim,)
        sh[p + "temporal_fc.weight"] = (dim, dim); sh[p + "temporal_fc.bias"] = (dim,)
        sh[p + "mlp.fc1.weight"] = (4 * dim, dim); sh[p + "mlp.fc1.bias"] = (4 * dim,)
        sh[p + "mlp.fc2.weight"] = (dim, 4 * dim); sh[p + "mlp.fc2.bias"] = (dim,)
    return sh


def seeded_state(shapes, seed):
    """Deterministic synthetic weights keyed by NAME (sorted), so two implementations with the same keys get
    identical tensors: LayerNorm gains ~ 1 + 0.1 N, matrices / embeddings ~ 0.02 N (0.05 for biases)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        r = torch.randn(shp, generator=g)
        if k.endswith("logit_scale"):
            sd[k] = torch.tensor(math.log(1 / 0.07))
        elif (".norm" in k or k.startswith("norm") or ".ln_" in k or "ln_final" in k or "temporal_norm1" in k) and k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * r
        elif k.endswith("bias"):
            sd[k] = 0.05 * r
        elif k.endswith("positional_embedding") or "Embedding" in k or "embedding.weight" in k:
            sd[k] = 0.02 * r
        elif "text_projection" in k or "in_proj_weight" in k:
            sd[k] = r * (shp[-1] ** -0.5)
        else:
            sd[k] = 0.02 * r
    return sd


def stack_shapes(pre, layers, w):
    """parameter shapes of a CLIP-style residual attention stack (tfm_model.py:32-67 names)"""
    sh = {}
    for i in range(layers):
        q = f"{pre}resblocks.{i}."
        sh.update({q + "attn.in_proj_weight": (3 * w, w), q + "attn.in_proj_bias": (3 * w,),
                   q + "attn.out_proj.weight": (w, w), q + "attn.out_proj.bias": (w,),
                   q + "ln_1.weight": (w,), q + "ln_1.bias": (w,), q + "ln_2.weight": (w,), q + "ln_2.bias": (w,),
                   q + "mlp.c_fc.weight": (4 * w, w), q + "mlp.c_fc.bias": (4 * w,),
                   q + "mlp.c_proj.weight": (w, 4 * w), q + "mlp.c_proj.bias": (w,)})
    return sh


def order_shapes(layers=4, w=512, L=9):
    """DiffusionTransformer parameters (tfm_model.py:70-104)"""
    sh = {"order_tfm.pad_embedding.weight": (1, w), "order_tfm.type_embedding.weight": (2, w),
          "order_tfm.temporalEmbedding.weight": (L, w), "order_tfm.time_mlp.1.weight": (w, w // 4),
          "order_tfm.time_mlp.1.bias": (w,), "order_tfm.time_mlp.3.weight": (w, w), "order_tfm.time_mlp.3.bias": (w,)}
    sh.update(stack_shapes("order_tfm.temporalModelling.", layers, w))
    return sh


def text_shapes(layers, w=512):
    """CLIP ViT-B/16 text tower parameters under CLIP's key names (vit.py:258-261)"""
    sh = {"text_model.token_embedding.weight": (49408, w), "text_model.positional_embedding": (77, w),
          "text_model.ln_final.weight": (w,), "text_model.ln_final.bias": (w,), "text_model.text_projection": (w, w),
          "text_model.logit_scale": ()}
    sh.update(stack_shapes("text_model.transformer.", layers, w))
    return sh


def host_cpu():
    """(model string, physical cores, logical cores visible to this process) of the host"""
    import os
    model, phys = "unknown CPU", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown CPU":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        logical = len(os.sched_getaffinity(0))
    except Exception:
        logical = os.cpu_count() or 1
    return model, (len(phys) or logical), logical


def timed_full_step(videos=2, frames=8, classes=9871, budget_s=60.0, threads=None):
    """SURVEY 8(d)'s CPU baseline: BASELINE configs[0] -- `videos` videos x 9 clips of 8 x 224^2, the reference's FULL pre-training
    step (vit.py:283-352 encoder + head + frozen 12-layer CLIP-text teacher + order / diffusion transformer, train_net.py:152-192
    top-5 KL + MSE, backward, AdamW over the trainable parameters) in eager fp32 on ALL cores visible to the process.  Method
    (SURVEY 8d): 1 warm-up step (first-call set-up of the math library is not the baseline), then the median of up to 3 timed
    steps -- at least ONE whatever the budget, more while `budget_s` allows.  Plus the 2-clip eval `forward_features` leg
    (`eval_leg`, 1 warm-up + median of 3)."""
    model, phys, logical = host_cpu()
    calib = ""
    if not threads:
        # Eager PyTorch does not scale to every core of a large host (each small op forks and joins all threads): time one encoder
        # block (2 clips, forward + backward) on all physical cores and on 32 threads and keep the faster -- the STRONGER baseline.
        cand = sorted({min(phys, logical), min(32, phys, logical)}, reverse=True)
        sd1 = {k: v.clone().requires_grad_(True) for k, v in seeded_state(encoder_shapes(1, frames), 0).items()}
        x1 = torch.randn(2, 3, frames, 224, 224)
        best = None
        for n in cand:
            torch.set_num_threads(n)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                forward_features(sd1, x1, 1).sum().backward()
                ts.append(time.perf_counter() - t0)
            calib += f"{n} threads {min(ts):.2f} s; "
            if best is None or min(ts) < best[0]:
                best = (min(ts), n)
        threads = best[1]
        calib = f"one encoder block fwd+bwd on 2 clips: {calib}the faster is used"
    threads = int(threads)
    torch.set_num_threads(max(1, threads))
    sh = encoder_shapes(12, frames)
    sh.update(order_shapes())
    sh.update(text_shapes(12))
    sd = seeded_state(sh, 0)
    params = {k: (v.clone().requires_grad_(True) if not k.startswith("text_model.") else v) for k, v in sd.items()}
    opt = torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=5e-5, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1)
    b, m = videos, 9
    inputs = torch.randn(b, m, 3, frames, 224, 224, generator=g)
    label = l2n(torch.randn(classes, 512, generator=g) * 0.38)
    L = torch.randint(8, 41, (b * m,), generator=g)
    ids = torch.zeros(b * m, 77, dtype=torch.long)
    for r in range(b * m):
        n = int(L[r])
        ids[r, 0] = 49406
        ids[r, 1:1 + n] = torch.randint(1, 49406, (n,), generator=g)
        ids[r, 1 + n] = 49407
    meta = {"clip_text_ids": ids, "clip_vis_feat": torch.randn(b * m, 512, generator=g) * 0.4}
    rng = {"mask_inds": torch.randint(0, m, (b,), generator=g), "pad_start": torch.randint(1, m + 1, (b,), generator=g).tolist(),
           "noises": [torch.randn(b, 512, generator=g) for _ in range(4)], "rand_inds": torch.randperm(b * m, generator=g)}
    times = []
    t_all = time.perf_counter()
    for it in range(4):
        t0 = time.perf_counter()
        opt.zero_grad()
        pred, teacher, mse = vit_forward_train(params, inputs, meta, label, 0.02, 12, m, 4, 12, rng)
        loss, _, _ = pretrain_loss(pred, teacher, mse, 5)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        if it >= 1 and time.perf_counter() - t_all + times[-1] > budget_s:     # another step would not fit the budget
            break
    timed = times[1:]
    dt = sorted(timed)[len(timed) // 2]
    clips = b * m
    # eval leg (1.15 clips/s on the real reference, 8 cores): forward_features of 2 clips, no gradients
    xe = inputs[0, :2].contiguous()
    te = []
    with torch.no_grad():
        for _ in range(4):
            t0 = time.perf_counter()
            forward_features(sd, xe, 12)
            te.append(time.perf_counter() - t0)
    dte = sorted(te[1:])[1]
    return {"value": round(clips / dt, 4), "unit": "clips/s", "cores": threads, "kind": "port", "timed_steps": len(timed),
            "step_s": [round(t, 2) for t in timed],
            "eval_leg": {"value": round(2 / dte, 4), "unit": "clips/s", "sample": f"2 clips x {frames}f x 224^2, forward_features only "
                         f"(no gradients), median of 3 after 1 warm-up, {dte:.2f} s per pass"},
            "sample": f"BASELINE configs[0]: {videos} videos x 9 = {clips} clips x {frames}f x 224^2, the reference's FULL pre-training step "
                      f"(encoder + head + frozen CLIP-text teacher + order transformer + top-5 KL + MSE, backward, AdamW), eager fp32 PyTorch "
                      f"oracle on {threads} threads (host: {phys} physical / {logical} logical cores visible to the process, {model}{'; ' + calib if calib else ''}); "
                      f"median of {len(timed)} timed step{'s' if len(timed) > 1 else ''} after 1 warm-up ({times[0]:.2f} s), "
                      f"{dt:.2f} s per step ({time.perf_counter() - t_all:.0f} s of wall time in all)"}


def timed_train_step(clips=2, frames=8, classes=9871, threads=8, repeats=3):
    """One full training step of BASELINE config 2's workload (encoder fwd, head + step logits + top-5 KL,
    backward, AdamW) on the host CPU in eager fp32, on a bounded sample of `clips` clips."""
    try:
        import os
        threads = min(int(threads), len(os.sched_getaffinity(0)))
    except Exception:
        threads = int(threads)
    torch.set_num_threads(max(1, threads))
    sd = seeded_state(encoder_shapes(12, frames), 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(params.values()), lr=5e-5, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(clips, 3, frames, 224, 224, generator=g)
    label = l2n(torch.randn(classes, 512, generator=g) * 0.38)
    teacher = torch.randn(clips, classes, generator=g) * 4
    times = []
    for _ in range(repeats + 1):   # first pass is the warm-up
        t0 = time.perf_counter()
        opt.zero_grad()
        feat = forward_features(params, x, 12)
        _, logits = head_logits(params, feat, label, 0.02)
        loss, _, _ = pretrain_loss(logits, teacher, None, 5)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]   # median
    return {"value": round(clips / dt, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{clips} clips x {frames}f x 224^2, one full train step (fwd+bwd+AdamW), eager fp32 PyTorch oracle, "
                      f"median of {repeats} after 1 warm-up, {dt:.2f} s per step ({sum(times):.0f} s of CPU work in all)"}
