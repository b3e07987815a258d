"""CPU fp32 restatement of the MViTv2 encoder path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(procedurevrl_amd/) never does.  Pinned: tests/golden/mvit_small.pt and mvit_block_shapes.pt are produced by the
UNMODIFIED reference `MViT_encoder` (lib/models/slowfast_mvit/mvit.py) in tests/golden/make_golden.py, and
tests/test_oracle_golden.py holds this file to them (forward features, per-block outputs, parameter gradients).

Scope = the configuration every shipped MViT yaml uses (configs/HowTo100M/procedurevrl_mvitv2_*.yaml):
MODE conv, CLS_EMBED_ON, no absolute position embedding, REL_POS_SPATIAL + REL_POS_TEMPORAL, RESIDUAL_POOLING,
DIM_MUL_IN_ATT, POOL_KVQ_KERNEL (3,3,3), adaptive KV stride, DROPOUT 0; DropPath (DROPPATH_RATE, per clip) is supported with pinned draws.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ geometry
def round_width(width, multiplier, min_width=1, divisor=1):
    """lib/models/slowfast_mvit/utils.py:7-20"""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    width_out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if width_out < 0.9 * width:
        width_out += divisor
    return int(width_out)


def plan(mv, num_frames, crop):
    """Per-block geometry exactly as MViT_encoder.__init__ derives it (mvit.py:139-239): returns
    (patch_dims, blocks) with blocks[i] = dict(dim, dim_out, heads, stride_q, stride_kv, in_thw)."""
    depth = int(mv["DEPTH"])
    ps = list(mv["PATCH_STRIDE"])
    thw = [num_frames // ps[0], crop // ps[1], crop // ps[2]]
    dim_mul = [1.0] * (depth + 1)
    head_mul = [1.0] * (depth + 1)
    for i, m in mv["DIM_MUL"]:
        dim_mul[int(i)] = float(m)
    for i, m in mv["HEAD_MUL"]:
        head_mul[int(i)] = float(m)
    stride_q = [[] for _ in range(depth)]
    for e in mv["POOL_Q_STRIDE"]:
        stride_q[int(e[0])] = [int(v) for v in e[1:]]
    stride_kv = [[] for _ in range(depth)]
    if mv.get("POOL_KV_STRIDE_ADAPTIVE") is not None:
        cur = [int(v) for v in mv["POOL_KV_STRIDE_ADAPTIVE"]]
        for i in range(depth):
            if len(stride_q[i]) > 0:
                cur = [max(cur[d] // stride_q[i][d], 1) for d in range(3)]
            stride_kv[i] = list(cur)
    else:
        for e in mv["POOL_KV_STRIDE"]:
            stride_kv[int(e[0])] = [int(v) for v in e[1:]]
    embed_dim, heads = int(mv["EMBED_DIM"]), int(mv["NUM_HEADS"])
    blocks = []
    size = list(thw)
    for i in range(depth):
        heads = round_width(heads, head_mul[i])
        if mv["DIM_MUL_IN_ATT"]:
            dim_out = round_width(embed_dim, dim_mul[i], divisor=round_width(heads, head_mul[i]))
        else:
            dim_out = round_width(embed_dim, dim_mul[i + 1], divisor=round_width(heads, head_mul[i + 1]))
        blocks.append(dict(dim=embed_dim, dim_out=dim_out, heads=heads, stride_q=stride_q[i] or [1, 1, 1],
                           stride_kv=stride_kv[i] or [1, 1, 1], in_thw=list(size)))
        if len(stride_q[i]) > 0:
            size = [s // st for s, st in zip(size, stride_q[i])]
        embed_dim = dim_out
    return thw, blocks


def rel_index(q_n, k_n):
    """dist table of cal_rel_pos_spatial / cal_rel_pos_temporal (attention.py:80-92,130-137): int index [q_n, k_n]."""
    q_ratio = max(k_n / q_n, 1.0)
    k_ratio = max(q_n / k_n, 1.0)
    d = torch.arange(q_n)[:, None] * q_ratio - torch.arange(k_n)[None, :] * k_ratio
    d = d + (k_n - 1) * k_ratio
    return d.long()


# ------------------------------------------------------------------------------------------------ forward
def ln(x, w, b, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention_pool(t, w, stride, thw, norm_w, norm_b):
    """attention.py:14-48 for mode 'conv' with a cls token: t [B, heads, 1+L, d]; depthwise Conv3d(k=3, pad=1)."""
    cls_tok, t = t[:, :, :1, :], t[:, :, 1:, :]
    B, N, L, C = t.shape
    T, H, W = thw
    t = t.reshape(B * N, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    t = F.conv3d(t, w, None, stride=tuple(stride), padding=tuple(k // 2 for k in w.shape[2:]), groups=C)
    out_thw = [t.shape[2], t.shape[3], t.shape[4]]
    t = t.reshape(B, N, C, -1).transpose(2, 3)
    t = torch.cat((cls_tok, t), dim=2)
    return ln(t, norm_w, norm_b), out_thw


def pool_skip(x, stride, thw):
    """MaxPool3d skip of MultiScaleBlock (attention.py:537-552, kernel s+1 where s>1, padding k//2), cls passes through."""
    if all(s == 1 for s in stride):
        return x
    k = [s + 1 if s > 1 else s for s in stride]
    cls_tok, t = x[:, :1, :], x[:, 1:, :]
    B, L, C = t.shape
    T, H, W = thw
    t = t.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3).contiguous()
    t = F.max_pool3d(t, k, tuple(stride), tuple(kk // 2 for kk in k), ceil_mode=False)
    t = t.reshape(B, C, -1).transpose(1, 2)
    return torch.cat((cls_tok, t), dim=1)


def msa(sd, pre, x, blk, taps=None):
    """MultiScaleAttention.forward (attention.py:307-442), pool_first False, separate_qkv False."""
    B, N, _ = x.shape
    h, dout = blk["heads"], blk["dim_out"]
    d = dout // h
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, h, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    thw = blk["in_thw"]
    q, q_thw = attention_pool(q, sd[pre + "pool_q.weight"], blk["stride_q"], thw, sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"])
    k, k_thw = attention_pool(k, sd[pre + "pool_k.weight"], blk["stride_kv"], thw, sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"])
    v, _ = attention_pool(v, sd[pre + "pool_v.weight"], blk["stride_kv"], thw, sd[pre + "norm_v.weight"], sd[pre + "norm_v.bias"])
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    qt, qh, qw = q_thw
    kt, kh, kw = k_thw
    r_q = q[:, :, 1:].reshape(B, h, qt, qh, qw, d)
    Rh = sd[pre + "rel_pos_h"][rel_index(qh, kh)]          # [qh, kh, d]   (attention.py:97-98; no interpolation
    Rw = sd[pre + "rel_pos_w"][rel_index(qw, kw)]          #  needed: table length = 2 max(q, k) - 1 by construction)
    Rt = sd[pre + "rel_pos_t"][rel_index(qt, kt)]
    rel_h = torch.einsum("bythwc,hkc->bythwk", r_q, Rh)
    rel_w = torch.einsum("bythwc,wkc->bythwk", r_q, Rw)
    rel_t = torch.einsum("bythwc,tkc->bythwk", r_q, Rt)
    bias = (rel_h[:, :, :, :, :, None, :, None] + rel_w[:, :, :, :, :, None, None, :] +
            rel_t[:, :, :, :, :, :, None, None]).reshape(B, h, qt * qh * qw, kt * kh * kw)
    attn = torch.cat((attn[:, :, :1], torch.cat((attn[:, :, 1:, :1], attn[:, :, 1:, 1:] + bias), dim=3)), dim=2)
    attn = attn.softmax(dim=-1)
    o = attn @ v
    o = torch.cat((o[:, :, :1], o[:, :, 1:] + q[:, :, 1:]), dim=2)            # residual pooling (attention.py:431-435)
    o = o.transpose(1, 2).reshape(B, -1, dout)
    if taps is not None:
        taps.update(q=q, k=k, v=v)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), q_thw


def block(sd, pre, x, blk, taps=None, dp=None):
    """MultiScaleBlock.forward (attention.py:545-568) with dim_mul_in_att, no layer scale.  dp = (s_attn, s_mlp): the
    per-clip DropPath factors floor(keep + U) / keep of common.py:38-52 (None = identity)."""
    xn = ln(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    xb, thw_new = msa(sd, pre + "attn.", xn, blk, taps)
    if blk["dim"] != blk["dim_out"]:
        x = F.linear(xn, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    if dp is not None:
        xb = xb * dp[0][:, None, None]
    x = pool_skip(x, blk["stride_q"], blk["in_thw"]) + xb
    xn = ln(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    h = F.gelu(F.linear(xn, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    y = F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if dp is not None:
        y = y * dp[1][:, None, None]
    return x + y


def forward_features(sd, x, mv, block_outputs=None, droppath=None):
    """MViT_encoder.forward (mvit.py:346-407): x fp32 [B, 3, T, H, W] -> [B, C_last] = norm(tokens)[:, 0].
    droppath: list over blocks of (s_attn [B], s_mlp [B]) or None per block."""
    _, blocks = plan(mv, x.shape[2], x.shape[3])
    t = F.conv3d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=tuple(mv["PATCH_STRIDE"]),
                 padding=tuple(mv["PATCH_PADDING"]))
    t = t.flatten(2).transpose(1, 2)                                             # stem_helper.py:319-321
    t = torch.cat((sd["cls_token"].expand(t.shape[0], -1, -1), t), dim=1)
    for i, blk in enumerate(blocks):
        t = block(sd, f"blocks.{i}.", t, blk, dp=None if droppath is None else droppath[i])
        if block_outputs is not None:
            block_outputs.append(t)
    t = ln(t, sd["norm.weight"], sd["norm.bias"])
    return t[:, 0]


def encoder_shapes(mv, num_frames, crop):
    """state_dict shapes of MViT_encoder for this cfg (names as the reference module produces them)."""
    thw, blocks = plan(mv, num_frames, crop)
    pk = tuple(mv["PATCH_KERNEL"])
    e0 = int(mv["EMBED_DIM"])
    sh = {"cls_token": (1, 1, e0), "patch_embed.proj.weight": (e0, 3) + pk, "patch_embed.proj.bias": (e0,)}
    for i, b in enumerate(blocks):
        p = f"blocks.{i}."
        dim, dout, h = b["dim"], b["dim_out"], b["heads"]
        d = dout // h
        q_n = b["in_thw"][1] // b["stride_q"][1]
        k_n = b["in_thw"][1] // b["stride_kv"][1]
        sh.update({p + "norm1.weight": (dim,), p + "norm1.bias": (dim,), p + "norm2.weight": (dout,), p + "norm2.bias": (dout,),
                   p + "attn.qkv.weight": (3 * dout, dim), p + "attn.qkv.bias": (3 * dout,),
                   p + "attn.proj.weight": (dout, dout), p + "attn.proj.bias": (dout,),
                   p + "attn.rel_pos_h": (2 * max(q_n, k_n) - 1, d), p + "attn.rel_pos_w": (2 * max(q_n, k_n) - 1, d),
                   p + "attn.rel_pos_t": (2 * b["in_thw"][0] - 1, d),
                   p + "mlp.fc1.weight": (4 * dout, dout), p + "mlp.fc1.bias": (4 * dout,),
                   p + "mlp.fc2.weight": (dout, 4 * dout), p + "mlp.fc2.bias": (dout,)})
        for n in ("q", "k", "v"):
            sh[p + f"attn.pool_{n}.weight"] = (d, 1, 3, 3, 3)
            sh[p + f"attn.norm_{n}.weight"] = (d,)
            sh[p + f"attn.norm_{n}.bias"] = (d,)
        if dim != dout:
            sh[p + "proj.weight"] = (dout, dim)
            sh[p + "proj.bias"] = (dout,)
    last = blocks[-1]["dim_out"]
    sh["norm.weight"] = (last,)
    sh["norm.bias"] = (last,)
    return sh
