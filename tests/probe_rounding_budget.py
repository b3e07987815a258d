"""ANALYSIS SCRIPT (test infrastructure, CPU only; not collected by pytest): where does the fp16-operand flavour's
logits error come from?  Runs the oracle with the HIP datapath's rounding points (oracle/rounded_oracle.py) on the
benchmark model (12 blocks, 8 x 224^2, 2 clips) with individual groups of rounding points switched off and prints the
relative L2 error of the features against the fp32 oracle.  `python tests/probe_rounding_budget.py [seed]`.

Used in round 5 to decide which rounding points to remove from the kernels.
"""
import os
import sys

import torch
import torch.nn.functional as F
from einops import rearrange

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import rounded_oracle as rorc          # noqa: E402
from oracle import timesformer_oracle as orc       # noqa: E402

OP = torch.float16


def rnd(t):
    return t.to(OP).to(torch.float32)


class V:
    """switches: every True is a rounding point (or an approximation) that stays on"""
    weights = True          # 16-bit weight copies
    cls_rows = True         # the cls rows' own h / q / o / g roundings and 16-bit weights (False: fp32 chain for cls rows)
    p_norm_rounded = False  # True: softmax normaliser = sum of the ROUNDED exp() (weights of P.V sum to the normaliser)
    h = True
    qkv = True
    o = True
    g = True
    p = True
    we = True               # fused temporal map rounded once more
    patch = True


def Rsel(t, on, cls_first):
    """round t; keep row 0 along dim 1 unrounded when the cls chain is fp32"""
    if not on:
        return t
    r = rnd(t)
    if cls_first and not V.cls_rows:
        r = torch.cat([t[:, :1], r[:, 1:]], 1)
    return r


def attn_core(qkv, B, S, C, H, mfma):
    qkv = qkv.reshape(B, S, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = (C // H) ** -0.5
    if not mfma:
        return ((q @ k.transpose(-2, -1)) * scale).softmax(-1).matmul(v).transpose(1, 2).reshape(B, S, C)
    s = q @ k.transpose(-2, -1)
    e = torch.exp((s - s.amax(-1, keepdim=True)) * scale)
    er = rnd(e) if V.p else e
    if not V.cls_rows:
        er = torch.cat([e[:, :, :1], er[:, :, 1:]], 2)
    den = (er if V.p_norm_rounded else e).sum(-1, keepdim=True)
    return ((er @ v) / den).transpose(1, 2).reshape(B, S, C)


def lin(sd, pre, n, t, cls_first=False):
    w, b = sd[pre + n + ".weight"], sd[pre + n + ".bias"]
    y = F.linear(t, rnd(w) if V.weights else w, b)
    if cls_first and not V.cls_rows and V.weights:
        y = torch.cat([F.linear(t[:, :1], w, b), y[:, 1:]], 1)
    return y


def block(sd, pre, x, B, T, W):
    N = (x.size(1) - 1) // T
    Hh = N // W
    C = x.shape[-1]
    ln = lambda t, n: F.layer_norm(t, (C,), sd[pre + n + ".weight"], sd[pre + n + ".bias"], orc.LN_EPS_VIT)
    xt = rearrange(x[:, 1:], "b (h w t) m -> (b h w) t m", b=B, h=Hh, w=W, t=T)
    h = Rsel(ln(xt, "temporal_norm1"), V.h, False)
    o = Rsel(attn_core(Rsel(lin(sd, pre, "temporal_attn.qkv", h), V.qkv, False), xt.shape[0], T, C, 12, mfma=(T != 8)), V.o, False)
    wf, wp = sd[pre + "temporal_fc.weight"], sd[pre + "temporal_attn.proj.weight"]
    we = (rnd(wf) @ rnd(wp)) if V.weights else wf @ wp
    if V.we:
        we = rnd(we)
    res_t = F.linear(o, we, wf @ sd[pre + "temporal_attn.proj.bias"])
    res_t = rearrange(res_t, "(b h w) t m -> b (h w t) m", b=B, h=Hh, w=W, t=T) + sd[pre + "temporal_fc.bias"]
    xt = x[:, 1:] + res_t
    init_cls = x[:, 0:1]
    cls = rearrange(init_cls.repeat(1, T, 1), "b t m -> (b t) m", b=B, t=T).unsqueeze(1)
    xs = torch.cat((cls, rearrange(xt, "b (h w t) m -> (b t) (h w) m", b=B, h=Hh, w=W, t=T)), 1)
    h = Rsel(ln(xs, "norm1"), V.h, True)
    qkv = Rsel(lin(sd, pre, "attn.qkv", h, True), V.qkv, True)
    o = Rsel(attn_core(qkv, xs.shape[0], xs.shape[1], C, 12, True), V.o, True)
    res_s = lin(sd, pre, "attn.proj", o, True)
    cls_new = rearrange(res_s[:, 0], "(b t) m -> b t m", b=B, t=T).mean(1, True)
    res_s = rearrange(res_s[:, 1:], "(b t) (h w) m -> b (h w t) m", b=B, h=Hh, w=W, t=T)
    x = torch.cat((init_cls, xt), 1) + torch.cat((cls_new, res_s), 1)
    h = Rsel(ln(x, "norm2"), V.h, True)
    g = Rsel(F.gelu(lin(sd, pre, "mlp.fc1", h, True)), V.g, True)
    return x + lin(sd, pre, "mlp.fc2", g, True)


def features(sd, x, depth):
    B, _, T, _, _ = x.shape
    xx = rearrange(rnd(x) if V.patch else x, "b c t h w -> (b t) c h w")
    w = sd["patch_embed.proj.weight"]
    xx = F.conv2d(xx, rnd(w) if V.weights else w, sd["patch_embed.proj.bias"], stride=16)
    W = xx.size(-1)
    xx = xx.flatten(2).transpose(1, 2)
    xx = torch.cat((sd["cls_token"].expand(xx.size(0), -1, -1), xx), 1) + sd["pos_embed"]
    cls = xx[:B, 0].unsqueeze(1)
    xx = rearrange(xx[:, 1:], "(b t) n m -> (b n) t m", b=B, t=T) + sd["time_embed"]
    xx = torch.cat((cls, rearrange(xx, "(b n) t m -> b (n t) m", b=B, t=T)), 1)
    for i in range(depth):
        xx = block(sd, f"blocks.{i}.", xx, B, T, W)
    return F.layer_norm(xx, (xx.shape[-1],), sd["norm.weight"], sd["norm.bias"], orc.LN_EPS_VIT)[:, 0]


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    depth, B = 12, 2
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed)
    label = torch.randn(9871, 512, generator=g) * 0.38
    label = label / label.norm(dim=1, keepdim=True)
    sd = orc.seeded_state(orc.encoder_shapes(depth), seed)
    x = torch.randn(B, 3, 8, 224, 224, generator=g)

    def logits(f):
        return orc.head_logits(sd, f, label, 0.02)[1]

    with torch.no_grad():
        ref = orc.forward_features(sd, x, depth)
        lref = logits(ref)
        cases = [("all rounding points (the shipped datapath)", {}),
                 ("cls chain fp32 (rows' own h/q/o/g/P and fp32 weights)", dict(cls_rows=False)),
                 ("normaliser = sum of rounded exp", dict(p_norm_rounded=True)),
                 ("both", dict(cls_rows=False, p_norm_rounded=True)),
                 ("no weight rounding", dict(weights=False, we=False)),
                 ("no h rounding", dict(h=False)),
                 ("no qkv rounding", dict(qkv=False)),
                 ("no o rounding", dict(o=False)),
                 ("no g rounding", dict(g=False)),
                 ("no P rounding", dict(p=False)),
                 ("no extra W_e rounding", dict(we=False)),
                 ("none (must be ~1e-6)", dict(weights=False, h=False, qkv=False, o=False, g=False, p=False, we=False, patch=False))]
        for name, kw in cases:
            saved = {k: getattr(V, k) for k in kw}
            for k, v in kw.items():
                setattr(V, k, v)
            f = features(sd, x, depth)
            print(f"{name:58s} features {rel(f, ref):.3e}   logits {rel(logits(f), lref):.3e}", flush=True)
            for k, v in saved.items():
                setattr(V, k, v)


if __name__ == "__main__":
    main()
