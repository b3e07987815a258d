"""CPU ORACLE WITH THE HIP DATAPATH'S ROUNDING POINTS -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

`oracle/timesformer_oracle.py` restates the reference in fp32 and is pinned to it by golden vectors.  The HIP path differs
from it in exactly one declared way: GEMM / attention operands and the activations stored between kernels are 16-bit
(bf16 by default), everything else -- accumulation, residual stream, LayerNorm and softmax statistics, head, logits,
losses, parameter gradients -- is fp32.  This module is the same restatement with a rounding to
the operand type inserted at every point where the HIP kernels round, and nowhere else.  It exists to answer one
question with a test instead of an argument: is operand rounding the ONLY source of the 2-6e-3 difference between the
HIP path and the fp32 reference?  If it is, the HIP outputs must agree with this model to well below 1e-3
(`tests/e2e_checks.py: check_rounding_model_*`), while this model with `operand=None` must equal the fp32 oracle exactly
(`tests/test_oracle_golden.py::test_rounded_oracle_without_rounding_is_the_oracle`), which pins it to the reference.

Rounding points (kernel that rounds -> where it appears below):
  forward   patchify (pixels -> operand)                         `R(x)` on the im2col rows
            weight operand copies (cast_weights_multi)           `RW(w)` on every GEMM weight; biases / LN params stay fp32
            LayerNorm output h (norm.hip, 16-bit output)         `R(ln(..))`
            qkv (gemm_nt 16-bit epilogue)                        `R(linear(h))`
            spatial attention: un-normalised exp() as the 2nd MFMA operand, 1/sum applied to the fp32 accumulator
                                                                 `AttnMFMA`; temporal attention (attn_t8) keeps P in fp32
            attention output o                                   `R(o)`
            temporal branch: proj and temporal_fc folded into ONE matrix W_e = R(RW(W_fc) @ RW(W_proj)) (engine._fused_temporal)
            MLP: pre-activation u stored 16-bit for backward, g = R(gelu(u_fp32))    `GeluStore`
            residual stream (round 6, RESID = "fwd" | "both"): the PATCH rows of x after the embedding prologue and after each of a
            block's three residual adds are stored in the operand type (PVRL_EPI_RESID_16 epilogues; the cls rows stay fp32)  `RX`
            EXCEPT the cls rows (round 5, csrc/cls_chain.hip): their attn.proj (from the 16-bit o) and their whole MLP run in fp32 on
            the master weights -- `linf`; their backward takes the 16-bit path, as in the kernels (`_value_of`)
  backward  the gradient operand of every GEMM is the 16-bit copy of (DropPath scale x fp32 residual gradient), or the
            16-bit output of the previous backward GEMM / attention kernel          backward half of `R`, and `RB`
            dGELU uses the 16-bit stored u; dS and P are rounded for the dQ/dK/dV MFMAs  `GeluStore`, `AttnMFMA`
            RESID = "both": the residual GRADIENT stream of the patch rows is 16-bit as well (ln_bwd's dx in / out)   backward half of `RX`
"""
import torch
import torch.nn.functional as F
from einops import rearrange

from . import timesformer_oracle as orc

OPERAND = None          # None (no rounding: identical to timesformer_oracle), torch.bfloat16 or torch.float16
# Round 6: the PATCH rows of the residual stream stored in the operand type between kernels (the cls rows stay
# fp32).  None: fp32 stream (rounds 1-5); "fwd": the values x0 / x1 / x2 / x3 of the patch rows are 16-bit (GEMM residual epilogues
# read and write them, LayerNorm reads them); "both": the residual GRADIENT stream dx of the patch rows as well (ln_bwd's dx in / out).
RESID = None
# fc1 epilogue variant that was costed and rejected (tests/probe_resid16.py): GELU derivative kept as an n-bit code instead of the
# 16-bit pre-activation u.  None: the shipped form (16-bit u).
DGELU_BITS = None


def _rnd(t):
    return t if OPERAND is None else t.to(OPERAND).to(torch.float32)


class _Round(torch.autograd.Function):
    """value rounded to the operand type; its gradient is rounded too (it is stored / consumed as a 16-bit operand)"""
    @staticmethod
    def forward(ctx, x):
        return _rnd(x)

    @staticmethod
    def backward(ctx, g):
        return _rnd(g)


class _RoundFwd(torch.autograd.Function):
    """weights: 16-bit operand copy in forward, fp32 gradient"""
    @staticmethod
    def forward(ctx, x):
        return _rnd(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """fp32 value whose incoming gradient is consumed as a 16-bit GEMM operand"""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rnd(g)


def R(x):
    return _Round.apply(x)


def RW(x):
    return _RoundFwd.apply(x)


def RB(x):
    return _RoundBwd.apply(x)


def RX(x):
    """patch rows of the residual stream (see RESID)"""
    if OPERAND is None or RESID is None:
        return x
    return (_Round if RESID == "both" else _RoundFwd).apply(x)


class GeluStore(torch.autograd.Function):
    """fc1 epilogue (gemm_nt GELU): g = gelu(u) from the fp32 accumulator, u kept as a 16-bit copy; backward multiplies by
    gelu'(stored u) (gemm_nt dGELU epilogue).  Returns the unrounded g (the caller rounds it)."""
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(_rnd(u))
        return F.gelu(u)

    @staticmethod
    def backward(ctx, dg):
        (u,) = ctx.saved_tensors
        cdf = 0.5 * (1.0 + torch.erf(u * 0.7071067811865476))
        pdf = torch.exp(-0.5 * u * u) * 0.3989422804014327
        d = cdf + u * pdf
        if DGELU_BITS is not None:      # gelu' in [-0.13, 1.13] as a uniform n-bit code
            q = 1.26 / (2 ** DGELU_BITS - 1)
            d = torch.round((d + 0.13) / q) * q - 0.13
        return dg * d


class AttnMFMA(torch.autograd.Function):
    """softmax(q k^T * scale) v as csrc/attn_mfma.hip computes it on 16-bit q, k, v [.., S, d]:
    forward: e = exp(scale * (s - max)) rounded as the second MFMA's operand, o = (R(e) v) / sum(e);
    backward: P = exp(scale*s - lse) in fp32, D = rowsum(dO * O_stored), dS = R(P * (dP - D) * scale), dV = R(P)^T dO."""
    @staticmethod
    def forward(ctx, q, k, v, scale):
        s = q @ k.transpose(-2, -1)
        m = s.amax(-1, keepdim=True)
        e = torch.exp((s - m) * scale)
        den = e.sum(-1, keepdim=True)
        o = (_rnd(e) @ v) / den
        ctx.save_for_backward(q, k, v, e / den, _rnd(o))
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, o = ctx.saved_tensors
        dp = do @ v.transpose(-2, -1)
        d = (do * o).sum(-1, keepdim=True)
        ds = _rnd(p * (dp - d) * ctx.scale)
        return ds @ k, ds.transpose(-2, -1) @ q, _rnd(p).transpose(-2, -1) @ do, None


def _value_of(exact, rounded):
    """the forward VALUE of `exact` with the backward of `rounded`: the cls rows' forward runs in fp32 (csrc/cls_chain.hip) while
    their backward takes the 16-bit kernels' path on the activations that path saved"""
    return rounded + (exact - rounded).detach()


def attention_core(qkv, B, N, C, num_heads, mfma):
    qkv = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = (C // num_heads) ** -0.5
    if mfma and OPERAND is not None:
        o = AttnMFMA.apply(q, k, v, scale)
    else:       # attn_t8.hip: fp32 VALU on the 16-bit inputs, P and dS never rounded
        o = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1) @ v
    return o.transpose(1, 2).reshape(B, N, C)


def block(sd, pre, x, B, T, W, num_heads=12, dp=None):
    """timesformer_oracle.block (Block.forward, vit.py:119-158) with the HIP datapath's rounding points."""
    num_spatial_tokens = (x.size(1) - 1) // T
    H = num_spatial_tokens // W
    C = x.shape[-1]
    s1, s2, s3 = dp if dp is not None else (None, None, None)
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[pre + n + ".weight"], sd[pre + n + ".bias"], orc.LN_EPS_VIT)
    lin = lambda t, n: F.linear(t, RW(sd[pre + n + ".weight"]), sd[pre + n + ".bias"])
    linf = lambda t, n: F.linear(t, sd[pre + n + ".weight"], sd[pre + n + ".bias"])
    # temporal (:129-135).  proj and temporal_fc are one matrix in the HIP engine (only a per-row DropPath scale sits
    # between them): x + fc(rs * proj(o)) = x + rs * (o W_e^T + W_fc b_proj) + b_fc, W_e = W_fc W_proj
    xt = x[:, 1:, :]
    xt = rearrange(xt, "b (h w t) m -> (b h w) t m", b=B, h=H, w=W, t=T)
    h = R(ln(xt, "temporal_norm1"))
    o = R(attention_core(R(lin(h, "temporal_attn.qkv")), xt.shape[0], T, C, num_heads, mfma=(T != 8)))
    if OPERAND is None:
        res_temporal = orc.drop_path_apply(lin(o, "temporal_attn.proj"), s1)
        res_temporal = rearrange(res_temporal, "(b h w) t m -> b (h w t) m", b=B, h=H, w=W, t=T)
        res_temporal = lin(res_temporal, "temporal_fc")
    else:
        wf, wp = sd[pre + "temporal_fc.weight"], sd[pre + "temporal_attn.proj.weight"]
        we = R(RW(wf) @ RW(wp))
        be = wf @ sd[pre + "temporal_attn.proj.bias"]
        res_temporal = orc.drop_path_apply(RB(F.linear(o, we, be)), s1)
        res_temporal = rearrange(res_temporal, "(b h w) t m -> b (h w t) m", b=B, h=H, w=W, t=T)
        res_temporal = res_temporal + sd[pre + "temporal_fc.bias"]
    xt = RX(x[:, 1:, :] + res_temporal)
    # spatial (:137-151)
    init_cls_token = x[:, 0, :].unsqueeze(1)
    cls_token = init_cls_token.repeat(1, T, 1)
    cls_token = rearrange(cls_token, "b t m -> (b t) m", b=B, t=T).unsqueeze(1)
    xs = rearrange(xt, "b (h w t) m -> (b t) (h w) m", b=B, h=H, w=W, t=T)
    xs = torch.cat((cls_token, xs), 1)
    h = R(ln(xs, "norm1"))
    o = R(attention_core(R(lin(h, "attn.qkv")), xs.shape[0], xs.shape[1], C, num_heads, mfma=True))
    res_spatial = RB(lin(o, "attn.proj"))
    if OPERAND is not None:     # the cls rows' projection runs on the fp32 master weight (csrc/cls_chain.hip), from the 16-bit o
        res_spatial = torch.cat((_value_of(linf(o[:, :1], "attn.proj"), res_spatial[:, :1]), res_spatial[:, 1:]), 1)
    res_spatial = orc.drop_path_apply(res_spatial, s2)
    cls_token = res_spatial[:, 0, :]
    cls_token = rearrange(cls_token, "(b t) m -> b t m", b=B, t=T)
    cls_token = torch.mean(cls_token, 1, True)
    res_spatial = res_spatial[:, 1:, :]
    res_spatial = rearrange(res_spatial, "(b t) (h w) m -> b (h w t) m", b=B, h=H, w=W, t=T)
    # merge + MLP (:155-157)
    x = torch.cat((init_cls_token + cls_token, RX(xt + res_spatial)), 1)
    hf = ln(x, "norm2")
    h = R(hf)
    u = RB(lin(h, "mlp.fc1"))
    g = RW(GeluStore.apply(u)) if OPERAND is not None else F.gelu(u)
    y = RB(lin(g, "mlp.fc2"))
    if OPERAND is not None:     # the cls rows' MLP in fp32 end to end: LayerNorm output, master weights, exact GELU (csrc/cls_chain.hip)
        y = torch.cat((_value_of(linf(F.gelu(linf(hf[:, :1], "mlp.fc1")), "mlp.fc2"), y[:, :1]), y[:, 1:]), 1)
    x = x + orc.drop_path_apply(y, s3)
    if OPERAND is not None and RESID is not None:
        x = torch.cat((x[:, :1], RX(x[:, 1:])), 1)
    return x


def forward_features(sd, x, depth, num_heads=12, droppath=None):
    """timesformer_oracle.forward_features (vit.py:365-423) with the rounding points of the HIP patch-embed GEMM."""
    B = x.shape[0]
    Bc, Cc, T, Hh, Ww = x.shape
    xx = rearrange(R(x), "b c t h w -> (b t) c h w")
    xx = RB(F.conv2d(xx, RW(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=16))
    W = xx.size(-1)
    xx = xx.flatten(2).transpose(1, 2)
    cls_tokens = sd["cls_token"].expand(xx.size(0), -1, -1)
    xx = torch.cat((cls_tokens, xx), dim=1)
    assert xx.size(1) == sd["pos_embed"].size(1) and T == sd["time_embed"].size(1), "resized embeddings: use the fp32 oracle"
    xx = xx + sd["pos_embed"]
    cls_tokens = xx[:B, 0, :].unsqueeze(1)
    xx = xx[:, 1:]
    xx = rearrange(xx, "(b t) n m -> (b n) t m", b=B, t=T)
    xx = xx + sd["time_embed"]
    xx = rearrange(xx, "(b n) t m -> b (n t) m", b=B, t=T)
    xx = torch.cat((cls_tokens, RX(xx)), dim=1)
    for i in range(depth):
        xx = block(sd, f"blocks.{i}.", xx, B, T, W, num_heads, None if droppath is None else droppath[i])
    xx = F.layer_norm(xx, (xx.shape[-1],), sd["norm.weight"], sd["norm.bias"], orc.LN_EPS_VIT)
    return xx[:, 0]


class operand:
    """`with rounded_oracle.operand(torch.bfloat16): ...`"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global OPERAND
        self.prev, OPERAND = OPERAND, self.dtype

    def __exit__(self, *a):
        global OPERAND
        OPERAND = self.prev
