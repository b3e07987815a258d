"""Kernel sequencing for CLIP-style residual attention stacks (width 512, 8 heads of 64).

Reference: `ResidualAttentionBlock` / `TemporalModelling`, lib/models/tfm_model.py:32-67 (itself an
adaptation of openai/CLIP's Transformer) -- used (a) as the 4-layer order / diffusion transformer
over the 9 clip embeddings of a video (tfm_model.py:70-204) and (b) with a causal mask and 12
layers as the frozen CLIP text tower behind `clip.encode_text` (lib/models/vit.py:258-261, 428).

    x = x + out_proj(MHA(ln_1(x)));   x = x + c_proj(QuickGELU(c_fc(ln_2(x))))

Rows are batch-first here (row = seq * S + t) so that the attention kernel addresses each
sequence as contiguous rows; the reference runs sequence-first [t, b, c] through
nn.MultiheadAttention, which is the same arithmetic.  LayerNorm eps is 1e-5 and computed in
fp32 (tfm_model.py:18-24), activations between GEMMs are bf16, the residual stream is fp32.
"""
import torch

from . import ops
from ._lib import lib

OP16 = ops.OP16
F32 = torch.float32
LN_EPS = 1e-5


class StackEngine:
    """Runs a list of residual attention blocks (objects with ln_1, attn.in_proj_weight/bias,
    attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj).  Shares the weight-copy cache of `enc` (EncoderEngine)."""

    def __init__(self, resblocks, weight_cache, grad_target=None, heads=8, grad_store=None):
        """`grad_store` (optional callable -> engine.GradStore): parameter gradients are then written in "fused" form -- the kernels
        take the fp16 flavour's backward scale out themselves and raise the optimiser's skip flag on a non-finite value
        (GradStore.target(..., fused=True)); without it `grad_target(p)` -> (tensor, beta) as before."""
        self.blocks = list(resblocks)
        self._weight = weight_cache
        self._target = grad_target
        self._gs = grad_store
        self.H = heads

    def forward(self, x, nseq, S, causal=False, kpm=None, save=True):
        """x fp32 [nseq*S, W] (batch-first rows).  Returns (y fp32 [nseq*S, W], saved)."""
        L = lib()
        Wd = x.shape[1]
        scale = (Wd // self.H) ** -0.5
        saved = []
        P = lambda t: t.detach()
        for blk in self.blocks:
            h1, m1, r1 = ops.layernorm_fwd(x, P(blk.ln_1.weight), P(blk.ln_1.bias), LN_EPS)
            qkv = ops.gemm_nt(h1, self._weight(blk.attn.in_proj_weight).w, L.PVRL_EPI_BF16, bias=P(blk.attn.in_proj_bias))
            o, _, lse = ops.attn_fwd(qkv, nseq, S, self.H, scale, mode=0, causal=causal, kpm=kpm)
            x1 = ops.gemm_nt(o, self._weight(blk.attn.out_proj.weight).w, L.PVRL_EPI_RESID_F32,
                             bias=P(blk.attn.out_proj.bias), aux=x)
            h2, m2, r2 = ops.layernorm_fwd(x1, P(blk.ln_2.weight), P(blk.ln_2.bias), LN_EPS)
            u, g = ops.gemm_nt(h2, self._weight(blk.mlp.c_fc.weight).w, L.PVRL_EPI_QGELU, bias=P(blk.mlp.c_fc.bias))
            x2 = ops.gemm_nt(g, self._weight(blk.mlp.c_proj.weight).w, L.PVRL_EPI_RESID_F32,
                             bias=P(blk.mlp.c_proj.bias), aux=x1)
            if save:
                saved.append(dict(x0=x, h1=h1, st1=(m1, r1), qkv=qkv, o=o, lse=lse, x1=x1, h2=h2, st2=(m2, r2), u=u, g=g))
            x = x2
        return x, dict(blocks=saved, nseq=nseq, S=S, causal=causal, kpm=kpm)

    def backward(self, dy, saved):
        """dy fp32 [rows, W] -> dx fp32; parameter gradients go to grad_target(p) -> (tensor, beta)."""
        L = lib()
        nseq, S, causal, kpm = saved["nseq"], saved["S"], saved["causal"], saved["kpm"]
        Wd = dy.shape[1]
        scale = (Wd // self.H) ** -0.5
        P = lambda t: t.detach()
        dx = dy.contiguous().clone()
        gs = self._gs() if self._gs is not None else None
        tgt = self._target if gs is None else (lambda p: gs.target(p, fused=True))
        gsc, bad = (gs.inv, gs.bad) if gs is not None else (None, None)

        def wgrad(d, xin, w, b):
            (dw, bw), (db, _) = tgt(w), tgt(b)
            ops.gemm_tn(d, xin, dw, db, beta=bw, gscale=gsc, nonfinite=bad)

        def lnbwd(dh, x, st, ln):
            (dg, bg), (db, _) = tgt(ln.weight), tgt(ln.bias)
            ops.layernorm_bwd(dh, x, st[0], st[1], P(ln.weight), dg, db, dx_in=dx, dx_out=dx, beta_acc=bg, gscale=gsc,
                              nonfinite=bad)

        for blk, s in zip(reversed(self.blocks), reversed(saved["blocks"])):
            d2 = ops.cast_scale(dx, None)
            wgrad(d2, s["g"], blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)
            du = ops.gemm_nt(d2, self._weight(blk.mlp.c_proj.weight).t, L.PVRL_EPI_DQGELU, aux=s["u"])
            wgrad(du, s["h2"], blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
            dh = ops.gemm_nt(du, self._weight(blk.mlp.c_fc.weight).t, L.PVRL_EPI_BF16)
            lnbwd(dh, s["x1"], s["st2"], blk.ln_2)
            d1 = ops.cast_scale(dx, None)
            wgrad(d1, s["o"], blk.attn.out_proj.weight, blk.attn.out_proj.bias)
            do = ops.gemm_nt(d1, self._weight(blk.attn.out_proj.weight).t, L.PVRL_EPI_BF16)
            dqkv, _ = ops.attn_bwd(s["qkv"], s["o"], None, do, None, s["lse"], nseq, S, self.H, scale, mode=0,
                                   causal=causal, kpm=kpm)
            wgrad(dqkv, s["h1"], blk.attn.in_proj_weight, blk.attn.in_proj_bias)
            dh = ops.gemm_nt(dqkv, self._weight(blk.attn.in_proj_weight).t, L.PVRL_EPI_BF16)
            lnbwd(dh, s["x0"], s["st1"], blk.ln_1)
        return dx
