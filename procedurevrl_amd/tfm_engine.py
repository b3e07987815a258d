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

    def level_buffers(self, levels, nseq, S, Wd, dev):
        """Saved-activation storage of `levels` independent passes over the SAME stack (the denoise levels of the pre-training head,
        tfm_model.py:165-204), level-major: pass i of forward(..., into=(bufs, i)) writes slice i, and the passes' saved tensors are
        then ONE row-wise concatenation without a copy (`merged`) -- what the batched backward over levels x nseq sequences reads."""
        rows = nseq * S
        e = lambda *shape, dtype=OP16: torch.empty((levels,) + shape, device=dev, dtype=dtype)
        return [dict(x0=e(rows, Wd, dtype=F32), h1=e(rows, Wd), st1=(e(rows, dtype=F32), e(rows, dtype=F32)), qkv=e(rows, 3 * Wd),
                     o=e(rows, Wd), lse=e(nseq, self.H, S, dtype=F32), x1=e(rows, Wd, dtype=F32), h2=e(rows, Wd),
                     st2=(e(rows, dtype=F32), e(rows, dtype=F32)), u=e(rows, 4 * Wd), g=e(rows, 4 * Wd))
                for _ in self.blocks]

    @staticmethod
    def merged(bufs, levels, nseq, S, causal, kpm):
        """the level buffers as the `saved` of one pass over levels x nseq sequences (views: flatten(0, 1) of contiguous tensors)"""
        fl = lambda t: t.flatten(0, 1)
        blocks = [{k: (tuple(fl(t) for t in v) if isinstance(v, tuple) else fl(v)) for k, v in b.items()} for b in bufs]
        return dict(blocks=blocks, nseq=nseq * levels, S=S, causal=causal, kpm=None if kpm is None else kpm.repeat(levels, 1))

    def forward(self, x, nseq, S, causal=False, kpm=None, save=True, into=None):
        """x fp32 [nseq*S, W] (batch-first rows).  Returns (y fp32 [nseq*S, W], saved).
        `into` = (level_buffers(...), i): the saved activations are written into slice i of the level buffers (x must BE
        bufs[0]["x0"][i]); `saved` is then None -- use `merged`."""
        L = lib()
        Wd = x.shape[1]
        scale = (Wd // self.H) ** -0.5
        saved = []
        P = lambda t: t.detach()
        nb = len(self.blocks)
        for j, blk in enumerate(self.blocks):
            if into is not None:
                bufs, i = into
                b = {k: (tuple(t[i] for t in v) if isinstance(v, tuple) else v[i]) for k, v in bufs[j].items()}
                assert j > 0 or x.data_ptr() == b["x0"].data_ptr()
                nxt = bufs[j + 1]["x0"][i] if j + 1 < nb else None
            else:
                b = dict(h1=None, st1=None, qkv=None, o=None, lse=None, x1=None, h2=None, st2=None, u=None, g=None)
                nxt = None
            h1, m1, r1 = ops.layernorm_fwd(x, P(blk.ln_1.weight), P(blk.ln_1.bias), LN_EPS, out=b["h1"], stats=b["st1"])
            qkv = ops.gemm_nt(h1, self._weight(blk.attn.in_proj_weight).w, L.PVRL_EPI_BF16, bias=P(blk.attn.in_proj_bias),
                              out0=b["qkv"])
            o, _, lse = ops.attn_fwd(qkv, nseq, S, self.H, scale, mode=0, causal=causal, kpm=kpm, o=b["o"], lse=b["lse"])
            x1 = ops.gemm_nt(o, self._weight(blk.attn.out_proj.weight).w, L.PVRL_EPI_RESID_F32,
                             bias=P(blk.attn.out_proj.bias), aux=x, out0=b["x1"])
            h2, m2, r2 = ops.layernorm_fwd(x1, P(blk.ln_2.weight), P(blk.ln_2.bias), LN_EPS, out=b["h2"], stats=b["st2"])
            u, g = ops.gemm_nt(h2, self._weight(blk.mlp.c_fc.weight).w, L.PVRL_EPI_QGELU, bias=P(blk.mlp.c_fc.bias),
                               out0=b["u"], out1=b["g"])
            x2 = ops.gemm_nt(g, self._weight(blk.mlp.c_proj.weight).w, L.PVRL_EPI_RESID_F32,
                             bias=P(blk.mlp.c_proj.bias), aux=x1, out0=nxt)
            if save and into is None:
                saved.append(dict(x0=x, h1=h1, st1=(m1, r1), qkv=qkv, o=o, lse=lse, x1=x1, h2=h2, st2=(m2, r2), u=u, g=g))
            x = x2
        if into is not None:
            return x, None
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

        # weight gradients: collected and issued as grouped launches of up to TN_GROUP_MAX problems (a four-layer stack: 16 problems
        # = two launches + two reduces instead of 16 + 16; ops.gemm_tn_grouped falls back to single launches for shapes off the
        # 256-multiples)
        wq = []

        def wgrad(d, xin, w, b):
            (dw, bw), (db, _) = tgt(w), tgt(b)
            wq.append((d, xin, dw, db, bw, gsc, bad))

        def flush():
            for a in range(0, len(wq), ops.TN_GROUP_MAX):
                ops.gemm_tn_grouped(wq[a:a + ops.TN_GROUP_MAX], ws_tag="tn_group_stack")
            del wq[:]

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
            if len(wq) >= ops.TN_GROUP_MAX:
                flush()
        flush()
        return dx
