"""MViTv2 video encoder + step-matching head, MI355X-native (BASELINE config 5).

Drop-in for the reference's `lib/models/mvit.py` (wrapper `VisionTransformer` :45, registered `MViT` :231) over
`lib/models/slowfast_mvit/mvit.py:MViT_encoder` (:29) / `attention.py:MultiScaleBlock` (:445) /
`MultiScaleAttention` (:162): same module tree and `state_dict()` keys (`model.video_encoder.blocks.{i}.attn.pool_q.weight`,
`...rel_pos_h`, ...), same cfg keys (`MVIT.*`), same call signature and outputs as the TimeSformer wrapper.
The sub-modules own parameters only; the arithmetic is `MViTEngine`'s kernel schedule over libpvrl_hip.so
(csrc/mvit.hip, csrc/attn_pool.hip and the shared bf16 MFMA GEMMs).  No PyTorch fallback.

Built for the configuration every shipped MViT yaml uses (configs/HowTo100M/procedurevrl_mvitv2_*.yaml): MODE conv,
CLS_EMBED_ON, no absolute position embedding, REL_POS_SPATIAL + REL_POS_TEMPORAL, RESIDUAL_POOLING, DIM_MUL_IN_ATT,
POOL_KVQ_KERNEL (3,3,3), head_dim 96, DROPOUT_RATE 0 (DROPPATH_RATE: any); other settings raise NotImplementedError.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from . import ops
from . import ops_mvit as om
from ._lib import lib
from .build import MODEL_REGISTRY
from .engine import EncoderEngine, GradStore, GraphReplay
from .vit import VisionTransformer as _StepMatchingModel, trunc_normal_

OP16 = ops.OP16
F32 = torch.float32
HD = 96


# ------------------------------------------------------------------------------------------------ geometry
def round_width(width, multiplier, min_width=1, divisor=1):
    """slowfast_mvit/utils.py:7-20"""
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def mvit_plan(cfg):
    """Per-block geometry as MViT_encoder.__init__ derives it (slowfast_mvit/mvit.py:96-239):
    -> (patch thw, [dict(dim, dim_out, heads, stride_q, stride_kv, in_thw, has_q_stride)])."""
    mv = cfg.MVIT
    depth = int(mv.DEPTH)
    ps = list(mv.PATCH_STRIDE)
    thw = [cfg.DATA.NUM_FRAMES // ps[0], cfg.DATA.TRAIN_CROP_SIZE // ps[1], cfg.DATA.TRAIN_CROP_SIZE // ps[2]]
    dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
    for e in mv.DIM_MUL:
        dim_mul[int(e[0])] = float(e[1])
    for e in mv.HEAD_MUL:
        head_mul[int(e[0])] = float(e[1])
    stride_q = [[] for _ in range(depth)]
    for e in mv.POOL_Q_STRIDE:
        stride_q[int(e[0])] = [int(v) for v in e[1:]]
    stride_kv = [[] for _ in range(depth)]
    if mv.POOL_KV_STRIDE_ADAPTIVE is not None:
        cur = [int(v) for v in mv.POOL_KV_STRIDE_ADAPTIVE]
        for i in range(depth):
            if len(stride_q[i]) > 0:
                cur = [max(cur[d] // stride_q[i][d], 1) for d in range(3)]
            stride_kv[i] = list(cur)
    else:
        for e in mv.POOL_KV_STRIDE:
            stride_kv[int(e[0])] = [int(v) for v in e[1:]]
    embed_dim, heads = int(mv.EMBED_DIM), int(mv.NUM_HEADS)
    blocks, size = [], list(thw)
    for i in range(depth):
        heads = round_width(heads, head_mul[i])
        dim_out = round_width(embed_dim, dim_mul[i], divisor=round_width(heads, head_mul[i]))
        blocks.append(dict(dim=embed_dim, dim_out=dim_out, heads=heads, stride_q=stride_q[i] or [1, 1, 1],
                           stride_kv=stride_kv[i] or [1, 1, 1], in_thw=list(size), has_q=len(stride_q[i]) > 0,
                           has_kv=len(stride_kv[i]) > 0))
        if len(stride_q[i]) > 0:
            size = [s // st for s, st in zip(size, stride_q[i])]
        embed_dim = dim_out
    return thw, blocks


def rel_index(q_n, k_n):
    """Index table of the decomposed relative position embedding (attention.py:80-92,130-137), int32 [q_n, k_n]."""
    q_ratio = max(k_n / q_n, 1.0)
    k_ratio = max(q_n / k_n, 1.0)
    d = torch.arange(q_n)[:, None] * q_ratio - torch.arange(k_n)[None, :] * k_ratio + (k_n - 1) * k_ratio
    return d.long().to(torch.int32).contiguous()


def _check_cfg(cfg):
    mv = cfg.MVIT
    bad = []
    if mv.MODE != "conv": bad.append("MODE != conv")
    if mv.POOL_FIRST: bad.append("POOL_FIRST")
    if not mv.CLS_EMBED_ON: bad.append("CLS_EMBED_ON False")
    if mv.USE_ABS_POS: bad.append("USE_ABS_POS")
    if not (mv.REL_POS_SPATIAL and mv.REL_POS_TEMPORAL): bad.append("REL_POS_SPATIAL/TEMPORAL off")
    if not mv.RESIDUAL_POOLING: bad.append("RESIDUAL_POOLING off")
    if not mv.DIM_MUL_IN_ATT: bad.append("DIM_MUL_IN_ATT off")
    if mv.SEPARATE_QKV: bad.append("SEPARATE_QKV")
    if mv.POOL_KVQ_KERNEL is None or list(mv.POOL_KVQ_KERNEL) != [3, 3, 3]: bad.append("POOL_KVQ_KERNEL != [3,3,3]")
    if float(mv.DROPOUT_RATE) != 0.0: bad.append("DROPOUT_RATE > 0")
    if float(mv.LAYER_SCALE_INIT_VALUE) != 0.0: bad.append("LAYER_SCALE_INIT_VALUE > 0")
    if mv.NORM_STEM or mv.USE_MEAN_POOLING or mv.PATCH_2D or mv.NORM != "layernorm": bad.append("NORM_STEM/USE_MEAN_POOLING/PATCH_2D/NORM")
    if not mv.QKV_BIAS: bad.append("QKV_BIAS False")
    if bad:
        raise NotImplementedError("MViT on the HIP path is built for the shipped MViTv2-S configuration "
                                  "(configs/HowTo100M/procedurevrl_mvitv2_*.yaml); unsupported here: " + ", ".join(bad))


# ------------------------------------------------------------------------------------------------ parameter containers
class PatchEmbed(nn.Module):
    """slowfast_mvit/stem_helper.py:290-321"""

    def __init__(self, dim_in, dim_out, kernel, stride, padding):
        super().__init__()
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=tuple(kernel), stride=tuple(stride), padding=tuple(padding))


class Mlp(nn.Module):
    """slowfast_mvit/common.py:7-35"""

    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class MultiScaleAttention(nn.Module):
    """slowfast_mvit/attention.py:162-305 (parameters only)"""

    def __init__(self, dim, dim_out, input_size, num_heads, stride_q, stride_kv, norm_layer):
        super().__init__()
        self.num_heads = num_heads
        self.dim_out = dim_out
        head_dim = dim_out // num_heads
        assert head_dim == HD, f"the pooling-attention kernels are built for head_dim 96, got {head_dim}"
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim_out * 3, bias=True)
        self.proj = nn.Linear(dim_out, dim_out)
        for n, st in (("q", stride_q), ("k", stride_kv), ("v", stride_kv)):
            setattr(self, "pool_" + n, nn.Conv3d(head_dim, head_dim, (3, 3, 3), stride=tuple(st), padding=(1, 1, 1),
                                                 groups=head_dim, bias=False))
            setattr(self, "norm_" + n, norm_layer(head_dim))
        size = input_size[1]
        q_size, kv_size = size // stride_q[1], size // stride_kv[1]
        rel_sp_dim = 2 * max(q_size, kv_size) - 1
        self.rel_pos_h = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
        self.rel_pos_w = nn.Parameter(torch.zeros(rel_sp_dim, head_dim))
        self.rel_pos_t = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
        trunc_normal_(self.rel_pos_h, std=0.02)
        trunc_normal_(self.rel_pos_w, std=0.02)
        trunc_normal_(self.rel_pos_t, std=0.02)


class MultiScaleBlock(nn.Module):
    """slowfast_mvit/attention.py:445-543 with dim_mul_in_att (parameters only)"""

    def __init__(self, dim, dim_out, num_heads, input_size, mlp_ratio, stride_q, stride_kv, norm_layer):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.norm1 = norm_layer(dim)
        self.attn = MultiScaleAttention(dim, dim_out, input_size, num_heads, stride_q, stride_kv, norm_layer)
        self.norm2 = norm_layer(dim_out)
        self.mlp = Mlp(dim_out, int(dim_out * mlp_ratio), dim_out)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)


class MViT_encoder(nn.Module):
    """slowfast_mvit/mvit.py:29-298 (parameters + geometry); forward = MViTEngine."""

    def __init__(self, cfg):
        super().__init__()
        _check_cfg(cfg)
        assert cfg.DATA.TRAIN_CROP_SIZE == cfg.DATA.TEST_CROP_SIZE
        self.cfg = cfg
        mv = cfg.MVIT
        self.patch_dims, self.plan = mvit_plan(cfg)
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.ln_eps = 1e-6
        e0 = int(mv.EMBED_DIM)
        self.patch_embed = PatchEmbed(cfg.DATA.INPUT_CHANNEL_NUM[0], e0, mv.PATCH_KERNEL, mv.PATCH_STRIDE, mv.PATCH_PADDING)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, e0))
        self.blocks = nn.ModuleList([
            MultiScaleBlock(b["dim"], b["dim_out"], b["heads"], b["in_thw"], float(mv.MLP_RATIO), b["stride_q"],
                            b["stride_kv"], norm_layer) for b in self.plan])
        self.norm = norm_layer(self.plan[-1]["dim_out"])
        self.drop_path_rates = [x.item() for x in torch.linspace(0, float(mv.DROPPATH_RATE), int(mv.DEPTH))]   # mvit.py:104-106
        trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """slowfast_mvit/mvit.py:283-291"""
        if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.constant_(m.bias, 0.02)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0.02)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        names = []
        if self.cfg.MVIT.ZERO_DECAY_POS_CLS:
            names += ["rel_pos_h", "rel_pos_w", "rel_pos_hw", "rel_pos_t", "cls_token"]
        return names


# ------------------------------------------------------------------------------------------------ engine
class _PW:
    """zero-padded bf16 operand copies of one weight: w [Np, Kp] (forward), t [Kp, Np] (data gradient), bias [Np] fp32"""
    __slots__ = ("w", "t", "b", "ver", "N", "K")


class MViTEngine(GraphReplay):
    """Kernel schedule of MViT_encoder.forward (slowfast_mvit/mvit.py:346-407) and its hand-written backward.
    Token matrices are fp32 [B*L + B, pad128(C)]: patch tokens (b, t, h, w) first, the B cls tokens last.
    The ~3,000 launches of a step are replayed from HIP graphs (engine.GraphReplay; with a data-parallel gradient hook
    installed the backward is one graph per block, the hook running between them)."""

    _weight = EncoderEngine._weight      # un-padded bf16 copies for the width-512 stacks (order transformer, text tower)
    refresh_params = EncoderEngine.refresh_params

    def __init__(self, owner, enc):
        self.m = owner                   # the wrapper (weights_epoch, grad_target)
        self.enc = enc
        self._w = {}
        self._pw = {}
        self._idx = {}
        self.saved = None
        self.grad_hook = None
        # only the cls row of the last block's output is read (norm + x[:, 0], slowfast_mvit/mvit.py:400-407): its projection and MLP run
        # on the B cls rows alone, forward and backward (engine.EncoderEngine.prune_last; PVRL_PRUNE_LAST=0: A/B runs)
        self.prune_last = os.environ.get("PVRL_PRUNE_LAST", "1") == "1"
        self.dxn16 = os.environ.get("PVRL_MVIT_DXN16", "1") == "1"
        self._graph_init()

    # -------------------------------------------------------------- HIP graphs (engine.GraphReplay)
    def _eager_forward(self, frames, training, save):
        return self._forward(frames, training, save, None)

    def _eager_backward(self, dfeat):
        return self._backward(dfeat)

    def _graph_key(self, frames, training, save):
        enc = self.enc
        return (tuple(frames.shape), frames.dtype, bool(training), bool(save), frames.device.index, tuple(enc.drop_path_rates),
                enc.blocks[0].attn.qkv.weight.data_ptr(), enc.norm.weight.data_ptr(), self.m.grad_store().flat.data_ptr())

    def _enc_params(self):
        return [p for p in self.enc.parameters() if p.requires_grad]

    def forward(self, frames, training, save=True, droppath=None):
        if self.use_graphs and droppath is None and isinstance(frames, torch.Tensor) and frames.is_cuda:
            return self._graph_forward(frames, training, save)
        self._gkey = None
        return self._forward(frames, training, save, droppath)

    def backward(self, dfeat):
        if self._gkey is not None:
            return self._graph_backward(dfeat)
        return self._backward(dfeat)

    # -------------------------------------------------------------- weights
    def _wpad(self, weight, bias=None, Np=None, Kp=None):
        e = self._pw.get(id(weight))
        ver = (weight._version, bias._version if bias is not None else -1, getattr(self.m, "weights_epoch", 0),
               weight.data_ptr())
        if self._capturing == "bwd":      # the forward graph of the same step refreshed the padded copies
            assert e is not None
            return e
        if self._capturing == "fwd" or e is None or e.ver != ver or e.w.device != weight.device:
            w2 = weight.detach().reshape(weight.shape[0], -1).contiguous()
            N, K = w2.shape
            Np = om.pad128(N) if Np is None else Np
            Kp = om.pad128(K) if Kp is None else Kp
            if e is None or e.w.device != weight.device or tuple(e.w.shape) != (Np, Kp):
                e = _PW()
                e.w = torch.zeros((Np, Kp), device=weight.device, dtype=OP16)
                e.t = torch.zeros((Kp, Np), device=weight.device, dtype=OP16)
                e.b = torch.zeros(Np, device=weight.device, dtype=F32)
                self._pw[id(weight)] = e
            lib().call("pvrl_cast_weight_pad_bf16", ops._ptr(w2), ops._ptr(e.w), Kp, ops._ptr(e.t), Np, N, K,
                       ops._ptr(bias.detach()) if bias is not None else None, ops._ptr(e.b) if bias is not None else None,
                       ops._stream())
            e.N, e.K, e.ver = N, K, ver
        return e

    def _rel_idx(self, i, blk, q_thw, k_thw, dev):
        c = self._idx.get(i)
        if c is None or c[0].device != dev:
            c = (rel_index(q_thw[1], k_thw[1]).to(dev), rel_index(q_thw[2], k_thw[2]).to(dev), rel_index(q_thw[0], k_thw[0]).to(dev))
            self._idx[i] = c
        return c

    # -------------------------------------------------------------- gradient plumbing
    def _grad(self, p):
        return self.m.grad_target(p)

    def _wgrad(self, P, Q, weight, bias, e):
        """dW = P^T Q on the padded operands, reduced straight into the [N, K] weight.grad (+ bias.grad from the column sums)"""
        gw, bw = self._grad(weight)
        gb, bb = self._grad(bias) if bias is not None else (None, 0.0)
        ops.gemm_tn_into(P, Q, gw.view(e.N, e.K), e.N, e.K, dbias=None if gb is None else gb.view(-1), beta=bw, beta_bias=bb)

    def _acc_target(self, p):
        """fp32 buffer that a kernel ACCUMULATES into (atomicAdd): zeroed first unless it already holds this step's sum"""
        g, beta = self._grad(p)
        if beta == 0.0:
            g.zero_()
        return g

    # -------------------------------------------------------------- forward
    def draw_droppath(self, B, device, training):
        """Per block (s_attn, s_mlp), each [B] = floor(keep + U[0,1)) / keep (slowfast_mvit/common.py:38-52: one draw per
        DropPath call, per sample), or None where the block's rate is 0 / in eval mode."""
        out = []
        for rate in self.enc.drop_path_rates:
            if not training or rate == 0.0:
                out.append(None)
            else:
                keep = 1.0 - rate
                out.append(tuple(torch.floor(keep + torch.rand(B, device=device)) / keep for _ in range(2)))
        return out

    def _forward(self, frames, training, save=True, droppath=None):
        L = lib()
        enc = self.enc
        mv = enc.cfg.MVIT
        eps = enc.ln_eps
        frames = frames.contiguous()
        B = frames.shape[0]
        dev = frames.device
        e0 = enc.plan[0]["dim"]
        Cp = om.pad128(e0)
        wpe = self._wpad(enc.patch_embed.proj.weight, enc.patch_embed.proj.bias, Np=Cp, Kp=512 * ((enc.patch_embed.proj.weight[0].numel() + 511) // 512))
        a_pe, thw = om.im2col3d(frames, tuple(mv.PATCH_KERNEL), tuple(mv.PATCH_STRIDE), tuple(mv.PATCH_PADDING), wpe.w.shape[1])
        assert list(thw) == list(enc.patch_dims), (thw, enc.patch_dims)
        R = a_pe.shape[0]
        x = torch.empty((R + B, Cp), device=dev, dtype=F32)
        ops.gemm_nt(a_pe, wpe.w, L.PVRL_EPI_F32, bias=wpe.b, out0=x[:R])
        x[R:].zero_()
        x[R:, :e0] = enc.cls_token.detach()[0, 0]
        sv = dict(B=B, a_pe=a_pe if save else None, blocks=[])
        if droppath is None:
            droppath = self.draw_droppath(B, dev, training)
        for i, (blk, pl) in enumerate(zip(enc.blocks, enc.plan)):
            x = self._block_fwd(i, blk, pl, x, B, sv, save, droppath[i])
        Cl = enc.plan[-1]["dim_out"]
        Rl = x.shape[0] - B
        feat, mean, rstd = om.ln_fwd(x[Rl:], Cl, enc.norm.weight.detach(), enc.norm.bias.detach(), eps, out_dtype=F32)
        if save:
            sv.update(x_final=x, f_mean=mean, f_rstd=rstd)
            self.saved = sv
        return feat

    def _block_fwd(self, i, blk, pl, x, B, sv, save, dp=None):
        L = lib()
        eps = self.enc.ln_eps
        dim, dout, H = pl["dim"], pl["dim_out"], pl["heads"]
        thw, sq, skv = pl["in_thw"], pl["stride_q"], pl["stride_kv"]
        Cpi, Cpo = om.pad128(dim), om.pad128(dout)
        a = blk.attn
        P = lambda t: t.detach()
        xn, mean1, rstd1 = om.ln_fwd(x, dim, P(blk.norm1.weight), P(blk.norm1.bias), eps, Cpad=Cpi)
        wqkv = self._wpad(a.qkv.weight, a.qkv.bias)
        qkv = ops.gemm_nt(xn, wqkv.w, L.PVRL_EPI_BF16, bias=wqkv.b)
        pw = lambda c: P(c.weight).reshape(HD, 27)
        q, cq = om.pool_fwd(qkv, 0, B, H, thw, sq, pw(a.pool_q), P(a.norm_q.weight), P(a.norm_q.bias), eps)
        k, ck = om.pool_fwd(qkv, dout, B, H, thw, skv, pw(a.pool_k), P(a.norm_k.weight), P(a.norm_k.bias), eps)
        v, cv = om.pool_fwd(qkv, 2 * dout, B, H, thw, skv, pw(a.pool_v), P(a.norm_v.weight), P(a.norm_v.bias), eps)
        q_thw, k_thw = om.pool_out_thw(thw, sq), om.pool_out_thw(thw, skv)
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        ih, iw, it = self._rel_idx(i, blk, q_thw, k_thw, x.device)
        rel = om.rel_fwd(q, B * H, q_thw, k_thw, P(a.rel_pos_h), P(a.rel_pos_w), P(a.rel_pos_t), ih, iw, it,
                         out_scale=1.0 / a.scale)
        o, lse = om.attn_fwd(q, k, v, rel, B, H, Lq, k_thw, a.scale, Cpo)
        if dim != dout:
            wsk = self._wpad(blk.proj.weight, blk.proj.bias)
            xs = ops.gemm_nt(xn, wsk.w, L.PVRL_EPI_F32, bias=wsk.b)
        else:
            xs = x
        pooled = max(sq) > 1
        if pooled:
            assert sq[0] == 1 and sq[1] == sq[2], "max-pool skip kernel is built for stride (1, s, s)"
            if save:
                xres, amax = om.maxpool_fwd(xs, B, thw, sq[1], dout, want_argmax=True)
            else:
                xres, amax = om.maxpool_fwd(xs, B, thw, sq[1], dout), None
        else:
            xres, amax = xs, None
        # DropPath: one factor per clip, expanded to the output rows (patch tokens (b, l) then the cls rows)
        rs_a = rs_m = None
        if dp is not None:
            rs_a = torch.cat((dp[0].float().repeat_interleave(Lq), dp[0].float())).contiguous()
            rs_m = torch.cat((dp[1].float().repeat_interleave(Lq), dp[1].float())).contiguous()
        wproj = self._wpad(a.proj.weight, a.proj.bias)
        w1 = self._wpad(blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        w2 = self._wpad(blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        prune = self.prune_last and i == len(self.enc.blocks) - 1
        if prune:
            # the last block: x1 / x2 are defined on the cls rows only; xn2 / mean2 / rstd2 / u / g are [B, .] tensors (_block_bwd)
            Rc = o.shape[0] - B
            cs = lambda t: None if t is None else t[Rc:]
            x1 = torch.empty((o.shape[0], Cpo), device=x.device, dtype=F32)
            ops.gemm_nt(o[Rc:], wproj.w, L.PVRL_EPI_RESID_F32, bias=wproj.b, rowscale=cs(rs_a), aux=xres[Rc:], out0=x1[Rc:])
            xn2, mean2, rstd2 = om.ln_fwd(x1[Rc:], dout, P(blk.norm2.weight), P(blk.norm2.bias), eps, Cpad=Cpo)
            u, g = ops.gemm_nt(xn2, w1.w, L.PVRL_EPI_GELU, bias=w1.b)
            x2 = torch.empty_like(x1)
            ops.gemm_nt(g, w2.w, L.PVRL_EPI_RESID_F32, bias=w2.b, rowscale=cs(rs_m), aux=x1[Rc:], out0=x2[Rc:])
        else:
            x1 = ops.gemm_nt(o, wproj.w, L.PVRL_EPI_RESID_F32, bias=wproj.b, rowscale=rs_a, aux=xres)
            xn2, mean2, rstd2 = om.ln_fwd(x1, dout, P(blk.norm2.weight), P(blk.norm2.bias), eps, Cpad=Cpo)
            u, g = ops.gemm_nt(xn2, w1.w, L.PVRL_EPI_GELU, bias=w1.b)
            x2 = ops.gemm_nt(g, w2.w, L.PVRL_EPI_RESID_F32, bias=w2.b, rowscale=rs_m, aux=x1)
        if save:
            sv["blocks"].append(dict(x=x, xn=xn, mean1=mean1, rstd1=rstd1, qkv=qkv, q=q, k=k, v=v, cq=cq, ck=ck, cv=cv,
                                     rel=rel, o=o, lse=lse, xs=xs if pooled else None, amax=amax, x1=x1, xn2=xn2, mean2=mean2,
                                     rstd2=rstd2, u=u, g=g, q_thw=q_thw, k_thw=k_thw, pooled=pooled, rs_a=rs_a, rs_m=rs_m,
                                     pruned=prune))
        return x2

    # -------------------------------------------------------------- backward
    # The backward in three kinds of stages -- begin (final norm), one per block, end (stem) -- so that engine.GraphReplay can
    # capture one HIP graph per block and run the data-parallel reducer's hook between them (staged capture); without a
    # hook the whole backward is one graph.
    def _bwd_begin(self, dfeat):
        enc = self.enc
        sv = self.saved
        if sv is None:
            raise RuntimeError("MViTEngine.backward without a saved forward")
        self.saved = None
        B = sv["B"]
        xf = sv["x_final"]
        Cl = enc.plan[-1]["dim_out"]
        Rl = xf.shape[0] - B
        dx = torch.zeros_like(xf)
        from .engine import SCALED_GRADS
        gs = self.m.grad_store() if SCALED_GRADS else None
        if gs is not None:
            dfeat = gs.begin_scaled(dfeat.float())     # fp16-operand flavour: backward in S-scaled units (GradStore.begin_scaled)
        # every encoder parameter receives a gradient below, most of them through accumulating kernels: one fill per
        # contiguous run of the flat gradient buffer instead of ~320 five-microsecond fills
        self.m.grad_store().prezero([p for p in enc.parameters() if p.requires_grad])
        dgn, dbn = self._acc_target(enc.norm.weight), self._acc_target(enc.norm.bias)
        dx[Rl:] = om.ln_bwd(dfeat.contiguous().float(), xf[Rl:], Cl, sv["f_mean"], sv["f_rstd"], enc.norm.weight.detach(),
                            dgn, dbn, Cpad=xf.shape[1])
        return dict(sv=sv, B=B, dx=dx, dx16=None, gs=gs)

    def _bwd_block(self, st, i):
        enc, sv = self.enc, st["sv"]
        # the block's last kernel also writes the 16-bit operand copy the block below starts from (its DropPath factor in)
        nxt = sv["blocks"][i - 1]["rs_m"] if i > 0 else None
        st["dx"], st["dx16"] = self._block_bwd(i, enc.blocks[i], enc.plan[i], sv["blocks"][i], st["dx"], st["B"], st["dx16"],
                                               True, nxt)
        sv["blocks"][i] = None
        if st["gs"] is not None:
            st["gs"].unscale()

    def _bwd_end(self, st):
        # patch embed (weight gradient only: the input needs none) and cls token
        enc, sv, B, dx, dx16 = self.enc, st["sv"], st["B"], st["dx"], st["dx16"]
        e0 = enc.plan[0]["dim"]
        R = dx.shape[0] - B
        pw = enc.patch_embed.proj.weight      # same padded shape as the forward asked for (one cache entry, not two that evict each other)
        wpe = self._wpad(pw, enc.patch_embed.proj.bias, Np=om.pad128(e0), Kp=512 * ((pw[0].numel() + 511) // 512))
        dxb = dx16[:R] if dx16 is not None else ops.cast_scale(dx[:R])     # block 0's last LayerNorm backward wrote the 16-bit copy
        self._wgrad(dxb, sv["a_pe"], enc.patch_embed.proj.weight, enc.patch_embed.proj.bias, wpe)
        gc, bc = self._grad(enc.cls_token)
        s = ops.batch_sum(dx[R:], B, 1)
        om.copy2d(s.view(1, -1), gc.view(1, -1), 1, e0, beta=bc)
        if st["gs"] is not None:
            st["gs"].end_scaled()

    def join_side_stream(self):
        pass                                   # this engine issues everything on one stream

    def _backward(self, dfeat):
        st = self._bwd_begin(dfeat)
        for i in range(len(self.enc.blocks) - 1, -1, -1):
            self._bwd_block(st, i)
            if self.grad_hook is not None:
                self.grad_hook(i)
        self._bwd_end(st)

    def _block_bwd(self, i, blk, pl, s, dx2, B, dx2_16=None, want16=False, next_rs=None):
        L = lib()
        eps = self.enc.ln_eps
        dim, dout, H = pl["dim"], pl["dim_out"], pl["heads"]
        thw, sq, skv = pl["in_thw"], pl["stride_q"], pl["stride_kv"]
        Cpi, Cpo = om.pad128(dim), om.pad128(dout)
        a = blk.attn
        P = lambda t: t.detach()
        # ---- MLP
        w1 = self._wpad(blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        w2 = self._wpad(blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        wproj = self._wpad(a.proj.weight, a.proj.bias)
        if s.get("pruned"):
            # the last block under prune_last: dx2 is zero outside the B cls rows, so the MLP, norm2 and the projection back-propagate
            # those rows alone (the saved xn2 / u / g are [B, .]); no gradient reaches the patch queries' attention outputs
            assert dx2_16 is None
            Rc = dx2.shape[0] - B
            cs = lambda t: None if t is None else t[Rc:]
            dyb = ops.cast_scale(dx2[Rc:], rowscale=cs(s["rs_m"]))
            self._wgrad(dyb, s["g"], blk.mlp.fc2.weight, blk.mlp.fc2.bias, w2)
            du = ops.gemm_nt(dyb, w2.t, L.PVRL_EPI_DGELU, aux=s["u"])
            self._wgrad(du, s["xn2"], blk.mlp.fc1.weight, blk.mlp.fc1.bias, w1)
            dxn2 = ops.gemm_nt(du, w1.t, L.PVRL_EPI_BF16)
            dx1c, dx1b = om.ln_bwd(dxn2, s["x1"][Rc:], dout, s["mean2"], s["rstd2"], P(blk.norm2.weight),
                                   self._acc_target(blk.norm2.weight), self._acc_target(blk.norm2.bias), dres=dx2[Rc:], Cpad=Cpo,
                                   want16=True, rowscale16=cs(s["rs_a"]))
            dx1 = dx2                      # zero on the patch rows already
            dx1[Rc:] = dx1c
            self._wgrad(dx1b, s["o"][Rc:], a.proj.weight, a.proj.bias, wproj)
            d_o = torch.zeros_like(s["o"])
            ops.gemm_nt(dx1b, wproj.t, L.PVRL_EPI_BF16, out0=d_o[Rc:])
        else:
            dyb = dx2_16 if dx2_16 is not None else ops.cast_scale(dx2, rowscale=s["rs_m"])
            self._wgrad(dyb, s["g"], blk.mlp.fc2.weight, blk.mlp.fc2.bias, w2)
            du = ops.gemm_nt(dyb, w2.t, L.PVRL_EPI_DGELU, aux=s["u"])
            self._wgrad(du, s["xn2"], blk.mlp.fc1.weight, blk.mlp.fc1.bias, w1)
            dxn2 = ops.gemm_nt(du, w1.t, L.PVRL_EPI_BF16)
            dx1, dx1b = om.ln_bwd(dxn2, s["x1"], dout, s["mean2"], s["rstd2"], P(blk.norm2.weight), self._acc_target(blk.norm2.weight),
                                  self._acc_target(blk.norm2.bias), dres=dx2, Cpad=Cpo, want16=True, rowscale16=s["rs_a"])
            # ---- attention output projection
            self._wgrad(dx1b, s["o"], a.proj.weight, a.proj.bias, wproj)
            d_o = ops.gemm_nt(dx1b, wproj.t, L.PVRL_EPI_BF16)
        # ---- pooling attention, rel-pos terms, pooling convs
        q_thw, k_thw = s["q_thw"], s["k_thw"]
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        dq, dk, dv, drel = om.attn_bwd(s["q"], s["k"], s["v"], s["rel"], B, H, Lq, k_thw, a.scale, s["o"], d_o, s["lse"])
        ih, iw, it = self._rel_idx(i, blk, q_thw, k_thw, dx2.device)
        om.rel_bwd(drel, s["q"], dq, B * H, q_thw, k_thw, P(a.rel_pos_h), P(a.rel_pos_w), P(a.rel_pos_t), ih, iw, it,
                   self._acc_target(a.rel_pos_h), self._acc_target(a.rel_pos_w), self._acc_target(a.rel_pos_t))
        qkv = s["qkv"]
        dqkv = torch.empty_like(qkv)       # the three pool backwards overwrite every row of their column slices
        if qkv.shape[1] > 3 * dout:
            dqkv[:, 3 * dout:].zero_()     # zero padding columns of the GEMM operand
        pw = lambda c: P(c.weight).reshape(HD, 27)
        for (d, c, col0, st, pool, norm) in ((dq, s["cq"], 0, sq, a.pool_q, a.norm_q), (dk, s["ck"], dout, skv, a.pool_k, a.norm_k),
                                             (dv, s["cv"], 2 * dout, skv, a.pool_v, a.norm_v)):
            gw = self._acc_target(pool.weight)
            om.pool_bwd(d, c, qkv, dqkv, col0, B, H, thw, st, pw(pool), P(norm.weight), eps, gw.view(HD, 27),
                        self._acc_target(norm.weight), self._acc_target(norm.bias))
        wqkv = self._wpad(a.qkv.weight, a.qkv.bias)
        self._wgrad(dqkv, s["xn"], a.qkv.weight, a.qkv.bias, wqkv)
        # (16-bit where nothing is added to it before norm1's backward reads it -- 13 of MViTv2-S's 16 blocks; round 6: the fp32 form wrote and
        #  re-read 411 MB on block 0's 803k rows.  PVRL_MVIT_DXN16=0: A/B)
        dxn = ops.gemm_nt(dqkv, wqkv.t, L.PVRL_EPI_BF16 if (dim == dout and self.dxn16) else L.PVRL_EPI_F32)
        # ---- skip path
        dxs = om.maxpool_bwd(s["xs"], dx1, B, thw, sq[1], dout, argmax=s["amax"]) if s["pooled"] else dx1
        dres = None
        if dim != dout:
            wsk = self._wpad(blk.proj.weight, blk.proj.bias)
            dxsb = ops.cast_scale(dxs)
            self._wgrad(dxsb, s["xn"], blk.proj.weight, blk.proj.bias, wsk)
            dxn = ops.gemm_nt(dxsb, wsk.t, L.PVRL_EPI_RESID_F32, aux=dxn)
        else:
            dres = dxs
        out = om.ln_bwd(dxn, s["x"], dim, s["mean1"], s["rstd1"], P(blk.norm1.weight), self._acc_target(blk.norm1.weight),
                        self._acc_target(blk.norm1.bias), dres=dres, Cpad=Cpi, want16=want16, rowscale16=next_rs)
        return out if want16 else (out, None)


class MViTFn(torch.autograd.Function):
    """frames -> norm(tokens)[:, 0] through the MViT encoder (MViTEngine)."""

    @staticmethod
    def forward(ctx, anchor, frames, owner, droppath=None):
        need = bool(ctx.needs_input_grad[0])
        # (DropPath belongs to the encoder's blocks: `video_encoder.eval()` of the linear-probing loop, tools/train_net.py:72-85, switches it off)
        feat = owner.engine.forward(frames, training=owner.video_encoder.training, save=need, droppath=droppath)
        ctx.owner = owner
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        ctx.owner.engine.backward(dfeat)
        return None, None, None, None


# ------------------------------------------------------------------------------------------------ wrapper (lib/models/mvit.py)
class VisionTransformer(_StepMatchingModel):
    """lib/models/mvit.py:45-229: the TimeSformer wrapper with `self.video_encoder = MViT_encoder(cfg)` in place of the
    patch-embed / blocks / norm trio.  Heads, teacher, order transformer, forward(): inherited (identical code)."""

    def __init__(self, num_classes=1000, label_emb="", mlp=0, text_model="", num_seg=0, cfg=None, **unused):
        nn.Module.__init__(self)
        assert cfg.MODEL.MODEL_NAME == "MViT"
        self.cfg = cfg
        self.num_classes = num_classes
        self.temp = cfg.DEV.TEMP
        self.order_pretrain = cfg.DEV.ORDER_PRETRAIN_ENABLED
        self.order_max_len = cfg.DEV.ORDER_PRETRAIN_MAX_LEN
        self.order_tfm_layers = cfg.DEV.ORDER_TFM_LAYERS
        self.order_recog_batch = cfg.DEV.ORDER_RECOG_BATCH
        self.depth = cfg.MVIT.DEPTH
        self.video_encoder = MViT_encoder(cfg)
        embed_dim = self.video_encoder.norm.weight.shape[0]
        self.num_features = self.embed_dim = embed_dim
        self.ln_eps = 1e-6
        self._init_heads(embed_dim, label_emb, mlp, text_model, num_seg, num_classes, cfg)
        self.engine = MViTEngine(self, self.video_encoder)
        self.weight_cache = self.engine._weight
        self._grad_store = None
        self._label_cache = None
        if hasattr(self, "order_tfm"):
            self.order_tfm.bind(self)
        if hasattr(self, "text_model"):
            self.text_model.bind(self)

    block_prefix = "video_encoder.blocks."      # distributed.GradReducer: per-block slices of the flat gradient buffer

    @property
    def cls_token(self):
        return self.video_encoder.cls_token

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "time_embed"}

    def forward_features(self, x, cls=True, droppath=None):
        if not x.is_cuda:
            raise RuntimeError("procedurevrl_amd runs on the HIP path only: move the model and inputs to the GPU "
                               "(the CPU restatement lives in oracle/ and is test infrastructure)")
        from .transform import DecodedClips
        if isinstance(x, DecodedClips):       # decoded uint8 clips: GPU-side normalise / rescale / crop / flip, then the 3-D im2col
            x = ops.frames_u8_to_f32(x)
        if self.training and torch.is_grad_enabled():
            self.grad_store()
        return MViTFn.apply(self.video_encoder.cls_token, x.float(), self, droppath)


@MODEL_REGISTRY.register()
class MViT(nn.Module):
    """lib/models/mvit.py:231-266"""

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.pretrained = cfg.MODEL.PRETRAINED
        self.model = VisionTransformer(num_classes=cfg.MODEL.NUM_CLASSES, label_emb=cfg.TRAIN.LABEL_EMB, mlp=cfg.MODEL.MLP,
                                       text_model=cfg.MODEL.TEXT_MODEL, num_seg=cfg.MODEL.NUM_SEG, cfg=cfg)
        self.attention_type = cfg.TIMESFORMER.ATTENTION_TYPE
        if self.pretrained:
            from .checkpoint import load_pretrained_mvit
            load_pretrained_mvit(self.model, cfg)
        else:
            print("not loading any pretrained weights!")

    def forward(self, x, rng=None):
        return self.model(x) if rng is None else self.model(x, rng=rng)
