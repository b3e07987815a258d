"""torch.autograd glue: each Function's forward / backward is a schedule of C-ABI kernel calls.

PyTorch's autograd is used only to connect the pieces in the order the reference's eager graph does
(lib/models/vit.py:283-352, tools/train_net.py:147-181); no Function computes with ATen kernels
except for trivial shape plumbing (transposes of <1 MB operands, bias column sums).
Parameter gradients of the kernel-scheduled stacks are written straight into the model's flat
gradient buffer (engine.GradStore); the Functions return None for them.
"""
import torch

from . import ops

F32 = torch.float32


class EncoderFn(torch.autograd.Function):
    """frames -> norm(x)[:, 0] through the TimeSformer encoder (engine.EncoderEngine)."""

    @staticmethod
    def forward(ctx, anchor, frames, owner, droppath):
        eng = owner.engine
        need = bool(ctx.needs_input_grad[0])
        # DropPath lives in the blocks (vit.py:110-117): `model.blocks.eval()` of the linear-probing loop (tools/train_net.py:72-85)
        # switches it off while the wrapper stays in train mode, so the blocks' flag decides, not the wrapper's
        feat = eng.forward(frames, training=owner.blocks.training, droppath=droppath, save=need)
        ctx.owner = owner
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        ctx.owner.engine.backward(dfeat.contiguous())
        return None, None, None, None


class StackFn(torch.autograd.Function):
    """x [nseq*S, W] -> residual attention stack (tfm_engine.StackEngine)."""

    @staticmethod
    def forward(ctx, x, owner, resblocks, nseq, S, causal, kpm, heads, anchor):
        from .tfm_engine import StackEngine
        eng = StackEngine(resblocks, owner.weight_cache, owner.grad_target, heads=heads, grad_store=getattr(owner, "grad_store", None))
        need = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[8])
        y, saved = eng.forward(x.contiguous(), nseq, S, causal=causal, kpm=kpm, save=need)
        ctx.eng, ctx.saved_acts, ctx.owner_ref = eng, saved, owner
        return y

    @staticmethod
    def backward(ctx, dy):
        from .engine import SCALED_GRADS
        gs = ctx.owner_ref.grad_store() if SCALED_GRADS else None
        if gs is not None:
            dy = gs.begin_scaled(dy)                 # fp16-operand flavour: the stack's backward runs in S-scaled units
        dx = ctx.eng.backward(dy.contiguous(), ctx.saved_acts)
        if gs is not None:
            dx = dx * gs.end_scaled()
        ctx.saved_acts = None
        return dx, None, None, None, None, None, None, None, None


class LinearF32Fn(torch.autograd.Function):
    """y = x W^T + b in fp32 (projection head, vit.py:299; time_mlp, tfm_model.py:89-94)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.gemm_nt_f32(x.contiguous(), weight.detach().contiguous(), bias.detach() if bias is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt_f32(dy, w.detach().t().contiguous())
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_nt_f32(dy.t().contiguous(), x.detach().t().contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear_f32(x, weight, bias=None):
    return LinearF32Fn.apply(x, weight, bias)


class L2NormFn(torch.autograd.Function):
    """x / x.norm(dim=1, keepdim=True)   (vit.py:300, 303, 331, 339, 431)"""

    @staticmethod
    def forward(ctx, x):
        y, inv = ops.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(dy.contiguous(), y, inv)


def l2norm(x):
    return L2NormFn.apply(x)


class GeluF32Fn(torch.autograd.Function):
    """nn.GELU() (exact erf) on a small fp32 tensor: the order transformer's time_mlp (tfm_model.py:89-94)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.gelu_f32(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_f32(x, dy.contiguous())


def gelu_f32(x):
    return GeluF32Fn.apply(x)


class StepLogitsFn(torch.autograd.Function):
    """x @ label_emb.t() / temp   (vit.py:307); label_emb is a constant (no gradient)."""

    @staticmethod
    def forward(ctx, x, label_emb, label_emb_t, inv_temp):
        ctx.label_emb_t = label_emb_t
        ctx.inv_temp = inv_temp
        return ops.gemm_nt_f32(x.contiguous(), label_emb, alpha=inv_temp)

    @staticmethod
    def backward(ctx, dy):
        return ops.gemm_nt_f32(dy.contiguous(), ctx.label_emb_t, alpha=ctx.inv_temp), None, None, None


def step_logits(x, label_emb, label_emb_t, temp):
    return StepLogitsFn.apply(x, label_emb, label_emb_t, 1.0 / temp)


class KLTopkLossFn(torch.autograd.Function):
    """tools/train_net.py:152-160: KLDivLoss(batchmean)(log_softmax(pred), renormalised top-k softmax(teacher))."""

    @staticmethod
    def forward(ctx, pred, teacher, topk):
        rows = pred.shape[0]
        row_loss, dpred, _ = ops.kl_topk(pred.contiguous(), teacher.detach().contiguous(), topk, grad_scale=1.0 / rows)
        ctx.save_for_backward(dpred)
        return row_loss.sum() / rows

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


def kl_topk_loss(pred, teacher, topk):
    return KLTopkLossFn.apply(pred, teacher, topk)


class MSELossFn(torch.autograd.Function):
    """nn.MSELoss(reduction='mean')(a, b) with gradient to both operands (tools/train_net.py:161)."""

    @staticmethod
    def forward(ctx, a, b):
        loss, da, db = ops.mse(a.contiguous(), b.contiguous(), grad_scale=1.0)
        ctx.save_for_backward(da, db)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return da * g, db * g


def mse_loss(a, b):
    return MSELossFn.apply(a, b)


class MatmulNTFn(torch.autograd.Function):
    """a @ b.t() in fp32 with gradients to both operands (similarity matrix of the contrastive loss)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.gemm_nt_f32(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        da = ops.gemm_nt_f32(dy, b.t().contiguous()) if ctx.needs_input_grad[0] else None
        db = ops.gemm_nt_f32(dy.t().contiguous(), a.t().contiguous()) if ctx.needs_input_grad[1] else None
        return da, db


class MILNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n, C):
        nom, den, dx = ops.milnce(x.contiguous(), n, C, grad_scale=1.0 / n)
        ctx.save_for_backward(dx)
        return (den - nom).sum() / n

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None


def milnce_loss(video_embd, text_embd):
    """MILNCELoss.forward (lib/models/losses.py:15-23): video [n, D], text [n*C, D]."""
    n = video_embd.shape[0]
    C = text_embd.shape[0] // n
    x = MatmulNTFn.apply(video_embd, text_embd)
    return MILNCEFn.apply(x, n, C)
