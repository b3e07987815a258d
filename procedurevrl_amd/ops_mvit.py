"""Tensor-level wrappers over the MViTv2 entry points of the C ABI (include/pvrl.h, csrc/mvit.hip, csrc/attn_pool.hip).
Same rules as ops.py: device tensors in, device tensors out, raw pointers to libpvrl_hip.so on the current stream,
no PyTorch compute fallback.  Pooled per-head tensors are [B*H, L+1, 96] bf16 with the cls token LAST."""
import torch

from ._lib import lib
from .ops import OP16, F32, _ptr, _stream

I32 = torch.int32
HD = 96


def pad128(c):
    return (int(c) + 127) // 128 * 128


def im2col3d(frames, kernel, stride, padding, ldo):
    """fp32 [B,Cin,T,H,W] -> (bf16 [B*To*Ho*Wo, ldo], (To, Ho, Wo))"""
    L = lib()
    assert frames.is_cuda and frames.dtype == F32 and frames.is_contiguous()
    B, Cin, T, H, W = frames.shape
    out_thw = tuple((s + 2 * p - k) // st + 1 for s, k, st, p in zip((T, H, W), kernel, stride, padding))
    out = torch.empty((B * out_thw[0] * out_thw[1] * out_thw[2], ldo), device=frames.device, dtype=OP16)
    L.call("pvrl_im2col3d_bf16", _ptr(frames), B, Cin, T, H, W, *kernel, *stride, *padding, _ptr(out), ldo, _stream())
    return out, out_thw


def ln_fwd(x, C, gamma, beta, eps, out_dtype=OP16, Cpad=None, stats=True):
    """x fp32 [M, ld>=C] -> (y [M, Cpad] (zeros beyond C), mean, rstd)"""
    L = lib()
    assert x.is_cuda and x.dtype == F32 and x.stride(1) == 1
    M = x.shape[0]
    Cpad = C if Cpad is None else Cpad
    y = torch.empty((M, Cpad), device=x.device, dtype=out_dtype)
    mean = torch.empty(M, device=x.device, dtype=F32) if stats else None
    rstd = torch.empty(M, device=x.device, dtype=F32) if stats else None
    L.call("pvrl_layernorm_g_fwd", _ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), _ptr(y), y.stride(0),
           int(out_dtype == F32), M, C, Cpad, _ptr(mean), _ptr(rstd), _stream())
    return y, mean, rstd


def ln_bwd(dy, x, C, mean, rstd, gamma, dgamma, dbeta, dres=None, Cpad=None, want16=False, rowscale16=None):
    """-> dx fp32 [M, Cpad] = dres + dLN(dy); dgamma / dbeta (fp32 [C]) are accumulated into.
    want16: -> (dx, dx16) with dx16 = (rowscale16[row] *) dx in the operand dtype, written by the same kernel."""
    L = lib()
    M = x.shape[0]
    Cpad = C if Cpad is None else Cpad
    dx = torch.empty((M, Cpad), device=x.device, dtype=F32)
    dx16 = torch.empty((M, Cpad), device=x.device, dtype=OP16) if want16 else None
    assert dy.dtype in (F32, OP16) and dy.stride(1) == 1 and dgamma.dtype == F32 and dgamma.is_contiguous()
    from .ops import workspace
    ws = workspace(L.call("pvrl_layernorm_g_bwd_workspace_bytes", M, C), x.device, "mvit_ln_g")
    L.call("pvrl_layernorm_g_bwd", _ptr(dy), dy.stride(0), int(dy.dtype == F32), _ptr(x), x.stride(0), _ptr(mean),
           _ptr(rstd), _ptr(gamma), _ptr(dres), dres.stride(0) if dres is not None else 0, _ptr(dx), dx.stride(0),
           _ptr(dx16), Cpad if want16 else 0, _ptr(rowscale16) if want16 else None, M, C, Cpad, _ptr(dgamma), _ptr(dbeta),
           _ptr(ws), ws.numel(), _stream())
    return (dx, dx16) if want16 else dx


def pool_out_thw(thw, stride):
    return tuple((s + 2 - 3) // st + 1 for s, st in zip(thw, stride))


def pool_fwd(qkv, col0, B, H, thw, stride, w, gamma, beta, eps):
    """-> (y, conv_out) both bf16 [B*H, Lo+1, 96]"""
    L = lib()
    To, Ho, Wo = pool_out_thw(thw, stride)
    n = B * H * (To * Ho * Wo + 1)
    y = torch.empty((B * H, To * Ho * Wo + 1, HD), device=qkv.device, dtype=OP16)
    c = torch.empty_like(y)
    L.call("pvrl_mvit_pool_fwd", _ptr(qkv), qkv.stride(0), col0, B, H, *thw, *stride, _ptr(w), _ptr(gamma), _ptr(beta),
           float(eps), _ptr(y), _ptr(c), _stream())
    return y, c


def pool_bwd(dy, conv_out, qkv, dqkv, col0, B, H, thw, stride, w, gamma, eps, dw, dgamma, dbeta):
    L = lib()
    scratch = torch.empty_like(conv_out)
    assert dy.dtype == OP16 and dy.is_contiguous() and dw.is_contiguous() and dw.dtype == F32
    from .ops import workspace
    ws = workspace(L.call("pvrl_mvit_pool_bwd_workspace_bytes"), dy.device, "mvit_pool_bwd")
    L.call("pvrl_mvit_pool_bwd", _ptr(dy), _ptr(conv_out), _ptr(qkv), _ptr(dqkv), qkv.stride(0), col0, B, H, *thw,
           *stride, _ptr(w), _ptr(gamma), float(eps), _ptr(scratch), _ptr(dw), _ptr(dgamma), _ptr(dbeta), _ptr(ws),
           ws.numel(), _stream())


def maxpool_fwd(x, B, thw, s, C, want_argmax=False):
    """-> y, or (y, argmax uint8 [B*T*Ho*Wo, C]) with `want_argmax` (saved for maxpool_bwd)"""
    L = lib()
    T, H, W = thw
    k = s + 1
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    y = torch.empty((B * T * Ho * Wo + B, x.shape[1]), device=x.device, dtype=F32)
    if x.shape[1] > C:
        y[:, C:].zero_()
    am = torch.empty((B * T * Ho * Wo, C), device=x.device, dtype=torch.uint8) if want_argmax else None
    L.call("pvrl_mvit_maxpool_fwd", _ptr(x), x.stride(0), B, T, H, W, s, C, _ptr(y), y.stride(0), _ptr(am), _stream())
    return (y, am) if want_argmax else y


def maxpool_bwd(x, dy, B, thw, s, C, argmax=None):
    L = lib()
    T, H, W = thw
    dx = torch.empty_like(x)
    if x.shape[1] > C:
        dx[:, C:].zero_()          # the kernel writes the C real columns of every row; the padding stays zero
    L.call("pvrl_mvit_maxpool_bwd", _ptr(x), x.stride(0), _ptr(dy), dy.stride(0), B, T, H, W, s, C, _ptr(dx), _ptr(argmax),
           _stream())
    return dx


def rel_width(k_thw):
    """columns of one half (hi or lo) of the rel operand rows: 32 or 64"""
    return int(lib().call("pvrl_mvit_rel_width", *k_thw))


def rel_fwd(Q, BH, q_thw, k_thw, Rh, Rw, Rt, ih, iw, it, out_scale=1.0):
    """-> relp [BH, Lq, 2*JP] (operand dtype): out_scale * rel as a hi | lo 16-bit pair, the form attn_fwd / attn_bwd take
    (with out_scale = 1 / attention scale); `rel_unpack` turns it back into fp32 [BH, Lq, J]"""
    L = lib()
    Lq = q_thw[0] * q_thw[1] * q_thw[2]
    JP = rel_width(k_thw)
    relp = torch.empty((BH, Lq, 2 * JP), device=Q.device, dtype=OP16)
    L.call("pvrl_mvit_rel_fwd", _ptr(Q), BH, *q_thw, *k_thw, _ptr(Rh), _ptr(Rw), _ptr(Rt), _ptr(ih), _ptr(iw), _ptr(it),
           float(out_scale), _ptr(relp), _stream())
    return relp


def rel_unpack(relp, k_thw, out_scale=1.0):
    JP = relp.shape[-1] // 2
    J = k_thw[0] + k_thw[1] + k_thw[2]
    return (relp[..., :JP].float() + relp[..., JP:].float())[..., :J] / out_scale


def rel_pack(rel, k_thw, out_scale=1.0):
    """fp32 rel [BH, Lq, J] -> the operand form (what rel_fwd writes), e.g. to feed the attention kernels a given bias"""
    JP = rel_width(k_thw)
    x = torch.zeros(rel.shape[:-1] + (JP,), device=rel.device, dtype=F32)
    x[..., :rel.shape[-1]] = rel.float() * out_scale
    hi = x.to(OP16)
    lo = (x - hi.float()).to(OP16)
    return torch.cat((hi, lo), dim=-1).contiguous()


_KEYMAPS = {}


def keymap(k_thw, device):
    """the key geometry's 0/1 map E[key][j] as MFMA tile images (pvrl_mvit_attn_keymap), built once per geometry"""
    key = (tuple(k_thw), str(device), str(OP16))
    km = _KEYMAPS.get(key)
    if km is None:
        # a buffer cached process-wide must not be born inside a HIP-graph capture (it would live in the graph's private
        # pool and only be filled on replay): MViTEngine builds every geometry's map in its eager warm-up calls
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("ops_mvit.keymap(): first use of a key geometry inside a HIP-graph capture; run the shape "
                               "eagerly once first (GraphReplay.GRAPH_WARMUP >= 1)")
        L = lib()
        km = torch.empty(L.call("pvrl_mvit_attn_keymap_bytes", *k_thw), device=device, dtype=torch.uint8)
        L.call("pvrl_mvit_attn_keymap", *k_thw, _ptr(km), _stream())
        _KEYMAPS[key] = km
    return km


def rel_bwd(drel, Q, dQ, BH, q_thw, k_thw, Rh, Rw, Rt, ih, iw, it, dRh, dRw, dRt):
    L = lib()
    from .ops import workspace
    ws = workspace(L.call("pvrl_mvit_rel_bwd_workspace_bytes", BH, *q_thw, *k_thw), drel.device, "mvit_rel_bwd")
    assert dRh.is_contiguous() and dRw.is_contiguous() and dRt.is_contiguous()
    L.call("pvrl_mvit_rel_bwd", _ptr(drel), _ptr(Q), _ptr(dQ), BH, *q_thw, *k_thw, _ptr(Rh), _ptr(Rw), _ptr(Rt), _ptr(ih),
           _ptr(iw), _ptr(it), dRh.shape[0], dRw.shape[0], dRt.shape[0], _ptr(dRh), _ptr(dRw), _ptr(dRt), _ptr(ws),
           ws.numel(), _stream())


def attn_fwd(q, k, v, relp, B, H, Lq, k_thw, scale, ldo):
    """relp = rel_fwd(..., out_scale=1/scale) -> (o bf16 [B*Lq + B, ldo] token-major (zeros beyond H*96), lse fp32 [B*H, Lq+1])"""
    L = lib()
    o = torch.empty((B * Lq + B, ldo), device=q.device, dtype=OP16)
    if ldo > H * HD:
        o[:, H * HD:].zero_()          # the kernel writes the H*96 real columns of every row
    lse = torch.empty((B * H, Lq + 1), device=q.device, dtype=F32)
    assert relp.dtype == OP16 and relp.shape[-1] == 2 * rel_width(k_thw)
    L.call("pvrl_mvit_attn_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(relp), _ptr(keymap(k_thw, q.device)), B, H, Lq, *k_thw,
           float(scale), _ptr(o), ldo,
           _ptr(lse), _stream())
    return o, lse


def attn_bwd(q, k, v, relp, B, H, Lq, k_thw, scale, o, d_o, lse):
    """-> (dq, dk, dv bf16 like q / k / v, drel fp32 [B*H, Lq, J]: the gradient of the unscaled rel)"""
    L = lib()
    assert d_o.dtype == OP16 and d_o.stride(0) == o.stride(0)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    assert relp.dtype == OP16 and relp.shape[-1] == 2 * rel_width(k_thw)
    drel = torch.empty((B * H, Lq, k_thw[0] + k_thw[1] + k_thw[2]), device=q.device, dtype=F32)
    delta = torch.empty_like(lse)
    from .ops import workspace
    nbytes = L.call("pvrl_mvit_attn_bwd_workspace_bytes", B, H, Lq, *k_thw)
    ws = workspace(nbytes, q.device, "mvit_attn_bwd")
    L.call("pvrl_mvit_attn_bwd", _ptr(q), _ptr(k), _ptr(v), _ptr(relp), _ptr(keymap(k_thw, q.device)), B, H, Lq, *k_thw,
           float(scale), _ptr(o), _ptr(d_o),
           o.stride(0), _ptr(lse), _ptr(delta), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(drel), _ptr(ws), ws.numel(), _stream())
    return dq, dk, dv, drel


def copy2d(src, dst, R, C, beta=0.0):
    """dst[:R, :C] = beta*dst[:R, :C] + src[:R, :C] (fp32, row-major with their own leading dimensions)"""
    L = lib()
    L.call("pvrl_copy2d_f32", _ptr(src), src.stride(0), _ptr(dst), dst.stride(0) if dst.dim() > 1 else C, R, C,
           float(beta), _stream())
    return dst
