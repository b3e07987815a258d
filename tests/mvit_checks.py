"""Parity checks of the MViTv2 kernels (csrc/mvit.hip, csrc/attn_pool.hip) against CPU fp32 restatements built from
oracle/mvit_oracle.py pieces and autograd.  Inputs are rounded to bf16 first where the kernel consumes bf16, so the
tolerances only cover accumulation order and the bf16 rounding of outputs (2^-8 relative)."""
import math
import os

import torch
import torch.nn.functional as F

from oracle import mvit_oracle as mo

from procedurevrl_amd._lib import OPERAND
BF = torch.bfloat16 if OPERAND == "bf16" else torch.float16
DEV = "cuda:0"
# End-to-end tolerances per operand flavour, set to <= 2x what is observed on MI355X (gpurun_out/mvit_obs_*.txt, round 2):
# bf16: features / logits 4.9-6.2e-3, losses <= 1.1e-3, gradients <= 1.8e-2 (rel_pos_h 4.2e-2: signed sums, cancellation);
# f16 (the flavour held to north_star's 1e-3): features / logits 6.1-7.8e-4, losses <= 1.3e-4, gradients <= 1.8e-3.
F16 = OPERAND != "bf16"
TOL_FEAT = 1e-3 if F16 else 1e-2
TOL_LOSS = 1e-3 if F16 else 2.5e-3
GSC = 1.0 / 3.0 if F16 else 1.0          # scale of every gradient tolerance below (f16: 8x smaller rounding, ~4x smaller errors)


def bf(x):
    return x.to(BF).float()


def rel(a, b):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def check_mvit_im2col_ln():
    from procedurevrl_amd import ops_mvit as om
    g = torch.Generator().manual_seed(1)
    out = []
    B, T, H, W = 2, 4, 20, 24
    x = bf(torch.randn(B, 3, T, H, W, generator=g))
    a, thw = om.im2col3d(x.to(DEV), (3, 7, 7), (2, 4, 4), (1, 3, 3), 512)
    xp = F.pad(x, (3, 3, 3, 3, 1, 1))
    u = xp.unfold(2, 3, 2).unfold(3, 7, 4).unfold(4, 7, 4)            # [B,C,To,Ho,Wo,kt,kh,kw]
    ref = u.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(-1, 441)
    out.append(("im2col3d columns", rel(a[:, :441], ref), 1e-6))
    out.append(("im2col3d zero padding", float(a[:, 441:].float().abs().max()), 0.0))
    out.append(("im2col3d geometry", float(thw != (2, 5, 6)), 0.0))
    for C, Cpad, M in [(96, 128, 300), (192, 256, 77), (384, 384, 65), (768, 768, 9)]:
        xx = torch.randn(M, Cpad, generator=g); xx[:, C:] = 0
        gm = 1 + 0.1 * torch.randn(C, generator=g); bt = 0.1 * torch.randn(C, generator=g)
        dy = torch.randn(M, C, generator=g); dres = torch.randn(M, Cpad, generator=g)
        xr = xx[:, :C].clone().requires_grad_(True); gr = gm.clone().requires_grad_(True); br = bt.clone().requires_grad_(True)
        yr = F.layer_norm(xr, (C,), gr, br, 1e-6)
        yr.backward(dy)
        y, mean, rstd = om.ln_fwd(xx.to(DEV), C, gm.to(DEV), bt.to(DEV), 1e-6, out_dtype=torch.float32, Cpad=Cpad)
        out.append((f"ln_g fwd C={C}", rel(y[:, :C], yr), 1e-5))
        if Cpad > C:
            out.append((f"ln_g fwd pad zeros C={C}", float(y[:, C:].abs().max()), 0.0))
        dg = torch.zeros(C, device=DEV); db = torch.zeros(C, device=DEV)
        dyp = torch.zeros(M, Cpad); dyp[:, :C] = dy
        dx = om.ln_bwd(dyp.to(DEV), xx.to(DEV), C, mean, rstd, gm.to(DEV), dg, db, dres=dres.to(DEV), Cpad=Cpad)
        out.append((f"ln_g bwd dx C={C}", rel(dx[:, :C], xr.grad + dres[:, :C]), 1e-5))
        out.append((f"ln_g bwd dgamma C={C}", rel(dg, gr.grad), 1e-5))
        out.append((f"ln_g bwd dbeta C={C}", rel(db, br.grad), 1e-5))
        # the fused 16-bit operand copy (what the next backward GEMM reads): rowscale * dx rounded once, zero padding
        rsc = 0.5 + torch.rand(M, generator=g)
        dg2 = torch.zeros(C, device=DEV); db2 = torch.zeros(C, device=DEV)
        dxb, dx16 = om.ln_bwd(dyp.to(DEV), xx.to(DEV), C, mean, rstd, gm.to(DEV), dg2, db2, dres=dres.to(DEV), Cpad=Cpad,
                              want16=True, rowscale16=rsc.to(DEV))
        prod = dxb[:, :C] * rsc.to(DEV)[:, None]            # one rounding of the product (v_fma_mix: not fp32 first)
        half_ulp = 2.0 ** -8 if BF == torch.bfloat16 else 2.0 ** -11
        out.append((f"ln_g bwd fused 16-bit copy C={C}", float(((dx16[:, :C].float() - prod).abs() / prod.abs().clamp_min(1e-3)).max()), 1.01 * half_ulp))
        out.append((f"ln_g bwd fused copy: fp32 dx unchanged, padding zero C={C}",
                    0.0 if torch.equal(dxb, dx) and (Cpad == C or float(dx16[:, C:].float().abs().max()) == 0.0) else 1.0, 0.0))
        yb, _, _ = om.ln_fwd(xx.to(DEV), C, gm.to(DEV), bt.to(DEV), 1e-6, Cpad=Cpad)
        out.append((f"ln_g fwd bf16 C={C}", rel(yb[:, :C], yr), 4e-3))
    return out


def _pool_ref(t, w, stride, thw, gm, bt):
    """t [B, H, 1+L, 96] with cls FIRST (reference order) -> pooled, same convention"""
    return mo.attention_pool(t, w, stride, thw, gm, bt)


def check_mvit_pool():
    from procedurevrl_amd import ops_mvit as om
    g = torch.Generator().manual_seed(2)
    out = []
    # temporal stride 1 -> the t-sliding forward kernel (incl. a single frame); (2, 2, 2) -> the general one
    for (B, H, thw, stride) in [(2, 2, (2, 8, 8), (1, 2, 2)), (1, 1, (3, 6, 10), (1, 1, 1)), (2, 4, (2, 8, 8), (1, 4, 4)),
                                (1, 1, (1, 6, 6), (1, 1, 1)), (1, 2, (4, 6, 6), (2, 2, 2)), (3, 1, (8, 14, 14), (1, 1, 1))]:
        T, Hh, Ww = thw
        L = T * Hh * Ww
        dout = H * 96
        ld = om.pad128(3 * dout)
        qkv = torch.zeros(B * L + B, ld)
        qkv[:, :3 * dout] = bf(torch.randn(B * L + B, 3 * dout, generator=g))
        w = torch.randn(96, 1, 3, 3, 3, generator=g) * 0.2
        gm = 1 + 0.1 * torch.randn(96, generator=g); bt = 0.1 * torch.randn(96, generator=g)
        col0 = dout          # the "k" slice
        # reference tensor [B, H, 1+L, 96], cls first
        tok = qkv[:B * L, col0:col0 + dout].reshape(B, L, H, 96).permute(0, 2, 1, 3)
        cls = qkv[B * L:, col0:col0 + dout].reshape(B, 1, H, 96).permute(0, 2, 1, 3)
        tr = torch.cat((cls, tok), dim=2).clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True); gr = gm.clone().requires_grad_(True); br = bt.clone().requires_grad_(True)
        yr, othw = _pool_ref(tr, wr, stride, thw, gr, br)
        Lo = othw[0] * othw[1] * othw[2]
        dy = bf(torch.randn(B * H, Lo + 1, 96, generator=g))
        dyr = torch.cat((dy[:, Lo:], dy[:, :Lo]), dim=1).reshape(B, H, Lo + 1, 96)
        yr.backward(dyr)
        y, c = om.pool_fwd(qkv.to(DEV, BF), col0, B, H, thw, stride, w.reshape(96, 27).contiguous().to(DEV), gm.to(DEV),
                           bt.to(DEV), 1e-6)
        yr2 = yr.reshape(B * H, Lo + 1, 96)
        out.append((f"pool fwd tokens {thw}/{stride}", rel(y[:, :Lo], yr2[:, 1:]), 6e-3))
        out.append((f"pool fwd cls {thw}/{stride}", rel(y[:, Lo], yr2[:, 0]), 6e-3))
        dqkv = torch.zeros(B * L + B, ld, device=DEV, dtype=BF)
        dqkv[:, col0:col0 + dout] = 777.0      # the backward OVERWRITES every row of its slice (the engine hands it torch.empty)
        dw = torch.zeros(96, 27, device=DEV); dg = torch.zeros(96, device=DEV); db = torch.zeros(96, device=DEV)
        om.pool_bwd(dy.to(DEV, BF), c, qkv.to(DEV, BF), dqkv, col0, B, H, thw, stride,
                    w.reshape(96, 27).contiguous().to(DEV), gm.to(DEV), 1e-6, dw, dg, db)
        gt = tr.grad                                            # [B, H, 1+L, 96]
        ref_tok = gt[:, :, 1:].permute(0, 2, 1, 3).reshape(B * L, dout)
        ref_cls = gt[:, :, 0].reshape(B, dout)
        out.append((f"pool bwd d tokens {thw}/{stride}", rel(dqkv[:B * L, col0:col0 + dout], ref_tok), 1.5e-2))
        out.append((f"pool bwd d cls {thw}/{stride}", rel(dqkv[B * L:, col0:col0 + dout], ref_cls), 1.5e-2))
        out.append((f"pool bwd other columns untouched {thw}", float(dqkv[:, :col0].float().abs().max()), 0.0))
        out.append((f"pool bwd dw {thw}/{stride}", rel(dw, wr.grad.reshape(96, 27)), 1.5e-2))
        out.append((f"pool bwd dgamma {thw}/{stride}", rel(dg, gr.grad), 1.5e-2))
        out.append((f"pool bwd dbeta {thw}/{stride}", rel(db, br.grad), 1.5e-2))
    return out


def check_mvit_maxpool_rel():
    from procedurevrl_amd import ops_mvit as om
    g = torch.Generator().manual_seed(3)
    out = []
    B, thw, C, Cp = 2, (2, 8, 6), 192, 256
    T, H, W = thw
    L = T * H * W
    x = torch.zeros(B * L + B, Cp); x[:, :C] = torch.randn(B * L + B, C, generator=g)
    xr = torch.cat((x[B * L:, :C].reshape(B, 1, C), x[:B * L, :C].reshape(B, L, C)), dim=1).clone().requires_grad_(True)
    yr = mo.pool_skip(xr, (1, 2, 2), thw)
    Lo = yr.shape[1] - 1
    dy = torch.randn(B * Lo + B, Cp, generator=g); dy[:, C:] = 0
    yr.backward(torch.cat((dy[B * Lo:, :C].reshape(B, 1, C), dy[:B * Lo, :C].reshape(B, Lo, C)), dim=1))
    y = om.maxpool_fwd(x.to(DEV), B, thw, 2, C)
    out.append(("maxpool fwd tokens", rel(y[:B * Lo, :C], yr[:, 1:].reshape(B * Lo, C)), 0.0))
    out.append(("maxpool fwd cls", rel(y[B * Lo:, :C], yr[:, 0]), 0.0))
    dx = om.maxpool_bwd(x.to(DEV), dy.to(DEV), B, thw, 2, C)
    out.append(("maxpool bwd tokens", rel(dx[:B * L, :C], xr.grad[:, 1:].reshape(B * L, C)), 1e-6))
    out.append(("maxpool bwd cls", rel(dx[B * L:, :C], xr.grad[:, 0]), 0.0))
    y2, am = om.maxpool_fwd(x.to(DEV), B, thw, 2, C, want_argmax=True)          # the path the engine takes: saved winners
    dx2 = om.maxpool_bwd(x.to(DEV), dy.to(DEV), B, thw, 2, C, argmax=am)
    out.append(("maxpool fwd with argmax == without", 0.0 if torch.equal(y2, y) else 1.0, 0.0))
    out.append(("maxpool bwd routed by saved argmax == re-scanned", 0.0 if torch.equal(dx2, dx) else 1.0, 0.0))
    # relative-position tables
    for q_thw, k_thw in [((2, 8, 8), (2, 2, 2)), ((2, 4, 4), (2, 4, 4)), ((3, 4, 8), (3, 4, 2))]:
        BH = 3
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        Q = bf(torch.randn(BH, Lq + 1, 96, generator=g))
        nh = 2 * max(q_thw[1], k_thw[1]) - 1; nw = 2 * max(q_thw[2], k_thw[2]) - 1; nt = 2 * max(q_thw[0], k_thw[0]) - 1
        Rh = torch.randn(nh, 96, generator=g) * 0.1; Rw = torch.randn(nw, 96, generator=g) * 0.1; Rt = torch.randn(nt, 96, generator=g) * 0.1
        ih = mo.rel_index(q_thw[1], k_thw[1]); iw = mo.rel_index(q_thw[2], k_thw[2]); it = mo.rel_index(q_thw[0], k_thw[0])
        Qr = Q.clone().requires_grad_(True); Rhr = Rh.clone().requires_grad_(True); Rwr = Rw.clone().requires_grad_(True); Rtr = Rt.clone().requires_grad_(True)
        rq = Qr[:, :Lq].reshape(BH, *q_thw, 96)
        ref = torch.cat((torch.einsum("bthwc,hkc->bthwk", rq, Rhr[ih]), torch.einsum("bthwc,wkc->bthwk", rq, Rwr[iw]),
                         torch.einsum("bthwc,tkc->bthwk", rq, Rtr[it])), dim=-1).reshape(BH, Lq, -1)
        d = lambda t: t.to(DEV)
        di = lambda t: t.to(DEV, torch.int32).contiguous()
        for osc in (1.0, 96 ** 0.5):              # the operand form (hi | lo pair of out_scale * rel) decodes to rel
            relp = om.rel_fwd(d(Q).to(BF), BH, q_thw, k_thw, d(Rh), d(Rw), d(Rt), di(ih), di(iw), di(it), out_scale=osc)
            out.append((f"rel fwd {q_thw}x{k_thw} out_scale {osc:.2f}", rel(om.rel_unpack(relp, k_thw, osc), ref), 2e-5))
            JP, J = relp.shape[-1] // 2, ref.shape[-1]
            pad = torch.cat((relp[..., J:JP], relp[..., JP + J:]), dim=-1)
            out.append((f"rel fwd {q_thw}x{k_thw} padding columns zero", float(pad.float().abs().max()), 0.0))
        drel = torch.randn(ref.shape, generator=g)
        ref.backward(drel)
        dQ0 = bf(torch.randn(BH, Lq + 1, 96, generator=g))
        dQ = d(dQ0).to(BF)
        dRh = torch.zeros(nh, 96, device=DEV); dRw = torch.zeros(nw, 96, device=DEV); dRt = torch.zeros(nt, 96, device=DEV)
        om.rel_bwd(d(drel), d(Q).to(BF), dQ, BH, q_thw, k_thw, d(Rh), d(Rw), d(Rt), di(ih), di(iw), di(it), dRh, dRw, dRt)
        out.append((f"rel bwd dQ {q_thw}x{k_thw}", rel(dQ, dQ0 + Qr.grad), 6e-3))
        out.append((f"rel bwd dRh {q_thw}x{k_thw}", rel(dRh, Rhr.grad), 1e-4))
        out.append((f"rel bwd dRw {q_thw}x{k_thw}", rel(dRw, Rwr.grad), 1e-4))
        out.append((f"rel bwd dRt {q_thw}x{k_thw}", rel(dRt, Rtr.grad), 1e-4))
    return out


def _attn_ref(q, k, v, relb, q_thw, k_thw, scale):
    """q [BH, Lq+1, 96], k / v [BH, Lk+1, 96] (cls LAST), relb [BH, Lq, J] -> out [BH, Lq+1, 96] incl. residual pooling"""
    BH, Lq1, _ = q.shape
    Lq, Lk = Lq1 - 1, k.shape[1] - 1
    kt, kh, kw = k_thw
    s = (q * scale) @ k.transpose(1, 2)
    j = torch.arange(Lk)
    bias = relb[:, :, (j // kw) % kh] + relb[:, :, kh + j % kw] + relb[:, :, kh + kw + j // (kw * kh)]
    s = torch.cat((torch.cat((s[:, :Lq, :Lk] + bias, s[:, :Lq, Lk:]), dim=2), s[:, Lq:]), dim=1)
    o = s.softmax(-1) @ v
    return torch.cat((o[:, :Lq] + q[:, :Lq], o[:, Lq:]), dim=1)


def check_mvit_attention():
    from procedurevrl_amd import ops_mvit as om
    g = torch.Generator().manual_seed(4)
    out = []
    # the last two have kh + kw + kt > 32 (the 64-column rel operand: MViTv2-S blocks 1, 3, 14) and > 1 key tile
    for (B, H, q_thw, k_thw) in [(2, 2, (2, 8, 8), (2, 2, 2)), (1, 1, (2, 6, 6), (2, 6, 6)), (2, 4, (1, 3, 5), (1, 3, 5)),
                                 (1, 2, (4, 8, 8), (4, 4, 4)), (1, 2, (2, 5, 7), (8, 14, 14)), (2, 1, (8, 7, 7), (7, 13, 14))]:
        BH = B * H
        Lq = q_thw[0] * q_thw[1] * q_thw[2]; Lk = k_thw[0] * k_thw[1] * k_thw[2]
        J = k_thw[1] + k_thw[2] + k_thw[0]
        q = bf(torch.randn(BH, Lq + 1, 96, generator=g)); k = bf(torch.randn(BH, Lk + 1, 96, generator=g))
        v = bf(torch.randn(BH, Lk + 1, 96, generator=g)); relb = torch.randn(BH, Lq, J, generator=g)
        scale = 96 ** -0.5
        qr, kr, vr, rr = (t.clone().requires_grad_(True) for t in (q, k, v, relb))
        ref = _attn_ref(qr, kr, vr, rr, q_thw, k_thw, scale)
        ldo = om.pad128(H * 96)
        d = lambda t: t.to(DEV)
        relp = om.rel_pack(d(relb), k_thw, 1.0 / scale)
        o, lse = om.attn_fwd(d(q).to(BF), d(k).to(BF), d(v).to(BF), relp, B, H, Lq, k_thw, scale, ldo)
        # token-major [B*Lq + B, ldo] -> [BH, Lq+1, 96]
        ot = torch.cat((o[:B * Lq, :H * 96].float().reshape(B, Lq, H, 96), o[B * Lq:, :H * 96].float().reshape(B, 1, H, 96)), dim=1)
        ot = ot.permute(0, 2, 1, 3).reshape(BH, Lq + 1, 96)
        tag = f"B{B} H{H} q{q_thw} k{k_thw}"
        out.append((f"attn fwd {tag}", rel(ot, ref), 6e-3))
        if ldo > H * 96:
            out.append((f"attn fwd pad zero {tag}", float(o[:, H * 96:].float().abs().max()), 0.0))
        do_ = bf(torch.randn(BH, Lq + 1, 96, generator=g))
        ref.backward(do_)
        dot = do_.reshape(B, H, Lq + 1, 96).permute(0, 2, 1, 3)             # [B, Lq+1, H, 96]
        d_o = torch.zeros(B * Lq + B, ldo)
        d_o[:B * Lq, :H * 96] = dot[:, :Lq].reshape(B * Lq, H * 96)
        d_o[B * Lq:, :H * 96] = dot[:, Lq].reshape(B, H * 96)
        dq, dk, dv, drel = om.attn_bwd(d(q).to(BF), d(k).to(BF), d(v).to(BF), relp, B, H, Lq, k_thw, scale, o,
                                       d(d_o).to(BF), lse)
        out.append((f"attn bwd dq {tag}", rel(dq, qr.grad), 1.5e-2))
        out.append((f"attn bwd dk {tag}", rel(dk, kr.grad), 1.5e-2))
        out.append((f"attn bwd dv {tag}", rel(dv, vr.grad), 1.5e-2))
        out.append((f"attn bwd drel {tag}", rel(drel, rr.grad), 1.5e-2))
    return out


def _load(name):
    import os
    return torch.load(os.path.join(os.path.dirname(__file__), "golden", name + ".pt"), weights_only=False)


def _mvit_cfg(mvit_dict, frames, crop, K=64):
    from procedurevrl_amd.config import get_cfg
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "MViT"
    cfg.MODEL.ARCH = "mvit"
    cfg.MODEL.PRETRAINED = False
    cfg.MODEL.NUM_CLASSES = K
    cfg.MODEL.LOSS_FUNC = "kldiv"
    cfg.MODEL.TEXT_MODEL = ""
    cfg.DATA.NUM_FRAMES = frames
    cfg.DATA.TRAIN_CROP_SIZE = cfg.DATA.TEST_CROP_SIZE = crop
    cfg.DATA.INPUT_CHANNEL_NUM = [3]
    cfg.DEV.MATCH_LANG_EMB = True
    cfg.DEV.ORDER_PRETRAIN_ENABLED = False
    cfg.NUM_GPUS = 1
    for k, v in mvit_dict.items():
        setattr(cfg.MVIT, k, v)
    return cfg


def _build_mvit(g, frames, crop, K=64):
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.datasets import synthetic_label_emb
    from oracle import timesformer_oracle as orc
    cfg = _mvit_cfg(g["mvit"], frames, crop, K)
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(K, 512, seed=1)
    model = build_model(cfg, gpu_id=0)
    sd = orc.seeded_state(mo.encoder_shapes(g["mvit"], frames, crop), g["seed"])
    enc = model.model.video_encoder
    missing, unexpected = enc.load_state_dict(sd, strict=True), None
    return model.to(DEV), sd


def check_mvit_encoder_small_golden():
    """The HIP MViT encoder vs the REFERENCE MViT_encoder (tests/golden/mvit_small.pt: reduced 4-block geometry with every
    block flavour): state_dict keys, features, parameter gradients.  bf16 datapath: same tolerances as the TimeSformer
    end-to-end checks (1e-2 features, 3e-2 gradients; softmax-invariant / near-zero gradients use an absolute floor)."""
    g = _load("mvit_small")
    c = g["cfg"]
    model, sd = _build_mvit(g, c["frames"], c["crop"])
    vt = model.model
    out = [("mvit state_dict keys == reference", float(sorted(vt.video_encoder.state_dict().keys()) != g["keys"]), 0.0)]
    model.train()
    feat = vt.forward_features(g["x"].to(DEV))
    out.append(("mvit small features vs reference", rel(feat, g["feat"]), TOL_FEAT))
    (feat * g["gout"].to(DEV)).sum().backward()
    params = dict(vt.video_encoder.named_parameters())
    for n, ref in g["grads"].items():
        got = params[n].grad
        got = got[:64] if got.dim() == 2 else got
        err = float((got.detach().float().cpu() - ref).norm())
        if n.endswith("norm_k.bias"):
            # a common shift of every key is softmax-invariant: the true gradient is 0 (the reference holds 4e-6 of fp32
            # noise); the bf16 path's residue is bounded against the sibling gain gradient instead
            tol = GSC * (5e-2 * float(g["grads"].get(n.replace("norm_k.bias", "norm_q.weight"), ref).norm()) + 1e-3)
        elif "rel_pos" in n:
            tol = GSC * (6e-2 * float(ref.norm()) + 1e-4 * ref.numel() ** 0.5)   # signed sums over all queries: cancellation
        else:
            tol = GSC * (3e-2 * float(ref.norm()) + 1e-4 * ref.numel() ** 0.5)
        out.append((f"mvit small d {n} (abs err / allowed)", err / tol, 1.0))
    return out


def check_mvit_e2e_golden():
    """The registered `MViT` model end to end vs the REFERENCE's (tests/golden/mvit_e2e.pt): full pre-training forward
    (encoder + CLIP-text teacher + order transformer + output assembly) with the reference's RNG draws pinned, KL + MSE
    losses and parameter gradients."""
    import test_oracle_golden as tg
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.vit import pretrain_loss
    from oracle import timesformer_oracle as orc
    f = _load("mvit_e2e")
    c = f["cfg"]
    cfg = _mvit_cfg(f["mvit"], c["frames"], c["crop"], f["K"])
    cfg.MODEL.TEXT_MODEL = "clip_vit_b_16"
    cfg.SYNTHETIC.TEXT_LAYERS = f["text_layers"]
    cfg.DEV.ORDER_PRETRAIN_ENABLED = True
    cfg.TRAIN.LABEL_EMB = f["label_emb"].clone()
    model = build_model(cfg, gpu_id=0)
    sh = {"video_encoder." + k: v for k, v in mo.encoder_shapes(f["mvit"], c["frames"], c["crop"]).items()}
    last = sh["video_encoder.norm.weight"][0]
    sh.update({"head.weight": (512, last), "head.bias": (512,)})
    sh.update(tg.orc_order_shapes())
    sh.update(tg.orc_text_shapes(f["text_layers"]))
    full = orc.seeded_state({"model." + k: v for k, v in sh.items()}, f["seed"])
    out = [("mvit e2e state_dict keys == reference", float(sorted(full.keys()) != f["state_keys"]), 0.0)]
    model.load_state_dict(full, strict=True)
    model.to(DEV).train()
    meta = {"clip_text_ids": f["clip_text_ids"].to(DEV), "clip_vis_feat": f["clip_vis_feat"].to(DEV)}
    rng = dict(order=dict(mask_inds=f["rng"]["mask_inds"].to(DEV), pad_start=f["rng"]["pad_start"].to(DEV),
                          noises=[n.to(DEV) for n in f["rng"]["noises"]]), rand_inds=f["rng"]["rand_inds"].to(DEV))
    pred, teacher, mse = model([f["inputs"].to(DEV), meta], rng=rng)
    out += [("mvit e2e pred logits vs reference", rel(pred, f["pred"]), TOL_FEAT),
            ("mvit e2e teacher logits vs reference", rel(teacher, f["teacher"]), TOL_FEAT),
            ("mvit e2e mse target vs reference", rel(mse[0], f["mse0"]), TOL_FEAT),
            ("mvit e2e mse pred vs reference", rel(mse[1], f["mse1"]), TOL_FEAT)]
    loss, l1, l2 = pretrain_loss(pred, teacher, mse, cfg)
    out.append(("mvit e2e loss1 (KL)", abs(float(l1.detach()) - f["loss1"]) / abs(f["loss1"]), TOL_LOSS))
    out.append(("mvit e2e loss2 (MSE)", abs(float(l2.detach()) - f["loss2"]) / abs(f["loss2"]), TOL_LOSS))
    for p in model.parameters():
        p.grad = None
    loss.backward()
    named = dict(model.named_parameters())
    for k, g in f["grads"].items():
        tol = (6e-2 if "rel_pos" in k else 3e-2) * (0.4 if F16 and "rel_pos" in k else (0.1 if F16 else 1.0))
        out.append((f"mvit e2e grad {k[6:]}", rel(named[k].grad, g), tol))
    return out


def check_mvit_droppath_golden():
    """DROPPATH_RATE 0.2 in train mode with the reference's draws pinned (tests/golden/mvit_droppath.pt): features and
    gradients of the HIP encoder vs the reference."""
    import test_oracle_golden as tg
    g = _load("mvit_droppath")
    c = g["cfg"]
    model, sd = _build_mvit(g, c["frames"], c["crop"])
    vt = model.model
    model.train()
    dp = [None if d is None else (d[0].to(DEV), d[1].to(DEV)) for d in tg.mvit_droppath_scales(g)]
    feat = vt.forward_features(g["x"].to(DEV), droppath=dp)
    out = [("mvit droppath features vs reference", rel(feat, g["feat"]), TOL_FEAT)]
    (feat * g["gout"].to(DEV)).sum().backward()
    params = dict(vt.video_encoder.named_parameters())
    for n, ref in g["grads"].items():
        got = params[n].grad
        got = got[:64] if got.dim() == 2 else got
        out.append((f"mvit droppath d {n}", rel(got, ref), 2.5e-3 if F16 else 2e-2))
    return out


def check_mvit_s_features():
    """MViTv2-S geometry (16 x 224^2, 16 blocks, 34 M parameters): one clip's features vs the reference's."""
    g = _load("mvit_s")
    model, sd = _build_mvit(g, 16, 224)
    model.eval()
    x = torch.randn(1, 3, 16, 224, 224, generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        feat = model.model.forward_features(x.to(DEV))
    return [("mvit-S features vs reference", rel(feat, g["feat"]), TOL_FEAT)]


def check_mvit_pretrain_steps():
    """The registered `MViT` model through the reference's call surface: model([inputs, meta]) -> (pred, teacher, mse)
    with the frozen text tower and the order transformer, KL + MSE loss, backward, fused AdamW -- five steps on one
    fixed batch of 2 videos x 9 clips: finite everywhere and the loss goes down."""
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.datasets import SyntheticHowTo100M, synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    from procedurevrl_amd.vit import pretrain_loss
    g = _load("mvit_small")
    cfg = _mvit_cfg(g["mvit"], 4, 64, K=128)
    cfg.MODEL.TEXT_MODEL = "clip_vit_b_16"
    cfg.SYNTHETIC.TEXT_LAYERS = 2
    cfg.DEV.ORDER_PRETRAIN_ENABLED = True
    cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
    cfg.SOLVER.BASE_LR = 1e-4
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(128, 512, seed=1)
    torch.manual_seed(0)
    model = build_model(cfg, gpu_id=0).train()
    model.model.text_model.eval()
    opt = construct_optimizer(model, cfg)
    set_lr(opt, 1e-4)
    ds = SyntheticHowTo100M(cfg, num_videos=2, seed=3)
    items = [ds[i] for i in range(2)]
    inputs = torch.stack([it[0] for it in items]).to(DEV)
    meta = {k: torch.stack([it[3][k] for it in items]).to(DEV) for k in ("clip_text_ids", "clip_vis_feat")}
    meta = {k: v.view(-1, v.shape[-1]) for k, v in meta.items()}
    losses = []
    for _ in range(5):
        pred, teacher, mse = model([inputs, meta])
        loss, l1, l2 = pretrain_loss(pred, teacher, mse, cfg)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        model.model.adopt_grads()
        opt.step()
        losses.append(float(loss))
    gs = model.model.grad_store()
    return [("mvit pretraining: non-finite loss", float(not all(math.isfinite(v) for v in losses)), 0.0),
            ("mvit pretraining: non-finite gradient", float(not bool(torch.isfinite(gs.flat).all())), 0.0),
            ("mvit pretraining: pred rows = 13 b", float(pred.shape[0] != 26), 0.0),
            ("mvit pretraining: loss after 5 steps / first loss", losses[-1] / losses[0], 0.999)]


def check_mvit_hip_graph_replay():
    """MViT encoder step replayed from HIP graphs (engine.GraphReplay) vs the eager launches of the same kernels: logits
    AND gradients bit-identical (every reduction of the MViT path has a fixed order too: no floating-point atomics)."""
    from procedurevrl_amd.build import build_model
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.functional import kl_topk_loss
    g = _load("mvit_small")
    cfg = _mvit_cfg(g["mvit"], 4, 64, K=128)
    cfg.TRAIN.LABEL_EMB = synthetic_label_emb(128, 512, seed=1)
    torch.manual_seed(0)
    model = build_model(cfg, gpu_id=0).train()
    eng = model.model.engine
    gen = torch.Generator().manual_seed(6)
    xs = [torch.randn(3, 3, 4, 64, 64, generator=gen).to(DEV) for _ in range(2)]
    teacher = (torch.randn(3, 128, generator=gen) * 3).to(DEV)

    def step(x):
        model.zero_grad(set_to_none=True)
        pred = model(x)
        kl_topk_loss(pred, teacher, 5).backward()
        return pred.detach().clone(), model.model.adopt_grads().flat.clone()

    eng.use_graphs = False
    ref = [step(x) for x in xs]
    eng.use_graphs = True
    for _ in range(eng.GRAPH_WARMUP + 1):
        step(xs[0])
    out = [("mvit graphs were captured (0 = yes)", 0.0 if len(eng._graphs) == 1 else 1.0, 0.5)]
    for k, i in enumerate((1, 0, 1, 0, 1)):     # several replays: state must not leak from one replay into the next
        pred, grads = step(xs[i])                # (a captured hipMemsetAsync did not re-zero a max-pool gradient buffer)
        out.append((f"mvit graph replay {k}, input {i}: logits differ (count)", float((pred != ref[i][0]).sum()), 0.0))
        out.append((f"mvit graph replay {k}, input {i}: gradients differ (count)", float((grads != ref[i][1]).sum()), 0.0))
    # with a per-block gradient hook (the data-parallel reducer): one graph per block, the hook between them, same bits
    calls = []
    eng.grad_hook = lambda blk: calls.append(blk)
    nb = len(model.model.video_encoder.blocks)
    for k, i in enumerate((0, 1, 0, 1)):
        del calls[:]
        pred, grads = step(xs[i])
        out.append((f"mvit staged graph replay {k}, input {i}: gradients differ (count)", float((grads != ref[i][1]).sum()), 0.0))
        out.append((f"mvit staged graph replay {k}: hook order (0 = last block first, every block once)",
                    0.0 if calls == list(range(nb - 1, -1, -1)) else 1.0, 0.0))
    gk = next(iter(eng._graphs.values()))
    out.append(("mvit staged backward was captured as one graph per block (0 = yes)",
                0.0 if gk.get("bwd_staged") and len(gk["bwd_staged"]["graphs"]) == nb else 1.0, 0.0))
    eng.grad_hook = None
    return out


def check_mvit_s_full_size_step_vs_oracle():
    """MViTv2-S at BASELINE config-5 size (16 x 224^2, 16 blocks): one clip's features AND parameter gradients of the HIP
    path against the CPU oracle (pinned to the reference by the golden tests) -- the 25,089-query / 1,569-key attention
    shapes, every stage transition and the 34 M-parameter backward at real size."""
    from oracle import timesformer_oracle as orc
    g = _load("mvit_s")
    model, sd = _build_mvit(g, 16, 224)
    vt = model.model
    model.train()
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(1, 3, 16, 224, 224, generator=gen)
    gout = torch.randn(1, 768, generator=gen)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = mo.forward_features(p, x, g["mvit"])
    (ref * gout).sum().backward()
    feat = vt.forward_features(x.to(DEV))
    out = [("mvit-S full-size features vs oracle", rel(feat, ref), TOL_FEAT)]
    (feat * gout.to(DEV)).sum().backward()
    params = dict(vt.video_encoder.named_parameters())
    for n in ("patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.pool_q.weight", "blocks.1.proj.weight",
              "blocks.1.attn.rel_pos_t", "blocks.3.attn.pool_k.weight", "blocks.7.mlp.fc1.weight", "blocks.14.attn.qkv.bias",
              "blocks.15.mlp.fc2.weight", "norm.weight", "cls_token"):
        r = p[n].grad
        err = float((params[n].grad.detach().float().cpu() - r).norm())
        # the stem's gradient has passed through all 16 blocks of bf16 activations / gradients: observed 5.9e-2 relative
        # (2.1e-2 after 4 blocks in the reduced geometry); every later parameter is below 3e-2
        tol = GSC * (8e-2 if n.startswith("patch_embed") else 5e-2)
        out.append((f"mvit-S full-size d {n} (abs err / allowed)", err / (tol * float(r.norm()) + 1e-5 * r.numel() ** 0.5), 1.0))
    return out


def bench_parity_two_clips(frames=16):
    """bench.py --arch mvit --parity-probe: MViTv2-S at its timed geometry (16 x 224^2, 16 blocks) on 2 clips -- features, a fixed
    linear functional of them (the "loss") and the gradients of every encoder parameter vs the CPU oracle.  -> dict for the bench line."""
    g = _load("mvit_s")
    model, sd = _build_mvit(g, frames, 224)
    vt = model.model
    model.train()
    gen = torch.Generator().manual_seed(19)
    x = torch.randn(2, 3, frames, 224, 224, generator=gen)
    gout = torch.randn(2, 768, generator=gen)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    try:
        torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    ref = mo.forward_features(p, x, g["mvit"])
    lref = 0.5 * (ref * ref).sum() + (ref * gout).sum()          # (a positive, well-conditioned functional: no cancellation in its value)
    lref.backward()
    feat = vt.forward_features(x.to(DEV))
    loss = 0.5 * (feat * feat).sum() + (feat * gout.to(DEV)).sum()
    loss.backward()
    worst, wk = 0.0, ""
    gmax = max(float(v.grad.norm()) / v.numel() ** 0.5 for v in p.values() if v.grad is not None)
    for n, q in vt.video_encoder.named_parameters():
        # (gradients that are zero in exact arithmetic -- the last block's key-side biases: softmax ignores a vector added to every
        #  key -- are rounding noise in both implementations: only tensors whose reference rms is within 1e-4 of the largest count)
        if n in p and p[n].grad is not None and q.grad is not None and float(p[n].grad.norm()) / p[n].numel() ** 0.5 > 1e-4 * gmax:
            e = rel(q.grad, p[n].grad)
            if e > worst:
                worst, wk = e, n
    return {"features_rel_err": float(f"{rel(feat, ref):.3e}"), "loss_rel_err": float(f"{abs(float(loss) - float(lref)) / abs(float(lref)):.3e}"),
            "worst_grad_rel_err": float(f"{worst:.3e}"), "worst_grad": wk,
            "sample": f"2 clips of {frames} x 224^2, MViTv2-S encoder (16 blocks): features, the functional 0.5 |f|^2 + <f, fixed weights> "
                      "and every encoder gradient of it vs oracle/mvit_oracle.py (fp32 CPU)"}


def check_mvit_timed_config_train_step():
    """MViTv2-S at the size bench.py times it (BASELINE configs[4]): 32 clips of 16 x 224^2 in ONE HIP step -- M up to 803k
    token rows, the padded split-M weight-gradient reductions and the pool kernels at B = 32 -- against the oracle run in
    micro-batches of 2 clips on the host cores (the loss is linear in the features, so the micro-batch gradients add up to
    the batch's).  Features and a dozen named gradients incl. rel_pos_h / rel_pos_w and the stem."""
    g = _load("mvit_s")
    model, sd = _build_mvit(g, 16, 224)
    vt = model.model
    model.train()
    B, MB = 32, 2
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, 16, 224, 224, generator=gen)
    gout = torch.randn(B, 768, generator=gen)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    try:
        torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    except Exception:
        pass
    refs = []
    for a in range(0, B, MB):
        f = mo.forward_features(p, x[a:a + MB], g["mvit"])
        (f * gout[a:a + MB]).sum().backward()
        refs.append(f.detach())
    ref = torch.cat(refs)
    feat = vt.forward_features(x.to(DEV))
    out = [("mvit-S 32 clips (timed config): features vs oracle", rel(feat, ref), TOL_FEAT)]
    (feat * gout.to(DEV)).sum().backward()
    params = dict(vt.video_encoder.named_parameters())
    names = ("patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.pool_q.weight", "blocks.0.attn.rel_pos_h",
             "blocks.0.attn.rel_pos_w", "blocks.1.proj.weight", "blocks.1.attn.rel_pos_t", "blocks.3.attn.pool_k.weight",
             "blocks.7.mlp.fc1.weight", "blocks.14.attn.qkv.bias", "blocks.15.mlp.fc2.weight", "norm.weight", "cls_token")
    obs = []
    for n in names:
        r = p[n].grad
        err = float((params[n].grad.detach().float().cpu() - r).norm())
        obs.append(f"{n} {err / float(r.norm()):.3e}")
        tol = 1.5 * TIMED_GRAD_OBS[n]            # per tensor: 1.5x the value observed on MI355X (a regression that doubles one fails)
        out.append((f"mvit-S 32 clips d {n} (rel err {err / float(r.norm()):.2e}; abs err / allowed)",
                    err / (tol * float(r.norm()) + 1e-5 * r.numel() ** 0.5), 1.0))
    try:        # observed values, for setting the bounds (scratch output)
        os.makedirs("gpurun_out", exist_ok=True)
        open(f"gpurun_out/r3_mvit_timed_obs_{OPERAND}.txt", "w").write(f"features {out[0][1]:.3e}\n" + "\n".join(obs) + "\n")
    except OSError:
        pass
    return out


# per-tensor relative errors of the 32-clip step OBSERVED on MI355X (fp16: round 6, gpurun_out/r3_mvit_timed_obs_f16.txt of the run that
# set them; bf16: round 3); the check allows 1.5x each.  The stem and the rel_pos_h / rel_pos_w tables (signed sums over the 25,088
# queries of 32 clips: cancellation) carry the largest values; features: 7.4e-4 (fp16) / 6.0e-3 (bf16).
TIMED_GRAD_OBS = ({"patch_embed.proj.weight": 2.12e-2, "blocks.0.attn.qkv.weight": 7.9e-3, "blocks.0.attn.pool_q.weight": 6.2e-3,
                   "blocks.0.attn.rel_pos_h": 2.12e-2, "blocks.0.attn.rel_pos_w": 2.16e-2, "blocks.1.proj.weight": 7.9e-3,
                   "blocks.1.attn.rel_pos_t": 1.47e-2, "blocks.3.attn.pool_k.weight": 2.6e-3, "blocks.7.mlp.fc1.weight": 1.52e-3,
                   "blocks.14.attn.qkv.bias": 6.6e-4, "blocks.15.mlp.fc2.weight": 8.4e-4, "norm.weight": 7.7e-4, "cls_token": 1.32e-3} if F16 else
                  {"patch_embed.proj.weight": 6.13e-2, "blocks.0.attn.qkv.weight": 2.41e-2, "blocks.0.attn.pool_q.weight": 2.03e-2,
                   "blocks.0.attn.rel_pos_h": 6.32e-2, "blocks.0.attn.rel_pos_w": 6.41e-2, "blocks.1.proj.weight": 2.43e-2,
                   "blocks.1.attn.rel_pos_t": 4.73e-2, "blocks.3.attn.pool_k.weight": 1.6e-2, "blocks.7.mlp.fc1.weight": 9.3e-3,
                   "blocks.14.attn.qkv.bias": 5.2e-3, "blocks.15.mlp.fc2.weight": 7.2e-3, "norm.weight": 6.0e-3, "cls_token": 1.01e-2})

ALL_CHECKS = [check_mvit_timed_config_train_step, check_mvit_s_full_size_step_vs_oracle, check_mvit_hip_graph_replay, check_mvit_encoder_small_golden, check_mvit_droppath_golden, check_mvit_e2e_golden, check_mvit_pretrain_steps, check_mvit_s_features, check_mvit_im2col_ln, check_mvit_pool, check_mvit_maxpool_rel, check_mvit_attention]
