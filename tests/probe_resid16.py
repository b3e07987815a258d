"""ANALYSIS SCRIPT (test infrastructure, CPU only; not collected by pytest) -- VERDICT r5 item 1, step 1.

What does a 16-bit residual stream on the PATCH rows (cls rows stay fp32) cost in logits / loss / gradient error?
Runs one training step of the benchmark model (12 blocks, 8 x 224^2, K = 9871, 2 clips: the `parity` probe of bench.py)
through the fp32 oracle and through the oracle with the HIP datapath's rounding points (oracle/rounded_oracle.py) in
four variants and prints the relative L2 errors against the fp32 oracle:

  shipped (r5)      fp32 residual stream (rounds 1-5)
  resid fwd         x0 / x1 / x2 / x3 of the patch rows stored in the operand type
  resid both        ... and the residual gradient stream dx of the patch rows as well
  dgelu 8 / 12 bit  the fc1 epilogue keeping an n-bit code of gelu'(u) instead of the 16-bit u (item 1, step 2)

The backward runs in S-scaled units as the engine's does (GradStore.begin_scaled: S = 2^floor(log2(256 / max|dfeat|))), so the
fp16 gradient operands sit mid-range here too.   python tests/probe_resid16.py [seed] [depth] [operand]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import rounded_oracle as rorc          # noqa: E402
from oracle import timesformer_oracle as orc       # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def step(sd, x, label, teacher, depth, rounded, resid=None, dgelu_bits=None):
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rorc.RESID, rorc.DGELU_BITS = resid, dgelu_bits
    try:
        if rounded is None:
            feat = orc.forward_features(params, x, depth)
        else:
            with rorc.operand(rounded):
                feat = rorc.forward_features(params, x, depth)
        f2 = feat.detach().requires_grad_(True)
        _, logits = orc.head_logits(params, f2, label, 0.02)
        loss, _, _ = orc.pretrain_loss(logits, teacher, None, 5)
        loss.backward()                                   # head gradients + d loss / d feat
        dfeat = f2.grad
        S = 2.0 ** float(torch.floor(torch.log2(256.0 / dfeat.abs().max())))
        if rounded is None:
            feat.backward(dfeat * S)
        else:
            with rorc.operand(rounded):
                feat.backward(dfeat * S)
    finally:
        rorc.RESID, rorc.DGELU_BITS = None, None
    grads = {}
    for k, p in params.items():
        if p.grad is None:
            continue
        grads[k] = p.grad if k.startswith("head.") else p.grad / S
    return logits.detach(), float(loss), grads


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    op = dict(f16=torch.float16, bf16=torch.bfloat16)[sys.argv[3] if len(sys.argv) > 3 else "f16"]
    B, K, crop = 2, 9871, 224
    g = torch.Generator().manual_seed(seed)
    label = torch.randn(K, 512, generator=g) * 0.38
    label = label / label.norm(dim=1, keepdim=True)
    sd = orc.seeded_state(orc.encoder_shapes(depth, 8, (crop // 16) ** 2), seed)
    x = torch.randn(B, 3, 8, crop, crop, generator=g)
    teacher = torch.randn(B, K, generator=g) * 4
    t0 = time.time()
    lg0, loss0, g0 = step(sd, x, label, teacher, depth, None)
    print(f"# fp32 oracle step: {time.time() - t0:.1f} s; seed {seed}, depth {depth}, operand {op}", flush=True)
    cases = [("shipped (r5): fp32 residual stream", dict()),
             ("resid fwd: 16-bit x on the patch rows", dict(resid="fwd")),
             ("resid both: 16-bit x and dx on the patch rows", dict(resid="both")),
             ("dgelu code 8 bit (fp32 stream)", dict(dgelu_bits=8)),
             ("dgelu code 12 bit (fp32 stream)", dict(dgelu_bits=12))]
    print(f"{'variant':52s} {'logits':>9s} {'loss':>9s} {'worst grad':>10s}  (which)    median grad   patch_embed.w   blocks.0.fc2.w")
    for name, kw in cases:
        lg, loss, gr = step(sd, x, label, teacher, depth, op, **kw)
        errs = {k: rel(gr[k], g0[k]) for k in g0}
        wk = max(errs, key=errs.get)
        med = sorted(errs.values())[len(errs) // 2]
        print(f"{name:52s} {rel(lg, lg0):9.2e} {abs(loss - loss0) / abs(loss0):9.2e} {errs[wk]:10.2e}  {wk:28s} {med:9.2e} "
              f"{errs['patch_embed.proj.weight']:9.2e} {errs['blocks.0.mlp.fc2.weight']:9.2e}", flush=True)


if __name__ == "__main__":
    main()
