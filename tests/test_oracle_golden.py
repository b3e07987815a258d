"""The CPU oracle (oracle/timesformer_oracle.py) against the golden vectors produced by the UNMODIFIED reference
(tests/golden/make_golden.py).  This is what pins the oracle; fp32 vs fp32, so the bar is 1e-5 relative."""
import os

import pytest
import torch

from oracle import timesformer_oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def load(name):
    return torch.load(os.path.join(G, name + ".pt"), weights_only=False)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values()))


@pytest.mark.parametrize("name", ["attn_s8", "attn_s197"])
def test_attention(name):
    f = load(name)
    shapes = {"qkv.weight": (2304, 768), "qkv.bias": (2304,), "proj.weight": (768, 768), "proj.bias": (768,)}
    sd = orc.seeded_state(shapes, f["seed"])
    assert abs(checksum(sd) - f["wsum"]) < 1e-6 * f["wsum"]
    assert rel(orc.attention(sd, "", f["x"]), f["y"]) < TOL


def block_state(seed):
    sh = {k[len("blocks.0."):]: v for k, v in orc.encoder_shapes(1).items() if k.startswith("blocks.0.")}
    return orc.seeded_state(sh, seed)


def test_block_forward_backward():
    f = load("block")
    sd = {k: v.requires_grad_(True) for k, v in block_state(f["seed"]).items()}
    assert abs(checksum(sd) - f["wsum"]) < 1e-6 * f["wsum"]
    x = f["x"].clone().requires_grad_(True)
    y = orc.block(sd, "", x, f["B"], f["T"], f["W"])
    assert rel(y, f["y"]) < TOL
    y.backward(f["dy"])
    assert rel(x.grad, f["dx"]) < TOL
    for k, g in f["grads"].items():
        assert rel(sd[k].grad, g) < 1e-4, k
    for k, s in f["grad_sums"].items():
        assert abs(float(sd[k].grad.double().abs().sum()) - s) < 1e-3 * max(s, 1e-6), k


def e2e_state(f):
    K = f["K"]
    sh = orc.encoder_shapes(f["depth"], 8, (f["crop"] // 16) ** 2)
    sh.update(orc_order_shapes())
    sh.update(orc_text_shapes(f["text_layers"]))
    sh = {"model." + k: v for k, v in sh.items()}
    return sh


def orc_order_shapes(layers=4, w=512, L=9):
    sh = {"order_tfm.pad_embedding.weight": (1, w), "order_tfm.type_embedding.weight": (2, w),
          "order_tfm.temporalEmbedding.weight": (L, w), "order_tfm.time_mlp.1.weight": (w, w // 4),
          "order_tfm.time_mlp.1.bias": (w,), "order_tfm.time_mlp.3.weight": (w, w), "order_tfm.time_mlp.3.bias": (w,)}
    sh.update(stack_shapes("order_tfm.temporalModelling.", layers, w))
    return sh


def stack_shapes(pre, layers, w):
    sh = {}
    for i in range(layers):
        p = f"{pre}resblocks.{i}."
        sh.update({p + "attn.in_proj_weight": (3 * w, w), p + "attn.in_proj_bias": (3 * w,),
                   p + "attn.out_proj.weight": (w, w), p + "attn.out_proj.bias": (w,),
                   p + "ln_1.weight": (w,), p + "ln_1.bias": (w,), p + "ln_2.weight": (w,), p + "ln_2.bias": (w,),
                   p + "mlp.c_fc.weight": (4 * w, w), p + "mlp.c_fc.bias": (4 * w,),
                   p + "mlp.c_proj.weight": (w, 4 * w), p + "mlp.c_proj.bias": (w,)})
    return sh


def orc_text_shapes(layers, w=512):
    sh = {"text_model.token_embedding.weight": (49408, w), "text_model.positional_embedding": (77, w),
          "text_model.ln_final.weight": (w,), "text_model.ln_final.bias": (w,), "text_model.text_projection": (w, w),
          "text_model.logit_scale": ()}
    sh.update(stack_shapes("text_model.transformer.", layers, w))
    return sh


def test_e2e_state_dict_keys_match_reference():
    f = load("e2e")
    assert sorted(e2e_state(f).keys()) == f["state_keys"]


def test_e2e_train_forward_loss_grads():
    f = load("e2e")
    full = orc.seeded_state(e2e_state(f), f["seed"])
    assert abs(checksum(full) - f["wsum"]) < 1e-6 * f["wsum"]
    sd = {k[len("model."):]: v.requires_grad_(not k.startswith("model.text_model")) for k, v in full.items()}
    meta = {"clip_text_ids": f["clip_text_ids"], "clip_vis_feat": f["clip_vis_feat"]}
    pred, teacher, mse = orc.vit_forward_train(sd, f["inputs"], meta, f["label_emb"], 0.02, f["depth"], 9, 4,
                                               f["text_layers"], f["rng"])
    assert rel(pred, f["pred"]) < 1e-4
    assert rel(teacher, f["teacher"]) < 1e-4
    assert rel(mse[0], f["mse0"]) < 1e-4 and rel(mse[1], f["mse1"]) < 1e-4
    loss, l1, l2 = orc.pretrain_loss(pred, teacher, mse, 5)
    assert abs(float(l1) - f["loss1"]) < 1e-4 * abs(f["loss1"])
    assert abs(float(l2) - f["loss2"]) < 1e-4 * abs(f["loss2"])
    loss.backward()
    for k, g in f["grads"].items():
        assert rel(sd[k[len("model."):]].grad, g) < 2e-3, k
    for k, s in f["grad_sums"].items():
        got = float(sd[k[len("model."):]].grad.double().abs().sum())
        assert abs(got - s) < 5e-3 * max(s, 1e-6), (k, got, s)


def test_forward_features_eval():
    f = load("features")
    e = load("e2e")
    full = orc.seeded_state(e2e_state(e), f["seed"])
    sd = {k[len("model."):]: v for k, v in full.items()}
    with torch.no_grad():
        feat = orc.forward_features(sd, f["x"], e["depth"])
    assert rel(feat, f["feat"]) < TOL


def test_milnce():
    f = load("milnce")
    assert abs(float(orc.milnce(f["v"], f["t"])) - f["loss"]) < 1e-5 * abs(f["loss"])


def test_droppath_formula():
    f = load("droppath_formula")
    assert rel(orc.drop_path_apply(f["x"], f["mask"]), f["y"]) < 1e-6


def test_allgather_semantics():
    f = load("allgather")
    out, grads = orc.allgather_forward_backward([r["x"] for r in f], [r["w"] for r in f])
    for r, g in zip(f, grads):
        assert torch.equal(out, r["y"])
        assert torch.equal(g, r["grad"])


def test_lr_table():
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd import lr_policy
    tab = load("lr_table")
    sched = {"procedurevrl_sgd": dict(BASE_LR=0.005, STEPS=[0, 2, 4], MAX_EPOCH=5),
             "procedurevrl_adamw": dict(BASE_LR=5e-5, STEPS=[0, 15, 23], MAX_EPOCH=25)}
    for name, t in tab.items():
        cfg = get_cfg()
        cfg.SOLVER.LR_POLICY = "steps_with_relative_lrs"
        cfg.SOLVER.LRS = [1, 0.1, 0.01]
        for k, v in sched[name].items():
            cfg.SOLVER[k] = v
        for ep, lr in zip(t["epochs"], t["lrs"]):
            assert abs(orc.lr_at_epoch(cfg, ep) - lr) < 1e-12
            assert abs(lr_policy.get_lr_at_epoch(cfg, ep) - lr) < 1e-12


def forecast_state(f):
    sh = orc.encoder_shapes(f["depth"], 8, (f["crop"] // 16) ** 2)
    sh.update(orc_order_shapes())
    return {"model." + k: v for k, v in sh.items()}


def test_forecast_eval():
    f = load("forecast")
    shapes = forecast_state(f)
    assert sorted(shapes.keys()) == f["state_keys"]
    full = orc.seeded_state(shapes, f["seed"])
    assert abs(checksum(full) - f["wsum"]) < 1e-6 * f["wsum"]
    sd = {k[len("model."):]: v for k, v in full.items()}
    with torch.no_grad():
        probs = orc.vit_forward_forecast_eval(sd, f["x"], f["label_emb"], 0.02, f["depth"], 8)
    assert rel(probs, f["probs"]) < 1e-4


def test_input_pipeline_oracle_and_host_draws_match_reference():
    """tests/golden/input_pipeline.pt: the reference's tensor_normalize + spatial_sampling on random uint8 clips.
    (1) the host mirror draws the same (size, offsets, flip) from the same numpy seed; (2) the oracle restatement of
    the chain reproduces the reference's output from those draws."""
    import numpy as np
    from procedurevrl_amd.transform import spatial_sampling_params
    g = load("input_pipeline")
    for c in g["cases"]:
        np.random.seed(c["seed"])
        prm = spatial_sampling_params(c["H0"], c["W0"], c["spatial_idx"], c["min_scale"], c["max_scale"], c["crop"],
                                      c["flip"], c["inv"])
        got = orc.input_pipeline(c["frames"], prm, g["mean"], g["std"], c["crop"])
        assert got.shape == c["out"].shape
        assert torch.allclose(got, c["out"], atol=1e-6, rtol=1e-6), (prm, (got - c["out"]).abs().max())


def _mvit_state(g, frames, crop):
    from oracle import mvit_oracle as mo
    sd = orc.seeded_state(mo.encoder_shapes(g["mvit"], frames, crop), g["seed"])
    assert abs(sum(float(v.double().abs().sum()) for v in sd.values()) - g["checksum"]) < 1e-6 * g["checksum"]
    return sd


def test_mvit_oracle_small_forward_blocks_and_grads():
    """oracle/mvit_oracle.py vs the reference MViT_encoder on the reduced 4-block geometry (tests/golden/mvit_small.pt):
    state_dict keys, every block's output, the features and the gradients of every kind of parameter."""
    from oracle import mvit_oracle as mo
    g = load("mvit_small")
    c = g["cfg"]
    sd = _mvit_state(g, c["frames"], c["crop"])
    assert sorted(sd) == g["keys"]
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    outs = []
    feat = mo.forward_features(p, g["x"], g["mvit"], block_outputs=outs)
    for i, (a, b) in enumerate(zip(outs, g["blocks"])):
        assert torch.allclose(a, b, atol=2e-5, rtol=2e-5), (i, (a - b).abs().max())
    assert torch.allclose(feat, g["feat"], atol=2e-5, rtol=2e-5)
    (feat * g["gout"]).sum().backward()
    for n, ref in g["grads"].items():
        got = p[n].grad[:64] if p[n].grad.dim() == 2 else p[n].grad
        # norm_k.bias shifts every key by the same vector: softmax-invariant, its true gradient is 0 and the reference's
        # value is rounding noise (4e-6 in norm) -> absolute floor next to the relative bound
        assert (got - ref).norm() < 2e-4 * ref.norm() + 1e-6 * ref.numel() ** 0.5, (n, float((got - ref).norm()), float(ref.norm()))


def test_mvit_oracle_full_size_features():
    """MViTv2-S geometry (16 x 224^2, 16 blocks, 34 M parameters): parameter shapes and one clip's features."""
    from oracle import mvit_oracle as mo
    g = load("mvit_s")
    assert {k: tuple(v) for k, v in g["shapes"].items()} == mo.encoder_shapes(g["mvit"], 16, 224)
    sd = _mvit_state(g, 16, 224)
    x = torch.randn(1, 3, 16, 224, 224, generator=torch.Generator().manual_seed(g["x_seed"]))
    with torch.no_grad():
        feat = mo.forward_features(sd, x, g["mvit"])
    assert torch.allclose(feat, g["feat"], atol=5e-5, rtol=5e-5), (feat - g["feat"]).abs().max()


def test_mvit_wrapper_e2e_train_forward_loss_grads():
    """The reference's registered MViT model (tests/golden/mvit_e2e.pt): the oracle's wrapper restatement around the
    MViT encoder restatement reproduces (pred, teacher, mse), both losses and the stored gradients."""
    from oracle import mvit_oracle as mo
    f = load("mvit_e2e")
    c = f["cfg"]
    sh = {"video_encoder." + k: v for k, v in mo.encoder_shapes(f["mvit"], c["frames"], c["crop"]).items()}
    last = sh["video_encoder.norm.weight"][0]
    sh.update({"head.weight": (512, last), "head.bias": (512,)})
    sh.update(orc_order_shapes())
    sh.update(orc_text_shapes(f["text_layers"]))
    full = orc.seeded_state({"model." + k: v for k, v in sh.items()}, f["seed"])
    assert sorted(full.keys()) == f["state_keys"]
    assert abs(checksum(full) - f["wsum"]) < 1e-6 * f["wsum"]
    sd = {k[len("model."):]: v.requires_grad_(not k.startswith("model.text_model")) for k, v in full.items()}
    enc = {k[len("video_encoder."):]: v for k, v in sd.items() if k.startswith("video_encoder.")}
    meta = {"clip_text_ids": f["clip_text_ids"], "clip_vis_feat": f["clip_vis_feat"]}
    pred, teacher, mse = orc.vit_forward_train(sd, f["inputs"], meta, f["label_emb"], 0.02, None, 9, 4, f["text_layers"], f["rng"],
                                               encoder=lambda x: mo.forward_features(enc, x, f["mvit"]))
    assert rel(pred, f["pred"]) < 1e-4 and rel(teacher, f["teacher"]) < 1e-4
    assert rel(mse[0], f["mse0"]) < 1e-4 and rel(mse[1], f["mse1"]) < 1e-4
    loss, l1, l2 = orc.pretrain_loss(pred, teacher, mse, 5)
    assert abs(float(l1) - f["loss1"]) < 1e-4 * abs(f["loss1"]) and abs(float(l2) - f["loss2"]) < 1e-4 * abs(f["loss2"])
    loss.backward()
    for k, g in f["grads"].items():
        assert rel(sd[k[len("model."):]].grad, g) < 2e-3, k


def test_pos_time_embed_resize():
    """Input with another frame count / patch grid than the model was built for (tests/golden/embed_interp.pt): the
    nearest-neighbour resize of pos_embed and time_embed (vit.py:374-386,398-402)."""
    f = load("embed_interp")
    full = orc.seeded_state(e2e_state(f), f["seed"])
    assert abs(checksum(full) - f["wsum"]) < 1e-6 * f["wsum"]
    sd = {k[len("model."):]: v for k, v in full.items()}
    with torch.no_grad():
        feat = orc.forward_features(sd, f["x"], f["depth"])
    assert rel(feat, f["feat"]) < 1e-5


def mvit_droppath_scales(g):
    """the reference's captured torch.rand draws -> per-block (s_attn, s_mlp) = floor(keep + u) / keep (common.py:38-52)"""
    depth = g["cfg"]["depth"]
    rates = [x.item() for x in torch.linspace(0, g["rate"], depth)]
    draws = list(g["rand"])
    dp = []
    for r in rates:
        if r == 0.0:
            dp.append(None)
        else:
            keep = 1.0 - r
            dp.append((torch.floor(keep + draws.pop(0)) / keep, torch.floor(keep + draws.pop(0)) / keep))
    assert not draws
    return dp


def test_mvit_oracle_droppath():
    """Reference MViT_encoder in train mode with DROPPATH_RATE 0.2 (tests/golden/mvit_droppath.pt, rand draws captured)."""
    from oracle import mvit_oracle as mo
    g = load("mvit_droppath")
    c = g["cfg"]
    sd = _mvit_state(g, c["frames"], c["crop"])
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat = mo.forward_features(p, g["x"], g["mvit"], droppath=mvit_droppath_scales(g))
    assert torch.allclose(feat, g["feat"], atol=2e-5, rtol=2e-5)
    (feat * g["gout"]).sum().backward()
    for n, ref in g["grads"].items():
        got = p[n].grad[:64] if p[n].grad.dim() == 2 else p[n].grad
        assert (got - ref).norm() < 2e-4 * ref.norm() + 1e-6 * ref.numel() ** 0.5, n


def test_rounded_oracle_without_rounding_is_the_oracle():
    """oracle/rounded_oracle.py (the oracle + the HIP datapath's rounding points) with rounding switched off must BE the
    fp32 oracle -- forward and backward, DropPath included -- which pins it to the reference through the goldens above; and
    against the reference's own block it must sit where a 16-bit operand datapath sits (a few 1e-3), not elsewhere."""
    from oracle import rounded_oracle as rorc
    f = load("block")
    outs = {}
    for name, blockfn, dtype in (("orc", orc.block, None), ("off", rorc.block, None), ("bf16", rorc.block, torch.bfloat16),
                                 ("f16", rorc.block, torch.float16)):
        sd = {k: v.requires_grad_(True) for k, v in block_state(f["seed"]).items()}
        x = f["x"].clone().requires_grad_(True)
        B, T = f["B"], f["T"]
        N = (x.shape[1] - 1) // T
        g = torch.Generator().manual_seed(0)
        dp = tuple(torch.floor(0.7 + torch.rand(n, generator=g)) / 0.7 for n in (B * N, B * T, B))
        with rorc.operand(dtype):
            y = blockfn(sd, "", x, B, T, f["W"], dp=dp)
        y.backward(f["dy"])
        outs[name] = (y.detach(), x.grad, {k: v.grad for k, v in sd.items()})
    y0, dx0, g0 = outs["orc"]
    y1, dx1, g1 = outs["off"]
    assert rel(y1, y0) < 1e-6 and rel(dx1, dx0) < 1e-6
    for k in g0:
        assert rel(g1[k], g0[k]) < 1e-5, k
    e_bf, e_f16 = rel(outs["bf16"][0], y0), rel(outs["f16"][0], y0)
    assert 2e-4 < e_bf < 1e-2, e_bf                 # bf16 operands: 2^-9 per rounding
    assert e_f16 < e_bf / 4, (e_f16, e_bf)          # fp16 operands: three more mantissa bits


def test_rounded_oracle_features_match_oracle_when_off():
    from oracle import rounded_oracle as rorc
    f = load("features")
    e = load("e2e")
    full = orc.seeded_state(e2e_state(e), f["seed"])
    sd = {k[len("model."):]: v for k, v in full.items()}
    with torch.no_grad():
        a = orc.forward_features(sd, f["x"], e["depth"])
        b = rorc.forward_features(sd, f["x"], e["depth"])
    assert rel(b, a) < 1e-6


def test_clip_text_oracle_matches_hf_clip_port():
    """Row T1's arithmetic lives in third-party openai/CLIP (`clip.load("ViT-B/16")`, reference lib/models/vit.py:258),
    which is absent here.  The `transformers` package in this image carries an independent, widely used port of that
    text encoder (CLIPTextModelWithProjection, validated upstream against openai's weights): with the SAME random weights
    mapped by name, at the real geometry (12 layers, width 512, 8 heads, context 77, vocabulary 49,408, quick_gelu,
    causal mask, EOT pooling, text_projection), the oracle's `clip_encode_text` must reproduce its text embeddings."""
    transformers = pytest.importorskip("transformers")
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    from procedurevrl_amd.datasets import synthetic_text_ids
    layers = 12
    sd = orc.seeded_state(orc_text_shapes(layers), 5)
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512,
                         num_hidden_layers=layers, num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu",
                         layer_norm_eps=1e-5, attention_dropout=0.0, eos_token_id=49407, bos_token_id=49406, pad_token_id=0)
    cfg._attn_implementation = "eager"
    hf = CLIPTextModelWithProjection(cfg).eval()
    m = {"text_model.embeddings.token_embedding.weight": sd["text_model.token_embedding.weight"],
         "text_model.embeddings.position_embedding.weight": sd["text_model.positional_embedding"],
         "text_model.final_layer_norm.weight": sd["text_model.ln_final.weight"],
         "text_model.final_layer_norm.bias": sd["text_model.ln_final.bias"],
         "text_projection.weight": sd["text_model.text_projection"].t().contiguous()}
    for i in range(layers):
        src, dst = f"text_model.transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        w, b = sd[src + "attn.in_proj_weight"], sd[src + "attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            m[dst + f"self_attn.{n}.weight"] = w[512 * j:512 * (j + 1)]
            m[dst + f"self_attn.{n}.bias"] = b[512 * j:512 * (j + 1)]
        for a, c in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                     ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            m[dst + c + ".weight"] = sd[src + a + ".weight"]
            m[dst + c + ".bias"] = sd[src + a + ".bias"]
    missing, unexpected = hf.load_state_dict(m, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    ids = synthetic_text_ids(6, torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = hf(input_ids=ids).text_embeds
        got = orc.clip_encode_text(sd, "text_model.", ids, layers)
    assert rel(got, want) < 2e-5, rel(got, want)


def test_rounded_oracle_split_residual_stream_switch():
    """Round 6's rounding points -- the patch rows of the residual stream (and of its gradient) stored in the operand type, RESID = "fwd" /
    "both" -- are inert without an operand type (the model stays THE oracle), change values only on the patch rows' path, and cost what
    one more 16-bit rounding per residual add costs (features within 2x of the stream-less model's error), not more."""
    from oracle import rounded_oracle as rorc
    f = load("features")
    e = load("e2e")
    full = orc.seeded_state(e2e_state(e), f["seed"])
    sd = {k[len("model."):]: v for k, v in full.items()}
    try:
        with torch.no_grad():
            a = orc.forward_features(sd, f["x"], e["depth"])
            rorc.RESID = "both"
            off = rorc.forward_features(sd, f["x"], e["depth"])             # no operand type: nothing is rounded
            with rorc.operand(torch.float16):
                both = rorc.forward_features(sd, f["x"], e["depth"])
                rorc.RESID = None
                none = rorc.forward_features(sd, f["x"], e["depth"])
        assert rel(off, a) < 1e-6
        assert 0 < rel(both, none) < 1e-3                                     # the switch does something, and something small
        assert rel(both, a) < 2.0 * rel(none, a) + 1e-5, (rel(both, a), rel(none, a))
        # gradients: "fwd" leaves the gradient stream alone, "both" rounds it -- both stay a 16-bit datapath's distance from fp32
        outs = {}
        for mode in (None, "fwd", "both"):
            p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
            rorc.RESID = mode
            with rorc.operand(torch.float16):
                (rorc.forward_features(p, f["x"], e["depth"]) * 64.0).sum().backward()
            outs[mode] = p["blocks.0.mlp.fc1.weight"].grad
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        (orc.forward_features(p, f["x"], e["depth"]) * 64.0).sum().backward()
        ref = p["blocks.0.mlp.fc1.weight"].grad
        for mode in ("fwd", "both"):
            assert rel(outs[mode], ref) < 2.0 * rel(outs[None], ref) + 1e-4, (mode, rel(outs[mode], ref), rel(outs[None], ref))
    finally:
        rorc.RESID = None
