"""Order / diffusion transformer over clip embeddings and the CLIP-text tower, on the HIP path.

Module and parameter names follow the reference `lib/models/tfm_model.py` (ResidualAttentionBlock :32,
TemporalModelling :56, DiffusionTransformer :70) so `state_dict()` keys match
(`order_tfm.temporalModelling.resblocks.{j}.attn.in_proj_weight` ...).  The modules only own
parameters; the arithmetic runs through `tfm_engine.StackEngine` (LayerNorm / MFMA GEMM / MFMA
attention kernels).  Every random draw of the reference forward (`mask_inds` :145, per-sample
`pad_start` :283, four noise tensors :180) is an explicit, overridable input so parity tests can pin
them.
"""
import math
import os
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from .functional import StackFn, gelu_f32, linear_f32


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x) (tfm_model.py:27-29); fused into the c_fc GEMM epilogue on the HIP path."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, dropout=0.0, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head, dropout=dropout)  # parameter holder (same key names)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)
        self.attn_mask = attn_mask


class TemporalModelling(nn.Module):
    def __init__(self, width, layers, heads, dropout=0.0, attn_mask=None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.heads = heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, dropout, attn_mask) for _ in range(layers)])


def linear_beta_schedule(timesteps):
    """lib/models/diffusion_model.py:328-331"""
    return torch.linspace(0.0001, 0.02, timesteps)


def sinusoidal_embedding(time, dim):
    """SinusoidalPositionEmbeddings, lib/models/diffusion_model.py:34-47"""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=time.device) * -k)
    e = time[:, None].float() * freqs[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


class DiffusionTransformer(nn.Module):
    def __init__(self, num_seg=8, tfm_layers=4, tfm_heads=8, hidden_size=512, dropout=0.0, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.dropout = 0.0  # the reference hard-sets 0.0 (tfm_model.py:74)
        self.hidden_size = hidden_size
        self.num_seg = num_seg
        self.tfm_layers = tfm_layers
        self.tfm_heads = tfm_heads
        self.max_len = cfg.DEV.ORDER_PRETRAIN_MAX_LEN
        self.pad_embedding = nn.Embedding(1, hidden_size)
        self.type_embedding = nn.Embedding(2, hidden_size)
        self.temporalEmbedding = nn.Embedding(self.max_len, hidden_size)
        self.temporalModelling = TemporalModelling(hidden_size, tfm_layers, tfm_heads, self.dropout)
        # index 0 of the reference Sequential is the parameter-free sinusoidal embedding: keep the numbering
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(hidden_size // 4, hidden_size), nn.GELU(),
                                      nn.Linear(hidden_size, hidden_size))
        self.initialize_parameters()
        self.total_levels = tfm_layers
        self.level_batch = tfm_layers
        betas = linear_beta_schedule(self.total_levels)
        alphas_cumprod = torch.cumprod(1.0 - betas, 0)
        self.sqrt_alphas_cumprod = torch.sqrt(alphas_cumprod)                    # tfm_model.py:121
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - alphas_cumprod)    # tfm_model.py:122
        self._engine = None

    def initialize_parameters(self):  # tfm_model.py:251-263
        nn.init.normal_(self.pad_embedding.weight, std=0.01)
        nn.init.normal_(self.temporalEmbedding.weight, std=0.01)
        width, layers = self.temporalModelling.width, self.temporalModelling.layers
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = width ** -0.5
        fc_std = (2 * width) ** -0.5
        for block in self.temporalModelling.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)

    # ------------------------------------------------------------------------------------
    def bind(self, owner):
        """owner: the VisionTransformer that provides the weight cache / gradient store."""
        self._owner = [owner]

    def _stack(self, x_bt, nseq, S, kpm):
        own = self._owner[0]
        return StackFn.apply(x_bt, own, tuple(self.temporalModelling.resblocks), nseq, S, False, kpm, self.tfm_heads,
                             own.anchor())

    def _time_mlp(self, t):
        e = sinusoidal_embedding(t, self.hidden_size // 4)
        h = linear_f32(e, self.time_mlp[1].weight, self.time_mlp[1].bias)
        h = gelu_f32(h)
        return linear_f32(h, self.time_mlp[3].weight, self.time_mlp[3].bias)

    def _ennoise(self, x_start, noise, t_index):  # tfm_model.py:291-302
        a = float(self.sqrt_alphas_cumprod[t_index])
        b = float(self.sqrt_one_minus_alphas_cumprod[t_index])
        return a * x_start + b * noise

    def draw(self, batch_size, device, dtype=torch.float32):
        """The reference's RNG draws for one pre-training forward (tfm_model.py:145, 279-286, 180)."""
        L = self.max_len
        mask_inds = torch.randint(0, L, (batch_size,), device=device)
        lo = mask_inds + 1                       # randint(mask + 1, max_len) when the mask is not the last token
        span = L - lo
        r = torch.floor(torch.rand(batch_size, device=device) * span.clamp(min=1)).long()
        pad_start = torch.where(span > 0, torch.clamp(lo + r, max=L - 1), torch.full_like(lo, L))
        noises = [torch.randn(batch_size, self.hidden_size, device=device, dtype=dtype) for _ in range(self.tfm_layers)]
        return dict(mask_inds=mask_inds, pad_start=pad_start, noises=noises)

    def forward(self, x, is_pretrain=False, rng=None):
        if self.training and is_pretrain:
            return self.forward_pretrain(x, rng)
        return self.diffusion_signal_forecast(x)

    def forward_pretrain(self, x, rng=None):
        """x fp32 [(b t), c] video embeddings (t = max_len clips per video).  Returns
        (denoised [b,c], mask_inds [b], [x0 repeated [levels*b, c], intermediate [levels*b, c]], intermediate)
        exactly like tfm_model.py:129-156."""
        L, C = self.max_len, self.hidden_size
        dev = x.device
        b = x.shape[0] // L
        if rng is None:
            rng = self.draw(b, dev)
        mask_inds, pad_start = rng["mask_inds"].to(dev), rng["pad_start"].to(dev)
        bs = torch.arange(b, device=dev)
        feats = x.view(b, L, C)                                  # batch-first view of '(b t) c'
        rows = bs * L + mask_inds                                # row of each video's masked clip in the '(b t) c' matrix
        # (index_select: its backward is index_add_, which a HIP graph can hold; advanced indexing's backward is a
        #  sort-based index_put that crashes stream capture on ROCm 7.x)
        x0 = x.index_select(0, rows)                             # clip_feats_x0 (copy), tfm_model.py:148
        pos = torch.arange(L, device=dev)[None, :]
        pad_mask = pos >= pad_start[:, None]                     # [b, L] bool, True = padded key
        feats = torch.where(pad_mask[:, :, None], self.pad_embedding.weight[0][None, None, :], feats)
        kpm = pad_mask.to(torch.uint8).contiguous()
        temb = self.temporalEmbedding.weight[None, :, :]         # positions 0..L-1
        is_mask = (pos == mask_inds[:, None])[:, :, None]
        type_emb = torch.where(is_mask, self.type_embedding.weight[1][None, None, :],
                               self.type_embedding.weight[0][None, None, :])
        intermediate = []
        denoised = None
        for time_i in range(self.tfm_layers):
            t_index = self.total_levels - 1 - time_i
            src = x0.detach() if time_i == 0 else denoised.detach()
            noisy = self._ennoise(src, rng["noises"][time_i].to(dev), t_index)
            cur = torch.where(is_mask, noisy[:, None, :], feats)
            t = torch.full((b,), t_index, device=dev, dtype=torch.long)
            cur = cur + type_emb + temb + self._time_mlp(t)[:, None, :]
            out = self._stack(cur.reshape(b * L, C).contiguous(), b, L, kpm).view(b, L, C)
            denoised = out.reshape(b * L, C).index_select(0, rows)
            intermediate.append(denoised)
        x0_rep = x0.unsqueeze(0).expand(self.total_levels, -1, -1).reshape(-1, C)
        inter = torch.cat(intermediate)
        return denoised, mask_inds, [x0_rep, inter], inter

    def diffusion_signal_forecast(self, x):
        """Zero-shot / fine-tuning step forecasting, tfm_model.py:206-249: append a zero 'noise' token after the
        num_seg observed clips and denoise it through the levels (noise is all-zero in the reference)."""
        L, C = self.max_len, self.hidden_size
        dev = x.device
        n = self.num_seg
        b = x.shape[0] // n
        assert n + 1 == L, "forecast appends one token: num_seg + 1 must equal ORDER_PRETRAIN_MAX_LEN"
        feats = torch.cat([x.view(b, n, C), torch.zeros(b, 1, C, device=dev, dtype=x.dtype)], 1)
        pos = torch.arange(L, device=dev)[None, :]
        is_mask = (pos == L - 1).expand(b, L)[:, :, None]
        temb = self.temporalEmbedding.weight[None, :, :]
        type_emb = torch.where(is_mask, self.type_embedding.weight[1][None, None, :],
                               self.type_embedding.weight[0][None, None, :])
        cur = feats
        denoised = None
        for time_i in range(self.tfm_layers):
            t_index = self.total_levels - 1 - time_i
            if time_i != 0:
                noisy = self._ennoise(denoised.detach(), torch.zeros_like(denoised), t_index)
                cur = torch.where(is_mask, noisy[:, None, :], feats)
            t = torch.full((b,), t_index, device=dev, dtype=torch.long)
            inp = cur + type_emb + temb + self._time_mlp(t)[:, None, :]
            out = self._stack(inp.reshape(b * L, C).contiguous(), b, L, None).view(b, L, C)
            denoised = out[:, L - 1]
        return denoised


class ClipTextModel(nn.Module):
    """The text half of openai/CLIP ViT-B/16 (third-party, un-vendored in the reference: `clip.load`,
    lib/models/vit.py:258-261).  Published architecture: 12 layers, width 512, 8 heads, context 77,
    vocab 49408, causal mask, ln_final, EOT-token (argmax id) pooling, text_projection.  Parameter
    names follow CLIP's so a released ProcedureVRL checkpoint (`text_model.*`) loads."""

    def __init__(self, layers=12, width=512, heads=8, context_length=77, vocab_size=49408, embed_dim=512):
        super().__init__()
        self.context_length = context_length
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = TemporalModelling(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=width ** -0.5)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=(2 * width) ** -0.5)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=width ** -0.5)
        self.use_graphs = os.environ.get("PVRL_HIP_GRAPHS", "1") == "1"
        self._graphs, self._gseen = {}, {}

    def bind(self, owner):
        self._owner = [owner]

    def release_graphs(self):       # (engine.GraphReplay.release_graphs)
        from .engine import drop_graphs_quietly
        drop_graphs_quietly(self._graphs)

    @torch.no_grad()
    def encode_text(self, text):
        """text int64 [n, 77] -> fp32 [n, embed_dim]; frozen, forward only.  The ~110 launches of the tower are replayed
        from a HIP graph after two eager calls of a shape (same switch as the encoder: PVRL_HIP_GRAPHS=0 disables)."""
        if not (text.is_cuda and self.use_graphs):
            return self._encode_text(text)
        blk0 = self.transformer.resblocks[0]
        key = (tuple(text.shape), text.dtype, text.device.index, self.text_projection.data_ptr(),
               self.token_embedding.weight.data_ptr(), self.text_projection._version, blk0.attn.in_proj_weight._version,
               blk0.mlp.c_fc.weight._version)      # a checkpoint load (in-place copy) is a new key
        g = self._graphs.get(key)
        if g is None:
            n = self._gseen.get(key, 0)
            self._gseen[key] = n + 1
            if n < 2 or len(self._graphs) >= 4:
                return self._encode_text(text)
            try:
                st = text.clone()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self._encode_text(st)
                g = self._graphs[key] = (graph, st, out)
            except Exception as e:          # never fatal: the eager launches are the same kernels
                import warnings
                warnings.warn(f"HIP graph capture of the text tower failed ({type(e).__name__}: {e}); launching eagerly")
                self.use_graphs = False
                torch.cuda.synchronize()
                return self._encode_text(text)
        graph, st, out = g
        st.copy_(text)
        graph.replay()
        return out.clone()

    def _encode_text(self, text):
        from .tfm_engine import StackEngine
        own = self._owner[0]
        n, S = text.shape
        x = self.token_embedding.weight[text] + self.positional_embedding[None, :S]
        eng = StackEngine(self.transformer.resblocks, own.weight_cache, None, heads=self.transformer.heads)
        y, _ = eng.forward(x.reshape(n * S, -1).contiguous().float(), n, S, causal=True, save=False)
        eot = text.argmax(dim=-1) + torch.arange(n, device=text.device) * S
        pooled = y[eot].contiguous()
        pooled, _, _ = ops.layernorm_fwd(pooled, self.ln_final.weight, self.ln_final.bias, 1e-5,
                                         out_dtype=torch.float32, save_stats=False)
        return ops.gemm_nt_f32(pooled, self.text_projection.t().contiguous())
