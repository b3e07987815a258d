"""Kernel schedule of the pre-training head -- everything between the encoder's clip features and the loss
(reference lib/models/vit.py:298-352 + lib/models/tfm_model.py:129-204, 272-302) -- with a HAND-WRITTEN backward, so that
forward and backward are two HIP graphs instead of ~1,000 eager launches wired through torch.autograd.

    feat [9b, 768] --head--> x --l2norm--> v [9b, 512] --step logits--> logits [9b, K]
    v --order / diffusion transformer (mask one clip per video, pad a tail, 4 denoise levels of a 4-layer stack)--> inter [4b, 512]
    inter --l2norm, step logits--> inter_pred [4b, K];  pred = cat(logits[perm], inter_pred), mse = [x0 repeated per level, inter]
    teacher = cat(teacher[perm], teacher[masked rows] per level): `assemble_teacher`, OUTSIDE the graphs -- nothing above reads the
    teacher logits, so the frozen text tower that produces them runs on its side stream UNDER this head's forward (vit._teacher_begin)

Round 2 left this part eager: capturing it needed torch.autograd INSIDE a stream capture, which crashes on ROCm 7.x
(an AccumulateGrad node is bound to the default stream).  Here autograd sees ONE Function (`PretrainHeadFn`): its forward and
backward replay graphs captured from the plain kernel schedule below; parameter gradients are written into the model's flat
gradient buffer like the encoder engines do.  The reference's random draws (mask_inds, pad_start, the noise tensors, the
output permutation) are explicit INPUTS of the graphs, so pinned draws (parity tests) and fresh ones take the same path.
"""
import os

import torch

from . import ops
from .engine import SCALED_GRADS
from .tfm_engine import StackEngine

F32 = torch.float32


class PretrainHeadEngine:
    GRAPH_WARMUP = 2
    GRAPH_MAX_KEYS = 2

    def __init__(self, owner):
        self.o = owner
        self.use_graphs = os.environ.get("PVRL_HIP_GRAPHS", "1") == "1"
        self._graphs, self._gseen = {}, {}
        self._pool = None
        self._cap = None
        self._cap_seen = set()
        self._gkey = None
        self.saved = None
        self._tconst = None

    def release_graphs(self):       # (engine.GraphReplay.release_graphs)
        from .engine import drop_graphs_quietly
        drop_graphs_quietly(self._graphs)
        self._gkey = None

    # ------------------------------------------------------------------ plumbing
    def params(self):
        o = self.o
        return [p for p in list(o.head.parameters()) + list(o.order_tfm.parameters()) if p.requires_grad]

    def _wc(self, p):
        """16-bit operand copies of a stack weight (the encoder engine's cache).  While the forward is being captured the
        first use of every weight re-casts it INSIDE the graph: a replay must refresh the copies after an optimiser step."""
        eng = self.o.engine
        if self._cap == "fwd" and id(p) not in self._cap_seen:
            self._cap_seen.add(id(p))
            e = eng._w.get(id(p))
            if e is not None:
                e.ver = None
        return eng._weight(p)

    def draws(self, rng, b, n, dev):
        """the reference's random draws of one forward (tfm_model.py:145, 279-286, 180; vit.py:345), pinned by `rng` or fresh"""
        ot = self.o.order_tfm
        d = (rng or {}).get("order") or ot.draw(b, dev)
        rand = (rng or {}).get("rand_inds")
        if rand is None:        # torch.randperm syncs; argsort of uniforms is the same distribution
            rand = torch.rand(n, device=dev).argsort()
        noises = torch.stack([x.to(dev, F32) for x in d["noises"]])
        return dict(mask_inds=d["mask_inds"].to(dev).long(), pad_start=d["pad_start"].to(dev).long(), noises=noises,
                    rand_inds=rand.to(dev).long())

    def _time_consts(self, dev):
        ot = self.o.order_tfm
        if self._tconst is None or self._tconst.device != dev:
            from .tfm_model import sinusoidal_embedding
            t = torch.tensor([ot.total_levels - 1 - i for i in range(ot.tfm_layers)], device=dev)
            self._tconst = sinusoidal_embedding(t, ot.hidden_size // 4).contiguous()       # [levels, C/4]
        return self._tconst

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def _forward(self, feat, dr, save=True):
        o, ot = self.o, self.o.order_tfm
        L, C, lv = ot.max_len, ot.hidden_size, ot.tfm_layers
        dev = feat.device
        n = feat.shape[0]
        b = n // L
        le, le_t = o._labels(dev)
        it = 1.0 / o.temp
        x = ops.gemm_nt_f32(feat.contiguous(), o.head.weight.detach().contiguous(), o.head.bias.detach())
        v, vinv = ops.l2norm_fwd(x)
        o.last_video_emb = v
        logits = ops.gemm_nt_f32(v, le, alpha=it)
        mask_inds, pad_start, noises, rand_inds = dr["mask_inds"], dr["pad_start"], dr["noises"], dr["rand_inds"]
        rows = torch.arange(b, device=dev) * L + mask_inds              # row of each video's masked clip in '(b t) c'
        x0 = v.index_select(0, rows)
        pos = torch.arange(L, device=dev)[None, :]
        pad_mask = (pos >= pad_start[:, None])[:, :, None]             # [b, L, 1] True = padded key
        is_mask = (pos == mask_inds[:, None])[:, :, None]
        pw, tw = ot.pad_embedding.weight.detach(), ot.type_embedding.weight.detach()
        feats = torch.where(pad_mask, pw[0][None, None, :], v.view(b, L, C))
        kpm = pad_mask[:, :, 0].to(torch.uint8).contiguous()
        type_emb = torch.where(is_mask, tw[1][None, None, :], tw[0][None, None, :])
        temb = ot.temporalEmbedding.weight.detach()[None, :, :]
        # time_mlp (tfm_model.py:89-94) of the levels' constant time indices, all levels in one go (the reference evaluates
        # it per sample; the b rows of a level are identical)
        E = self._time_consts(dev)
        m1, m3 = ot.time_mlp[1], ot.time_mlp[3]
        h1 = ops.gemm_nt_f32(E, m1.weight.detach().contiguous(), m1.bias.detach())
        g1 = ops.gelu_f32(h1)
        tm = ops.gemm_nt_f32(g1, m3.weight.detach().contiguous(), m3.bias.detach())           # [levels, C]
        eng = StackEngine(ot.temporalModelling.resblocks, self._wc, o.grad_target, heads=ot.tfm_heads, grad_store=o.grad_store)
        self._refresh_stack_weights()
        # the levels' saved activations go straight into level-major buffers: the batched backward reads them as one row-wise
        # concatenation without the ~45 torch.cat launches that used to build it; the levels' outputs likewise (`inter`)
        bufs = eng.level_buffers(lv, b, L, C, dev) if save else None
        inter = torch.empty((lv, b, C), device=dev, dtype=F32)
        denoised = None
        for i in range(lv):
            t_index = ot.total_levels - 1 - i
            src = x0 if i == 0 else denoised
            noisy = float(ot.sqrt_alphas_cumprod[t_index]) * src + float(ot.sqrt_one_minus_alphas_cumprod[t_index]) * noises[i]
            cur = torch.where(is_mask, noisy[:, None, :], feats)
            xin = bufs[0]["x0"][i] if save else torch.empty((b * L, C), device=dev, dtype=F32)
            torch.add(cur + type_emb + temb, tm[i][None, None, :], out=xin.view(b, L, C))
            out, _ = eng.forward(xin, b, L, causal=False, kpm=kpm, save=save, into=(bufs, i) if save else None)
            denoised = inter[i]
            torch.index_select(out, 0, rows, out=denoised)
        levels = (bufs, lv, b, L, kpm)
        inter = inter.view(lv * b, C)                                                        # [levels * b, C]
        x0_rep = x0.unsqueeze(0).expand(lv, -1, -1).reshape(-1, C)
        inter_n, inter_inv = ops.l2norm_fwd(inter)
        inter_pred = ops.gemm_nt_f32(inter_n, le, alpha=it)
        n_keep = b * o.order_recog_batch
        ri = rand_inds[:n_keep].contiguous()
        pred = torch.cat((logits.index_select(0, ri), inter_pred), dim=0)
        if save:
            self.saved = dict(feat=feat, v=v, vinv=vinv, rows=rows, pad_mask=pad_mask, is_mask=is_mask, ri=ri, n_keep=n_keep,
                              inter_n=inter_n, inter_inv=inter_inv, E=E, h1=h1, g1=g1, levels=levels, eng=eng, b=b, n=n)
        return pred, x0_rep, inter, rows, ri

    def assemble_teacher(self, teacher_x, rows, ri):
        """the teacher half of the output assembly (vit.py:336-350): rows [0, n_keep) are teacher[perm], then the masked clips'
        teacher rows (get_mask_samples, vit.py:360-363) once per denoise level.  Four launches outside the head's graphs: the first
        point of the step that reads the text tower's result."""
        lv = self.o.order_tfm.tfm_layers
        masked_teacher = teacher_x.index_select(0, rows)
        inter_teacher = masked_teacher.unsqueeze(0).expand(lv, -1, -1).reshape(-1, teacher_x.shape[1])
        return torch.cat((teacher_x.index_select(0, ri), inter_teacher), dim=0)

    # ------------------------------------------------------------------ backward
    @torch.no_grad()
    def _backward(self, d_pred, d_x0rep, d_inter_mse):
        """gradients of (pred, mse target, mse prediction) -> d feat; parameter gradients go to the flat gradient buffer"""
        o, ot, sv = self.o, self.o.order_tfm, self.saved
        assert sv is not None, "PretrainHeadEngine.backward() without a saved forward()"
        L, C, lv = ot.max_len, ot.hidden_size, ot.tfm_layers
        b, n = sv["b"], sv["n"]
        dev = d_pred.device
        le, le_t = o._labels(dev)
        it = 1.0 / o.temp
        gs = o.grad_store()
        K = d_pred.shape[1]
        n_keep = sv["n_keep"]
        # output assembly (vit.py:345-350): pred rows [0, n_keep) are logits[perm], the rest the intermediate predictions
        dlogits = torch.zeros((n, K), device=dev, dtype=F32)
        dlogits.index_copy_(0, sv["ri"], d_pred[:n_keep])
        d_inter_n = ops.gemm_nt_f32(d_pred[n_keep:].contiguous(), le_t, alpha=it)
        d_inter = ops.l2norm_bwd(d_inter_n, sv["inter_n"], sv["inter_inv"])
        if d_inter_mse is not None:
            d_inter = d_inter + d_inter_mse
        # the denoise levels: each level's input is detached from the previous level's output (tfm_model.py:176-178), so the
        # levels back-propagate independently -- as ONE stack backward over levels x b sequences (the saved activations of the
        # levels concatenated row-wise: 36-row tensors), a quarter of the launches of four passes; the shared stack's parameter
        # gradients are then sums over all levels' rows inside one weight-gradient launch each
        eng = sv["eng"]
        d_out = torch.zeros((lv, b * L, C), device=dev, dtype=F32)
        d_out.index_copy_(1, sv["rows"], d_inter.view(lv, b, C))
        d_out = d_out.view(lv * b * L, C)
        bufs, lvn, nb, Ln, kpm = sv["levels"]
        allsv = eng.merged(bufs, lvn, nb, Ln, False, kpm)
        if SCALED_GRADS:                         # fp16-operand flavour: the stack's backward runs in S-scaled units
            d_cur = eng.backward(gs.begin_scaled(d_out), allsv)
            d_cur = d_cur * gs.end_scaled()
        else:
            d_cur = eng.backward(d_out, allsv)
        sv["levels"] = None
        d_cur = d_cur.view(lv, b, L, C)
        d_tm = d_cur.sum((1, 2))
        D = d_cur.sum(0)
        is_mask, pad_mask = sv["is_mask"], sv["pad_mask"]
        zero = torch.zeros((), device=dev, dtype=F32)
        d_type = torch.stack((torch.where(is_mask, zero, D).sum((0, 1)), torch.where(is_mask, D, zero).sum((0, 1))))
        d_feats = torch.where(is_mask, zero, D)                     # cur = where(is_mask, noisy(detached), feats)
        d_pad = torch.where(pad_mask, d_feats, zero).sum((0, 1)).view(1, C)
        d_v = torch.where(pad_mask, zero, d_feats).reshape(n, C).contiguous()
        if d_x0rep is not None:                                     # x0 = v[rows] is the MSE target of every level, with gradient
            d_v.index_add_(0, sv["rows"], d_x0rep.view(lv, b, C).sum(0))
        d_v += ops.gemm_nt_f32(dlogits, le_t, alpha=it)
        d_x = ops.l2norm_bwd(d_v, sv["v"], sv["vinv"])
        feat = sv["feat"]
        d_feat = ops.gemm_nt_f32(d_x, o.head.weight.detach().t().contiguous())
        grads = [(o.head.weight, ops.gemm_nt_f32(d_x.t().contiguous(), feat.t().contiguous())), (o.head.bias, d_x.sum(0)),
                 (ot.pad_embedding.weight, d_pad), (ot.type_embedding.weight, d_type), (ot.temporalEmbedding.weight, D.sum(0))]
        # time_mlp backward (Linear -> exact GELU -> Linear on the [levels, C/4] sinusoidal rows)
        m1, m3 = ot.time_mlp[1], ot.time_mlp[3]
        dg1 = ops.gemm_nt_f32(d_tm, m3.weight.detach().t().contiguous())
        dh1 = ops.gelu_f32(sv["h1"], dg1.contiguous())
        grads += [(m3.weight, ops.gemm_nt_f32(d_tm.t().contiguous(), sv["g1"].t().contiguous())), (m3.bias, d_tm.sum(0)),
                  (m1.weight, ops.gemm_nt_f32(dh1.t().contiguous(), sv["E"].t().contiguous())), (m1.bias, dh1.sum(0))]
        for p, g in grads:
            if not p.requires_grad:
                continue
            t, beta = gs.target(p)
            if beta == 0.0:
                t.copy_(g.view_as(t))
            else:
                t.add_(g.view_as(t))
        self.saved = None
        return d_feat

    def _refresh_stack_weights(self):
        """16-bit operand copies of the stack's 16 weight matrices in ONE launch (engine.refresh_params) instead of one per matrix on
        first use; inside the forward capture every copy is re-cast (a replay must refresh them after an optimiser step)"""
        ot = self.o.order_tfm
        plist = []
        for blk in ot.temporalModelling.resblocks:
            plist += [(blk.attn.in_proj_weight, True), (blk.attn.out_proj.weight, True), (blk.mlp.c_fc.weight, True),
                      (blk.mlp.c_proj.weight, True)]
        self.o.engine.refresh_params(plist, force=self._cap == "fwd")
        if self._cap == "fwd":
            self._cap_seen.update(id(p) for p, _ in plist)

    # ------------------------------------------------------------------ HIP graphs
    def _key(self, feat):
        o = self.o
        return (tuple(feat.shape), feat.device.index, o.head.weight.data_ptr(),
                o.order_tfm.time_mlp[1].weight.data_ptr(), o.grad_store().flat.data_ptr(), o._labels(feat.device)[0].data_ptr())

    def forward(self, feat, dr, save=True):
        """-> (pred, mse target, mse prediction, rows of the masked clips, kept permutation) -- the last two for `assemble_teacher`"""
        if not (self.use_graphs and feat.is_cuda and save):
            self._gkey = None
            return self._forward(feat, dr, save)
        key = self._key(feat)
        g = self._graphs.get(key)
        if g is None:
            k = self._gseen.get(key, 0)
            self._gseen[key] = k + 1
            if k < self.GRAPH_WARMUP or len(self._graphs) >= self.GRAPH_MAX_KEYS:
                self._gkey = None
                return self._forward(feat, dr, save)
            try:
                g = self._capture_forward(key, feat, dr)
            except Exception as e:          # never fatal: the eager launch sequence is the same kernels
                self._failed("forward", e)
                self._gkey = None
                return self._forward(feat, dr, save)
        g["feat"].copy_(feat)
        for k in ("mask_inds", "pad_start", "noises", "rand_inds"):
            g["dr"][k].copy_(dr[k])
        g["fwd"].replay()
        self.saved = dict(g["saved"])
        self._gkey = key
        return tuple(t.clone() for t in g["out"])       # the graph's own output buffers are overwritten by the next replay

    def _failed(self, what, e):
        import warnings
        warnings.warn(f"HIP graph capture of the pre-training head {what} failed ({type(e).__name__}: {e}); continuing with "
                      "eager launches")
        self.use_graphs = False
        self._cap = None
        self._graphs = {}
        torch.cuda.synchronize()

    def _capture_forward(self, key, feat, dr):
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        sf = feat.clone()
        sd = {k: v.clone() for k, v in dr.items()}
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self._cap, self._cap_seen = "fwd", set()
        try:
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                out = self._forward(sf, sd, True)
        finally:
            self._cap = None
        g = dict(fwd=graph, feat=sf, dr=sd, out=out, saved=self.saved, bwd=None)
        self._graphs[key] = g
        return g

    def backward(self, d_pred, d_x0rep, d_inter):
        g = self._graphs.get(self._gkey) if self._gkey is not None else None
        self._gkey = None
        if g is None:
            return self._backward(d_pred, d_x0rep, d_inter)
        params = self.params()
        if any(p.grad is not None for p in params):     # accumulation into existing gradients (beta = 1 launches): eager
            return self._backward(d_pred, d_x0rep, d_inter)
        lvC = g["out"][1].shape
        z = lambda t: torch.zeros(lvC, device=d_pred.device, dtype=F32) if t is None else t
        d_x0rep, d_inter = z(d_x0rep), z(d_inter)
        if g["bwd"] is None:
            try:
                sp, sx, si = d_pred.contiguous().clone(), d_x0rep.contiguous().clone(), d_inter.contiguous().clone()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                saved = self.saved
                self._cap = "bwd"
                try:
                    with torch.cuda.graph(graph, pool=self._pool, capture_error_mode="thread_local"):
                        d_feat = self._backward(sp, sx, si)
                finally:
                    self._cap = None
                touched = [(p, p.grad) for p in params if p.grad is not None]
                for p, _ in touched:        # the capture ran no kernel: undo its host-side effect
                    p.grad = None
                g["bwd"] = dict(graph=graph, d_pred=sp, d_x0rep=sx, d_inter=si, d_feat=d_feat, touched=touched)
                self.saved = saved
            except Exception as e:
                self._failed("backward", e)
                for p in params:
                    p.grad = None
                self.saved = dict(g["saved"])
                return self._backward(d_pred, d_x0rep, d_inter)
        gb = g["bwd"]
        gb["d_pred"].copy_(d_pred)
        gb["d_x0rep"].copy_(d_x0rep)
        gb["d_inter"].copy_(d_inter)
        gb["graph"].replay()
        for p, v in gb["touched"]:
            p.grad = v
        self.saved = None
        return gb["d_feat"].clone()


class PretrainHeadFn(torch.autograd.Function):
    """(feat, draws) -> (pred, mse target, mse prediction, masked rows, kept permutation): vit.py:298-352 as ONE autograd node (the
    teacher half of the output assembly follows outside, PretrainHeadEngine.assemble_teacher)"""

    @staticmethod
    def forward(ctx, anchor, feat, owner, dr):
        need = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        pred, x0_rep, inter, rows, ri = owner.head_engine.forward(feat.contiguous(), dr, save=need)
        ctx.owner = owner
        ctx.mark_non_differentiable(rows, ri)
        return pred, x0_rep, inter, rows, ri

    @staticmethod
    def backward(ctx, d_pred, d_x0rep, d_inter, _d_rows, _d_ri):
        eng = ctx.owner.head_engine
        if d_pred is None:
            sv = eng.saved
            d_pred = torch.zeros((sv["n_keep"] + sv["inter_n"].shape[0], ctx.owner._labels(sv["v"].device)[0].shape[0]),
                                 device=sv["v"].device, dtype=F32)
        d_feat = eng.backward(d_pred.contiguous(), None if d_x0rep is None else d_x0rep.contiguous(),
                              None if d_inter is None else d_inter.contiguous())
        return None, d_feat, None, None
