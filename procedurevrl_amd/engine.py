"""Kernel sequencing for the TimeSformer divided space-time encoder (forward + backward).

This is the host side of the hot path `VisionTransformer.forward_features`
(reference lib/models/vit.py:365-423) and its autograd backward, restated as an explicit
schedule of C-ABI kernel launches (include/pvrl.h) on torch's current HIP stream.

Token layout (the residual stream x has M = B*N*T + B rows of 768 channels):
    rows [0, R)   patch tokens ordered (b, n, t), t innermost      R = B*N*T     16-bit operand type since round 6 (`_X`, resid16)
    rows [R, M)   the cls token of clip b                                        fp32
The reference keeps [B, 1 + N*T, C] and re-gathers it with einops for every branch
(vit.py:130-151); here temporal sequences are 8 consecutive rows, spatial sequences are
addressed in place by the attention kernel, and the cls rows are a small suffix, so the
three residual branches of a block are 3 LayerNorms + 7 GEMMs + 2 attention launches and no
copy kernels.  Activations are GEMM operands in the library's 16-bit type (fp16 by default); the cls rows of the residual stream,
LayerNorm statistics, softmax statistics and all parameter gradients are fp32.
"""
import math
import os

import torch

from . import ops
from ._lib import lib
from .transform import DecodedClips

OP16 = ops.OP16
F32 = torch.float32
SCALED_GRADS = OP16 == torch.float16     # fp16-operand flavour: engines scale their internal gradients (GradStore.begin_scaled)


class GradStore:
    """Flat fp32 gradient buffer; every trainable parameter's .grad is a view into it so the
    data-parallel all-reduce works on large contiguous chunks (reference: DDP buckets,
    lib/models/build.py:49-53)."""

    def __init__(self, named_params, device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        sizes = [p.numel() for p in self.params]
        # 64-element alignment keeps every view 256-byte aligned
        self.offsets = []
        off = 0
        for s in sizes:
            self.offsets.append(off)
            off += (s + 63) // 64 * 64
        self.end = off                     # end of the parameter gradients
        # tail: one float per parameter, "this rank produced a gradient for it" -- summed over ranks inside the last
        # chunk of the data-parallel all-reduce (distributed.GradReducer, DDP's find_unused_parameters bookkeeping) -- and one
        # control slot behind them ("a rank asks for a re-synchronisation of the replicas", GradReducer find_unused="cached")
        # ... and behind it `bad`, the optimiser's "skip this step" flag (csrc/optim.hip): raised on the device by the kernels that
        # write parameter gradients when a value is inf / nan and by the training loop when a loss is; summed over ranks with the
        # rest of the tail, so every replica drops the same step (misc.check_nan_losses, tools/train_net.py:174)
        self.flat = torch.zeros(off + (len(sizes) + 2 + 63) // 64 * 64, device=device, dtype=F32)
        self.used = self.flat[off:off + len(sizes)]
        self.ctl = self.flat[off + len(sizes):off + len(sizes) + 1]
        self.bad = self.flat[off + len(sizes) + 1:off + len(sizes) + 2]
        self.fused_checked = set()     # parameters whose gradient producers raise `bad` themselves (target(..., fused=True))
        self._unchecked = set()        # ... parameters with at least one writer that does not (target(..., checks=False))
        self.reduced_over_ranks = False   # set by distributed.GradReducer.finish(): the buffer holds SUMS over ranks (the optimiser then scans all of it)
        self.views = [self.flat[o:o + s].view(p.shape) for o, s, p in zip(self.offsets, sizes, self.params)]
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.scale = None       # fp16 flavour: device scalar S while an engine's backward runs with S-scaled gradients
        self.inv = None         # ... and 1 / S: the `gscale` of the kernels that write parameter gradients (include/pvrl.h)
        self._touched = []

    def span(self, i):
        """[a, b) of parameter i in the flat buffer, padding included"""
        return self.offsets[i], (self.offsets[i + 1] if i + 1 < len(self.offsets) else self.end)

    def target(self, p, fused=False, checks=True):
        """-> (grad tensor to write into, beta).  beta = 0 overwrites, 1 accumulates.
        `fused`: the caller's kernel computes  grad = beta * grad + self.inv * (its S-scaled result)  itself (`gscale`): nothing is
        registered for unscale(), what is already there stays in true units.  `checks` (with `fused`): that kernel also raises `self.bad`
        on a non-finite value it writes (`nonfinite`), so the optimiser's scan may skip the parameter (`fused_checked`); a writer that
        takes no `nonfinite` argument says checks=False and its parameter stays in the scan."""
        i = self.index[id(p)]
        v = self.views[i]
        if p.grad is None:
            p.grad = v
            t, beta = v, 0.0
        elif p.grad.data_ptr() == v.data_ptr():
            t, beta = v, 1.0
        else:       # a foreign .grad tensor (someone else allocated it): accumulate into it
            t, beta = p.grad, 1.0
        if fused:
            if not checks:                       # one unchecked writer is enough to keep the parameter in the scan, whatever the order
                self._unchecked.add(i)
                self.fused_checked.discard(i)
            elif t is v and i not in self._unchecked:
                self.fused_checked.add(i)
            return t, beta
        if self.scale is not None:
            key = i if t is v else t
            seen = self._seen_idx if t is v else self._seen_ptr
            tag = i if t is v else t.data_ptr()
            if tag not in seen:
                seen.add(tag)
                if beta == 1.0:
                    t.mul_(self.scale)      # what is already there joins the S-scaled units until the engine is done
            self._touched.append(key)
        return t, beta

    def prezero(self, params):
        """An engine that is about to ACCUMULATE into every one of `params` (atomics / beta = 1 kernels): the ones without a
        gradient yet get their zeroed view now, contiguous runs of the flat buffer in one fill each, instead of one small
        fill per parameter at its first use.  Only for parameters the caller is certain to write: .grad stops being None."""
        idx = sorted(self.index[id(p)] for p in params if p.grad is None and id(p) in self.index)
        runs = []
        for i in idx:
            a, b = self.span(i)
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        for a, b in runs:
            self.flat[a:b].zero_()
        for i in idx:
            self.params[i].grad = self.views[i]
            if self.scale is not None:           # zeros need no conversion to S-scaled units, but are unscaled with the rest
                self._seen_idx.add(i)
                self._touched.append(i)

    # ---- fp16-operand flavour: gradient scaling inside an engine's backward -------------------------------------------
    # fp16 has 5 exponent bits: the 16-bit gradient operands of the backward GEMMs (rms 1e-6 .. 1e-4 at the benchmark
    # shapes) would underflow.  An engine therefore multiplies the gradient it receives by a power of two S, chosen on the
    # DEVICE from that gradient's magnitude (no host sync, capturable in a HIP graph), runs its whole backward in S-scaled
    # units -- exact in fp32, and the 16-bit operands sit mid-range -- and multiplies every parameter gradient it produced
    # by 1/S before anyone outside the engine sees it.  Nothing outside the engine ever holds a scaled value.
    SCALE_TARGET = 256.0          # S * max|incoming gradient|; the largest internal operand stays ~100x below fp16's 65504

    def begin_scaled(self, g):
        """g: the fp32 gradient entering the engine -> g * S; registers S for target() / unscale()"""
        # a non-finite incoming gradient (amax = inf / nan) must not turn S into 0 and 1/S into inf: S stays a finite power
        # of two, so the non-finite values flow through to the loss check (train_epoch) instead of poisoning gradients
        # accumulated by earlier micro-iterations
        self._touched, self._seen_idx, self._seen_ptr = [], set(), set()
        g = g.detach()
        if g.is_cuda and g.dtype == F32 and g.is_contiguous() and g.numel() <= (1 << 20):      # one launch (csrc/optim.hip)
            out = torch.empty_like(g)
            self.scale = torch.empty(1, device=g.device, dtype=F32)
            self.inv_row = torch.empty(4096, device=g.device, dtype=F32)       # 1 / S, also as a GEMM epilogue's per-row scale
            self.inv = self.inv_row[:1]
            lib().call("pvrl_grad_scale_begin", ops._ptr(g), g.numel(), float(self.SCALE_TARGET), ops._ptr(out), ops._ptr(self.scale),
                       ops._ptr(self.inv_row), 4096, ops._stream())
            return out
        amax = torch.nan_to_num(g.abs().max(), nan=1.0, posinf=3e38).clamp(1e-30, 3e38)
        self.scale = torch.exp2(torch.floor(torch.log2(self.SCALE_TARGET / amax)).clamp(-100.0, 100.0)).reshape(1)
        self.inv = 1.0 / self.scale
        self.inv_row = self.inv.expand(4096).contiguous()      # 1 / S as a GEMM epilogue's per-row scale (EncoderEngine._temporal_chain)
        return g * self.scale

    def unscale(self):
        """multiply every gradient written since the last call by 1/S (contiguous runs of the flat buffer in one op each)"""
        if self.scale is None or not self._touched:
            return
        inv = self.inv
        idx = sorted(set(k for k in self._touched if isinstance(k, int)))
        runs = []
        for i in idx:
            a, b = self.span(i)
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        for a, b in runs:
            self.flat[a:b].mul_(inv)
        done = set()
        for k in self._touched:
            if not isinstance(k, int) and k.data_ptr() not in done:
                done.add(k.data_ptr())
                k.mul_(inv)
        self._touched = []

    def end_scaled(self):
        """-> 1/S (device scalar) for gradients the engine hands back to autograd (StackEngine's dx)"""
        self.unscale()
        inv = self.inv
        self.scale = self.inv = None
        return inv


class _W:
    """bf16 operand copies of one weight matrix: `w` = [N, K] for forward, `t` = [K, N] for the data gradient."""
    __slots__ = ("w", "t", "ver", "be")

    def __init__(self):
        self.w = None
        self.t = None
        self.ver = -1
        self.be = None      # fused temporal map only: fp32 [N] bias W_fc b_proj


class _X:
    """One stage of the encoder's residual stream (or of its gradient): patch rows `p` [R, C], cls rows `c` [B, C] fp32.
    Split stream (EncoderEngine.resid16, round 6): `p` is its own tensor in the 16-bit operand type -- GEMM residual epilogues
    (PVRL_EPI_RESID_16) read and write it, the LayerNorm kernels take the two parts as one matrix (ops.SplitRows / pvrl_rows).
    Otherwise `p` and `c` are views of ONE fp32 buffer `full` [R + B, C] (rounds 1-5)."""
    __slots__ = ("p", "c", "full")

    def __init__(self, p, c, full=None):
        self.p, self.c, self.full = p, c, full

    @staticmethod
    def new(R, B, C, device, split, zero=False, zero_p=True):
        mk = torch.zeros if zero else torch.empty
        if split:
            return _X((mk if zero_p else torch.empty)((R, C), device=device, dtype=OP16), mk((B, C), device=device, dtype=F32))
        full = mk((R + B, C), device=device, dtype=F32)
        return _X(full[:R], full[R:], full)

    def all(self, p_zero=False):
        """every row, as the LayerNorm entry points take it (`p_zero`: the patch rows are known to be zero and are not read)"""
        if self.full is not None:
            return self.full
        return ops.SplitRows(None if p_zero else self.p, self.c, n_lo=self.p.shape[0])

    def patch(self):
        return self.p if self.full is not None else ops.SplitRows(self.p, None)


def drop_graphs_quietly(graphs):
    """clear a dict / list that holds captured HIP graphs, with the device idle (see GraphReplay.release_graphs)"""
    if not graphs:
        return
    try:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:
        pass
    graphs.clear()
    try:        # ... and their memory pool goes back NOW, still with the device idle, not whenever the allocator next trims its cache
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass


class GraphReplay:
    """HIP-graph replay of an encoder engine's training / inference step.

    One training step of the encoder is ~1,400 kernel launches (~7 ms of Python on an idle host, several times that
    when the node's cores are contended -- measured up to 95 ms, more than the 57 ms the GPU needs).  The launch
    sequence of a given (shape, mode) never changes, so it is captured once (torch.cuda.CUDAGraph = hipGraph; the
    C ABI launches on the capturing stream like on any other) and replayed: forward and backward are one graph each
    (backward: one per block when a data-parallel gradient hook is installed and the engine has staged backward),
    sharing a memory pool so the activations saved by the forward graph are the backward graph's inputs.  What a
    replay cannot express falls back to the eager path: pinned DropPath draws, DecodedClips inputs, gradient
    accumulation into existing .grad tensors.

    The engine provides: _eager_forward(frames, training, save), _eager_backward(dfeat), _graph_key(frames, training,
    save), _enc_params() (the parameters whose gradients backward writes), `saved`, `grad_hook`, and consults
    `self._capturing` ("fwd": refresh every weight copy inside the graph, "bwd": reuse them, and keep side-stream
    operands alive until the join).  Optional: _bwd_begin / _bwd_block / _bwd_end + join_side_stream for staged capture.
    """
    GRAPH_WARMUP = 2      # eager calls of a key before it is captured (lazy workspaces / caches settle)
    GRAPH_MAX_KEYS = 4    # captured (shape, mode) combinations kept; others run eagerly
    _capturing = None     # "fwd" / "bwd" while a HIP graph of that pass is being captured
    _staged = False       # ... the backward as one graph per block (a gradient hook runs between the replays)
    _refreshed = False    # EncoderEngine: every weight copy was just rebuilt by _refresh_weights()

    def _graph_init(self):
        self.use_graphs = os.environ.get("PVRL_HIP_GRAPHS", "1") == "1"
        self._capturing = None
        self._graphs = {}
        self._gpool = None
        self._gkey = None
        self._gseen = {}

    def _graph_reset_host_state(self):
        pass

    def release_graphs(self):
        """Drop every captured HIP graph of this engine (and give their memory pool back) WITH THE DEVICE IDLE -- call it, then
        `gc.collect()`, before a long-lived process lets go of a model it has trained.  Round 6: with the MViT tests in front, the
        train-loop tests aborted or hung inside the HIP runtime (ROCm 7.2, no message) in 3 of 4 runs: the dead models' ~20 captured
        graphs and their pools were being torn down by Python's cyclic garbage collector at arbitrary points of the NEXT model's steps.
        Releasing every test's GPU objects at a quiet point between tests (tests/conftest.py) removed it (5 of 5); doing the same from
        a `__del__` did not (3 of 6 still failed: the collector still picks the moment), so there is no destructor hook."""
        graphs = getattr(self, "_graphs", None)
        if not graphs:
            return
        drop_graphs_quietly(graphs)
        self._gkey = None
        self._gpool = None

    grad_hook_group = None      # optional: the hook for a list of blocks at once (distributed.GradReducer merges their all-reduces)

    def _group_of(self, i, nb):
        """-> (lo, hi): the blocks [lo, hi] whose gradient hook runs together with block i's (`hook_group` blocks per group: fewer,
        larger collectives).  The LAST group of a backward -- blocks 0 .. hook_group - 1 -- runs per block (PVRL_HOOK_TAIL_SPLIT=0:
        A/B): nothing is left to overlap its collective with but the embedding stage, so the exposed tail round stays one block's
        ~45 MB instead of three blocks' ~135 MB."""
        grp = max(1, int(getattr(self, "hook_group", 1)))
        if i < grp and getattr(self, "hook_tail_split", True):
            return i, i
        lo = (i // grp) * grp
        return lo, min(lo + grp, nb) - 1

    def _run_hooks(self, blocks):
        if not blocks or self.grad_hook is None:
            return
        if self.grad_hook_group is not None and len(blocks) > 1:
            self.grad_hook_group(list(blocks))
        else:
            for j in blocks:
                self.grad_hook(j)

    def _bwd_group_end(self, state):
        """staged backward: a group of blocks is done -- an engine that defers work of its blocks finishes it here (default: nothing)"""

    @staticmethod
    def _saved_copy(saved):
        """backward() consumes its `saved` dict (frees block entries as it goes): replays hand it a shallow copy"""
        sv = dict(saved)
        sv["blocks"] = list(saved["blocks"])
        return sv

    def _graph_forward(self, frames, training, save):
        key = self._graph_key(frames, training, save)
        g = self._graphs.get(key)
        if g is None:
            n = self._gseen.get(key, 0)
            self._gseen[key] = n + 1
            if n < self.GRAPH_WARMUP or len(self._graphs) >= self.GRAPH_MAX_KEYS:
                self._gkey = None
                return self._eager_forward(frames, training, save)
            try:
                g = self._capture_forward(key, frames, training, save)
            except Exception as e:      # never fatal: the eager launch sequence is the same kernels
                self._graph_failed("forward", e)
                self._gkey = None
                return self._eager_forward(frames, training, save)
        g["frames"].copy_(frames)
        g["fwd"].replay()
        self.saved = g["saved"]
        self._gkey = key if save else None
        return g["feat"].clone()        # the graph's own output buffer is overwritten by the next replay

    def _graph_failed(self, what, e):
        import warnings
        warnings.warn(f"HIP graph capture of the encoder {what} failed ({type(e).__name__}: {e}); continuing with eager "
                      "launches")
        self.use_graphs = False
        self._capturing = None
        self._graphs = {}
        self._graph_reset_host_state()
        torch.cuda.synchronize()

    def _capture_forward(self, key, frames, training, save):
        if self._gpool is None:
            self._gpool = torch.cuda.graph_pool_handle()
        st = torch.empty_like(frames)
        st.copy_(frames)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self._capturing = "fwd"
        try:
            with torch.cuda.graph(graph, pool=self._gpool, capture_error_mode="thread_local"):
                feat = self._eager_forward(st, training, save)
        finally:
            self._capturing = None
        g = dict(fwd=graph, frames=st, feat=feat, saved=self.saved if save else None)
        self._graphs[key] = g
        return g

    def _grads_fresh(self):
        return all(p.grad is None for p in self._enc_params())

    def _graph_backward(self, dfeat):
        g = self._graphs[self._gkey]
        staged_ok = hasattr(self, "_bwd_begin")
        if not self._grads_fresh() or (self.grad_hook is not None and not staged_ok):
            # accumulation into existing gradients (beta = 1 launches), or a hook this engine cannot stage: eager launches
            self.saved = self._saved_copy(g["saved"])
            self._gkey = None
            return self._eager_backward(dfeat)
        staged = self.grad_hook is not None             # cut the graph where the data-parallel reducer hooks in
        nb = len(g["saved"]["blocks"])
        slot = "bwd_staged" if staged else "bwd"
        if g.get(slot) is None and not self._capture_backward(g, slot, staged, nb, dfeat):
            self.saved = self._saved_copy(g["saved"])
            self._gkey = None
            return self._eager_backward(dfeat)
        gb = g[slot]
        gb["dfeat"].copy_(dfeat)
        for p, v in gb["touched"]:      # before the hooks run: the reducer treats a parameter without .grad as unused
            p.grad = v
        for graph, blocks in gb["graphs"]:
            graph.replay()
            self._run_hooks(blocks)     # (staged: the blocks this graph finished, last block first)
        self.saved = None
        self._gkey = None

    def _capture_backward(self, g, slot, staged, nb, dfeat):
        """-> True when g[slot] holds the captured graph(s); False after a failed capture (graphs are then switched off)"""
        st_in = torch.empty_like(dfeat.contiguous())
        st_in.copy_(dfeat)
        torch.cuda.synchronize()
        hook, self.grad_hook = self.grad_hook, None
        params = self._enc_params()
        graphs = []
        self._capturing = "bwd"
        self._staged = staged          # the hook is detached while capturing: a stage must still finish its block's gradients itself
        err = None
        try:
            self.saved = self._saved_copy(g["saved"])
            if not staged:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, pool=self._gpool, capture_error_mode="thread_local"):
                    self._eager_backward(st_in)
                graphs.append((graph, []))
            else:
                # one graph per GROUP of `hook_group` blocks (the first also holds the final-norm stage, the last the embedding stage):
                # the hook runs for a group's blocks after its replay.  A capture cannot end with side-stream work in flight, so each
                # stage joins its weight gradients
                state = None
                i = nb - 1
                while i >= 0:
                    lo = self._group_of(i, nb)[0]
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, pool=self._gpool, capture_error_mode="thread_local"):
                        if state is None:
                            state = self._bwd_begin(st_in)
                        for j in range(i, lo - 1, -1):
                            self._bwd_block(state, j)
                        self._bwd_group_end(state)
                        if lo == 0:
                            self._bwd_end(state)
                        else:
                            self.join_side_stream()
                    graphs.append((graph, list(range(i, lo - 1, -1))))
                    i = lo - 1
        except Exception as e:          # never fatal: the eager launch sequence is the same kernels
            err = e
        finally:
            self._capturing = None
            self._staged = False
            self.grad_hook = hook
        touched = [(p, p.grad) for p in params if p.grad is not None]     # all were None (_grads_fresh)
        for p, _ in touched:            # capture ran no kernel: undo its host-side effect
            p.grad = None
        if err is not None:
            self._graph_failed("backward", err)
            return False
        g[slot] = dict(graphs=graphs, dfeat=st_in, touched=touched)
        return True


class EncoderEngine(GraphReplay):
    def __init__(self, model):
        """`model` is a procedurevrl_amd.vit.VisionTransformer (same parameter names as the reference)."""
        self.m = model
        self.C = model.embed_dim
        self.H = model.num_heads
        self.scale = (self.C // self.H) ** -0.5
        self.eps = model.ln_eps
        self._w = {}
        self._grads = None
        self.saved = None
        self.grad_hook = None
        # ONE STREAM since round 5.  Until round 4 the weight-gradient GEMMs ran on a side stream next to the dgrad chain and the
        # fused temporal maps W_e were built on it at the start of the forward (PVRL_WGRAD_OVERLAP=1 / PVRL_PREFETCH_FUSED=1 bring
        # both back).  Measured (profiles/r5_hw_queues.txt, two passes): single stream 636.5 clips/s, W_e prefetch only 632.3, both
        # 622.7 -- and only the single-stream step is independent of how HIP maps streams to hardware queues: with the side
        # stream it is 58 ms instead of 51 at GPU_MAX_HW_QUEUES=6 (the branches of its graph truly co-run), the capture segfaults
        # inside ROCm 7 at 1-2 queues, and under a foreign stream's long kernel (RCCL) it stalls behind whatever shares its queue
        # (tools/probe/comm_cus_ab.py).  Grouped weight-gradient launches fill the chip on their own; there is nothing to overlap.
        # the 768^3 GEMMs of the fused temporal branch (W_e per block in forward; dW_fc, dW_proj per block in backward) batched into
        # one launch per dozen (ops.gemm_nt_batched): 36 tiles apiece cannot fill 256 CUs (PVRL_BATCH_FUSED=0: one launch each, A/B runs)
        self.batch_fused = os.environ.get("PVRL_BATCH_FUSED", "1") == "1"
        # with a gradient hook (data parallel): blocks per group -- the hook runs (and the deferred launches go out, batched) once per
        # group of this many blocks instead of per block: four all-reduce rounds of ~135 MB per backward instead of twelve of 45
        self.hook_group = max(1, int(os.environ.get("PVRL_HOOK_GROUP", "3")))
        self.hook_tail_split = os.environ.get("PVRL_HOOK_TAIL_SPLIT", "1") == "1"      # the backward's last group per block (_group_of)
        self._fused_fresh = set()
        self._chain = []
        self._ln_defer = []
        self.overlap_wgrad = os.environ.get("PVRL_WGRAD_OVERLAP", "0") == "1"
        self.prefetch_fused = os.environ.get("PVRL_PREFETCH_FUSED", "0") == "1"
        self._side = None
        self.group_wgrad = True       # one grouped launch for a block's seven weight gradients
        # the B cls rows' projection + MLP in fp32 on the master weights (csrc/cls_chain.hip; PVRL_CLS_FP32=0: A/B runs)
        self.cls_fp32 = os.environ.get("PVRL_CLS_FP32", "1") == "1"
        # Only the cls row of the LAST block's output is read (final norm + `x[:, 0]`, vit.py:418-421): that block's spatial projection
        # and MLP touch the cls rows alone, forward and backward -- the reference computes (and back-propagates zeros through) all 1,569
        # tokens of every clip there.  Same features, loss and gradients; PVRL_PRUNE_LAST=0: A/B runs.
        self.prune_last = os.environ.get("PVRL_PRUNE_LAST", "1") == "1"
        self.prune_attn = os.environ.get("PVRL_PRUNE_ATTN", "1") == "1"     # ... and its spatial attention the cls query only (attn_cls.hip)
        # SPLIT residual stream (round 6): the patch rows of x and of its gradient dx live in the 16-bit operand type between kernels,
        # the B cls rows stay fp32 -- per block 18 passes over the 50k token rows move 77 instead of 154 MB.  What it costs in error is
        # what the rounding-model oracle says (profiles/r6_resid16_rounding.txt: logits 2.84e-4 -> 2.96e-4, worst gradient 1.09e-3 ->
        # 1.16e-3 at 12 blocks, fp16 operands).  PVRL_RESID16=0: the fp32 stream of rounds 1-5 (A/B runs).
        self.resid16 = os.environ.get("PVRL_RESID16", "1") == "1"
        # Debug aid (VERDICT r5 weak #11 / ADVICE): with the last block pruned, the patch rows of its x2 / x3, the patch queries of its qkv,
        # o_s[:R], the patch rows of the gradient stream ahead of its norm1 backward and the query third of its dqkv are never written --
        # correct only while nothing reads them.  PVRL_DEBUG_NAN_UNDEFINED=1 fills every such region with NaN, so a reader that should
        # not exist turns the loss / gradients NaN (tests/e2e_checks.check_train_step_undefined_rows_nan_filled runs the suite's steps so).
        self.debug_nan_undefined = os.environ.get("PVRL_DEBUG_NAN_UNDEFINED", "0") == "1"
        self._wq = []
        self._wpost = []
        self._side_keep = []
        self._keep = None
        self._fe_events = {}
        self._graph_init()            # HIP-graph replay of the step (GraphReplay)
        self._refreshed = False
        assert self.C == 768 and self.C // self.H == 64, "kernels are built for ViT-B (C=768, head_dim=64)"

    # ------------------------------------------------------------------ weights
    def _weight(self, p, need_t=True):
        e = self._w.get(id(p))
        if e is None:
            e = _W()
            self._w[id(p)] = e
        # the fused optimiser updates trainable parameters through its flat buffer (no _version bump) and advances
        # weights_epoch instead; frozen parameters (text tower) only change through versioned in-place copies
        ver = (p._version, getattr(self.m, "weights_epoch", 0) if p.requires_grad else 0, p.data_ptr())
        if self._capturing == "bwd":      # the forward graph of the same step refreshed the copies
            assert e.w is not None and (e.t is not None or not need_t)
            return e
        if (self._capturing == "fwd" and not self._refreshed) or e.ver != ver or e.w is None or e.w.device != p.device:
            w2 = p.detach().reshape(p.shape[0], -1).contiguous()
            same = e.w is not None and e.w.device == p.device
            e.w, t = ops.cast_weight(w2, out=e.w if same else None, out_t=e.t if same else None, need_t=need_t)
            if need_t:
                e.t = t
            e.ver = ver
        return e

    def _refresh_weights(self):
        """Bring the bf16 operand copies of every encoder weight up to date in ONE launch (pvrl_cast_weights_multi_bf16)
        instead of one 6-us launch per matrix on first use; afterwards _weight() finds every version current."""
        m = self.m
        plist = [(m.patch_embed.proj.weight, False)]
        for blk in m.blocks:
            plist += [(blk.temporal_attn.qkv.weight, True), (blk.temporal_attn.proj.weight, True),
                      (blk.temporal_fc.weight, True), (blk.attn.qkv.weight, True), (blk.attn.proj.weight, True),
                      (blk.mlp.fc1.weight, True), (blk.mlp.fc2.weight, True)]
        self.refresh_params(plist, force=self._capturing == "fwd")

    def refresh_params(self, plist, force=False):
        """operand copies of the (parameter, transposed copy wanted) pairs of `plist` in ONE launch; `force`: re-cast the current ones
        as well (inside a forward capture: a replay must refresh the copies after an optimiser step)"""
        todo = []
        epoch = getattr(self.m, "weights_epoch", 0)
        for p, need_t in plist:
            e = self._w.get(id(p))
            if e is None:
                e = _W()
                self._w[id(p)] = e
            ver = (p._version, epoch if p.requires_grad else 0, p.data_ptr())
            fresh = e.ver == ver and e.w is not None and e.w.device == p.device and (e.t is not None or not need_t)
            if fresh and not force:
                continue
            w2 = p.detach().reshape(p.shape[0], -1)
            if not w2.is_contiguous():
                continue                       # left to _weight()
            if e.w is None or e.w.device != p.device:
                e.w = torch.empty(w2.shape, device=p.device, dtype=OP16)
                e.t = None
            if need_t and e.t is None:
                e.t = torch.empty((w2.shape[1], w2.shape[0]), device=p.device, dtype=OP16)
            todo.append((w2, e.w, e.t if need_t else None, e, ver))
        if todo:
            ops.cast_weights_multi([(w2, w, t) for w2, w, t, _, _ in todo])
            for _, _, _, e, ver in todo:
                e.ver = ver

    def _fused_temporal(self, blk):
        """The temporal branch applies two linear maps back to back (vit.py:131-134: temporal_attn.proj, DropPath, then
        temporal_fc), so  x + fc(rs * proj(o)) = x + rs * (o W_e^T + b_e) + b_fc  with  W_e = W_fc W_proj,  b_e = W_fc b_proj:
        ONE 50k-row GEMM forward (and one data-gradient / one weight-gradient GEMM backward) instead of two each.
        W_e is rebuilt from the bf16 operand copies whenever either weight changed (a 768^3 MFMA GEMM); the parameter
        gradients are recovered from dW_e in backward (`_temporal_chain`)."""
        wf, wp = blk.temporal_fc.weight, blk.temporal_attn.proj.weight
        e = self._w.get(("fused_t", id(wf)))
        if e is None:
            e = _W()
            self._w[("fused_t", id(wf))] = e
        ef, ep = self._weight(wf), self._weight(wp)
        ver = (ef.ver, ep.ver, blk.temporal_attn.proj.bias._version)
        if self._capturing == "bwd" or id(blk) in self._fe_events or id(blk) in self._fused_fresh:     # (built a moment ago by
            return e                                                     #  _prefetch_fused_temporal / _build_fused_all)
        if self._capturing == "fwd" or e.ver != ver or e.w is None:
            L = lib()
            we = ops.gemm_nt(ef.w, ep.t, L.PVRL_EPI_F32)                    # [out, in] = W_fc [out, mid] . W_proj [mid, in]
            e.w, e.t = ops.cast_weight(we, out=e.w, out_t=e.t)
            e.be = ops.gemv_rows(wf.detach(), blk.temporal_attn.proj.bias.detach(), out=e.be)                 # [out]
            e.ver = ver
        return e

    def _prefetch_fused_temporal(self, device):
        """W_e / b_e of every block depend on the weights only, yet building them where they are used puts a 768^3 GEMM, a
        cast and a GEMV (~50 us on 36 CUs) in front of every block's temporal GEMM: 0.6 ms per forward on the critical
        path.  They are built for all blocks on the side stream at the start of the forward, under the patch embedding and
        the first block's LayerNorm / QKV / attention; each block waits for its own event."""
        self._fe_events = {}
        side = self.side_stream(device, force=True) if self.prefetch_fused and device.type == "cuda" else None
        if side is None:
            return
        stale = [blk for blk in self.m.blocks if self._fused_temporal_stale(blk)]
        if not stale:
            return
        ev0 = torch.cuda.current_stream().record_event()      # behind the weight-copy refresh
        with torch.cuda.stream(side):
            side.wait_event(ev0)
            for blk in stale:
                self._fused_temporal(blk)
                self._fe_events[id(blk)] = side.record_event()      # (registered AFTER the build: see _fused_temporal)

    def _fused_temporal_stale(self, blk):
        if self._capturing == "fwd":
            return True
        e = self._w.get(("fused_t", id(blk.temporal_fc.weight)))
        if e is None or e.w is None:
            return True
        ef = self._w.get(id(blk.temporal_fc.weight))
        ep = self._w.get(id(blk.temporal_attn.proj.weight))
        return ef is None or ep is None or e.ver != (ef.ver, ep.ver, blk.temporal_attn.proj.bias._version)

    def _temporal_chain(self, blk, gs, dwe, dbe):
        """dW_e [out, in], db_e [out] (fp32, from the weight-gradient GEMM of the fused map) -> gradients of the four
        parameters:  dW_fc = dW_e W_proj^T + db_e b_proj^T,  dW_proj = W_fc^T dW_e,  db_proj = W_fc^T db_e  (db_fc comes from the LayerNorm
        backward's column sums).  Runs on the stream of the weight-gradient launch, right behind it."""
        L = lib()
        wf, wp = blk.temporal_fc.weight, blk.temporal_attn.proj.weight
        ef, ep = self._weight(wf), self._weight(wp)
        dwe_b, dwe_t = ops.cast_weight(dwe)                                   # bf16 [out, in] and [in, out]
        # dW_e / db_e are in the backward's S-scaled units (fp16 flavour; their 16-bit copies above must stay mid-range); the scale
        # is taken out where the four parameter gradients are written: a row scale of 1 / S in the GEMM epilogues, `gscale` below
        rs = gs.inv_row if gs.inv is not None else None
        # (gemm_nt_batched / rank1_add / gemv_rows take no `nonfinite` argument: checks=False keeps these parameters in the optimiser's scan)
        tg = [(gs.target(lin_w, fused=True, checks=False), A, W) for lin_w, A, W in ((wf, dwe_b, ep.w), (wp, ef.t, dwe_t))]   # [out,mid] = dW_e.W_p^T ; [mid,in] = W_f^T.dW_e
        if self.batch_fused and tg[0][0][1] == tg[1][0][1]:                   # both in one launch (72 tiles instead of 2 x 36)
            beta = tg[0][0][1]
            ops.gemm_nt_batched([dict(A=A, W=W, rowscale=rs, out0=g, aux=g if beta else None) for (g, _), A, W in tg],
                                L.PVRL_EPI_RESID_F32 if beta else L.PVRL_EPI_F32)
        else:
            for (g, beta), A, W in tg:
                if beta == 0.0:
                    ops.gemm_nt(A, W, L.PVRL_EPI_F32, rowscale=rs, out0=g)
                else:
                    ops.gemm_nt(A, W, L.PVRL_EPI_RESID_F32, rowscale=rs, aux=g, out0=g)
        # proj's bias rides through temporal_fc too (b_e = W_fc b_proj): its share of dW_fc is the outer product db_e x b_proj
        ops.rank1_add(gs.target(wf, fused=True, checks=False)[0], dbe, blk.temporal_attn.proj.bias.detach(), gscale=gs.inv)
        gb, beta = gs.target(blk.temporal_attn.proj.bias, fused=True, checks=False)
        ops.gemv_rows(ef.t, dbe, out=gb, beta=beta, gscale=gs.inv)            # [mid] = W_fc^T db_e (bf16 operand copy)

    def _build_fused_all(self):
        """W_e = W_fc W_proj (and b_e) of every block whose weights changed, at the start of a forward: ONE batched launch of the twelve
        768^3 GEMMs and one of the twelve casts instead of a 16.6-us GEMM + a cast in front of every block's temporal GEMM."""
        self._fused_fresh = set()
        if not self.batch_fused or self.prefetch_fused:
            return
        stale = [blk for blk in self.m.blocks if self._fused_temporal_stale(blk)]
        if len(stale) < 2:
            return
        L = lib()
        ents, probs = [], []
        for blk in stale:
            wf, wp = blk.temporal_fc.weight, blk.temporal_attn.proj.weight
            e = self._w.get(("fused_t", id(wf)))
            if e is None:
                e = _W()
                self._w[("fused_t", id(wf))] = e
            ef, ep = self._weight(wf), self._weight(wp)
            ents.append((blk, e, ef, ep))
            probs.append(dict(A=ef.w, W=ep.t))                              # [out, in] = W_fc [out, mid] . W_proj [mid, in]
        wes = ops.gemm_nt_batched(probs, L.PVRL_EPI_F32)
        items = []
        for we, (blk, e, ef, ep) in zip(wes, ents):
            if e.w is None or e.w.device != we.device:
                e.w = torch.empty(we.shape, device=we.device, dtype=OP16)
                e.t = torch.empty((we.shape[1], we.shape[0]), device=we.device, dtype=OP16)
            items.append((we, e.w, e.t))
        ops.cast_weights_multi(items)
        for blk, e, ef, ep in ents:
            if e.be is None or e.be.device != wes[0].device:
                e.be = torch.empty(self.C, device=wes[0].device, dtype=F32)
        ops.gemv_rows_batched([blk.temporal_fc.weight.detach() for blk, _, _, _ in ents],
                              [blk.temporal_attn.proj.bias.detach() for blk, _, _, _ in ents], [e.be for _, e, _, _ in ents],
                              [0.0] * len(ents))                               # b_e = W_fc b_proj
        for blk, e, ef, ep in ents:
            e.ver = (ef.ver, ep.ver, blk.temporal_attn.proj.bias._version)
            self._fused_fresh.add(id(blk))

    def _temporal_chain_all(self, gs):
        """`_temporal_chain` for every block queued by the backward (no gradient hook installed: nobody needs a block's gradients
        before the end): the 2 x 12 GEMMs dW_fc = dW_e W_proj^T, dW_proj = W_fc^T dW_e as batched launches, the twelve casts as one."""
        chain, self._chain = self._chain, []
        if not chain:
            return
        L = lib()
        rs = gs.inv_row if gs.inv is not None else None
        dev = chain[0][1].device
        C = self.C
        items = []
        for blk, dwe, dbe in chain:
            items.append((dwe, torch.empty((C, C), device=dev, dtype=OP16), torch.empty((C, C), device=dev, dtype=OP16)))
        ops.cast_weights_multi(items)
        groups = {0.0: [], 1.0: []}
        for (blk, dwe, dbe), (_, dwe_b, dwe_t) in zip(chain, items):
            wf, wp = blk.temporal_fc.weight, blk.temporal_attn.proj.weight
            ef, ep = self._weight(wf), self._weight(wp)
            for lin_w, A, W in ((wf, dwe_b, ep.w), (wp, ef.t, dwe_t)):
                g, beta = gs.target(lin_w, fused=True, checks=False)
                groups[beta].append(dict(A=A, W=W, rowscale=rs, out0=g, aux=g if beta else None))
        if groups[0.0]:
            ops.gemm_nt_batched(groups[0.0], L.PVRL_EPI_F32)
        if groups[1.0]:
            ops.gemm_nt_batched(groups[1.0], L.PVRL_EPI_RESID_F32)
        ops.rank1_add_batched([gs.target(blk.temporal_fc.weight, fused=True, checks=False)[0] for blk, _, _ in chain], [dbe for _, _, dbe in chain],
                              [blk.temporal_attn.proj.bias.detach() for blk, _, _ in chain], gscale=gs.inv)
        tb = [gs.target(blk.temporal_attn.proj.bias, fused=True, checks=False) for blk, _, _ in chain]
        ops.gemv_rows_batched([self._weight(blk.temporal_fc.weight).t for blk, _, _ in chain], [dbe for _, _, dbe in chain],
                              [t for t, _ in tb], [b for _, b in tb], gscale=gs.inv)

    def _finish_deferred(self, gs):
        """what the blocks' backward left for the end (no gradient hook): the fused temporal chains and the LayerNorm partial reduces"""
        self._temporal_chain_all(gs)
        items, self._ln_defer = self._ln_defer, []
        ops.layernorm_bwd_reduce_batched(items, gscale=gs.inv, nonfinite=gs.bad)

    def grad_store(self):
        return self.m.grad_store()

    # ------------------------------------------------------------------ side stream for weight gradients
    def side_stream(self, device, force=False):
        """Nothing in backward depends on a weight gradient until the optimiser step, while the data-gradient chain
        (dgrad GEMM -> LayerNorm bwd -> attention bwd ...) is strictly serial.  With `overlap_wgrad` the weight-gradient GEMMs are
        issued on a second HIP stream: they fill CUs left idle by the ragged last wave of the big-tile dgrad GEMMs
        and overlap the HBM-bound LayerNorm / cast kernels.  (`force`: the stream itself, for the forward's W_e prefetch.)"""
        if not self.overlap_wgrad and not force:
            return None
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _wgrad(self, dy, xin, dw, dbias, beta, post=None, gscale=None, nonfinite=None):
        """queue dW = beta*dW + gscale * dy^T xin (and dbias); `flush_wgrads` issues what is queued.  `post` (optional callable)
        runs right behind the launch on the same stream (consumers of dW)."""
        self._wq.append((dy, xin, dw, dbias, beta, gscale, nonfinite))
        if post is not None:
            self._wpost.append(post)
        if not self.group_wgrad:
            self.flush_wgrads()

    def flush_wgrads(self):
        """Issue the queued weight gradients as one grouped launch (ops.gemm_tn_grouped) -- on the side stream when
        overlap_wgrad, ordered after everything the current stream has produced so far."""
        q, self._wq = self._wq, []
        post, self._wpost = self._wpost, []
        if not q and not post:
            return
        side = self.side_stream(q[0][0].device if q else torch.device("cuda", torch.cuda.current_device()))
        if side is None:
            ops.gemm_tn_grouped(q, ws_tag="tn")
            for f in post:
                f()
            return
        ev = torch.cuda.current_stream().record_event()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops.gemm_tn_grouped(q, ws_tag="tn_side")
            for f in post:
                f()
            done = side.record_event()
        # The operands must outlive the side-stream kernels.  Tensor.record_stream would do that, but every pending
        # record makes EACH later allocation poll its event (measured: torch.empty 2 -> 52 us with ~170 records in flight,
        # 20 ms of host time per step); instead the references are parked here until the launch's event has completed,
        # or until join_side_stream() has ordered the main stream behind the side stream.
        self._side_keep.append((done, [t for pr in q for t in pr[:4] + pr[5:6] if t is not None]))
        while not self._capturing and self._side_keep and self._side_keep[0][0].query():
            self._side_keep.pop(0)        # (no event queries while capturing: there everything is held until the join)

    def join_side_stream(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._side_keep.clear()      # later allocations are ordered behind the join on the current stream

    # ------------------------------------------------------------------ drop path
    def _droppath_all(self, B, N, T, device, training):
        """DropPath row scales of every block, lib/models/vit_utils.py:140-155: floor(keep + U[0,1)) / keep per dim-0 row
        of each branch (temporal: per (b h w); spatial: per (b t); mlp: per b).  All blocks are drawn and expanded to
        token rows with ONE set of launches (a per-block version costs ~20 tiny kernels x depth on the critical path)."""
        rates = [float(r) for r in self.m.drop_path_rates]
        nb = len(rates)
        if not training or all(r == 0.0 for r in rates):
            return [None] * nb
        if self._keep is None or self._keep[0] != (tuple(rates), device):   # cached: a host->device copy blocks the host
            self._keep = ((tuple(rates), device), torch.tensor([1.0 - r for r in rates], device=device, dtype=F32).view(nb, 1))
        keep = self._keep[1]
        n1, n2, n3 = B * N, B * T, B
        sc = torch.floor(keep + torch.rand((nb, n1 + n2 + n3), device=device)) / keep
        s1, s2, s3 = sc[:, :n1], sc[:, n1:n1 + n2], sc[:, n1 + n2:]
        s1_tok = s1.repeat_interleave(T, dim=1)                                       # [nb, B*N*T]
        s2_seq = s2.contiguous()
        s2_tok = s2.reshape(nb, B, 1, T).expand(nb, B, N, T).reshape(nb, -1)          # [nb, B*N*T]
        s3_all = torch.cat([s3.repeat_interleave(N * T, dim=1), s3], 1)               # [nb, B*N*T + B]
        s2_mean = s2.reshape(nb, B, T).mean(2)                                        # [nb, B]: the cls rows' bias factor
        return [None if rates[i] == 0.0 else dict(s1_tok=s1_tok[i], s2_seq=s2_seq[i], s2_tok=s2_tok[i], s3_all=s3_all[i],
                                                  s2_mean=s2_mean[i])
                for i in range(nb)]

    @staticmethod
    def expand_droppath(s1, s2, s3, B, N, T):
        s1_tok = s1.repeat_interleave(T).contiguous()
        s2_tok = s2.view(B, 1, T).expand(B, N, T).reshape(-1).contiguous()
        s3_all = torch.cat([s3.repeat_interleave(N * T), s3]).contiguous()
        return dict(s1_tok=s1_tok, s2_seq=s2.contiguous(), s2_tok=s2_tok, s3_all=s3_all, s2_mean=s2.view(B, T).mean(1))

    # ------------------------------------------------------------------ embeddings
    def _pos_time(self, N, T, Wp):
        """nearest-neighbour resize of pos/time embeddings when the input differs (vit.py:375-386,398-402)."""
        m = self.m
        pos = m.pos_embed[0]
        if pos.shape[0] != N + 1:
            P = int(math.isqrt(pos.shape[0] - 1))
            Hn = N // Wp
            other = pos[1:].t().reshape(1, self.C, P, P)
            other = torch.nn.functional.interpolate(other, size=(Hn, Wp), mode="nearest").flatten(2)[0].t()
            pos = torch.cat([pos[:1], other], 0)
        tim = m.time_embed[0]
        if tim.shape[0] != T:
            tim = torch.nn.functional.interpolate(tim.t().unsqueeze(0), size=T, mode="nearest")[0].t()
        return pos.contiguous(), tim.contiguous()

    # ------------------------------------------------------------------ forward
    def forward(self, frames, training, droppath=None, save=True):
        """frames fp32 [B, 3, T, H, W] on the GPU (or transform.DecodedClips) -> (feat fp32 [B, C] = norm(x)[:, 0]).
        `droppath`: optional list (per block) of dicts from expand_droppath, to pin the RNG draws.
        With `use_graphs` the launch sequence of a (shape, mode) is captured once into a HIP graph and replayed."""
        if self.use_graphs and droppath is None and isinstance(frames, torch.Tensor) and frames.is_cuda:
            return self._graph_forward(frames, training, save)
        self._gkey = None
        return self._forward(frames, training, droppath, save)

    def _forward(self, frames, training, droppath=None, save=True):
        L = lib()
        m = self.m
        self._refresh_weights()
        self._refreshed = True
        self._build_fused_all()
        self._prefetch_fused_temporal(frames.device)
        B, _, T, HI, WI = frames.shape
        Wp = WI // 16
        N = (HI // 16) * Wp
        R = B * N * T
        M = R + B
        C = self.C
        dev = frames.device
        sv = dict(B=B, T=T, N=N, R=R, M=M, Wp=Wp, blocks=[])

        if isinstance(frames, DecodedClips):     # decoded uint8 clips: GPU-side normalise/rescale/crop/flip + im2col
            a_pe = ops.frames_u8_patchify(frames)
        else:
            a_pe = ops.patchify(frames.contiguous())
        pos, tim = self._pos_time(N, T, Wp)
        E = ops.embed_table(pos, tim, m.patch_embed.proj.bias.detach(), N, T)
        x = _X.new(R, B, C, dev, self.resid16)
        sv["split"] = self.resid16
        wpe = self._weight(m.patch_embed.proj.weight, need_t=False)
        ops.gemm_nt(a_pe, wpe.w, self._epi_resid(), aux=E, aux_rowmod=N * T, out0=x.p)
        x.c[:] = (m.cls_token.detach()[0, 0] + pos[0]).unsqueeze(0)
        sv["a_pe"] = a_pe if save else None

        if droppath is None:
            droppath = self._droppath_all(B, N, T, dev, training)
        for i, blk in enumerate(m.blocks):
            x = self._block_fwd(blk, x, sv, droppath[i], save, last=i == len(m.blocks) - 1)

        feat, mean, rstd = ops.layernorm_fwd(x.c, m.norm.weight.detach(), m.norm.bias.detach(), self.eps,
                                             out_dtype=F32)
        self._refreshed = False
        if save:
            sv["x_final"] = x
            sv["norm_stats"] = (mean, rstd)
            self.saved = sv
            if SCALED_GRADS:
                # fp16 operands: a 16-bit forward activation past 65504 (qkv, the MLP's u / g, attention outputs) is an inf that reaches
                # every later block's cls row through the attention, i.e. these B x C features: one few-microsecond check raises the
                # optimiser's skip flag on the device, in the step that overflowed, whatever the caller does with its loss
                gs = self.grad_store()
                lib().call("pvrl_nonfinite_flag_f32", ops._ptr(feat), feat.numel(), ops._ptr(gs.bad), ops._stream())
        return feat

    def stream_from_rows(self, x_rows, B):
        """fp32 [B*N*T + B, C] (patch rows, then cls rows) -> a stage of the residual stream as this engine keeps it (tests / probes)"""
        R = x_rows.shape[0] - B
        if not self.resid16:
            full = x_rows.contiguous()
            return _X(full[:R], full[R:], full)
        return _X(x_rows[:R].to(OP16).contiguous(), x_rows[R:].contiguous())

    @staticmethod
    def stream_to_rows(x):
        return x.full if x.full is not None else torch.cat([x.p.float(), x.c], 0)

    def _undef(self, t):
        """`t` is deliberately left unwritten (see debug_nan_undefined)"""
        if self.debug_nan_undefined and t is not None and t.numel():
            t.fill_(float("nan"))
        return t

    def _epi_resid(self):
        L = lib()
        return L.PVRL_EPI_RESID_16 if self.resid16 else L.PVRL_EPI_RESID_F32

    def _block_fwd(self, blk, x0, sv, dp, save, last=False):
        """`last`: the encoder's last block (with prune_last: the patch rows of x2 / x3 are neither computed nor defined)"""
        prune = last and self.prune_last
        L = lib()
        B, T, N, R, M = sv["B"], sv["T"], sv["N"], sv["R"], sv["M"]
        C, H = self.C, self.H
        dev = x0.c.device
        split, epi_res = sv["split"], self._epi_resid()
        s1_tok = dp["s1_tok"] if dp else None
        s2_seq = dp["s2_seq"] if dp else None
        s2_tok = dp["s2_tok"] if dp else None
        s3_all = dp["s3_all"] if dp else None
        P = lambda t: t.detach()

        # ---- temporal branch (vit.py:129-135), rows [0, R) ----
        h_t, mean_t, rstd_t = ops.layernorm_fwd(x0.patch(), P(blk.temporal_norm1.weight), P(blk.temporal_norm1.bias), self.eps)
        qkv_t = ops.gemm_nt(h_t, self._weight(blk.temporal_attn.qkv.weight).w, L.PVRL_EPI_BF16,
                            bias=P(blk.temporal_attn.qkv.bias))
        lse_t = None
        if T == 8:
            o_t = ops.attn_t8_fwd(qkv_t, B * N, H, self.scale)
        else:
            o_t, _, lse_t = ops.attn_fwd(qkv_t, B * N, T, H, self.scale, mode=0)
        if split:       # the cls rows pass the temporal branch unchanged: the same tensor (nothing writes a stage's rows twice)
            x1 = _X(torch.empty_like(x0.p), x0.c)
        else:
            x1 = _X.new(R, B, C, dev, False)
        fe = self._fused_temporal(blk)          # proj then temporal_fc as one linear map
        ev = self._fe_events.pop(id(blk), None)
        if ev is not None:                      # W_e of this block was built on the side stream (_prefetch_fused_temporal)
            torch.cuda.current_stream().wait_event(ev)
        ops.gemm_nt(o_t, fe.w, epi_res, bias=fe.be, rowscale=s1_tok, bias2=P(blk.temporal_fc.bias),
                    aux=x0.p, out0=x1.p)
        if not split:
            x1.c[:] = x0.c

        # ---- spatial branch (vit.py:137-151), all rows; cls of clip b is token 0 of its T sequences ----
        h_s, mean_s, rstd_s = ops.layernorm_fwd(x1.all(), P(blk.norm1.weight), P(blk.norm1.bias), self.eps)
        o_s = torch.empty((R + B * T, C), device=dev, dtype=OP16)
        cls_attn = prune and self.prune_attn
        wqkv = self._weight(blk.attn.qkv.weight).w
        if cls_attn:
            # the last block: only the cls query's output is read (csrc/attn_cls.hip) -- keys and values of every token, queries of the
            # B cls rows; the patch rows' query third of qkv_s and o_s[:R] stay undefined and are never read
            qkv_s = torch.empty((M, 3 * C), device=dev, dtype=OP16)
            self._undef(qkv_s[:R, :C])
            self._undef(o_s[:R])
            bq = P(blk.attn.qkv.bias)
            ops.gemm_nt(h_s, wqkv[C:], L.PVRL_EPI_BF16, bias=bq[C:], out0=qkv_s[:, C:])
            ops.gemm_nt(h_s[R:], wqkv[:C], L.PVRL_EPI_BF16, bias=bq[:C], out0=qkv_s[R:, :C])
            _, lse_s = ops.attn_cls_fwd(qkv_s, B * T, N + 1, H, self.scale, T, R, o_cls=o_s[R:])
        else:
            qkv_s = ops.gemm_nt(h_s, wqkv, L.PVRL_EPI_BF16, bias=P(blk.attn.qkv.bias))
            _, _, lse_s = ops.attn_fwd(qkv_s, B * T, N + 1, H, self.scale, mode=1, T=T, cls_base=R, o=o_s[:R], o_cls=o_s[R:])
        x2 = _X.new(R, B, C, dev, split)      # (pruned last block: its patch rows are neither computed nor defined)
        if prune:
            self._undef(x2.p)
        wproj = self._weight(blk.attn.proj.weight).w
        if not prune:
            ops.gemm_nt(o_s[:R], wproj, epi_res, bias=P(blk.attn.proj.bias), rowscale=s2_tok, aux=x1.p, out0=x2.p)
        if self.cls_fp32:
            # the cls rows' own chain in fp32 on the master weights (csrc/cls_chain.hip): the projection is linear, so the mean over
            # the T frames (vit.py:147-149) is taken first -- B rows instead of B * T
            om = ops.group_reduce(o_s[R:], B, T, scale=s2_seq, alpha=1.0 / T)
            ops.cls_linear(om, P(blk.attn.proj.weight), P(blk.attn.proj.bias), biasscale=dp["s2_mean"] if dp else None,
                           aux=x1.c, out=x2.c)
        else:
            pc = ops.gemm_nt(o_s[R:], wproj, L.PVRL_EPI_F32, bias=P(blk.attn.proj.bias))
            ops.group_reduce(pc, B, T, scale=s2_seq, alpha=1.0 / T, resid=x1.c, out=x2.c)

        # ---- MLP (vit.py:155-157) ----
        x3 = _X.new(R, B, C, dev, split)
        if prune:
            self._undef(x3.p)
        s3c = s3_all[R:] if s3_all is not None else None
        w2 = self._weight(blk.mlp.fc2.weight).w
        if prune:
            # the B cls rows only; h_m / st_m / u / g are then [B, .] tensors (what the backward of this block reads, _block_bwd)
            h_m = mean_m = rstd_m = u = g = None
            if save or not self.cls_fp32:
                h_m, mean_m, rstd_m = ops.layernorm_fwd(x2.c, P(blk.norm2.weight), P(blk.norm2.bias), self.eps)
                u, g = ops.gemm_nt(h_m, self._weight(blk.mlp.fc1.weight).w, L.PVRL_EPI_GELU, bias=P(blk.mlp.fc1.bias))
            if not self.cls_fp32:
                ops.gemm_nt(g, w2, L.PVRL_EPI_RESID_F32, bias=P(blk.mlp.fc2.bias), rowscale=s3c, aux=x2.c, out0=x3.c)
        else:
            h_m, mean_m, rstd_m = ops.layernorm_fwd(x2.all(), P(blk.norm2.weight), P(blk.norm2.bias), self.eps)
            u, g = ops.gemm_nt(h_m, self._weight(blk.mlp.fc1.weight).w, L.PVRL_EPI_GELU, bias=P(blk.mlp.fc1.bias))
            if split:       # the patch rows with the 16-bit residual epilogue; the cls rows are the fp32 chain's below (or their own GEMM)
                ops.gemm_nt(g[:R], w2, epi_res, bias=P(blk.mlp.fc2.bias), rowscale=s3_all[:R] if s3_all is not None else None,
                            aux=x2.p, out0=x3.p)
                if not self.cls_fp32:
                    ops.gemm_nt(g[R:], w2, L.PVRL_EPI_RESID_F32, bias=P(blk.mlp.fc2.bias), rowscale=s3c, aux=x2.c, out0=x3.c)
            else:
                ops.gemm_nt(g, w2, L.PVRL_EPI_RESID_F32, bias=P(blk.mlp.fc2.bias), rowscale=s3_all, aux=x2.full, out0=x3.full)
        if self.cls_fp32:       # (the 16-bit path's cls rows of h_m / u / g stay what the backward reads; x3's are replaced)
            hc, _, _ = ops.layernorm_fwd(x2.c, P(blk.norm2.weight), P(blk.norm2.bias), self.eps, out_dtype=F32,
                                         save_stats=False)
            gc = ops.cls_linear(hc, P(blk.mlp.fc1.weight), P(blk.mlp.fc1.bias), gelu=True)
            ops.cls_linear(gc, P(blk.mlp.fc2.weight), P(blk.mlp.fc2.bias), rowscale=s3c, biasscale=s3c, aux=x2.c,
                           out=x3.c)
        if save:
            sv["blocks"].append(dict(x0=x0, x1=x1, x2=x2, h_t=h_t, st_t=(mean_t, rstd_t), qkv_t=qkv_t, o_t=o_t,
                                     lse_t=lse_t, h_s=h_s, st_s=(mean_s, rstd_s), qkv_s=qkv_s, o_s=o_s,
                                     lse_s=lse_s, h_m=h_m, st_m=(mean_m, rstd_m), u=u, g=g, dp=dp, pruned=prune, cls_attn=cls_attn))
        return x3

    # ------------------------------------------------------------------ HIP graphs (GraphReplay)
    def _eager_forward(self, frames, training, save):
        return self._forward(frames, training, None, save)

    def _eager_backward(self, dfeat):
        return self._backward(dfeat)

    def _graph_key(self, frames, training, save):
        m = self.m
        # parameter / gradient storage is baked into a graph: a re-homed parameter (optimizer flat buffer, .to()) is a new key
        return (tuple(frames.shape), frames.dtype, bool(training), bool(save), frames.device.index, tuple(m.drop_path_rates),
                m.blocks[0].attn.qkv.weight.data_ptr(), m.norm.weight.data_ptr(), self.grad_store().flat.data_ptr())

    def _graph_reset_host_state(self):
        self._wq = []
        self._wpost = []
        self._side_keep = []
        self._fe_events = {}
        self._chain = []
        self._ln_defer = []
        self._fused_fresh = set()

    def _enc_params(self):
        """the parameters whose gradients backward() writes"""
        m = self.m
        ps = [m.cls_token, m.pos_embed, m.time_embed] + list(m.patch_embed.parameters()) + list(m.norm.parameters())
        for blk in m.blocks:
            ps += list(blk.parameters())
        return [p for p in ps if p.requires_grad]

    # ------------------------------------------------------------------ backward
    def backward(self, dfeat):
        """dfeat fp32 [B, C]: gradient of the loss w.r.t. forward()'s output.  Writes every encoder
        parameter gradient into the GradStore views (p.grad) and returns nothing."""
        if self._gkey is not None:
            return self._graph_backward(dfeat)
        return self._backward(dfeat)

    def _backward(self, dfeat):
        st = self._bwd_begin(dfeat)
        nb = len(self.m.blocks)
        for i in range(nb - 1, -1, -1):
            self._bwd_block(st, i)
            lo, hi = self._group_of(i, nb)
            if self.grad_hook is not None and i == lo:
                self._bwd_group_end(st)      # the group's deferred chains / reduces: its blocks' parameter gradients are final now,
                self._run_hooks(range(hi, lo - 1, -1))      # the reducer may start their all-reduce
        self._bwd_end(st)

    def _bwd_group_end(self, st):
        self._finish_deferred(st["gs"])

    # the three stages of backward(); a stage boundary is where the gradient hook may run (and where a staged HIP-graph
    # capture is cut, see _graph_backward)
    def _bwd_begin(self, dfeat):
        m = self.m
        sv = self.saved
        assert sv is not None, "backward() without a saved forward()"
        gs = self.grad_store()
        R, M = sv["R"], sv["M"]
        self._chain = []
        self._ln_defer = []
        if SCALED_GRADS:
            dfeat = gs.begin_scaled(dfeat)
        last = len(m.blocks) - 1
        pruned = bool(sv["blocks"][last].get("pruned"))
        split = sv["split"]
        # the residual gradient stream: zero outside the cls rows until the last block's spatial branch.  Split stream + pruned last
        # block: the patch rows are not even cleared -- the first kernel that would read them (that block's norm1 backward) takes them
        # as zeros (`dxp_zero`)
        dxp_zero = split and pruned
        dx = _X.new(R, M - R, self.C, dfeat.device, split, zero=True, zero_p=not dxp_zero)
        if dxp_zero:
            self._undef(dx.p)
        mean, rstd = sv["norm_stats"]
        (dg, bg), (db, bb) = gs.target(m.norm.weight, fused=True), gs.target(m.norm.bias, fused=True)
        ops.layernorm_bwd(dfeat.contiguous(), sv["x_final"].c, mean, rstd, m.norm.weight.detach(), dg, db,
                          dx_out=dx.c, beta_acc=bg, gscale=gs.inv, nonfinite=gs.bad)
        # dy = bf16(DropPath-scale * dx) is the operand of each block's first backward GEMMs; after the first block it is
        # emitted by the previous block's last LayerNorm-backward kernel instead of a separate cast pass.
        s3 = sv["blocks"][last]["dp"]["s3_all"] if sv["blocks"][last]["dp"] else None
        if pruned:                                # dx is zero outside the cls rows here: the last block's MLP / projection backward
            dy = ops.cast_scale(dx.c, s3[R:] if s3 is not None else None)       # runs on those rows alone (_block_bwd)
        elif split:
            dy = torch.zeros((M, self.C), device=dfeat.device, dtype=OP16)
            ops.cast_scale(dx.c, s3[R:] if s3 is not None else None, out=dy[R:])
        else:
            dy = ops.cast_scale(dx.full, s3)
        return dict(sv=sv, gs=gs, dx=dx, dy=dy, dxp_zero=dxp_zero)

    def _bwd_block(self, st, i):
        sv = st["sv"]
        nxt = sv["blocks"][i - 1]["dp"] if i > 0 else None
        st["dy"] = self._block_bwd(self.m.blocks[i], sv["blocks"][i], sv, st["dx"], st["gs"], st["dy"], i > 0, nxt,
                                   dxp_zero=st.get("dxp_zero", False))
        st["dxp_zero"] = False
        sv["blocks"][i] = None  # free activations as we go

    def _bwd_end(self, st):
        # ---- embedding prologue + patch embed (vit.py:174-180, 370-407) ----
        m = self.m
        sv, gs, dx, dy = st["sv"], st["gs"], st["dx"], st["dy"]
        B, T, N, R, C = sv["B"], sv["T"], sv["N"], sv["R"], self.C
        dz = dy[:R]     # block 0's last LayerNorm backward emitted the unscaled bf16 copy of dx[:R]
        w = m.patch_embed.proj.weight
        (dw, bw), (dbias, _) = gs.target(w, fused=True), gs.target(m.patch_embed.proj.bias, fused=True)
        self._wgrad(dz, sv["a_pe"], dw.view(C, -1), dbias, bw, gscale=gs.inv, nonfinite=gs.bad)
        G = ops.batch_sum(dx.p, B, N * T).view(N, T, C)
        dcls_rows = dx.c.sum(0)
        if gs.inv is not None:       # the three small embedding gradients below are sums of these: the scale is taken out here
            G = G * gs.inv           # (a non-finite value in dx reaches the patch-embed weight gradient above, which raises gs.bad)
            dcls_rows = dcls_rows * gs.inv
        self._acc(gs, m.cls_token, dcls_rows.view(1, 1, C))
        dpos = torch.cat([dcls_rows.unsqueeze(0), G.sum(1)], 0)
        dtime = G.sum(0)
        pos_p, tim_p = m.pos_embed, m.time_embed
        if pos_p.shape[1] != N + 1 or tim_p.shape[1] != T:
            raise NotImplementedError("training with resized pos/time embeddings is not supported (reference "
                                      "resizes at inference only, vit.py:374)")
        self._acc(gs, pos_p, dpos.unsqueeze(0))
        self._acc(gs, tim_p, dtime.unsqueeze(0))
        self.flush_wgrads()
        self._finish_deferred(gs)
        self.join_side_stream()
        if gs.scale is not None:
            gs.end_scaled()
        self.saved = None

    @staticmethod
    def _acc(gs, p, g):
        tgt, beta = gs.target(p, fused=True, checks=False)      # (g is in true units already; see _bwd_end.  torch copy_ / add_: no flag raised)
        if beta == 0.0:
            tgt.copy_(g.view_as(tgt))
        else:
            tgt.add_(g.view_as(tgt))

    def _block_bwd(self, blk, s, sv, dx, gs, dy, has_prev, prev_dp, dxp_zero=False):
        L = lib()
        B, T, N, R, M = sv["B"], sv["T"], sv["N"], sv["R"], sv["M"]
        C, H = self.C, self.H
        dev = dx.c.device
        dp = s["dp"]
        s1_tok = dp["s1_tok"] if dp else None
        s2_seq = dp["s2_seq"] if dp else None
        s2_tok = dp["s2_tok"] if dp else None
        s3_all = dp["s3_all"] if dp else None
        P = lambda t: t.detach()

        # every parameter gradient of the block is written by a kernel that takes the backward's scale out itself (`gscale` = 1 / S,
        # fp16 flavour) and raises the optimiser's skip flag on a non-finite value (`nonfinite`): no pass over the gradient buffer
        def wgrad(dy, xin, lin):
            (dw, bw), (dbias, _) = gs.target(lin.weight, fused=True), gs.target(lin.bias, fused=True)
            self._wgrad(dy, xin, dw, dbias, bw, gscale=gs.inv, nonfinite=gs.bad)

        # (the 7-us reduces of the LayerNorm partials are deferred: one launch for the whole backward, or per group of blocks under a gradient hook)
        defer = self._ln_defer if self.batch_fused else None

        def lnbwd(dh, x, st, ln, dx_in, dx_out, dxs=None, dxs_scale=None, dxsum=None, dxsum_beta=None):
            (dg, bg), (db, _) = gs.target(ln.weight, fused=True), gs.target(ln.bias, fused=True)
            ops.layernorm_bwd(dh, x, st[0], st[1], P(ln.weight), dg, db, dx_in=dx_in, dx_out=dx_out, beta_acc=bg,
                              dxs=dxs, dxs_scale=dxs_scale, dxsum=dxsum, dxsum_beta=dxsum_beta, gscale=gs.inv, nonfinite=gs.bad,
                              defer=defer)

        if s.get("pruned"):
            # The encoder's last block under prune_last: dx is zero outside the B cls rows (only `x[:, 0]` of the final norm is read,
            # vit.py:418-421), so the MLP, norm2 and the spatial projection back-propagate those rows alone: dy, h_m, u, g are [B, .]
            # tensors here, the weight gradients sums over B (B * T for the projection) rows -- their own small grouped launch: in the
            # block's grouped launch they would force ONE row slice on its 50k-row problems.
            def wgrad_c(d, xin, lin):
                (dw, bw), (dbias, _) = gs.target(lin.weight, fused=True), gs.target(lin.bias, fused=True)
                return (d, xin, dw, dbias, bw, gs.inv, gs.bad)
            wq = [wgrad_c(dy, s["g"], blk.mlp.fc2)]
            du = ops.gemm_nt(dy, self._weight(blk.mlp.fc2.weight).t, L.PVRL_EPI_DGELU, aux=s["u"])
            wq.append(wgrad_c(du, s["h_m"], blk.mlp.fc1))
            dh = ops.gemm_nt(du, self._weight(blk.mlp.fc1.weight).t, L.PVRL_EPI_BF16)
            lnbwd(dh, s["x2"].c, s["st_m"], blk.norm2, dx.c, dx.c)
            dpc = ops.group_bcast(dx.c, B, T, scale=s2_seq, alpha=1.0 / T)
            wq.append(wgrad_c(dpc, s["o_s"][R:], blk.attn.proj))
            if not s.get("cls_attn"):
                ops.gemm_tn_grouped(wq, ws_tag="tn_cls")
            if s.get("cls_attn"):
                do = ops.gemm_nt(dpc, self._weight(blk.attn.proj.weight).t, L.PVRL_EPI_BF16)      # [B * T, C]: the cls queries' dO
            else:
                do = torch.zeros((R + B * T, C), device=dev, dtype=OP16)        # no gradient reaches the patch queries' outputs
                ops.gemm_nt(dpc, self._weight(blk.attn.proj.weight).t, L.PVRL_EPI_BF16, out0=do[R:])
            del du, dpc
        else:
            # ---- MLP ----   (dy = bf16(s3 * dx) arrives from the caller)
            wgrad(dy, s["g"], blk.mlp.fc2)
            du = ops.gemm_nt(dy, self._weight(blk.mlp.fc2.weight).t, L.PVRL_EPI_DGELU, aux=s["u"])
            wgrad(du, s["h_m"], blk.mlp.fc1)
            dh = ops.gemm_nt(du, self._weight(blk.mlp.fc1.weight).t, L.PVRL_EPI_BF16)
            del du
            dps = torch.empty((R + B * T, C), device=dev, dtype=OP16)
            lnbwd(dh, s["x2"].all(), s["st_m"], blk.norm2, dx.all(), dx.all(), dxs=dps[:R], dxs_scale=s2_tok)   # also emits bf16(s2 * dx[:R])

            # ---- spatial ----
            ops.group_bcast(dx.c, B, T, scale=s2_seq, alpha=1.0 / T, out=dps[R:])
            wgrad(dps, s["o_s"], blk.attn.proj)
            do = ops.gemm_nt(dps, self._weight(blk.attn.proj.weight).t, L.PVRL_EPI_BF16)
            del dps
        dqkv = torch.empty((M + B * T, 3 * C), device=dev, dtype=OP16)
        if s.get("cls_attn"):
            self._undef(dqkv[:R, :C])
            # dQ is non-zero for the cls rows alone: the query third of dqkv's patch rows is neither written nor read -- the qkv weight
            # gradient's query rows are sums over the B cls rows (into the block's small grouped launch), the data gradient of the patch
            # rows a K = 2 C product
            ops.attn_cls_bwd(s["qkv_s"], s["o_s"][R:], do, s["lse_s"], B * T, N + 1, H, self.scale, T, R, dqkv[:M], dqkv[M:],
                             zero_patch_dq=False)
            ops.group_reduce(dqkv[M:], B, T, out=dqkv[R:M])
            (dw, bw), (dbias, _) = gs.target(blk.attn.qkv.weight, fused=True), gs.target(blk.attn.qkv.bias, fused=True)
            wq.append((dqkv[R:M, :C], s["h_s"][R:], dw[:C], dbias[:C], bw, gs.inv, gs.bad))
            ops.gemm_tn_grouped(wq, ws_tag="tn_cls")
            self._wgrad(dqkv[:M, C:], s["h_s"], dw[C:], dbias[C:], bw, gscale=gs.inv, nonfinite=gs.bad)
            wt = self._weight(blk.attn.qkv.weight).t
            dh = torch.empty((M, C), device=dev, dtype=OP16)
            ops.gemm_nt(dqkv[:R, C:], wt[:, C:], L.PVRL_EPI_BF16, out0=dh[:R])
            ops.gemm_nt(dqkv[R:M], wt, L.PVRL_EPI_BF16, out0=dh[R:])
            del wq
        else:
            ops.attn_bwd(s["qkv_s"], s["o_s"][:R], s["o_s"][R:], do[:R], do[R:], s["lse_s"], B * T, N + 1, H, self.scale,
                         mode=1, T=T, cls_base=R, dqkv=dqkv[:M], dqkv_cls=dqkv[M:])
            ops.group_reduce(dqkv[M:], B, T, out=dqkv[R:M])
            wgrad(dqkv[:M], s["h_s"], blk.attn.qkv)
            dh = ops.gemm_nt(dqkv[:M], self._weight(blk.attn.qkv.weight).t, L.PVRL_EPI_BF16)
        del dqkv, do
        # also emits dz = bf16(s1 * dx[:R]) and, into temporal_fc.bias.grad, the unscaled column sums of dx[:R]
        dz = torch.empty((R, C), device=dev, dtype=OP16)
        dbf, bbf = gs.target(blk.temporal_fc.bias, fused=True)
        lnbwd(dh, s["x1"].all(), s["st_s"], blk.norm1, dx.all(p_zero=dxp_zero), dx.all(), dxs=dz, dxs_scale=s1_tok, dxsum=dbf,
              dxsum_beta=bbf)

        # ---- temporal (rows [0, R); cls rows pass straight through): proj + temporal_fc as ONE map W_e (_fused_temporal)
        fe = self._fused_temporal(blk)
        dwe = torch.empty((C, C), device=dev, dtype=F32)
        dbe = torch.empty(C, device=dev, dtype=F32)
        if self.batch_fused:                                   # nobody needs this block's gradients before the end of the backward (of its
            self._wgrad(dz, s["o_t"], dwe, dbe, 0.0)           # group, under a gradient hook): the chains run batched in _finish_deferred
            self._chain.append((blk, dwe, dbe))
        else:
            self._wgrad(dz, s["o_t"], dwe, dbe, 0.0, post=lambda b=blk, w=dwe, v=dbe: self._temporal_chain(b, gs, w, v))
        dot = ops.gemm_nt(dz, fe.t, L.PVRL_EPI_BF16)
        if T == 8:
            dqkv_t = ops.attn_t8_bwd(s["qkv_t"], dot, B * N, H, self.scale)
        else:
            dqkv_t, _ = ops.attn_bwd(s["qkv_t"], s["o_t"], None, dot, None, s["lse_t"], B * N, T, H, self.scale, mode=0)
        wgrad(dqkv_t, s["h_t"], blk.temporal_attn.qkv)
        dh = ops.gemm_nt(dqkv_t, self._weight(blk.temporal_attn.qkv.weight).t, L.PVRL_EPI_BF16)
        # the block's input gradient is final after this kernel: it also emits the bf16 operand of the next stage
        # (previous block's MLP backward with that block's DropPath scale, or the patch-embed weight gradient)
        s3p = prev_dp["s3_all"] if (has_prev and prev_dp) else None
        dy_next = torch.empty((M if has_prev else R, C), device=dev, dtype=OP16)
        lnbwd(dh, s["x0"].patch(), s["st_t"], blk.temporal_norm1, dx.patch(), dx.patch(), dxs=dy_next[:R], dxs_scale=s3p)
        if has_prev:
            ops.cast_scale(dx.c, s3p[R:] if s3p is not None else None, out=dy_next[R:])
        self.flush_wgrads()
        return dy_next
