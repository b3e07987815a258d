"""Distributed helpers: RCCL (torch.distributed backend "nccl" on ROCm) over xGMI, one process per GPU.

API surface of the reference `lib/utils/distributed.py` (AllGather :13-29, all_gather :31-50,
all_reduce :53-69, init_process_group :72-110, get_world_size / get_rank / is_master_proc ...), plus the
data-parallel gradient reduction that replaces DDP (lib/models/build.py:49-53):

  * gradients live in ONE flat fp32 buffer (engine.GradStore); `GradReducer` all-reduces it in
    per-block chunks (~28 MB each) launched as soon as a block's backward finishes, so the
    transfers overlap the remaining backward GEMMs; xGMI is point-to-point, so few large
    messages beat DDP's 25 MB bucket stream with per-bucket copies.
  * the three per-iteration metric scalars of tools/train_net.py:234 travel as ONE 12-byte
    all-reduce (`all_reduce_scalars`) instead of three collectives + `.item()` syncs.
"""
import functools
import os

import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def _gather_into(out, tensor, world):
    """all_gather straight into `out` (RCCL); the gloo test backend lacks the flat form, so it gets views of `out`."""
    if dist.get_backend() == "gloo":
        dist.all_gather(list(out.chunk(world, 0)), tensor)
    else:
        dist.all_gather_into_tensor(out, tensor)


class AllGather(torch.autograd.Function):
    """All-gather with the reference's backward: the LOCAL slice of the incoming gradient, no
    reduction (distributed.py:24-29)."""

    @staticmethod
    def forward(ctx, tensor):
        world = dist.get_world_size()
        tensor = tensor.contiguous()
        out = torch.empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]), device=tensor.device, dtype=tensor.dtype)
        _gather_into(out, tensor, world)           # one collective into the final buffer, no list + cat
        ctx.rank = dist.get_rank()
        ctx.batch_size = tensor.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None


def all_gather(tensors):
    """distributed.py:31-50: every tensor gathered along dim 0 from all ranks."""
    world = dist.get_world_size()
    out = []
    for t in tensors:
        t = t.contiguous()
        buf = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        _gather_into(buf, t, world)
        out.append(buf)
    return out


def all_reduce(tensors, average=True):
    """distributed.py:53-69 (in place, sum then optional 1/world)."""
    for t in tensors:
        dist.all_reduce(t, async_op=False)
    if average:
        w = dist.get_world_size()
        for t in tensors:
            t.mul_(1.0 / w)
    return tensors


def all_reduce_scalars(values, average=True):
    """Fused form of `du.all_reduce([loss, top1_err, top5_err])`: one collective, no host sync."""
    v = torch.stack([x.detach().float().reshape(()) if torch.is_tensor(x) else torch.tensor(float(x)) for x in values])
    if torch.is_tensor(values[0]):
        v = v.to(values[0].device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v)
        if average:
            v.mul_(1.0 / dist.get_world_size())
    return v


def init_process_group(local_rank, local_world_size, shard_id, num_shards, init_method, dist_backend="nccl"):
    """distributed.py:72-110"""
    proc_rank = local_rank + shard_id * local_world_size
    world_size = local_world_size * num_shards
    if dist_backend == "nccl":
        reserve_comm_cus(world_size)       # CUs for RCCL's channel kernels, before the first launch of this process
    dist.init_process_group(backend=dist_backend, init_method=init_method, world_size=world_size, rank=proc_rank)
    if dist_backend == "nccl":
        torch.cuda.set_device(local_rank)


def is_master_proc(num_gpus=8):
    return dist.get_rank() % num_gpus == 0 if dist.is_available() and dist.is_initialized() else True


def is_root_proc():
    return dist.get_rank() == 0 if dist.is_available() and dist.is_initialized() else True


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def synchronize():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def init_distributed_training(cfg):
    """distributed.py:284-299: one local process group per node."""
    global _LOCAL_PROCESS_GROUP
    if cfg.NUM_GPUS <= 1:
        return
    n = cfg.NUM_GPUS
    for i in range(dist.get_world_size() // n):
        pg = dist.new_group(list(range(i * n, (i + 1) * n)))
        if i == cfg.SHARD_ID:
            _LOCAL_PROCESS_GROUP = pg


def reserve_comm_cus(world, per_xcd=None):
    """Optionally keep r CUs per XCD out of this library's persistent grids for RCCL (r = PVRL_COMM_CUS, default 0 = off), BEFORE
    the first kernel launch and before `init_process_group` (both settings are read once per process):
    PVRL_COMPUTE_CUS = 32 - r CUs per XCD for `gemm_nt8` / `gemm_tn8` / `attn_bwd_fused` (csrc/common.h: one 512-thread workgroup per
    CU on a grid sized to the device) and NCCL_MAX_NCHANNELS = NCCL_MIN_NCHANNELS = 8 r channel workgroups for RCCL.

    Why it is OFF by default -- measured on one MI355X with a kernel that holds CUs on another stream through the whole backward the
    way a collective's channel kernels do (tools/probe/comm_cus_ab.py, profiles/r5_comm_cus.txt): the reservation costs 0.6-0.7 ms
    of a 50.4 ms step (r = 1) and buys nothing at ROCm's default of 4 hardware queues, where the step under a 34 ms CU-holding kernel
    takes 79.0 ms with r = 0 and 80.1 ms with r = 1: what is exposed is not a second wave of workgroups but HARDWARE-QUEUE sharing
    -- the foreign kernel shares one of the 4 HSA queues with streams of the step, and everything queued behind it waits for it.
    (GPU_MAX_HW_QUEUES=8 un-shares the queues -- +6.9 ms under the held CUs instead of +28.6, and there r = 1 is worth 1.3 ms -- but the
    step ALONE is 20 % slower, 60.2 vs 50.4 ms: its graph branches then truly co-run.)  So an all-reduce costs the step at most
    its own kernel time (~3-5 ms per step for 0.94 GB per GPU at 8 ranks, if all of it lands on the critical queue).  Values already in the
    environment win.  -> r"""
    r = int(os.environ.get("PVRL_COMM_CUS", "0") if per_xcd is None else per_xcd)
    if world <= 1 or r <= 0:
        return 0
    os.environ.setdefault("PVRL_COMPUTE_CUS", str(32 - r))
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(8 * r))
    os.environ.setdefault("NCCL_MIN_NCHANNELS", str(8 * r))
    return r


class GradReducer:
    """Sum the flat gradient buffer over ranks, overlapped with the encoder backward.

    Installs `engine.grad_hook`; during loss.backward() the hook fires after each block (last block first) and
    launches an async collective on that block's contiguous slice on a dedicated communication stream (ordered behind
    the main stream and the weight-gradient side stream, and behind nothing later).  `finish()` reduces whatever is
    left (embeddings, head, order transformer: they sit outside the block ranges) and makes the main stream wait.
    Averaging (1/world) is folded into the optimiser's `grad_scale`.

    Gradient accumulation (tools/train_net.py:176-192): `sync = False` on the non-final micro-iterations -- DDP's
    `no_sync()` -- keeps the hook silent so gradients accumulate locally, and the accumulated buffer is reduced ONCE
    during the final micro-iteration's backward.

    `grad_coll` (env PVRL_GRAD_COLL = "allreduce" | "rsag"): the collective per chunk.  "rsag" = reduce-scatter + all-gather of
    the reduced shards, both in place: on a fully-connected xGMI node every rank exchanges 1/W of the chunk with each of its
    W - 1 peers directly (all 7 links busy in both phases) instead of walking a ring (SURVEY 5 / 8e ESTIMATE, never measured -- no multi-GPU node was available: ~0.9 vs ~6 ms for the 538 MB
    of ViT-B at W = 8; keep the default "allreduce" until tests/test_rccl_multi_gpu.py[rsag] has run on real RCCL); chunks whose length W does not divide fall back to the all-reduce.

    `find_unused` (the reference builds DDP with find_unused_parameters=True, lib/models/build.py:51): a parameter
    that received no gradient on ANY rank keeps `.grad is None`, so the optimiser skips it exactly as torch.optim
    does; one that was used on some rank gets the summed gradient on every rank.  The per-parameter "used" flags ride
    in the tail of the flat buffer (no extra collective).  Modes:
      True / "sync"   read the reduced flags back every step (one host sync per step: the host cannot run ahead);
      "cached"        (train() default, env PVRL_FIND_UNUSED) the reduced flags are read back on the FIRST step with a
                      given local used-pattern; later steps with that pattern reuse the decision without a sync.  EVERY step's
                      reduced flags are copied to pinned memory and compared with what the step assumed exactly LAG steps later
                      (by then the copy is long complete: the host blocks on nothing it would not soon need) -- on every rank at
                      the same step.  A rank that finds a mismatch (another rank started / stopped using a parameter: one or
                      more optimiser steps ran with the stale set, so the replicas have diverged) raises a request slot that rides
                      in the same tail; LAG steps later every rank sees it, re-synchronises through `on_resync` (train() installs
                      a broadcast of parameters and optimiser state from rank 0) and continues in "sync" mode.  Without an
                      `on_resync` the mismatch raises: diverged replicas are never trained on silently.
      False / "off"   every parameter is known to be used every step: no flags, no sync (bench.py).

    `grad_comm` (env PVRL_GRAD_COMM = "f32" | "bf16"): payload type of the collective.  "bf16" casts each chunk into a persistent
    16-bit staging buffer on the communication stream, reduces that (269 instead of 538 MB per step for ViT-B) and writes the sum
    back into the fp32 buffer -- SURVEY 8e's half-payload option as a switch.
    """
    LAG = 2          # "cached": a step's flags are checked this many steps later

    def __init__(self, vt, enabled=None, find_unused=True, grad_comm=None, grad_coll=None):
        self.vt = vt
        self.enabled = (get_world_size() > 1) if enabled is None else enabled
        fu = {True: "sync", False: "off", None: "sync"}.get(find_unused, find_unused)
        assert fu in ("sync", "cached", "off"), fu
        self.find_unused = fu
        self.grad_comm = (grad_comm or os.environ.get("PVRL_GRAD_COMM", "f32")).lower()
        assert self.grad_comm in ("f32", "bf16"), self.grad_comm
        self.grad_coll = (grad_coll or os.environ.get("PVRL_GRAD_COLL", "allreduce")).lower()
        assert self.grad_coll in ("allreduce", "rsag"), self.grad_coll
        self.sync = True
        self.handles = []
        self.done = []           # block indices whose slice the hook has already sent this step
        self.on_resync = None    # "cached": called on EVERY rank at the same step when a rank saw a changed used-pattern
        self._comm = None
        self._masks = {}
        self._layout_of = None   # (grad store, per-block parameter indices, per-block span, parameters outside the blocks, spans outside)
        self._stage = None       # "bf16": persistent 16-bit image of the flat buffer (chunks are cast into their own slice)
        self._decided = {}       # "cached": local used-pattern -> the reduced (global) pattern seen with it
        self._queue = []         # "cached": [(event | None, host copy of the reduced tail, assumed pattern)] of the last LAG steps
        self._want_resync = False
        self.host_syncs = 0      # flag read-backs that blocked the host (tests / diagnostics)
        self.resyncs = 0
        # `diag` (bench.py --gpus N): per step, HIP events around (i) every chunk's collective on the communication stream and (ii) the
        # main stream's wait for that stream in finish() -- what a scaling run needs to tell exposed link time from compute (diag_summary)
        self.diag = False
        self._diag_steps = []    # [(e_wait0, e_wait1, [(bytes, e_ready, e_done), ...])]
        self._diag_cur = []
        vt.engine.grad_hook = self._hook if self.enabled else None
        vt.engine.grad_hook_group = self._hook_group if self.enabled else None      # (engines that run the hook per group of blocks)

    # ---- layout of the flat buffer: computed once per gradient store ----------------------------------------------------
    def _layout(self):
        gs = self.vt.grad_store()
        if self._layout_of is None or self._layout_of[0] is not gs:
            pre = getattr(self.vt, "block_prefix", "blocks.")
            blocks = {}
            for k, n in enumerate(gs.names):
                if n.startswith(pre):
                    blocks.setdefault(int(n[len(pre):].split(".", 1)[0]), []).append(k)
            spans = {i: (gs.span(idx[0])[0], gs.span(idx[-1])[1]) for i, idx in blocks.items()}
            inblock = set(k for idx in blocks.values() for k in idx)
            rest = [k for k in range(len(gs.params)) if k not in inblock]
            self._layout_of = (gs, blocks, spans, rest)
        return self._layout_of

    def _block_params(self, i):
        return self._layout()[1][i]

    def _block_range(self, i):
        return self._layout()[2][i]

    def _comm_stream(self, device):
        if self._comm is None or self._comm.device != device:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm

    def _staging(self, gs, a, b):
        if self._stage is None or self._stage.numel() != gs.flat.numel() or self._stage.device != gs.flat.device:
            self._stage = torch.empty(gs.flat.numel(), device=gs.flat.device, dtype=torch.bfloat16)
        return self._stage[a:b]

    def _collective(self, c):
        """sum `c` over the ranks in place; returns the async handles in issue order"""
        world = dist.get_world_size()
        n = c.numel()
        if self.grad_coll == "rsag" and world > 1 and n % world == 0:
            rank = dist.get_rank()
            own = c.view(world, n // world)[rank]
            if dist.get_backend() == "nccl":   # RCCL: in place (recv = send + rank * count for the reduce-scatter, send = recv + rank * count for the gather)
                return [dist.reduce_scatter_tensor(own, c, async_op=True), dist.all_gather_into_tensor(c, own, async_op=True)]
            shard = torch.empty_like(own)                       # gloo (tests): no aliasing guarantees -- through a private shard
            dist.reduce_scatter_tensor(shard, c)
            return [dist.all_gather_into_tensor(c, shard, async_op=True)]
        return [dist.all_reduce(c, async_op=True)]

    def _reduce(self, a, b):
        """async sum over ranks of flat[a:b], ordered after everything enqueued so far on the main stream and on the
        engine's weight-gradient side stream"""
        gs = self.vt.grad_store()
        t = gs.flat[a:b]
        half = self.grad_comm == "bf16"
        if not t.is_cuda:
            c = self._staging(gs, a, b) if half else t
            if half:
                c.copy_(t)
            self.handles.append((self._collective(c), t if half else None, c if half else None))
            return
        comm = self._comm_stream(t.device)
        comm.wait_event(torch.cuda.current_stream().record_event())
        side = getattr(self.vt.engine, "_side", None)
        if side is not None:
            comm.wait_event(side.record_event())
        with torch.cuda.stream(comm):
            c = self._staging(gs, a, b) if half else t
            if half:
                c.copy_(t)                       # cast, reduced and consumed on the communication stream; the slice is this chunk's own
            if self.diag:                        # "ready": everything the chunk depends on has run (and the previous chunk is done)
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(comm)
                self._diag_cur.append([c.numel() * c.element_size(), e0, None])
            hs = self._collective(c)
            if dist.get_backend() == "nccl":
                # RCCL: Work.wait() only orders the CURRENT stream -- the communication stream -- behind the collective (no host block),
                # so it is issued right here: the 16-bit payload is widened as soon as its sum is there instead of in finish(), and an
                # event behind the wait stamps the collective's end where it happens (issued in finish(), every chunk's "end" was the
                # moment the host got there: round 6's first --world1-rccl line read 22.7 ms for the first chunk and 0.01 for the rest)
                for h in hs:
                    h.wait()
                if half:
                    t.copy_(c)
                if self.diag:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(comm)
                    self._diag_cur[-1][2] = e1
                self.handles.append(([], None, None))
            else:                                # gloo (tests): wait() blocks the HOST -- deferred to finish() so that the backward keeps being issued
                self.handles.append((hs, t if half else None, c if half else None))

    def _wait_all(self, device):
        if device.type != "cuda":
            for hs, t, c in self.handles:
                for h in hs:
                    h.wait()
                if c is not None:
                    t.copy_(c)
        else:
            comm = self._comm_stream(device)
            with torch.cuda.stream(comm):
                for j, (hs, t, c) in enumerate(self.handles):
                    for h in hs:
                        h.wait()                # the communication stream waits for the collective ...
                    if c is not None:
                        t.copy_(c)              # ... and widens the 16-bit sum back into the fp32 buffer
                    if self.diag and j < len(self._diag_cur) and self._diag_cur[j][2] is None:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record(comm)
                        self._diag_cur[j][2] = e1
            main = torch.cuda.current_stream()
            if self.diag:                       # how long the main stream sits in this wait = communication NOT hidden under the backward
                w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                w0.record(main)
                main.wait_stream(comm)
                w1.record(main)
                self._diag_steps.append((w0, w1, self._diag_cur))
                self._diag_cur = []
            else:
                main.wait_stream(comm)
        self.handles = []

    def diag_reset(self):
        self._diag_steps, self._diag_cur = [], []

    def diag_summary(self):
        """-> dict for bench.py's `comm` object (call after a device synchronisation): per step the time the main stream waited for the
        communication stream (`exposed_ms_per_step`), and per chunk the time from "its inputs are ready AND the previous chunk is done" to
        "its collective is done" on the communication stream (`allreduce_ms_per_chunk`, with the chunk sizes), i.e. the link + RCCL
        kernel time of that chunk as it ran next to the backward."""
        if not self._diag_steps:
            return None
        exposed, per_chunk, sizes = [], None, None
        for w0, w1, chunks in self._diag_steps:
            exposed.append(w0.elapsed_time(w1))
            ms, prev = [], None
            for nbytes, e0, e1 in chunks:
                if e1 is None:
                    continue
                d = e0.elapsed_time(e1)
                if prev is not None:            # queued behind the previous chunk: count from its end
                    d = min(d, max(0.0, prev.elapsed_time(e1)))
                ms.append(d)
                prev = e1
            if per_chunk is None:
                per_chunk, sizes = [0.0] * len(ms), [round(c[0] / 2 ** 20, 1) for c in chunks]
            if len(ms) == len(per_chunk):
                per_chunk = [a + b for a, b in zip(per_chunk, ms)]
        n = len(self._diag_steps)
        return {"exposed_ms_per_step": round(sum(exposed) / n, 3), "exposed_ms_max_step": round(max(exposed), 3),
                "allreduce_ms_per_chunk": [round(a / n, 3) for a in (per_chunk or [])], "chunk_mb": sizes or [],
                "allreduce_ms_per_step": round(sum(per_chunk or []) / n, 3), "steps": n,
                "grad_coll": self.grad_coll, "grad_comm": self.grad_comm}

    def _settle(self, gs, k):
        """parameter k is about to be reduced: its slot of the flat buffer must hold THIS step's gradient of this rank
        -- zeros when the rank produced none (zero_grad(set_to_none=True) does not clear the buffer), the values of a
        gradient tensor somebody else allocated otherwise"""
        p, v = gs.params[k], gs.views[k]
        if p.grad is None:
            v.zero_()
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v

    def _hook(self, i):
        if not self.sync:
            return
        gs, blocks, spans, _ = self._layout()
        for k in blocks[i]:     # (on the main stream, i.e. before the event the collective is ordered behind)
            self._settle(gs, k)
        self._reduce(*spans[i])
        self.done.append(i)

    def _hook_group(self, blocks):
        """the hook for several blocks at once (engine.EncoderEngine.hook_group): their slices of the flat buffer are adjacent, so ONE
        collective covers them (a ~135 MB all-reduce per group of three ViT-B blocks instead of three of 45 MB)"""
        if not self.sync:
            return
        gs, blk, spans, _ = self._layout()
        for i in blocks:
            for k in blk[i]:
                self._settle(gs, k)
        runs = []
        for a, b in sorted(spans[i] for i in blocks):
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        for a, b in runs:
            self._reduce(a, b)
        self.done.extend(blocks)

    # ---- "cached": late check of every step's reduced flags -------------------------------------------------------------
    def _resync(self):
        self.resyncs += 1
        self._queue = []
        self._decided = {}
        self._want_resync = False
        self.find_unused = "sync"
        if self.on_resync is None:
            raise RuntimeError("GradReducer(find_unused='cached'): the set of parameters used on SOME rank changed between steps and "
                               "optimiser steps ran with the stale set -- the replicas have diverged, and no `on_resync` is installed "
                               "to restore them (train() broadcasts parameters and optimiser state from rank 0).")
        self.on_resync()

    def _check_one(self, block):
        """the oldest queued step: -> False when it is not complete yet (block = False)"""
        ev, host, assumed = self._queue[0]
        if ev is not None:
            if not block and not ev.query():
                return False
            ev.synchronize()
        self._queue.pop(0)
        vals = host.tolist()
        if vals[-1] > 0:                  # some rank asked for a re-synchronisation at that step: every rank acts now, at the same step
            import warnings
            warnings.warn("GradReducer(find_unused='cached'): a rank saw the set of used parameters change; re-synchronising parameters "
                          "and optimiser state from rank 0 and falling back to find_unused='sync' (one host sync per step).")
            self._resync()
            return True
        if assumed is not None and tuple(v > 0 for v in vals[:-1]) != assumed:
            self._want_resync = True       # rides in the next step's tail: all ranks re-synchronise together LAG steps after it
        return True

    def flush(self):
        """"cached" mode checks a step's reduced flags LAG steps later: before a checkpoint / an evaluation / the end of the run, check
        what is still queued NOW (blocking) and let a requested re-synchronisation happen -- a used-pattern change in the last LAG steps
        must not reach a checkpoint unnoticed.  Collective when a re-synchronisation is due: every rank calls it at the same point."""
        if not self.enabled or self.find_unused != "cached":
            return
        while self._queue and self.find_unused == "cached":
            self._check_one(block=True)
        if self._want_resync:       # this rank saw a changed pattern in the steps just checked: the request must reach the others
            flag = torch.tensor([1.0], device=self.vt.grad_store().flat.device)
        else:
            flag = torch.tensor([0.0], device=self.vt.grad_store().flat.device)
        dist.all_reduce(flag)
        if float(flag.item()) > 0 and self.find_unused == "cached":
            import warnings
            warnings.warn("GradReducer.flush: a rank saw the set of used parameters change in the last steps; re-synchronising parameters "
                          "and optimiser state from rank 0 and falling back to find_unused='sync'.")
            self._resync()

    def finish(self):
        if not self.enabled:
            return
        gs, blocks, spans, rest = self._layout()
        had = [p.grad is not None for p in gs.params]           # host-side knowledge, before the buffer is adopted
        sent = set(self.done)
        for i, idx in blocks.items():                           # block parameters the hook has not settled (and sent) this step
            if i not in sent:
                for k in idx:
                    self._settle(gs, k)
        for k in rest:
            self._settle(gs, k)
        if self.find_unused != "off":
            key = tuple(had)                 # a pageable host->device copy would block the host until the backward has
            m = self._masks.get(key)         # drained: the handful of distinct patterns are cached on the device
            if m is None or m.device != gs.used.device:
                m = torch.tensor([1.0 if h else 0.0 for h in had], device=gs.used.device)
                self._masks[key] = m
            gs.used.copy_(m)
            gs.ctl.fill_(1.0 if self._want_resync else 0.0)
            self._want_resync = False
        cur = 0
        for a, b in sorted(spans[i] for i in sent) + [(gs.flat.numel(), gs.flat.numel())]:
            if a > cur:
                self._reduce(cur, a)
            cur = max(cur, b)
        self._wait_all(gs.flat.device)                          # the current (main) stream waits for the collectives
        gs.reduced_over_ranks = dist.get_world_size() > 1       # (optimizer.FusedOptimizer: scan the whole buffer, not only the unchecked part)
        self.done = []
        if self.find_unused == "off":
            used = [True] * len(had)
        elif self.find_unused == "sync":
            used = tuple(bool(x) for x in (gs.used > 0).tolist())              # host sync
            self.host_syncs += 1
        else:
            n = len(had)
            tail = gs.flat[gs.end:gs.end + n + 1]                              # the reduced flags + the request slot
            used = self._decided.get(tuple(had))
            if used is None:                                                   # first step with this local pattern: read it now
                host = tail.to("cpu")                                          # host sync
                self.host_syncs += 1
                used = tuple(v > 0 for v in host[:n].tolist())
                self._decided[tuple(had)] = used
                self._queue.append((None, host, None))
            elif tail.is_cuda:
                host = torch.empty(n + 1, dtype=tail.dtype, pin_memory=True)
                host.copy_(tail, non_blocking=True)
                self._queue.append((torch.cuda.current_stream().record_event(), host, used))
            else:
                self._queue.append((None, tail.clone(), used))
            while len(self._queue) > self.LAG and self.find_unused == "cached":   # the step LAG steps back, on every rank alike
                self._check_one(block=True)
            if self.find_unused == "sync":                                     # a re-synchronisation just happened: this step's flags, read now
                used = tuple(bool(x) for x in (gs.used > 0).tolist())
                self.host_syncs += 1
        for p, v, u in zip(gs.params, gs.views, used):
            p.grad = v if u else None
