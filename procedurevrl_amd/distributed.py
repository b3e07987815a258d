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


class GradReducer:
    """Sum the flat gradient buffer over ranks, overlapped with the encoder backward.

    Installs `engine.grad_hook`; during loss.backward() the hook fires after each block (last block first) and
    launches an async all-reduce of that block's contiguous slice on a dedicated communication stream (ordered behind
    the main stream and the weight-gradient side stream, and behind nothing later).  `finish()` reduces whatever is
    left (embeddings, head, order transformer: they sit outside the block ranges) and makes the main stream wait.
    Averaging (1/world) is folded into the optimiser's `grad_scale`.

    Gradient accumulation (tools/train_net.py:176-192): `sync = False` on the non-final micro-iterations -- DDP's
    `no_sync()` -- keeps the hook silent so gradients accumulate locally, and the accumulated buffer is reduced ONCE
    during the final micro-iteration's backward.

    `find_unused` (the reference builds DDP with find_unused_parameters=True, lib/models/build.py:51): a parameter
    that received no gradient on ANY rank keeps `.grad is None`, so the optimiser skips it exactly as torch.optim
    does; one that was used on some rank gets the summed gradient on every rank.  The per-parameter "used" flags ride
    in the tail of the flat buffer (no extra collective).  Modes:
      True / "sync"   read the reduced flags back every step (one host sync per step: the host cannot run ahead);
      "cached"        (train() default, env PVRL_FIND_UNUSED) the reduced flags are read back on the FIRST step with a
                      given local used-pattern; later steps with that pattern reuse the decision without a sync, and
                      every step's flags are checked one step late from a pinned host copy (a changed pattern on some
                      other rank is then seen: warning + the mode drops to "sync" for the rest of the run);
      False / "off"   every parameter is known to be used every step: no flags, no sync (bench.py).

    `grad_comm` (env PVRL_GRAD_COMM = "f32" | "bf16"): payload type of the all-reduce.  "bf16" casts each chunk on the
    communication stream, reduces the 16-bit copy (269 instead of 538 MB per step for ViT-B) and writes the sum back
    into the fp32 buffer -- SURVEY 8e's half-payload option as a switch.
    """

    def __init__(self, vt, enabled=None, find_unused=True, grad_comm=None):
        self.vt = vt
        self.enabled = (get_world_size() > 1) if enabled is None else enabled
        fu = {True: "sync", False: "off", None: "sync"}.get(find_unused, find_unused)
        assert fu in ("sync", "cached", "off"), fu
        self.find_unused = fu
        self.grad_comm = (grad_comm or os.environ.get("PVRL_GRAD_COMM", "f32")).lower()
        assert self.grad_comm in ("f32", "bf16"), self.grad_comm
        self.sync = True
        self.handles = []
        self.done = []
        self._comm = None
        self._masks = {}
        self._decided = {}       # "cached": local used-pattern -> the reduced (global) pattern seen with it
        self._pending = None     # "cached": (event, pinned flags, assumed pattern) of the last unchecked step
        self.host_syncs = 0      # flag read-backs that blocked the host (tests / diagnostics)
        vt.engine.grad_hook = self._hook if self.enabled else None

    def _block_params(self, i):
        gs = self.vt.grad_store()
        pre = f"{getattr(self.vt, 'block_prefix', 'blocks.')}{i}."
        return [k for k, n in enumerate(gs.names) if n.startswith(pre)]

    def _block_range(self, i):
        gs = self.vt.grad_store()
        idx = self._block_params(i)
        return gs.span(idx[0])[0], gs.span(idx[-1])[1]

    def _comm_stream(self, device):
        if self._comm is None or self._comm.device != device:
            self._comm = torch.cuda.Stream(device=device)
        return self._comm

    def _reduce(self, t):
        """async all-reduce of a slice of the flat buffer, ordered after everything enqueued so far on the main stream
        and on the engine's weight-gradient side stream"""
        half = self.grad_comm == "bf16"
        if not t.is_cuda:
            if half:
                c = t.to(torch.bfloat16)
                dist.all_reduce(c)
                t.copy_(c)
            else:
                self.handles.append((dist.all_reduce(t, async_op=True), None, None))
            return
        comm = self._comm_stream(t.device)
        comm.wait_event(torch.cuda.current_stream().record_event())
        side = getattr(self.vt.engine, "_side", None)
        if side is not None:
            comm.wait_event(side.record_event())
        with torch.cuda.stream(comm):
            if half:
                c = t.to(torch.bfloat16)         # allocated, reduced and consumed on the communication stream
                self.handles.append((dist.all_reduce(c, async_op=True), t, c))
            else:
                self.handles.append((dist.all_reduce(t, async_op=True), None, None))

    def _wait_all(self, device):
        if device.type != "cuda":
            for h, _, _ in self.handles:
                h.wait()
        else:
            comm = self._comm_stream(device)
            with torch.cuda.stream(comm):
                for h, t, c in self.handles:
                    h.wait()                    # the communication stream waits for the collective ...
                    if c is not None:
                        t.copy_(c)              # ... and widens the 16-bit sum back into the fp32 buffer
            torch.cuda.current_stream().wait_stream(comm)
        self.handles = []

    def _settle(self, gs, k):
        """parameter k is about to be all-reduced: its slot of the flat buffer must hold THIS step's gradient of this rank
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
        gs = self.vt.grad_store()
        idx = self._block_params(i)
        for k in idx:           # (on the main stream, i.e. before the event the all-reduce is ordered behind)
            self._settle(gs, k)
        a, b = gs.span(idx[0])[0], gs.span(idx[-1])[1]
        self._reduce(gs.flat[a:b])
        self.done.append((a, b))

    def _check_pending(self, block=False):
        """"cached" mode: compare the flags of an earlier step (pinned host copy) with the pattern that step assumed"""
        if self._pending is None:
            return
        ev, host, assumed = self._pending
        if not block and not ev.query():
            return
        ev.synchronize()
        self._pending = None
        if tuple(bool(x) for x in (host > 0).tolist()) != assumed:
            import warnings
            warnings.warn("GradReducer(find_unused='cached'): the set of parameters used on SOME rank changed between steps; "
                          "one optimiser step ran with the previous set.  Falling back to find_unused='sync' (one host sync "
                          "per step).")
            self.find_unused = "sync"
            self._decided = {}

    def finish(self):
        if not self.enabled:
            return
        gs = self.vt.grad_store()
        had = [p.grad is not None for p in gs.params]           # host-side knowledge, before the buffer is adopted
        reduced = sorted(self.done)
        in_done = lambda a: any(x <= a < y for x, y in reduced)
        for k in range(len(gs.params)):                         # block parameters were settled (and sent) by the hook
            if not in_done(gs.offsets[k]):
                self._settle(gs, k)
        if self.find_unused != "off":
            key = tuple(had)                 # a pageable host->device copy would block the host until the backward has
            m = self._masks.get(key)         # drained: the handful of distinct patterns are cached on the device
            if m is None or m.device != gs.used.device:
                m = torch.tensor([1.0 if h else 0.0 for h in had], device=gs.used.device)
                self._masks[key] = m
            gs.used.copy_(m)
        cur = 0
        for a, b in reduced + [(gs.flat.numel(), gs.flat.numel())]:
            if a > cur:
                self._reduce(gs.flat[cur:a])
            cur = max(cur, b)
        self._wait_all(gs.flat.device)                          # the current (main) stream waits for the collectives
        self.done = []
        if self.find_unused == "off":
            used = [True] * len(had)
        else:
            if self.find_unused == "cached":
                self._check_pending()
            used = self._decided.get(tuple(had)) if self.find_unused == "cached" else None
            if used is None:
                self._check_pending(block=True)
                used = tuple(bool(x) for x in (gs.used > 0).tolist())          # host sync
                self.host_syncs += 1
                if self.find_unused == "cached":
                    self._decided[tuple(had)] = used
            elif gs.used.is_cuda:
                host = torch.empty(gs.used.shape, dtype=gs.used.dtype, pin_memory=True)
                host.copy_(gs.used, non_blocking=True)
                self._pending = (torch.cuda.current_stream().record_event(), host, used)
        for p, v, u in zip(gs.params, gs.views, used):
            p.grad = v if u else None
