"""Optimiser construction + fused HIP step.

`construct_optimizer(model, cfg)` reproduces the parameter grouping of the reference
lib/models/optimizer.py:10-91 (pre-training: [bn (empty for ViT), non-bn + text] with `lr_mult`;
fine-tuning: encoder group with TRAIN.MULT / BN.WEIGHT_DECAY and head+order group) and its
hyper-parameters (:93-118).  `get_epoch_lr` / `set_lr` are optimizer.py:121-142.  The step itself is
one fused kernel per contiguous parameter range of the model's flat buffers (csrc/optim.hip)
instead of torch.optim's per-tensor foreach loops; numerics follow torch.optim.{SGD, Adam, AdamW}.
"""
import ctypes
import os

import torch

from . import lr_policy
from ._lib import OPERAND, lib


def _inner(model):
    m = model.module if hasattr(model, "module") else model
    return m.model if hasattr(m, "model") and hasattr(m.model, "grad_store") else m


class FusedOptimizer(torch.optim.Optimizer):
    def __init__(self, params, model, method, lr, momentum=0.9, dampening=0.0, nesterov=True, weight_decay=0.0,
                 betas=(0.9, 0.999), eps=1e-8):
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, nesterov=nesterov, weight_decay=weight_decay,
                        betas=betas, eps=eps)
        super().__init__(params, defaults)
        assert method in ("sgd", "adam", "adamw")
        self.method = method
        self.vt = _inner(model)
        self.flat_p = None
        self.buf1 = None   # exp_avg / momentum buffer
        self.buf2 = None   # exp_avg_sq
        self.steps = 0             # optimiser steps taken (informational; the arithmetic uses the per-parameter counts)
        self.param_steps = None    # per parameter of the flat store: updates applied so far (torch.optim's state["step"];
        self.grad_scale = 1.0      # a parameter without a gradient is skipped and its count does not advance)
        # Device-side `misc.check_nan_losses` (tools/train_net.py:174 raises in front of optimizer.step()).  The "skip this step" flag
        # is a slot of the model's flat gradient buffer (GradStore.bad): RAISED on the device by `note_loss()` (a loss of this
        # iteration is not finite) and by the kernels that write parameter gradients (inf / nan in a value they write, include/pvrl.h
        # `nonfinite`); summed over ranks with the tail of the gradient all-reduce, so every replica drops the same step; READ by
        # the update kernels, which leave weights and state untouched when it is set; and CLEARED by step() itself once the update
        # kernels are queued (pvrl_flag_roll also counts the dropped step in `bad_steps`) -- its life cycle belongs to the step,
        # whatever loop calls it.  No host sync per iteration: a training loop reads `dropped_steps()` where it reads its statistics.
        # The HOST's per-parameter step counts (`param_steps`, Adam's bias correction) advance for a dropped step too: the host does not
        # know it was dropped -- until it asks: `dropped_steps()` (the loop's log point, `state_dict()`) takes every drop it finds back
        # from the counts, so at most the steps between two such calls run with a bias correction one step ahead per dropped step.
        # The reference raises at the bad iteration; `train_epoch` at its next log point.
        # `check_grads` (default for the fp16-operand flavour, whose scaled backward can overflow where the loss cannot) additionally
        # scans the gradients of parameters whose producers do not check themselves (GradStore.fused_checked lists the others).
        self.check_grads = os.environ.get("PVRL_CHECK_GRADS", "1" if OPERAND == "f16" else "0") == "1"
        self.bad_steps = None      # device counter of dropped steps
        self._dropped_seen = 0     # ... as of the last dropped_steps() call
        self._pending_idx = None   # parameters that had a gradient in EVERY step() since then (None: no step yet)

    @property
    def skip_flag(self):
        """the device flag (1-element view of the flat gradient buffer); non-zero = the next step() is dropped"""
        return self.vt.grad_store().bad

    def note_loss(self, loss):
        """raise the skip flag when `loss` (device scalar) is not finite -- misc.check_nan_losses without a host sync"""
        bad = self.skip_flag
        bad.copy_(torch.maximum(bad, (~loss.detach().isfinite()).to(bad.dtype).reshape(1)))

    def dropped_steps(self):
        """number of optimiser steps dropped so far because of a non-finite loss / gradient (host sync).  Also RECONCILES the host's
        per-parameter step counts (Adam's bias correction, the `step` of a checkpoint): the host advances them at every step() -- it
        does not know a step was dropped on the device -- so every drop found here is taken back from the parameters that had a
        gradient in the steps since the last call (`state_dict()` calls this first)."""
        n = 0 if self.bad_steps is None else int(self.bad_steps.item())
        k = n - self._dropped_seen
        if k > 0 and self.param_steps is not None:
            idx = range(len(self.param_steps)) if self._pending_idx is None else self._pending_idx
            for i in idx:
                self.param_steps[i] = max(0, self.param_steps[i] - k)
        self._dropped_seen = n
        self._pending_idx = None
        return n

    # -- flat storage -----------------------------------------------------------------------
    def _ensure_flat(self):
        gs = self.vt.grad_store()
        ok = self.flat_p is not None and self.flat_p.device == gs.flat.device and self.flat_p.numel() == gs.flat.numel()
        if ok:
            for p, o in zip(gs.params[:4], gs.offsets[:4]):
                ok &= p.data_ptr() == self.flat_p.data_ptr() + 4 * o
        if not ok:
            old1, old2 = self.buf1, self.buf2
            self.flat_p = torch.zeros_like(gs.flat)
            for p, o in zip(gs.params, gs.offsets):
                v = self.flat_p[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
            self.buf1 = torch.zeros_like(gs.flat) if old1 is None or old1.numel() != gs.flat.numel() else old1.to(gs.flat.device)
            self.buf2 = (torch.zeros_like(gs.flat) if old2 is None or old2.numel() != gs.flat.numel()
                         else old2.to(gs.flat.device)) if self.method != "sgd" else None
        return gs

    def _steps(self, gs):
        if self.param_steps is None or len(self.param_steps) != len(gs.params):
            self.param_steps = [0] * len(gs.params)
        return self.param_steps

    def _runs(self, gs, params, had_grad, by_step=True):
        """-> [[a, b, step]]: maximal contiguous ranges of the flat buffers whose parameters all have a gradient and have
        all been updated `step` times before (one kernel launch each; normally ONE range per parameter group)"""
        ps = self._steps(gs) if by_step else [0] * len(gs.params)
        idx = sorted(gs.index[id(p)] for p in params if id(p) in gs.index and had_grad[gs.index[id(p)]])
        runs = []
        for i in idx:
            a, b = gs.span(i)
            if runs and runs[-1][1] == a and runs[-1][2] == ps[i]:
                runs[-1][1] = b
            else:
                runs.append([a, b, ps[i]])
        return runs

    @torch.no_grad()
    def step(self, closure=None):
        L = lib()
        gs0 = self.vt.grad_store()
        had_grad = [p.grad is not None for p in gs0.params]
        self.vt.adopt_grads(keep_none=True, zero_unused=bool(os.environ.get("PVRL_ZERO_UNUSED")))   # parameters without a gradient are skipped below, as torch.optim does (PVRL_ZERO_UNUSED=1: A/B switch, zero their slots anyway)
        gs = self._ensure_flat()
        self.steps += 1
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        base_p, base_g = self.flat_p.data_ptr(), gs.flat.data_ptr()
        base_1 = self.buf1.data_ptr()
        base_2 = self.buf2.data_ptr() if self.buf2 is not None else 0
        vp = ctypes.c_void_p
        ps = self._steps(gs)
        if self.bad_steps is None or self.bad_steps.device != gs.flat.device:
            self.bad_steps = torch.zeros(1, device=gs.flat.device)
        skip = vp(gs.bad.data_ptr())
        if self.check_grads:       # gradients written by kernels that do not raise the flag themselves (head, glue, MViT engine)
            # ... and, under data parallelism, EVERYTHING: what sits in the buffer now are sums over ranks (an overflow of the sum, an inf
            # from the 16-bit payload's staging), which no producer kernel has seen -- one ~0.1 ms pass
            scan_all = getattr(gs, "reduced_over_ranks", False)
            rest = [p for i, p in enumerate(gs.params) if scan_all or i not in gs.fused_checked]
            for a, b, _ in self._runs(gs, rest, had_grad, by_step=False):
                L.call("pvrl_nonfinite_flag_f32", vp(base_g + 4 * a), b - a, skip, stream)
        stepped = set()
        for g in self.param_groups:
            for a, b, done in self._runs(gs, g["params"], had_grad):
                n, o = b - a, 4 * a
                if self.method == "sgd":
                    L.call("pvrl_sgd_step", vp(base_p + o), vp(base_g + o), vp(base_1 + o), n, float(g["lr"]),
                           float(g["momentum"]), float(g["dampening"]), float(g["weight_decay"]),
                           1 if g["nesterov"] else 0, 1 if done == 0 else 0, float(self.grad_scale), skip, stream)
                else:
                    L.call("pvrl_adam_step", vp(base_p + o), vp(base_g + o), vp(base_1 + o), vp(base_2 + o), n,
                           float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                           float(g["weight_decay"]), done + 1, float(self.grad_scale),
                           1 if self.method == "adamw" else 0, skip, stream)
            for p in g["params"]:
                i = gs.index.get(id(p))
                if i is not None and had_grad[i]:
                    ps[i] += 1
                    stepped.add(i)
        self._pending_idx = stepped if self._pending_idx is None else (self._pending_idx & stepped)
        L.call("pvrl_flag_roll", skip, vp(self.bad_steps.data_ptr()), stream)      # count a dropped step, re-arm the flag
        self._bump(gs)
        return None

    def broadcast_state(self, src=0):
        """every rank takes rank `src`'s parameters, optimiser state and per-parameter step counts (one broadcast per flat buffer):
        what restores identical replicas after ranks disagreed about which parameters a step updates
        (distributed.GradReducer find_unused="cached" -> `on_resync`)"""
        import torch.distributed as dist
        gs = self._ensure_flat()
        for t in (self.flat_p, self.buf1, self.buf2):
            if t is not None:
                dist.broadcast(t, src)
        steps = torch.tensor(self._steps(gs), dtype=torch.int64, device=self.flat_p.device)
        dist.broadcast(steps, src)
        self.param_steps = [int(x) for x in steps.tolist()]
        self._bump(gs)

    def _bump(self, gs):
        # parameters changed in place through the flat buffer: tell the bf16 operand cache to refresh
        self.vt.weights_epoch = getattr(self.vt, "weights_epoch", 0) + 1

    # -- torch.optim-compatible state (checkpoint `optimizer_state`, lib/utils/checkpoint.py:126-131) ---------
    def state_dict(self):
        self.dropped_steps()       # (host sync) the step counts saved below are the updates actually applied
        sd = {"param_groups": [], "state": {}, "fused": {"steps": self.steps, "method": self.method}}
        k = 0
        gs = self.vt.grad_store() if self.flat_p is not None else None
        for g in self.param_groups:
            pg = {x: y for x, y in g.items() if x != "params"}
            pg["params"] = list(range(k, k + len(g["params"])))
            for j, p in enumerate(g["params"]):
                if gs is not None and id(p) in gs.index:
                    i = gs.index[id(p)]
                    a, n = gs.offsets[i], p.numel()
                    done = self._steps(gs)[i]
                    if done == 0:
                        continue                       # torch.optim creates a parameter's state on its first update
                    if self.method == "sgd":
                        sd["state"][k + j] = {"momentum_buffer": self.buf1[a:a + n].view(p.shape).clone()}
                    else:
                        sd["state"][k + j] = {"step": torch.tensor(float(done)),
                                              "exp_avg": self.buf1[a:a + n].view(p.shape).clone(),
                                              "exp_avg_sq": self.buf2[a:a + n].view(p.shape).clone()}
            k += len(g["params"])
            sd["param_groups"].append(pg)
        return sd

    def load_state_dict(self, sd):
        gs = self._ensure_flat()
        self.steps = int(sd.get("fused", {}).get("steps", 0))
        ps = self._steps(gs)
        for i in range(len(ps)):
            ps[i] = 0
        k = 0
        for g, pg in zip(self.param_groups, sd["param_groups"]):
            for key, val in pg.items():
                if key != "params":
                    g[key] = val
            for j, p in enumerate(g["params"]):
                st = sd["state"].get(k + j)
                if st is None or id(p) not in gs.index:
                    continue
                i = gs.index[id(p)]
                a, n = gs.offsets[i], p.numel()
                if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                    self.buf1[a:a + n].copy_(st["momentum_buffer"].reshape(-1))
                    ps[i] = max(1, self.steps)         # a loaded buffer is never overwritten by the "first step" branch
                if "exp_avg" in st:
                    self.buf1[a:a + n].copy_(st["exp_avg"].reshape(-1))
                    self.buf2[a:a + n].copy_(st["exp_avg_sq"].reshape(-1))
                    ps[i] = int(float(st.get("step", 0)))
                    self.steps = max(self.steps, ps[i])
            k += len(g["params"])


def construct_optimizer(model, cfg):
    """Parameter groups per lib/models/optimizer.py:18-91."""
    text = []
    if cfg.TRAIN.MULT != 1.0 or cfg.TRAIN.LINEAR:           # fine-tuning
        enc, rest = [], []
        for name, p in model.named_parameters():
            if "head" not in name and "order" not in name:
                enc.append(p)
                if cfg.TRAIN.LINEAR:
                    p.requires_grad = False
            else:
                rest.append(p)
        if cfg.TRAIN.LINEAR:
            optim_params = [{"params": rest, "weight_decay": cfg.SOLVER.WEIGHT_DECAY, "lr": cfg.SOLVER.BASE_LR, "lr_mult": 1.0}]
        else:
            optim_params = [{"params": enc, "weight_decay": cfg.BN.WEIGHT_DECAY, "lr_mult": cfg.TRAIN.MULT},
                            {"params": rest, "weight_decay": cfg.SOLVER.WEIGHT_DECAY, "lr_mult": 1.0}]
    else:                                                    # pre-training
        bn, non_bn = [], []
        for name, p in model.named_parameters():
            if "bn" in name:
                bn.append(p)
            elif "text_model" in name or "text_module" in name:
                text.append(p)
                if cfg.TRAIN.MULT == 0:
                    p.requires_grad = False
            else:
                non_bn.append(p)
        optim_params = [{"params": bn, "weight_decay": cfg.BN.WEIGHT_DECAY, "lr_mult": 1.0},
                        {"params": non_bn + text, "weight_decay": cfg.SOLVER.WEIGHT_DECAY, "lr_mult": 1.0}]
    s = cfg.SOLVER
    if s.OPTIMIZING_METHOD not in ("sgd", "adam", "adamw"):
        raise NotImplementedError("Does not support {} optimizer".format(s.OPTIMIZING_METHOD))
    return FusedOptimizer(optim_params, model, s.OPTIMIZING_METHOD, lr=s.BASE_LR, momentum=s.MOMENTUM,
                          dampening=s.DAMPENING, nesterov=s.NESTEROV, weight_decay=s.WEIGHT_DECAY)


def get_epoch_lr(cur_epoch, cfg):
    return lr_policy.get_lr_at_epoch(cfg, cur_epoch)


def set_lr(optimizer, new_lr):
    for g in optimizer.param_groups:
        g["lr"] = new_lr * g["lr_mult"] if "lr_mult" in g else new_lr
