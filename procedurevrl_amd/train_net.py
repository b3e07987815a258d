"""Training loop of the pre-training path (reference: tools/train_net.py `train` :417-524, `train_epoch` :56-248).

Kept from the reference: per-iteration LR set (:123-124), `meta` reshape (:146), the model call and KL + MSE loss
(:147-162), NaN check (:174: the bad step is never applied -- device-side skip flag into the fused optimiser, raised at the
next log point), the accumulation branch to GLOBAL_BATCH_SIZE (:176-192, folded into the optimiser's
grad_scale), top-k error on the logits vs a dummy label (:226-231), the fine-tuning branch (:149-150, 163-169: cross entropy /
`smooth` on model(inputs); EPIC-Kitchens verb + noun heads with their accuracies :195-222), json_stats logging, checkpoint /
auto-resume.  Not built: MIXUP (no shipped ProcedureVRL config enables it).
Changed by design: gradients are reduced by `GradReducer` (flat buffer, overlapped with backward) instead of DDP,
the three metric scalars are one all-reduce, and the host reads them only every LOG_PERIOD iterations instead of
`.item()`-syncing every iteration (:234-236)."""
import json
import math
import os
import time

import torch

from . import checkpoint as cu
from . import distributed as du
from . import optimizer as optim
from .build import build_model
from .datasets import DevicePrefetcher, construct_loader, shuffle_dataset
from .vit import pretrain_loss


def topks_correct(preds, labels, ks):
    """lib/utils/metrics.py:10-41"""
    _, top = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)
    rep = labels.view(1, -1).expand_as(top.t())
    correct = top.t().eq(rep)
    return [correct[:k, :].float().sum() for k in ks]


def topk_accuracies(preds, labels, ks):
    """lib/utils/metrics.py:60-70: top-k accuracy in percent"""
    return [x / preds.size(0) * 100.0 for x in topks_correct(preds, labels, ks)]


def multitask_topk_accuracies(preds, labels, ks):
    """lib/utils/metrics.py:73-96 (EPIC action = verb AND noun): a sample counts for k when EVERY task's label is among
    that task's top-k predictions."""
    max_k = int(max(ks))
    all_correct = torch.ones(max_k, labels[0].size(0), dtype=torch.bool, device=labels[0].device)
    for output, label in zip(preds, labels):
        _, top = output.topk(min(max_k, output.shape[1]), 1, True, True)
        hit = top.t().eq(label.view(1, -1).expand_as(top.t()))
        if hit.shape[0] < max_k:                                     # fewer classes than k: the task is always "in the top k" past its width
            hit = torch.cat([hit, hit.any(0, keepdim=True).expand(max_k - hit.shape[0], -1)], 0)
        all_correct = all_correct & hit.cumsum(0).bool()
    return [all_correct[k - 1].float().sum() / labels[0].size(0) * 100.0 for k in ks]


class LabelSmoothingCrossEntropy(torch.nn.Module):
    """timm.loss.LabelSmoothingCrossEntropy as tools/train_net.py:127-128 uses it (`MODEL.LOSS_FUNC: smooth`, smoothing 0.2)"""

    def __init__(self, smoothing=0.1):
        super().__init__()
        self.smoothing = smoothing

    def forward(self, x, target):
        logp = torch.nn.functional.log_softmax(x, dim=-1)
        nll = -logp.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
        return ((1.0 - self.smoothing) * nll + self.smoothing * (-logp.mean(dim=-1))).mean()


def is_pretraining(cfg):
    """tools/train_net.py:146: the pre-training tuple (inputs, narration tokens) vs the fine-tuning call model(inputs)"""
    return cfg.TRAIN.LABEL_EMB != "" and cfg.TRAIN.TEXT != ""


def finetune_loss(preds, labels, cfg):
    """tools/train_net.py:126-136,163-169: cross entropy (or `smooth`) on the logits; EPIC-Kitchens: the mean of the verb and
    the noun loss.  Returns (loss, per-task losses or None)."""
    from .losses import get_loss_func
    if cfg.MIXUP.ENABLED:
        raise NotImplementedError("MIXUP.ENABLED (timm Mixup, tools/train_net.py:137-143) is not built; no shipped ProcedureVRL config enables it")
    loss_fun = LabelSmoothingCrossEntropy(0.2) if cfg.MODEL.LOSS_FUNC == "smooth" else get_loss_func(cfg.MODEL.LOSS_FUNC)(reduction="mean")
    if isinstance(labels, dict) and cfg.TRAIN.DATASET == "Epickitchens":
        lv, ln = loss_fun(preds[0], labels["verb"]), loss_fun(preds[1], labels["noun"])
        return 0.5 * (lv + ln), (lv, ln)
    return loss_fun(preds, labels), None


def log_json_stats(stats):
    """lib/utils/logging.py:83-95: floats rounded to 5 decimals, one `json_stats:` line."""
    stats = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in stats.items()}
    if du.is_master_proc():
        print("json_stats: {:s}".format(json.dumps(stats, sort_keys=True)), flush=True)


def train_epoch(train_loader, model, optimizer, reducer, cur_epoch, cfg, max_iters=None):
    model.train()
    vt = model.model
    if cfg.TRAIN.LINEAR:                                       # train_net.py:72-85: the frozen encoder runs in eval mode (DropPath off)
        if hasattr(vt, "pos_drop"):
            vt.pos_drop.eval()
            vt.blocks.eval()
        elif hasattr(vt, "video_encoder"):                     # MViTv2
            vt.video_encoder.eval()
    if hasattr(vt, "text_model"):
        vt.text_model.eval()                                   # train_net.py:89-95: no gradients on the text model
    data_size = len(train_loader)
    world = du.get_world_size()
    cur_global = cfg.NUM_SHARDS * cfg.TRAIN.BATCH_SIZE
    num_iters = max(1, cfg.GLOBAL_BATCH_SIZE // cur_global)
    accumulate = cur_global < cfg.GLOBAL_BATCH_SIZE
    optimizer.grad_scale = 1.0 / (world * (num_iters if accumulate else 1))
    dev = next(model.parameters()).device
    window = []
    dropped_before = float(optimizer.dropped_steps()) if hasattr(optimizer, "dropped_steps") else 0.0
    last_line = None
    pretrain = is_pretraining(cfg)
    t_last = time.perf_counter()
    for cur_iter, (inputs, labels, _index, meta) in enumerate(train_loader):
        if max_iters is not None and cur_iter >= max_iters:
            break
        inputs = inputs.to(dev, non_blocking=True)
        labels = {k: v.to(dev).view(-1) for k, v in labels.items()} if isinstance(labels, dict) else labels.to(dev).view(-1)
        meta = {k: v.to(dev, non_blocking=True) for k, v in meta.items()}
        lr = optim.get_epoch_lr(cur_epoch + float(cur_iter) / data_size, cfg)
        optim.set_lr(optimizer, lr)
        task_losses = None
        if pretrain:
            meta = {k: meta[k].view(-1, meta[k].shape[-1]) for k in meta}
            pred, teacher_pred, mse = model([inputs, meta])
            loss, loss1, loss2 = pretrain_loss(pred, teacher_pred, mse, cfg)
        else:                                                  # fine-tuning (train_net.py:149-150,163-169): logits -> cross entropy
            pred = model(inputs)
            loss, task_losses = finetune_loss(pred, labels, cfg)
        first_micro = not accumulate or cur_iter % num_iters == 0
        if first_micro:
            optimizer.zero_grad(set_to_none=True)
        last_micro = not accumulate or (cur_iter + 1) % num_iters == 0
        # misc.check_nan_losses(loss) (train_net.py:174) raises BEFORE the step of a bad iteration.  Here the test stays on the
        # device: a non-finite loss raises the optimiser's skip flag (as does a non-finite gradient, where it is written), the flag
        # is summed over ranks with the gradients, the update kernels do nothing when it is set, and the host raises at the next
        # log point, where it reads its statistics anyway.
        if hasattr(optimizer, "note_loss"):
            optimizer.note_loss(loss)
        reducer.sync = last_micro          # DDP no_sync() on the other micro-iterations: accumulate locally, reduce once
        loss.backward()
        if last_micro:
            reducer.finish()
            optimizer.step()
        with torch.no_grad():
            if task_losses is not None:                        # EPIC-Kitchens (train_net.py:195-222): verb / noun / action accuracies
                v1, v5 = topk_accuracies(pred[0], labels["verb"], (1, 5))
                n1, n5 = topk_accuracies(pred[1], labels["noun"], (1, 5))
                a1, a5 = multitask_topk_accuracies((pred[0], pred[1]), (labels["verb"], labels["noun"]), (1, 5))
                stats = du.all_reduce_scalars([loss.detach(), 100.0 - a1, 100.0 - a5, task_losses[0].detach(), task_losses[1].detach(),
                                               v1, v5, n1, n5, a1, a5])
            else:
                # "since labels during pretraining are not used, copy label to the same shape of prediction" (train_net.py:225-227:
                # whenever DEV.ORDER_PRETRAIN_ENABLED is set, pre-training or not)
                lab = labels[0].expand(pred.size(0)) if cfg.DEV.ORDER_PRETRAIN_ENABLED else labels
                assert lab.numel() == pred.size(0), (lab.shape, pred.shape)
                k5 = min(5, pred.shape[0], pred.shape[1])      # train_net.py:230 takes min(5, preds.shape[0])
                c1, c5 = topks_correct(pred, lab, (1, k5))
                stats = du.all_reduce_scalars([loss.detach(), (1.0 - c1 / pred.size(0)) * 100.0, (1.0 - c5 / pred.size(0)) * 100.0])
        window.append(stats)
        if (cur_iter + 1) % cfg.LOG_PERIOD == 0 or cur_iter + 1 == data_size:
            w = torch.stack(window)
            # one host sync per LOG_PERIOD: the medians, "every loss of the window is finite", "no step of the window was skipped"
            drops = optimizer.bad_steps if getattr(optimizer, "bad_steps", None) is not None else torch.zeros(1, device=w.device)
            vals = torch.cat([w.median(0).values, w[:, 0].isfinite().all().float().view(1),
                              (drops.to(w.device) - dropped_before).view(1)]).tolist()
            finite, dropped = vals[-2], vals[-1]
            vals = vals[:-2]
            if finite == 0.0 or dropped != 0.0 or any(math.isnan(v) or math.isinf(v) for v in vals):
                raise RuntimeError("ERROR: Got NaN losses {}".format(time.time()))   # misc.check_nan_losses, on EVERY loss of the window
            now = time.perf_counter()
            dt = (now - t_last) / len(window)
            t_last = now
            nclip = inputs.size(0) * (inputs.size(1) if inputs.dim() == 6 else 1)
            line = {"_type": "train_iter", "epoch": "{}/{}".format(cur_epoch + 1, cfg.SOLVER.MAX_EPOCH),
                    "iter": "{}/{}".format(cur_iter + 1, data_size), "dt": dt, "loss": vals[0],
                    "top1_err": vals[1], "top5_err": vals[2], "lr": lr, "clips_per_s": nclip * world / dt}
            if task_losses is not None:                        # EPICTrainMeter's extra columns (lib/utils/meters.py)
                m = vals
                line.update({"verb_loss": m[3], "noun_loss": m[4], "verb_top1_acc": m[5], "verb_top5_acc": m[6],
                             "noun_top1_acc": m[7], "noun_top5_acc": m[8], "top1_acc": m[9], "top5_acc": m[10]})
            log_json_stats(line)
            last_line = line
            window = []
    return last_line


class ValMeter:
    """lib/utils/meters.py:420-580 (single-label branch): windowed-median minibatch errors for the `val_iter` lines,
    sample-weighted epoch errors and their running minima for the `val_epoch` line."""

    def __init__(self, max_iter, cfg):
        self._cfg = cfg
        self.max_iter = max_iter
        self.min_top1_err = 100.0
        self.min_top5_err = 100.0
        self.reset()

    def reset(self):
        self.win1, self.win5 = [], []
        self.num_top1_mis = 0.0
        self.num_top5_mis = 0.0
        self.num_samples = 0
        self.all_preds, self.all_labels = [], []
        self.t0 = time.perf_counter()

    def update_stats(self, top1_err, top5_err, mb_size):
        p = self._cfg.LOG_PERIOD
        self.win1 = (self.win1 + [top1_err])[-p:]
        self.win5 = (self.win5 + [top5_err])[-p:]
        self.num_top1_mis += top1_err * mb_size
        self.num_top5_mis += top5_err * mb_size
        self.num_samples += mb_size

    def update_predictions(self, preds, labels):
        self.all_preds.append(preds)
        self.all_labels.append(labels)

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return
        med = lambda w: float(torch.tensor(w).median())
        log_json_stats({"_type": "val_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                        "iter": "{}/{}".format(cur_iter + 1, self.max_iter), "top1_err": med(self.win1),
                        "top5_err": med(self.win5)})

    def log_epoch_stats(self, cur_epoch):
        top1_err = self.num_top1_mis / self.num_samples
        top5_err = self.num_top5_mis / self.num_samples
        self.min_top1_err = min(self.min_top1_err, top1_err)
        self.min_top5_err = min(self.min_top5_err, top5_err)
        self.stats = {"_type": "val_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                      "time_diff": time.perf_counter() - self.t0, "top1_err": top1_err, "top5_err": top5_err,
                      "min_top1_err": self.min_top1_err, "min_top5_err": self.min_top5_err}
        log_json_stats(self.stats)
        return self.stats


class EPICValMeter:
    """lib/utils/meters.py:798-960: verb / noun / action (verb AND noun) accuracies -- windowed medians for the `val_iter` lines,
    sample-weighted epoch accuracies and their running maxima for the `val_epoch` line."""
    KEYS = ("verb_top1_acc", "noun_top1_acc", "top1_acc", "verb_top5_acc", "noun_top5_acc", "top5_acc")

    def __init__(self, max_iter, cfg):
        self._cfg = cfg
        self.max_iter = max_iter
        self.best = {"max_" + k: 0.0 for k in self.KEYS}
        self.reset()

    def reset(self):
        self.win = {k: [] for k in self.KEYS}
        self.cor = {k: 0.0 for k in self.KEYS}
        self.num_samples = 0
        self.t0 = time.perf_counter()

    def update_stats(self, top1_acc, top5_acc, mb_size):
        """top1_acc / top5_acc: (verb, noun, action) of the minibatch"""
        for k, v in zip(self.KEYS, tuple(top1_acc) + tuple(top5_acc)):
            self.win[k] = (self.win[k] + [v])[-self._cfg.LOG_PERIOD:]
            self.cor[k] += v * mb_size
        self.num_samples += mb_size

    def update_predictions(self, preds, labels):
        pass                                                    # (EPICValMeter.update_stats clears its lists: nothing is kept)

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self._cfg.LOG_PERIOD != 0:
            return
        line = {"_type": "val_iter", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                "iter": "{}/{}".format(cur_iter + 1, self.max_iter)}
        line.update({k: float(torch.tensor(w).median()) for k, w in self.win.items()})
        log_json_stats(line)

    def log_epoch_stats(self, cur_epoch):
        acc = {k: c / self.num_samples for k, c in self.cor.items()}
        for k, v in acc.items():
            self.best["max_" + k] = max(self.best["max_" + k], v)
        self.stats = {"_type": "val_epoch", "epoch": "{}/{}".format(cur_epoch + 1, self._cfg.SOLVER.MAX_EPOCH),
                      "time_diff": time.perf_counter() - self.t0, **acc, **self.best}
        log_json_stats(self.stats)
        return self.stats


@torch.no_grad()
def eval_epoch(val_loader, model, val_meter, cur_epoch, cfg):
    """tools/train_net.py:251-350: eval-mode forward (softmax probabilities, vit.py:355-356; (verb, noun) for EPIC-Kitchens),
    top-1 / top-5 error of the minibatch -- or the verb / noun / action accuracies of :296-322 -- averaged over ranks (`du.all_reduce`,
    here as ONE collective), fed to the meter; `val_iter` / `val_epoch` json lines."""
    model.eval()
    dev = next(model.parameters()).device
    world = du.get_world_size()
    for cur_iter, (inputs, labels, _index, _meta) in enumerate(val_loader):
        inputs = inputs.to(dev, non_blocking=True)
        labels = {k: v.to(dev).view(-1) for k, v in labels.items()} if isinstance(labels, dict) else labels.to(dev)
        preds = model(inputs)
        if isinstance(labels, dict) and cfg.TRAIN.DATASET == "Epickitchens":
            v1, v5 = topk_accuracies(preds[0], labels["verb"], (1, 5))
            n1, n5 = topk_accuracies(preds[1], labels["noun"], (1, 5))
            a1, a5 = multitask_topk_accuracies((preds[0], preds[1]), (labels["verb"], labels["noun"]), (1, 5))
            v1, n1, a1, v5, n5, a5 = du.all_reduce_scalars([v1, n1, a1, v5, n5, a5]).tolist()   # the reference syncs here too (:302-318)
            val_meter.update_stats((v1, n1, a1), (v5, n5, a5), inputs.size(0) * max(world, 1))
        else:
            if labels.dim() > 1 and labels.numel() == preds.size(0):       # [b, clips] labels of a clips-per-video batch
                labels = labels.reshape(-1)
            k5 = min(5, preds.shape[1])
            c1, c5 = topks_correct(preds, labels, (1, k5))
            errs = du.all_reduce_scalars([(1.0 - c1 / preds.size(0)) * 100.0, (1.0 - c5 / preds.size(0)) * 100.0])
            top1_err, top5_err = errs.tolist()                             # train_net.py:340: the reference syncs here too
            val_meter.update_stats(top1_err, top5_err, inputs.size(0) * max(world, 1))
        val_meter.update_predictions(preds, labels)
        val_meter.log_iter_stats(cur_epoch, cur_iter)
    stats = val_meter.log_epoch_stats(cur_epoch)
    val_meter.reset()
    return stats


def is_eval_epoch(cfg, cur_epoch):
    """lib/utils/misc.py:189-210 without the multigrid schedule (no shipped ViT / MViT config uses it)"""
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    return (cur_epoch + 1) % cfg.TRAIN.EVAL_PERIOD == 0


def train(cfg, max_iters=None):
    du.init_distributed_training(cfg)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg)
    optimizer = optim.construct_optimizer(model, cfg)
    start_epoch = cu.load_train_checkpoint(cfg, model, optimizer)
    # DDP(find_unused_parameters=True) of lib/models/build.py:51 without a host sync per step: see GradReducer
    reducer = du.GradReducer(model.model, find_unused=os.environ.get("PVRL_FIND_UNUSED", "cached"))
    reducer.on_resync = lambda: optimizer.broadcast_state(0)   # replicas that disagreed about a step's parameter set: rank 0's state wins
    train_loader = construct_loader(cfg, "train")
    val_loader = construct_loader(cfg, "val")
    val_meter = (EPICValMeter if cfg.TRAIN.DATASET == "Epickitchens" else ValMeter)(len(val_loader), cfg)   # train_net.py:470-475
    dev = next(model.parameters()).device
    if dev.type == "cuda":
        train_loader = DevicePrefetcher(train_loader, dev)       # H2D copy of batch i+1 under step i
    for cur_epoch in range(start_epoch, cfg.SOLVER.MAX_EPOCH):
        shuffle_dataset(train_loader, cur_epoch)               # train_net.py:503
        train_epoch(train_loader, model, optimizer, reducer, cur_epoch, cfg, max_iters)
        reducer.flush()       # "cached" used-parameter decisions are checked LAG steps late: settle them before a checkpoint / evaluation
        if cu.is_checkpoint_epoch(cfg, cur_epoch):
            cu.save_checkpoint(cfg.OUTPUT_DIR, model, optimizer, cur_epoch, cfg)
        if is_eval_epoch(cfg, cur_epoch):                      # train_net.py:516-518
            train.last_val_stats = eval_epoch(val_loader, model, val_meter, cur_epoch, cfg)
    return model, optimizer
