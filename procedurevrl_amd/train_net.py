"""Training loop of the pre-training path (reference: tools/train_net.py `train` :417-524, `train_epoch` :56-248).

Kept from the reference: per-iteration LR set (:123-124), `meta` reshape (:146), the model call and KL + MSE loss
(:147-162), NaN check (:174), the accumulation branch to GLOBAL_BATCH_SIZE (:176-192, folded into the optimiser's
grad_scale), top-k error on the logits vs a dummy label (:226-231), json_stats logging, checkpoint / auto-resume.
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


def log_json_stats(stats):
    """lib/utils/logging.py:83-95: floats rounded to 5 decimals, one `json_stats:` line."""
    stats = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in stats.items()}
    if du.is_master_proc():
        print("json_stats: {:s}".format(json.dumps(stats, sort_keys=True)), flush=True)


def train_epoch(train_loader, model, optimizer, reducer, cur_epoch, cfg, max_iters=None):
    model.train()
    vt = model.model
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
    t_last = time.perf_counter()
    for cur_iter, (inputs, labels, _index, meta) in enumerate(train_loader):
        if max_iters is not None and cur_iter >= max_iters:
            break
        inputs = inputs.to(dev, non_blocking=True)
        labels = labels.to(dev).view(-1)
        meta = {k: v.to(dev, non_blocking=True) for k, v in meta.items()}
        lr = optim.get_epoch_lr(cur_epoch + float(cur_iter) / data_size, cfg)
        optim.set_lr(optimizer, lr)
        meta = {k: meta[k].view(-1, meta[k].shape[-1]) for k in meta}
        pred, teacher_pred, mse = model([inputs, meta])
        loss, loss1, loss2 = pretrain_loss(pred, teacher_pred, mse, cfg)
        if not accumulate or cur_iter % num_iters == 0:
            optimizer.zero_grad(set_to_none=True)
        last_micro = not accumulate or (cur_iter + 1) % num_iters == 0
        reducer.sync = last_micro          # DDP no_sync() on the other micro-iterations: accumulate locally, reduce once
        loss.backward()
        if last_micro:
            reducer.finish()
            optimizer.step()
        with torch.no_grad():
            lab = labels[0].expand(pred.size(0))
            k5 = min(5, pred.shape[1])
            c1, c5 = topks_correct(pred, lab, (1, k5))
            stats = du.all_reduce_scalars([loss.detach(), (1.0 - c1 / pred.size(0)) * 100.0, (1.0 - c5 / pred.size(0)) * 100.0])
        window.append(stats)
        if (cur_iter + 1) % cfg.LOG_PERIOD == 0 or cur_iter + 1 == data_size:
            w = torch.stack(window)
            vals = torch.cat([w.median(0).values, w[:, 0].isfinite().all().float().view(1)]).tolist()   # one host sync per LOG_PERIOD
            if vals[3] == 0.0 or any(math.isnan(v) or math.isinf(v) for v in vals):
                raise RuntimeError("ERROR: Got NaN losses {}".format(time.time()))   # misc.check_nan_losses, on EVERY loss of the window
            now = time.perf_counter()
            dt = (now - t_last) / len(window)
            t_last = now
            log_json_stats({"_type": "train_iter", "epoch": "{}/{}".format(cur_epoch + 1, cfg.SOLVER.MAX_EPOCH),
                            "iter": "{}/{}".format(cur_iter + 1, data_size), "dt": dt, "loss": vals[0],
                            "top1_err": vals[1], "top5_err": vals[2], "lr": lr,
                            "clips_per_s": inputs.size(0) * inputs.size(1) * world / dt})
            window = []
    return None


def train(cfg, max_iters=None):
    du.init_distributed_training(cfg)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg)
    optimizer = optim.construct_optimizer(model, cfg)
    start_epoch = cu.load_train_checkpoint(cfg, model, optimizer)
    reducer = du.GradReducer(model.model)
    train_loader = construct_loader(cfg, "train")
    dev = next(model.parameters()).device
    if dev.type == "cuda":
        train_loader = DevicePrefetcher(train_loader, dev)       # H2D copy of batch i+1 under step i
    for cur_epoch in range(start_epoch, cfg.SOLVER.MAX_EPOCH):
        shuffle_dataset(train_loader, cur_epoch)               # train_net.py:503
        train_epoch(train_loader, model, optimizer, reducer, cur_epoch, cfg, max_iters)
        if cu.is_checkpoint_epoch(cfg, cur_epoch):
            cu.save_checkpoint(cfg.OUTPUT_DIR, model, optimizer, cur_epoch, cfg)
    return model, optimizer
