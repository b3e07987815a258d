"""Learning-rate schedules: same policies and formulas as the reference lib/utils/lr_policy.py:8-87
(`get_lr_at_epoch`, cosine, steps_with_relative_lrs, warm-up), restated."""
import math


def _cosine(cfg, cur_epoch):
    s = cfg.SOLVER
    assert s.COSINE_END_LR < s.BASE_LR
    return s.COSINE_END_LR + (s.BASE_LR - s.COSINE_END_LR) * (math.cos(math.pi * cur_epoch / s.MAX_EPOCH) + 1.0) * 0.5


def get_step_index(cfg, cur_epoch):
    steps = list(cfg.SOLVER.STEPS) + [cfg.SOLVER.MAX_EPOCH]
    ind = 0
    for ind, step in enumerate(steps):
        if cur_epoch < step:
            break
    return ind - 1


def _steps_with_relative_lrs(cfg, cur_epoch):
    return cfg.SOLVER.LRS[get_step_index(cfg, cur_epoch)] * cfg.SOLVER.BASE_LR


_POLICIES = {"cosine": _cosine, "steps_with_relative_lrs": _steps_with_relative_lrs}


def get_lr_func(lr_policy):
    if lr_policy not in _POLICIES:
        raise NotImplementedError("Unknown LR policy: {}".format(lr_policy))
    return _POLICIES[lr_policy]


def get_lr_at_epoch(cfg, cur_epoch):
    f = get_lr_func(cfg.SOLVER.LR_POLICY)
    lr = f(cfg, cur_epoch)
    if cur_epoch < cfg.SOLVER.WARMUP_EPOCHS:
        lr_start = cfg.SOLVER.WARMUP_START_LR
        lr_end = f(cfg, cfg.SOLVER.WARMUP_EPOCHS)
        alpha = (lr_end - lr_start) / cfg.SOLVER.WARMUP_EPOCHS
        lr = cur_epoch * alpha + lr_start
    return lr
