"""The fused optimiser step (csrc/optim.hip through optimizer.FusedOptimizer) against torch.optim on CPU fp32.

Reference: lib/models/optimizer.py:91-114 builds torch.optim.SGD(nesterov) / Adam / AdamW over parameter groups with an
`lr_mult`; tools/train_net.py:123-124 sets the LR every iteration; :176-192 accumulates `num_iters` micro-batches and
divides the gradients (`p.grad /= num_iters`) before the step.  Here: identical parameters and gradients go through both
implementations for five steps -- two parameter groups (different lr_mult and weight decay), a changing LR, the
accumulation division as `grad_scale`, and one parameter whose gradient is missing on one step (torch.optim skips it and
does not advance its `step`) -- and parameters + optimiser moments must agree to 1e-6 (relative to the tensor's max)."""
import pytest
import torch

TOL = 1e-6


def _run(method):
    import e2e_checks as ec
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    torch.manual_seed(0)
    cfg = ec.make_cfg(2, 32, 64)
    cfg.SOLVER.OPTIMIZING_METHOD = method
    cfg.SOLVER.WEIGHT_DECAY = 1e-2
    cfg.BN.WEIGHT_DECAY = 3e-3
    cfg.TRAIN.MULT = 0.1                       # fine-tuning grouping: encoder (lr_mult 0.1) / head + order transformer
    cfg.SOLVER.MOMENTUM, cfg.SOLVER.DAMPENING, cfg.SOLVER.NESTEROV = 0.9, 0.0, True
    model = ec.build(cfg, synthetic_label_emb(64, 512, seed=1))
    with torch.no_grad():
        for p in model.parameters():           # no parameter sits at exactly zero
            p.add_(torch.randn_like(p) * 0.02)
    opt = construct_optimizer(model, cfg)
    assert len([g for g in opt.param_groups if g["params"]]) == 2

    # the same thing on the CPU with torch.optim
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    cpu = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in named}
    by_id = {id(p): n for n, p in named}
    groups = []
    for g in opt.param_groups:
        ps = [cpu[by_id[id(p)]] for p in g["params"] if id(p) in by_id]
        groups.append({"params": ps, "weight_decay": g["weight_decay"], "lr_mult": g["lr_mult"]})
    s = cfg.SOLVER
    if method == "sgd":
        ref = torch.optim.SGD(groups, lr=s.BASE_LR, momentum=s.MOMENTUM, weight_decay=s.WEIGHT_DECAY, dampening=s.DAMPENING,
                              nesterov=s.NESTEROV)
    else:
        cls = torch.optim.Adam if method == "adam" else torch.optim.AdamW
        ref = cls(groups, lr=s.BASE_LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=s.WEIGHT_DECAY)

    gen = torch.Generator().manual_seed(5)
    skip_name = "model.time_embed"
    num_iters = 4
    opt.grad_scale = 1.0 / num_iters
    worst = {}
    for step in range(5):
        lr = 1e-3 * (1.0 + 0.37 * step)
        set_lr(opt, lr)
        for g in ref.param_groups:
            g["lr"] = lr * g["lr_mult"]
        opt.zero_grad(set_to_none=True)
        ref.zero_grad(set_to_none=True)
        for n, p in named:
            if n == skip_name and step == 2:
                continue                                   # no gradient this step: both sides must skip the parameter
            micro = [torch.randn(p.shape, generator=gen) * 0.05 for _ in range(num_iters)]
            acc = micro[0].clone()
            for m in micro[1:]:
                acc += m                                   # loss.backward() accumulating num_iters micro-batches
            p.grad = acc.to(p.device)
            cpu[n].grad = acc.clone()
            cpu[n].grad /= num_iters                       # tools/train_net.py:187-189
        opt.step()
        ref.step()
        torch.cuda.synchronize()
        for n, p in named:
            d = float((p.detach().cpu() - cpu[n].detach()).abs().max() / cpu[n].detach().abs().max())
            worst["param"] = max(worst.get("param", 0.0), d)
    sd = opt.state_dict()
    k = 0
    idx = {}
    for g in opt.param_groups:
        for j, p in enumerate(g["params"]):
            if id(p) in by_id:
                idx[by_id[id(p)]] = k + j
        k += len(g["params"])
    for n, _ in named:
        st_ref = ref.state[cpu[n]]
        st = sd["state"][idx[n]]
        keys = ["momentum_buffer"] if method == "sgd" else ["exp_avg", "exp_avg_sq"]
        for key in keys:
            a, b = st[key].cpu(), st_ref[key]
            worst[key] = max(worst.get(key, 0.0), float((a - b).abs().max() / b.abs().max()))
        if method != "sgd":
            want = 4.0 if n == skip_name else 5.0
            assert float(st["step"]) == float(st_ref["step"]) == want, (n, st["step"], st_ref["step"])
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["adamw", "adam", "sgd"])
def test_fused_step_matches_torch_optim(method):
    worst = _run(method)
    print(method, worst)
    for k, v in worst.items():
        assert v <= TOL, (method, k, v, worst)


@pytest.mark.gpu
def test_resume_from_a_torch_optim_sgd_checkpoint_keeps_the_momentum():
    """A reference checkpoint's `optimizer_state` is a torch.optim state_dict (lib/utils/checkpoint.py:126-131) with no
    private keys.  Loading it and stepping on must continue torch's trajectory: in particular the loaded SGD momentum
    buffers are used, not overwritten by the "first step" branch (round-1 ADVICE)."""
    import e2e_checks as ec
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    torch.manual_seed(1)
    cfg = ec.make_cfg(1, 32, 64)
    cfg.SOLVER.OPTIMIZING_METHOD = "sgd"
    cfg.SOLVER.MOMENTUM, cfg.SOLVER.NESTEROV, cfg.SOLVER.WEIGHT_DECAY = 0.9, True, 1e-4
    model = ec.build(cfg, synthetic_label_emb(64, 512, seed=1))
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    cpu = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in named}
    ref = torch.optim.SGD([{"params": []}, {"params": [cpu[n] for n, _ in named]}], lr=1e-2, momentum=0.9, nesterov=True,
                          weight_decay=1e-4)                                 # same group layout as construct_optimizer
    gen = torch.Generator().manual_seed(9)
    grads = [{n: torch.randn(p.shape, generator=gen) * 0.1 for n, p in named} for _ in range(3)]
    for k in range(2):                                                        # two steps in torch only
        for n, _ in named:
            cpu[n].grad = grads[k][n].clone()
        ref.step()
    with torch.no_grad():
        for n, p in named:
            p.copy_(cpu[n])
    opt = construct_optimizer(model, cfg)
    opt.load_state_dict(ref.state_dict())
    set_lr(opt, 1e-2)
    for n, p in named:
        p.grad = grads[2][n].to(p.device)
        cpu[n].grad = grads[2][n].clone()
    opt.step()
    ref.step()
    torch.cuda.synchronize()
    worst = max(float((p.detach().cpu() - cpu[n].detach()).abs().max() / cpu[n].detach().abs().max()) for n, p in named)
    assert worst <= TOL, worst


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["adamw", "sgd"])
def test_non_finite_step_is_dropped_on_the_device(method):
    """tools/train_net.py:174 `misc.check_nan_losses(loss)` raises in front of optimizer.step(): a bad iteration never touches the
    weights.  Here the check stays on the device: with an inf in ONE gradient (optimizer.check_grads) or a non-finite loss flag
    (optimizer.skip_flag set by the loop) the update kernels return at once -- parameters and state bit-equal -- and a clean
    step afterwards updates as usual."""
    import e2e_checks as ec
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.optimizer import construct_optimizer, set_lr
    torch.manual_seed(0)
    cfg = ec.make_cfg(2, 32, 64)
    cfg.SOLVER.OPTIMIZING_METHOD = method
    model = ec.build(cfg, synthetic_label_emb(64, 512, seed=1))
    opt = construct_optimizer(model, cfg)
    set_lr(opt, 1e-3)
    gs = model.model.grad_store()
    gen = torch.Generator(device=gs.flat.device).manual_seed(3)

    def fill():
        opt.zero_grad(set_to_none=True)
        for p, v in zip(gs.params, gs.views):
            v.copy_(torch.randn(v.shape, device=v.device, generator=gen) * 1e-2)
            p.grad = v

    fill(); opt.step()                                            # a clean first step (creates the state)
    snap = lambda: (opt.flat_p.clone(), opt.buf1.clone(), None if opt.buf2 is None else opt.buf2.clone())
    # (a) inf in one gradient element, gradient check on
    opt.check_grads = True
    before = snap()
    fill()
    gs.views[5].view(-1)[3] = float("inf")
    opt.step()
    after = snap()
    assert opt.param_steps[5] == 2, "the host counted the dropped step (it cannot know yet)"
    assert opt.dropped_steps() == 1
    assert opt.param_steps[5] == 1 and set(opt.param_steps) == {1}, "dropped_steps() takes the dropped step back from the host's counts"
    assert float(opt.skip_flag) == 0.0, "step() re-arms the flag it consumed"
    assert all(b is None or torch.equal(a, b) for a, b in zip(after, before)), "a step with an inf gradient changed weights or state"
    # (b) the loop's loss flag alone (optimizer.note_loss), gradients finite
    opt.check_grads = False
    fill()
    opt.note_loss(torch.tensor(float("nan"), device=gs.flat.device))
    assert float(opt.skip_flag) == 1.0
    opt.note_loss(torch.tensor(1.0, device=gs.flat.device))      # a finite loss of a later micro-iteration does not lower it
    assert float(opt.skip_flag) == 1.0
    opt.step()
    assert all(b is None or torch.equal(a, b) for a, b in zip(snap(), before))
    assert opt.dropped_steps() == 2
    # (c) the NEXT step of a loop that never touches the flag is applied (the flag's life cycle belongs to step())
    fill()
    opt.step()
    assert not torch.equal(snap()[0], before[0])
    assert opt.dropped_steps() == 2
    # (d) a kernel that writes a parameter gradient raises the flag itself: the weight-gradient reduce (and takes a scale out)
    from procedurevrl_amd import ops
    P = torch.randn(512, 256, device=gs.flat.device).to(ops.OP16)
    Q = torch.randn(512, 256, device=gs.flat.device).to(ops.OP16)
    dW = torch.zeros(256, 256, device=gs.flat.device)
    db = torch.zeros(256, device=gs.flat.device)
    flag = torch.zeros(1, device=gs.flat.device)
    half = torch.full((1,), 0.5, device=gs.flat.device)
    ops.gemm_tn(P, Q, dW, db, gscale=half, nonfinite=flag)
    ref = 0.5 * (P.float().t() @ Q.float())
    assert float((dW - ref).norm() / ref.norm()) < 1e-5 and float(flag) == 0.0
    ops.gemm_tn(P, Q, dW, db, beta=1.0, gscale=half, nonfinite=flag)              # what is there (beta) is not scaled again
    assert float((dW - 2 * ref).norm() / ref.norm()) < 1e-5
    P[7, 3] = float("inf")
    ops.gemm_tn(P, Q, dW, db, gscale=half, nonfinite=flag)
    assert float(flag) == 1.0
    # nan counts as well; finite values never raise the flag
    from procedurevrl_amd._lib import lib
    import ctypes
    flag = torch.zeros((), device=gs.flat.device)
    x = torch.randn(100003, device=gs.flat.device)[3:]            # (unaligned start is refused: the flat buffers are 256-byte aligned)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with pytest.raises(Exception):
        lib().call("pvrl_nonfinite_flag_f32", ctypes.c_void_p(x.data_ptr()), x.numel(), ctypes.c_void_p(flag.data_ptr()), stream)
    y = torch.randn(100003, device=gs.flat.device)
    lib().call("pvrl_nonfinite_flag_f32", ctypes.c_void_p(y.data_ptr()), y.numel(), ctypes.c_void_p(flag.data_ptr()), stream)
    assert float(flag) == 0.0
    y[-1] = float("nan")
    lib().call("pvrl_nonfinite_flag_f32", ctypes.c_void_p(y.data_ptr()), y.numel(), ctypes.c_void_p(flag.data_ptr()), stream)
    assert float(flag) == 1.0
