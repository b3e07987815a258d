"""World-size 4 and 8 (gloo, CPU) tests of the data-parallel path (BASELINE configs[2]: 8 ranks): AllGather slicing, the chunked
flat-gradient reducer with both collectives (all-reduce | reduce-scatter + all-gather) and both payload types, and the
find_unused="cached" mode when ONE rank starts using a parameter mid-run: every rank must detect it, re-synchronise
at the same step through `on_resync`, and continue in "sync" mode."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fill(gs, vt, red, value, skip=()):
    """one fake backward: every parameter (but `skip`) gets a gradient = `value`, the per-block hook fires last block first"""
    for p in gs.params:
        p.grad = None
    for k, (p, view) in enumerate(zip(gs.params, gs.views)):
        if k in skip:
            continue
        view.fill_(value)
        p.grad = view
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)
    red.finish()


def _worker(rank, world, port, results):
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from procedurevrl_amd import distributed as du
    out = {}
    # --- AllGather (lib/utils/distributed.py:13-29): rank r's rows land at [r * b, (r + 1) * b); backward = the local slice
    x = torch.full((2, 3), float(rank), requires_grad=True)
    y = du.AllGather.apply(x)
    w = torch.arange(world * 6, dtype=torch.float32).view(world * 2, 3)
    (y * w).sum().backward()
    out["ag_rows"] = y.detach()[:, 0].tolist()
    out["ag_bwd"] = bool(torch.equal(x.grad, w[2 * rank:2 * rank + 2]))
    g1, = du.all_gather([torch.full((1, 2), float(rank))])
    out["all_gather"] = g1[:, 0].tolist()
    # --- the chunked reducer: collective x payload
    from test_host_logic import _small_model
    torch.manual_seed(0)
    _, model = _small_model(text=False)
    vt = model.model
    gs = vt.grad_store()
    want = 0.25 * world * (world + 1) / 2
    for coll in ("allreduce", "rsag"):
        for comm in ("f32", "bf16"):
            red = du.GradReducer(vt, find_unused=False, grad_comm=comm, grad_coll=coll)
            _fill(gs, vt, red, 0.25 * (rank + 1))
            vals = sorted(set(torch.cat([v.reshape(-1) for v in gs.views]).tolist()))
            out[f"sum_{coll}_{comm}"] = vals
            # the same with gradient accumulation: the hook is silent on the first micro-iteration
            red.sync = False
            for i in reversed(range(len(vt.blocks))):
                vt.engine.grad_hook(i)
            red.sync = True
    out["want"] = want
    out["spans_cached"] = red._layout() is red._layout()
    # --- find_unused: a parameter unused everywhere stays None; one used on rank 1 only gets the sum everywhere
    names = gs.names
    i_none, i_half = names.index("time_embed"), names.index("cls_token")
    red = du.GradReducer(vt, find_unused="sync", grad_coll="rsag")
    _fill(gs, vt, red, float(rank + 1), skip=(i_none,) if rank == 1 else (i_none, i_half))
    out["unused_none"] = gs.params[i_none].grad is None
    out["half_used"] = sorted(set(gs.params[i_half].grad.reshape(-1).tolist()))          # rank 1's 2.0 + zeros
    # --- find_unused="cached" and a used-pattern that changes on ONE rank at step 4
    red = du.GradReducer(vt, find_unused="cached")
    weights = torch.full((4,), float(rank))                  # stand-in for replicas that have diverged
    events = []

    def resync():
        dist.broadcast(weights, 0)
        events.append(step)
    red.on_resync = resync
    seen = []
    for step in range(12):
        skip = (i_none, i_half) if (step < 4 or rank != 1) else (i_none,)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _fill(gs, vt, red, float(rank + 1), skip=skip)
        seen.append(gs.params[i_half].grad is not None)
    out["cached_seen"] = seen
    out["cached_resync_steps"] = events
    out["cached_mode_after"] = red.find_unused
    out["cached_syncs"] = red.host_syncs
    out["weights_after"] = weights.tolist()
    # --- flush(): a pattern change in the LAST step of a run (nothing would check it LAG steps later) is settled before a checkpoint
    red = du.GradReducer(vt, find_unused="cached")
    weights2 = torch.full((4,), float(rank))
    ev2 = []

    def resync2():
        dist.broadcast(weights2, 0)
        ev2.append(1)
    red.on_resync = resync2
    for step in range(3):
        skip = (i_none, i_half) if (step < 2 or rank != 1) else (i_none,)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _fill(gs, vt, red, float(rank + 1), skip=skip)
    out["flush_before"] = len(ev2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        red.flush()
    out["flush_after"] = len(ev2)
    out["flush_weights"] = weights2.tolist()
    out["flush_mode"] = red.find_unused
    results[rank] = out
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_n_rank_gloo(world):
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), results), nprocs=world, join=True)
    steps = None
    for rank in range(world):
        r = results[rank]
        assert r["ag_rows"] == [float(q) for q in range(world) for _ in range(2)] and r["ag_bwd"], r
        assert r["all_gather"] == [float(q) for q in range(world)]
        for coll in ("allreduce", "rsag"):
            for comm in ("f32", "bf16"):
                assert r[f"sum_{coll}_{comm}"] == [r["want"]], (coll, comm, r[f"sum_{coll}_{comm}"], r["want"])
        assert r["spans_cached"]
        assert r["unused_none"] and r["half_used"] == [2.0], r
        # the changed pattern: every rank re-synchronises exactly once, at the SAME step, within 2 * LAG + 1 steps of the change,
        # and from then on every step's flags are read (the parameter has its gradient on every rank)
        assert len(r["cached_resync_steps"]) == 1, r["cached_resync_steps"]
        steps = steps or r["cached_resync_steps"]
        assert r["cached_resync_steps"] == steps and 4 < steps[0] <= 4 + 2 * 2 + 1, (steps, r["cached_resync_steps"])
        assert r["cached_mode_after"] == "sync"
        assert all(r["cached_seen"][steps[0]:]) and not any(r["cached_seen"][:4]), r["cached_seen"]
        assert r["weights_after"] == [0.0] * 4                        # rank 0's copy everywhere
        assert r["flush_before"] == 0 and r["flush_after"] == 1 and r["flush_weights"] == [0.0] * 4 and r["flush_mode"] == "sync", r
    assert results[1]["cached_seen"][4]                               # the rank that changed reads the new flags at once
