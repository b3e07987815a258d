"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path: AllGather autograd op against the
reference's own 2-rank run (tests/golden/allgather.pt), fused scalar all-reduce, and the chunked flat-gradient reducer."""
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


def _worker(rank, world, port, results):
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from procedurevrl_amd import distributed as du
    out = {}
    # --- AllGather forward / backward with the reference's inputs
    gold = torch.load(os.path.join(HERE, "golden", "allgather.pt"), weights_only=False)[rank]
    x = gold["x"].clone().requires_grad_(True)
    y = du.AllGather.apply(x)
    (y * gold["w"]).sum().backward()
    out["ag_fwd"] = bool(torch.equal(y.detach(), gold["y"]))
    out["ag_bwd"] = bool(torch.equal(x.grad, gold["grad"]))
    # --- list all_gather / all_reduce API
    a, = du.all_gather([torch.full((2, 3), float(rank))])
    out["all_gather"] = a[:, 0].tolist()
    r, = du.all_reduce([torch.tensor([1.0 + rank])])
    out["all_reduce_avg"] = float(r)
    v = du.all_reduce_scalars([torch.tensor(2.0 * rank), 4.0, torch.tensor(1.0)])
    out["scalars"] = v.tolist()
    # --- chunked gradient reducer on the flat buffer
    from test_host_logic import _small_model
    torch.manual_seed(0)
    _, model = _small_model(text=False)
    vt = model.model
    red = du.GradReducer(vt)
    gs = vt.grad_store()
    for p, view in zip(gs.params, gs.views):
        p.grad = view
    gs.flat.fill_(float(rank + 1))
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)                      # what engine.backward() calls after each block
    vt.head.weight.grad = torch.full_like(vt.head.weight, float(rank + 1))   # a gradient autograd allocated itself
    red.finish()
    used = torch.cat([v.reshape(-1) for v in gs.views])
    out["reducer_all_3"] = bool(torch.all(used == 3.0))
    out["head_adopted"] = vt.head.weight.grad.data_ptr() == gs.views[gs.index[id(vt.head.weight)]].data_ptr()
    # --- gradient accumulation (tools/train_net.py:176-192) with world > 1: block gradients accumulate locally over the
    # micro-iterations (reducer.sync = False silences the hook, like DDP.no_sync) and are summed over ranks exactly once
    for p in gs.params:
        p.grad = None
    for p, view in zip(gs.params, gs.views):
        p.grad = view
    micro = [1.0 + rank, 10.0 * (1 + rank), 100.0 * (1 + rank)]         # this rank's gradient of three micro-iterations
    gs.flat.zero_()
    for k, g in enumerate(micro):
        red.sync = k == len(micro) - 1
        gs.flat[:gs.end].add_(g)                                          # loss.backward() accumulating into .grad
        for i in reversed(range(len(vt.blocks))):
            vt.engine.grad_hook(i)
        if red.sync:
            red.finish()
    used = torch.cat([v.reshape(-1) for v in gs.views])
    out["accum_sum"] = sorted(set(used.tolist()))                         # (1+10+100) * (1 + 2) everywhere
    # --- unused parameters (DDP find_unused_parameters=True, lib/models/build.py:51): a parameter without a gradient on
    # every rank keeps .grad None; one used on rank 0 only gets the sum on both ranks
    for p in gs.params:
        p.grad = None
    names = gs.names
    i_none, i_half = names.index("time_embed"), names.index("cls_token")
    for k, (p, view) in enumerate(zip(gs.params, gs.views)):
        if k == i_none or (k == i_half and rank == 1):
            continue
        view.fill_(float(rank + 1))
        p.grad = view
    red.sync = True
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)
    red.finish()
    out["unused_none"] = gs.params[i_none].grad is None
    out["half_used"] = sorted(set(gs.params[i_half].grad.reshape(-1).tolist()))    # rank 0's 1.0 + rank 1's zeros
    out["others_3"] = all(bool(torch.all(p.grad == 3.0)) for k, p in enumerate(gs.params) if k not in (i_none, i_half))
    # --- a BLOCK parameter unused on one rank: its slot still holds the previous step's values (zero_grad
    # (set_to_none) never clears the flat buffer) and is sent by the per-block hook, before finish(): the hook must zero it
    for p in gs.params:
        p.grad = None
    gs.flat[:gs.end].fill_(777.0)                                        # stale values from "the previous step"
    i_blk = names.index("blocks.1.mlp.fc1.weight")
    for k, (p, view) in enumerate(zip(gs.params, gs.views)):
        if k == i_blk and rank == 1:
            continue
        view.fill_(float(rank + 1))
        p.grad = view
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)
    red.finish()
    out["blk_half_used"] = sorted(set(gs.params[i_blk].grad.reshape(-1).tolist()))   # rank 0's 1.0 + zeros, not + 777
    out["blk_others_3"] = all(bool(torch.all(p.grad == 3.0)) for k, p in enumerate(gs.params) if k != i_blk)
    # --- find_unused="cached": the reduced flags are read back once per local pattern, not once per step
    red_c = du.GradReducer(vt, find_unused="cached")
    for step in range(4):
        for p in gs.params:
            p.grad = None
        for k, (p, view) in enumerate(zip(gs.params, gs.views)):
            if k == i_none:
                continue
            view.fill_(float(rank + 1))
            p.grad = view
        for i in reversed(range(len(vt.blocks))):
            vt.engine.grad_hook(i)
        red_c.finish()
    out["cached_syncs"] = red_c.host_syncs
    out["cached_none"] = gs.params[i_none].grad is None and all(p.grad is not None for k, p in enumerate(gs.params) if k != i_none)
    # --- 16-bit gradient payload (PVRL_GRAD_COMM=bf16): cast, reduce, widen back
    red_h = du.GradReducer(vt, find_unused=False, grad_comm="bf16")
    for p, view in zip(gs.params, gs.views):
        p.grad = view
    gs.flat[:gs.end].fill_(0.375 * (rank + 1))                            # exactly representable: the sum must be exact
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)
    red_h.finish()
    out["bf16_comm"] = sorted(set(torch.cat([v.reshape(-1) for v in gs.views]).tolist()))
    du.GradReducer(vt)                                                    # (re-installs the default hook)
    # --- per-rank data sharding (lib/datasets/utils.py:358-370 DistributedSampler, loader.py:140-157 set_epoch)
    from procedurevrl_amd.config import get_cfg
    from procedurevrl_amd import datasets as ds
    cfg = get_cfg()
    cfg.NUM_GPUS, cfg.TRAIN.BATCH_SIZE, cfg.DATA.NUM_FRAMES, cfg.DATA.TRAIN_CROP_SIZE = 2, 4, 2, 16
    cfg.SYNTHETIC.NUM_VIDEOS = 8
    torch.manual_seed(cfg.RNG_SEED)
    loader = ds.construct_loader(cfg, "train")
    seen = []
    for epoch in range(2):
        ds.shuffle_dataset(loader, epoch)
        seen.append([int(i) for _, _, idx, _ in loader for i in idx])
    # --- the optimiser's "skip this step" flag (GradStore.bad) rides in the reduced tail: raised on ONE rank (a non-finite loss or
    #     gradient there), every rank sees it after finish() and drops the same step (tools/train_net.py:174 raises on that rank's loss;
    #     the all-reduced NaN gradients must not be applied by the others)
    red_b = du.GradReducer(vt, find_unused=False)
    for p, view in zip(gs.params, gs.views):
        p.grad = view
    gs.flat[:gs.end].fill_(1.0)
    gs.bad.fill_(1.0 if rank == 1 else 0.0)
    for i in reversed(range(len(vt.blocks))):
        vt.engine.grad_hook(i)
    red_b.finish()
    out["bad_after"] = float(gs.bad)
    gs.bad.zero_()
    out["seen"] = seen
    out["batch"] = loader.batch_size
    results[rank] = out
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo():
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    for rank in (0, 1):
        r = results[rank]
        assert r["ag_fwd"] and r["ag_bwd"], r
        assert r["all_gather"] == [0.0, 0.0, 1.0, 1.0]
        assert r["all_reduce_avg"] == 1.5
        assert r["scalars"] == [1.0, 4.0, 1.0]
        assert r["reducer_all_3"] and r["head_adopted"], r
        assert r["accum_sum"] == [333.0], r["accum_sum"]
        assert r["unused_none"] and r["half_used"] == [1.0] and r["others_3"], r
        assert r["blk_half_used"] == [1.0] and r["blk_others_3"], r
        assert r["cached_syncs"] == 1 and r["cached_none"], r
        assert r["bf16_comm"] == [1.125], r
        assert r["bad_after"] == 1.0, r["bad_after"]                  # both ranks drop the step that was bad on rank 1
        assert r["batch"] == 2
    for epoch in range(2):      # the two ranks see disjoint videos that together cover the dataset; epochs are shuffled differently
        a, b = results[0]["seen"][epoch], results[1]["seen"][epoch]
        assert not set(a) & set(b) and sorted(a + b) == list(range(8)), (a, b)
    assert results[0]["seen"][0] != results[0]["seen"][1]
