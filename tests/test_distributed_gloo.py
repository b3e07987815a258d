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
