"""The N > 1 training path on ONE GPU: two processes (gloo backend, both on cuda:0) run data-parallel steps through
GradReducer -- per-block all-reduce from the engine's gradient hook -- for enough steps that the encoder is captured into
HIP graphs and the backward runs as one graph PER BLOCK with the hook between replays (engine.GraphReplay, staged).  After
each step the all-reduced gradient buffer of rank 0 must equal the SUM of the two ranks' own single-process gradients, and
the two ranks must hold identical buffers.  (RCCL itself needs one GPU per rank: an 8-GPU node is only available to the
driver; this covers everything above the collective.)"""
import os
import socket

import pytest
import torch


def _worker(rank, world, port, out_dir, backend="gloo", grad_comm="f32", grad_coll="allreduce"):
    """backend "gloo": both ranks on cuda:0 (one GPU is enough); "nccl": REAL RCCL, rank r on cuda:r
    (tests/test_rccl_multi_gpu.py, needs >= 2 GPUs)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    import e2e_checks as ec
    from procedurevrl_amd import distributed as du
    from procedurevrl_amd.datasets import synthetic_label_emb
    from procedurevrl_amd.functional import kl_topk_loss
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    ec.DEV = dev
    if backend == "nccl":
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    cfg = ec.make_cfg(2, 32, 64)
    model = ec.build(cfg, synthetic_label_emb(64, 512, seed=1)).to(dev).train()
    vt = model.model
    with torch.no_grad():
        for blk in vt.blocks:
            torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
    g = torch.Generator(device=dev).manual_seed(100 + rank)                 # a different batch per rank
    x = torch.randn(4, 3, 8, 32, 32, device=dev, generator=g)
    teacher = torch.randn(4, 64, device=dev, generator=g) * 3

    def step(reducer):
        model.zero_grad(set_to_none=True)
        kl_topk_loss(model(x), teacher, 5).backward()
        if reducer is not None:
            reducer.finish()
        gs = vt.adopt_grads()
        return gs.flat[:gs.end].clone()

    own = step(None)                                                         # this rank's gradients, no communication
    reducer = du.GradReducer(vt, grad_comm=grad_comm, grad_coll=grad_coll)
    assert reducer.enabled and vt.engine.grad_hook is not None and dist.get_world_size() == world
    res = []
    for _ in range(vt.engine.GRAPH_WARMUP + 3):                              # eager, eager, capture (staged), replay, replay
        res.append(step(reducer))
    staged = any("bwd_staged" in g for g in vt.engine._graphs.values())
    torch.save(dict(own=own.cpu(), reduced=[r.cpu() for r in res], staged=staged, backend=dist.get_backend(),
                    device=str(x.device)), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("grad_coll", ["allreduce", "rsag"])
def test_two_rank_data_parallel_steps_with_staged_graphs(tmp_path, grad_coll):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), "gloo", "f32", grad_coll), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["staged"] and r1["staged"], "the backward was not captured as per-block graphs"
    want = r0["own"] + r1["own"]
    scale = float(want.abs().max())
    for k, (a, b) in enumerate(zip(r0["reduced"], r1["reduced"])):
        assert torch.equal(a, b), f"step {k}: ranks disagree after the all-reduce"
        err = float((a - want).abs().max()) / scale
        assert err <= 1e-6, f"step {k}: all-reduced gradients differ from the sum of the ranks' gradients ({err:.2e})"


@pytest.mark.gpu
def test_allgather_infonce_two_ranks_gloo(tmp_path):
    """The contrastive leg of configs[2] -- AllGather of the video / text embeddings + MILNCELoss on the global batch -- with two
    processes on ONE GPU over gloo: the worker and the checks of tests/test_rccl_multi_gpu.py::test_allgather_infonce_two_ranks (which
    runs the moment a box shows two GPUs), everything but the RCCL transport."""
    import sys
    import torch.multiprocessing as mp
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    import test_rccl_multi_gpu as tr
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(tr._infonce_worker, args=(2, port, str(tmp_path), "gloo"), nprocs=2, join=True)
    tr.check_infonce_results(tmp_path)
