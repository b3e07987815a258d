"""REAL RCCL with more than one rank (one GPU per rank over xGMI) -- the tests the N > 1 path of BASELINE configs[2] waits
for.  They run whenever the box shows >= 2 GPUs and SKIP otherwise (the builder's gpurun boxes have one GPU; the gloo
variants of the same workers -- test_two_rank_gloo_gpu.py, test_launcher_gpu.py, test_distributed_gloo.py -- cover everything
above the collective there).  Reference: lib/utils/multiprocessing.py:49-58 (init_process_group per spawned process),
lib/models/build.py:49-53 (DDP), lib/utils/distributed.py:13-69."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                            reason="needs >= 2 GPUs (RCCL with one device per rank)")


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.gpu
@needs2
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` = the driver's launch (torch.distributed.run, one rank per GPU, backend nccl = RCCL)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PVRL_SINGLE_DEVICE", None); env.pop("PVRL_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-side", "--no-kernel-timing"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("RCCL", d["comm"])
    assert d["n_gpus"] == 2 and d["comm"]["backend"] == "nccl" and d["comm"]["ranks"] == 2 and d["comm"]["rccl"]
    assert d["config"]["global_batch"] == 8 and d["value"] > 0 and d["loss"] == d["loss"]


@pytest.mark.gpu
@needs2
@pytest.mark.parametrize("grad_comm,grad_coll", [("f32", "allreduce"), ("bf16", "allreduce"), ("f32", "rsag"), ("bf16", "rsag")])
def test_grad_reducer_two_ranks_rccl(tmp_path, grad_comm, grad_coll):
    """two ranks on two GPUs, five data-parallel steps (eager, captured, replayed with the per-block hook between the staged
    graphs): the reduced buffer equals the sum of the ranks' own gradients on both ranks -- through the all-reduce and through the
    in-place reduce-scatter + all-gather pair (PVRL_GRAD_COLL=rsag)"""
    import torch.multiprocessing as mp
    from test_two_rank_gloo_gpu import _worker
    mp.spawn(_worker, args=(2, _port(), str(tmp_path), "nccl", grad_comm, grad_coll), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["backend"] == "nccl" and {r0["device"], r1["device"]} == {"cuda:0", "cuda:1"}
    assert r0["staged"] and r1["staged"]
    want = r0["own"] + r1["own"]
    scale = float(want.abs().max())
    tol = 1e-6 if grad_comm == "f32" else 1.2e-2          # bf16 payload: each rank's chunk is rounded to 8 mantissa bits once
    for k, (a, b) in enumerate(zip(r0["reduced"], r1["reduced"])):
        assert torch.equal(a, b), f"step {k}: ranks disagree after the all-reduce"
        err = float((a - want).abs().max()) / scale
        assert err <= tol, f"step {k}: all-reduced gradients differ from the sum of the ranks' gradients ({err:.2e})"


def _comm_worker(rank, world, uid_path, out_dir):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import time
    from procedurevrl_amd._lib import lib
    L = lib()
    torch.cuda.set_device(rank)
    uid = ctypes.create_string_buffer(128)
    if rank == 0:
        L.call("pvrl_comm_unique_id", ctypes.cast(uid, ctypes.c_void_p))
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        for _ in range(600):
            if os.path.exists(uid_path):
                break
            time.sleep(0.1)
        uid = ctypes.create_string_buffer(open(uid_path, "rb").read(), 128)
    comm = ctypes.c_void_p()
    L.call("pvrl_comm_init", ctypes.cast(ctypes.byref(comm), ctypes.c_void_p), world, rank, ctypes.cast(uid, ctypes.c_void_p))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        x = torch.full((1 << 20,), float(rank + 1), device="cuda")
        stream = ctypes.c_void_p(st.cuda_stream)
        L.call("pvrl_comm_allreduce_f32", comm, ctypes.c_void_p(x.data_ptr()), x.numel(), stream)
        src = torch.full((4096,), rank + 7, device="cuda", dtype=torch.uint8)
        y = torch.empty(4096 * world, device="cuda", dtype=torch.uint8)
        L.call("pvrl_comm_allgather", comm, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(y.data_ptr()), 4096, stream)
        # reduce-scatter in place (recv = this rank's shard of send), then all-gather of the shards = the two-step all-reduce
        z = torch.arange(world * 1024, device="cuda", dtype=torch.float32) * (rank + 1)
        shard = z[rank * 1024:(rank + 1) * 1024]
        L.call("pvrl_comm_reducescatter_f32", comm, ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(shard.data_ptr()), 1024, stream)
        L.call("pvrl_comm_allgather", comm, ctypes.c_void_p(shard.data_ptr()), ctypes.c_void_p(z.data_ptr()), 4096, stream)
    st.synchronize()
    ok = bool(torch.all(x == sum(range(1, world + 1)))) and all(bool(torch.all(y[4096 * r:4096 * (r + 1)] == r + 7)) for r in range(world))
    ok = ok and bool(torch.equal(z.cpu(), torch.arange(world * 1024, dtype=torch.float32) * sum(range(1, world + 1))))
    L.call("pvrl_comm_destroy", comm)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write(str(ok))


@pytest.mark.gpu
@needs2
def test_pvrl_comm_cabi_world2(tmp_path):
    """pvrl_comm_* (include/pvrl.h, csrc/comm.hip) with two ranks: sum all-reduce and all-gather over RCCL"""
    import torch.multiprocessing as mp
    mp.spawn(_comm_worker, args=(2, str(tmp_path / "uid"), str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "True" and open(tmp_path / "ok1").read() == "True"


def _infonce_worker(rank, world, port, out_dir, backend="nccl"):
    """`backend` = "gloo": both ranks on device 0 (tests/test_two_rank_gloo_gpu.py runs it on the 1-GPU boxes)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from procedurevrl_amd import distributed as du
    from procedurevrl_amd.losses import MILNCELoss
    out = {}
    # AllGather forward / backward on the reference's own 2-rank inputs (tests/golden/allgather.pt), now over RCCL
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "allgather.pt"), weights_only=False)[rank]
    x = gold["x"].to(dev).requires_grad_(True)
    y = du.AllGather.apply(x)
    (y * gold["w"].to(dev)).sum().backward()
    out["ag_fwd"] = bool(torch.equal(y.detach().cpu(), gold["y"]))
    out["ag_bwd"] = bool(torch.equal(x.grad.cpu(), gold["grad"]))
    # the contrastive leg: each rank holds half of the videos / candidate texts of tests/golden/milnce.pt, gathers the other half
    # (forward all-gather, backward = the local slice: lib/utils/distributed.py:13-29) and evaluates MILNCELoss on the global batch
    m = torch.load(os.path.join(ROOT, "tests", "golden", "milnce.pt"), weights_only=False)
    n, nt = m["v"].shape[0] // world, m["t"].shape[0] // world
    v = m["v"][rank * n:(rank + 1) * n].to(dev).requires_grad_(True)
    t = m["t"][rank * nt:(rank + 1) * nt].to(dev).requires_grad_(True)
    loss = MILNCELoss()(du.AllGather.apply(v), du.AllGather.apply(t))
    loss.backward()
    out.update(loss=float(loss), dv=v.grad.cpu(), dt=t.grad.cpu())
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@needs2
def test_allgather_infonce_two_ranks(tmp_path):
    """AllGather + MILNCELoss at world 2 over RCCL: the gathered tensors and the slice-backward equal the reference's own 2-rank run
    (tests/golden/allgather.pt), the global InfoNCE loss equals the reference's value on the whole batch (tests/golden/milnce.pt) on
    both ranks, and each rank's dV / dT is ITS slice of the single-process gradient (oracle, fp32 CPU)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from oracle import timesformer_oracle as orc
    mp.spawn(_infonce_worker, args=(2, _port(), str(tmp_path)), nprocs=2, join=True)
    check_infonce_results(tmp_path)


def check_infonce_results(tmp_path):
    from oracle import timesformer_oracle as orc
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]
    m = torch.load(os.path.join(ROOT, "tests", "golden", "milnce.pt"), weights_only=False)
    v, t = m["v"].clone().requires_grad_(True), m["t"].clone().requires_grad_(True)
    ref = orc.milnce(v, t)
    ref.backward()
    assert abs(float(ref) - m["loss"]) <= 1e-5 * abs(m["loss"])
    n, nt = v.shape[0] // 2, t.shape[0] // 2
    for k in range(2):
        assert r[k]["ag_fwd"] and r[k]["ag_bwd"]
        assert abs(r[k]["loss"] - m["loss"]) <= 1e-5 * abs(m["loss"]), (k, r[k]["loss"], m["loss"])
        for got, want in ((r[k]["dv"], v.grad[k * n:(k + 1) * n]), (r[k]["dt"], t.grad[k * nt:(k + 1) * nt])):
            assert float((got - want).norm() / want.norm()) <= 1e-4
