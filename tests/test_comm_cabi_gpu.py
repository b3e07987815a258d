"""pvrl_comm_* (csrc/comm.hip): the C-ABI collectives over RCCL.  One GPU here, so the communicator has one rank: a
1-rank sum all-reduce / all-gather must be the identity, issued on the caller's stream like every other entry point.
(The N-rank behaviour is RCCL's; the N-rank *use* of collectives is covered through torch.distributed in
test_distributed_gloo.py / test_two_rank_gloo_gpu.py / test_launcher_gpu.py.)"""
import ctypes

import pytest
import torch


@pytest.mark.gpu
def test_comm_world1_roundtrip():
    from procedurevrl_amd._lib import lib
    L = lib()
    torch.cuda.set_device(0)
    uid = ctypes.create_string_buffer(128)
    L.call("pvrl_comm_unique_id", ctypes.cast(uid, ctypes.c_void_p))
    assert any(b != 0 for b in uid.raw)
    comm = ctypes.c_void_p()
    L.call("pvrl_comm_init", ctypes.cast(ctypes.byref(comm), ctypes.c_void_p), 1, 0, ctypes.cast(uid, ctypes.c_void_p))
    assert comm.value
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        x = torch.randn(1 << 20, device="cuda")
        want = x.clone()
        stream = ctypes.c_void_p(st.cuda_stream)
        L.call("pvrl_comm_allreduce_f32", comm, ctypes.c_void_p(x.data_ptr()), x.numel(), stream)
        y = torch.empty(4096, device="cuda", dtype=torch.uint8)
        src = torch.randint(0, 255, (4096,), device="cuda", dtype=torch.uint8)
        L.call("pvrl_comm_allgather", comm, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(y.data_ptr()), 4096, stream)
        z = torch.randn(1 << 16, device="cuda")
        zr = torch.zeros_like(z)
        L.call("pvrl_comm_reducescatter_f32", comm, ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(zr.data_ptr()), z.numel(), stream)
    st.synchronize()
    assert torch.equal(x, want) and torch.equal(y, src) and torch.equal(z, zr)
    L.call("pvrl_comm_destroy", comm)
    # argument errors are status codes, not crashes
    with pytest.raises(Exception):
        L.call("pvrl_comm_init", None, 1, 0, ctypes.cast(uid, ctypes.c_void_p))
