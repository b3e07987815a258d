"""ctypes access to tools/probe/libpvrl_probe.so (the measured-and-rejected GEMM variants): same argument conventions as
procedurevrl_amd.ops.gemm_nt / gemm_tn plus an explicit `tile` selector.  Benchmark / check tooling only."""
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from procedurevrl_amd import ops  # noqa: E402
from procedurevrl_amd._lib import lib  # noqa: E402

_dll = None
c_vp, c_i64, c_int, c_f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


def dll():
    global _dll
    if _dll is None:
        import build_probe
        _dll = ctypes.CDLL(build_probe.build())
        _dll.pvrl_probe_gemm_nt_bf16.argtypes = [c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_vp,
                                                 c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]
        _dll.pvrl_probe_gemm_tn_plan_splits.restype = c_i64
        _dll.pvrl_probe_gemm_tn_plan_splits.argtypes = [c_int, c_i64, c_i64, c_i64]
        _dll.pvrl_probe_gemm_tn_workspace_bytes.restype = c_i64
        _dll.pvrl_probe_gemm_tn_workspace_bytes.argtypes = [c_i64, c_i64, c_i64]
        _dll.pvrl_probe_gemm_tn_bf16.argtypes = [c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_vp, c_vp,
                                                 c_vp, c_i64, c_vp]
    return _dll


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def gemm_nt(tile, A, W, epi, bias=None, rowscale=None, aux=None, aux_rowmod=0, bias2=None, gm=0):
    L = lib()
    M, K = A.shape
    N = W.shape[0]
    f32_out = epi in (L.PVRL_EPI_RESID_F32, L.PVRL_EPI_F32)
    out0 = torch.empty((M, N), device=A.device, dtype=torch.float32 if f32_out else A.dtype)
    two = epi in (L.PVRL_EPI_GELU, L.PVRL_EPI_QGELU)
    out1 = torch.empty((M, N), device=A.device, dtype=A.dtype) if two else None
    rc = dll().pvrl_probe_gemm_nt_bf16(tile, gm, _p(A), A.stride(0), _p(W), W.stride(0), M, N, K, epi, _p(bias), _p(rowscale),
                                       _p(aux), aux.stride(0) if aux is not None else 0, aux_rowmod, _p(out0), out0.stride(0),
                                       _p(out1), out1.stride(0) if out1 is not None else 0, _p(bias2), ops._stream())
    if rc != 0:
        raise RuntimeError(f"pvrl_probe_gemm_nt_bf16(tile={tile}) -> {rc}")
    return (out0, out1) if two else out0


def tn_splits(tile, M, N, K):
    return dll().pvrl_probe_gemm_tn_plan_splits(tile, M, N, K)


def gemm_tn(tile, P, Q, dW, dbias=None, beta=0.0, splits=None):
    M, N = P.shape
    K = Q.shape[1]
    if splits is None:
        splits = tn_splits(tile, M, N, K)
    nbytes = dll().pvrl_probe_gemm_tn_workspace_bytes(N, K, splits)
    ws = ops.workspace(nbytes, P.device, "probe_tn")
    rc = dll().pvrl_probe_gemm_tn_bf16(tile, _p(P), P.stride(0), _p(Q), Q.stride(0), M, N, K, splits, float(beta), _p(dW),
                                       _p(dbias), _p(ws), ws.numel(), ops._stream())
    if rc != 0:
        raise RuntimeError(f"pvrl_probe_gemm_tn_bf16(tile={tile}) -> {rc}")
    return dW
