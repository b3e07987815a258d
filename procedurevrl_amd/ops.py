"""Tensor-level wrappers over the C ABI (include/pvrl.h).

PyTorch is used for device memory and streams only: every function here takes CUDA(ROCm)
tensors, passes raw pointers / leading dimensions to libpvrl_hip.so on torch's current
stream and returns tensors.  No function has a PyTorch compute fallback.
"""
import ctypes

import torch

from ._lib import lib, PvrlError, operand_torch_dtype

OP16 = operand_torch_dtype()     # the library flavour's 16-bit operand type: torch.bfloat16 (default) or torch.float16
F32 = torch.float32

# When set to a list, gemm launches are bracketed with events on the launch stream (bench.py roofline leg).
KERNEL_TIMING = None
_EPI_NAMES = {0: "op16", 1: "gelu", 2: "qgelu", 3: "resid_f32", 4: "f32", 5: "dgelu", 6: "dqgelu", 7: "resid_16"}


def _timed(name, flops, fn):
    if KERNEL_TIMING is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    KERNEL_TIMING.append((name, flops, e0, e1))
    return r


def collect_kernel_timing():
    """-> {"roofline": {...dominant kernel...}, "summary": {name: {calls, ms, tflops}}}"""
    if not KERNEL_TIMING:
        return None
    torch.cuda.synchronize()
    agg = {}
    for name, flops, e0, e1 in KERNEL_TIMING:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += flops
    summary = {k: {"calls": v[0], "ms": round(v[1], 3), "avg_us": round(1e3 * v[1] / v[0], 2),
                   "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else 0.0} for k, v in agg.items()}
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])

    def roof_of(item):
        name, (calls, ms, flops) = item
        ach = flops / (ms * 1e-3) / 1e12
        return {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                "frac": round(ach / 2500.0, 4), "traffic": None, "calls": calls, "avg_launch_us": round(1e3 * ms / calls, 2),
                "flops_per_launch_avg": flops / calls}
    roof = roof_of(order[0])
    if len(order) > 1:          # the two GEMM families of the step are within a few per cent of each other: report the second one too
        roof["runner_up"] = roof_of(order[1])
    return {"roofline": roof, "summary": summary}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = torch._C._cuda_getCurrentRawStream     # (device index) -> hipStream_t as int; ~20x cheaper than
_cur_device = torch._C._cuda_getDevice                # torch.cuda.current_stream().cuda_stream (8 us per call)


def _stream():
    return ctypes.c_void_p(_raw_stream(_cur_device()))


def _chk2d(t, dtype=None):
    if t.dim() != 2 or (t.stride(1) != 1 and t.shape[1] != 1):
        raise PvrlError(f"expected a row-major 2-D tensor, got shape {tuple(t.shape)} stride {t.stride()}")
    if not t.is_cuda:
        raise PvrlError("the HIP path needs device tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise PvrlError(f"expected dtype {dtype}, got {t.dtype}")
    return t


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ----------------------------------------------------------------------------------------
# GEMMs
# ----------------------------------------------------------------------------------------
def gemm_nt(A, W, epi, bias=None, rowscale=None, aux=None, aux_rowmod=0, out0=None, out1=None, bias2=None):
    """epilogue(A[M,K] @ W[N,K]^T).  Returns out0 (and out1 for the GELU epilogues)."""
    L = lib()
    _chk2d(A, OP16); _chk2d(W, OP16)
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K
    f32_out = epi in (L.PVRL_EPI_RESID_F32, L.PVRL_EPI_F32)
    if out0 is None:
        out0 = torch.empty((M, N), device=A.device, dtype=F32 if f32_out else OP16)
    if out0.dtype != (F32 if f32_out else OP16) or (epi == L.PVRL_EPI_RESID_16 and aux.dtype != (F32 if aux_rowmod else OP16)):
        raise PvrlError(f"gemm_nt epilogue {_EPI_NAMES[epi]}: out0 {out0.dtype}, aux {None if aux is None else aux.dtype}")
    two = epi in (L.PVRL_EPI_GELU, L.PVRL_EPI_QGELU)
    if two and out1 is None:
        out1 = torch.empty((M, N), device=A.device, dtype=OP16)
    # (few-row problems run gemm_nt_skinny_kernel, csrc/gemm_nt_skinny.h: their own family in the bench's per-kernel timing)
    fam = "gemm_nt_skinny<" if (M <= 192 and K % 256 == 0) else "gemm_nt_kernel<"
    _timed(fam + _EPI_NAMES[epi] + ">", 2.0 * M * N * K, lambda: L.call(
        "pvrl_gemm_nt_bf16", _ptr(A), _ld(A), _ptr(W), _ld(W), M, N, K, epi, _ptr(bias), _ptr(rowscale),
        _ptr(aux), _ld(aux) if aux is not None else 0, aux_rowmod, _ptr(out0), _ld(out0),
        _ptr(out1), _ld(out1) if out1 is not None else 0, _ptr(bias2), _stream()))
    return (out0, out1) if two else out0


def gemm_nt_batched(problems, epi):
    """problems: list of dicts (A, W[, bias, rowscale, aux, out0]) -> list of out0; one launch per 12 problems (pvrl_gemm_nt_batched_bf16):
    out0 = [aux +] rowscale * (A W^T + bias) with the 128 x 128-tile kernel.  `epi`: PVRL_EPI_BF16 | PVRL_EPI_F32 | PVRL_EPI_RESID_F32."""
    from ._lib import NtProblem
    L = lib()
    if not problems:
        return []
    f32_out = epi in (L.PVRL_EPI_RESID_F32, L.PVRL_EPI_F32)
    arr = (NtProblem * len(problems))()
    outs, flops = [], 0.0
    for a, pr in zip(arr, problems):
        A, W = _chk2d(pr["A"], OP16), _chk2d(pr["W"], OP16)
        M, K = A.shape
        N = W.shape[0]
        assert W.shape[1] == K
        out0 = pr.get("out0")
        if out0 is None:
            out0 = torch.empty((M, N), device=A.device, dtype=F32 if f32_out else OP16)
        aux, bias, rs = pr.get("aux"), pr.get("bias"), pr.get("rowscale")
        a.A, a.lda, a.W, a.ldw, a.M, a.N, a.K = A.data_ptr(), _ld(A), W.data_ptr(), _ld(W), M, N, K
        a.bias = None if bias is None else bias.data_ptr()
        a.rowscale = None if rs is None else rs.data_ptr()
        a.aux, a.aux_ld = (None, 0) if aux is None else (aux.data_ptr(), _ld(aux))
        a.out0, a.ld0 = out0.data_ptr(), _ld(out0)
        outs.append(out0)
        flops += 2.0 * M * N * K
    _timed("gemm_nt_batched<" + _EPI_NAMES[epi] + ">", flops, lambda: L.call(
        "pvrl_gemm_nt_batched_bf16", len(problems), ctypes.addressof(arr), epi, _stream()))
    return outs


def gemm_nt_f32(A, B, bias=None, alpha=1.0, out=None):
    L = lib()
    _chk2d(A, F32); _chk2d(B, F32)
    M, K = A.shape
    N = B.shape[0]
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=F32)
    nbytes = L.call("pvrl_gemm_nt_f32_small_workspace_bytes", M, N, K)
    ws = workspace(nbytes, A.device, "f32_small") if nbytes else None
    L.call("pvrl_gemm_nt_f32_small", _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(bias), float(alpha), _ptr(out), _ld(out),
           M, N, K, _ptr(ws), nbytes, _stream())
    return out


def cls_linear(X, W, bias=None, gelu=False, rowscale=None, biasscale=None, aux=None, out=None):
    """The cls rows' fp32 chain (pvrl_cls_linear_f32): out = aux + rowscale * (X W^T) + biasscale * bias, or GELU(X W^T + bias).
    X fp32 [M, K], W fp32 [N, K] (the MASTER weight, not its 16-bit operand copy)."""
    L = lib()
    _chk2d(X, F32); _chk2d(W, F32)
    M, K = X.shape
    N = W.shape[0]
    assert W.shape[1] == K
    if out is None:
        out = torch.empty((M, N), device=X.device, dtype=F32)
    nbytes = L.call("pvrl_cls_linear_f32_workspace_bytes", M, N, K)
    ws = workspace(nbytes, X.device, "cls_part") if nbytes else None
    L.call("pvrl_cls_linear_f32", _ptr(X), _ld(X), _ptr(W), _ld(W), _ptr(bias), M, N, K, 1 if gelu else 0, _ptr(rowscale),
           _ptr(biasscale), _ptr(aux), _ld(aux) if aux is not None else 0, _ptr(out), _ld(out), None, None, 0, _ptr(ws),
           ws.numel() if ws is not None else 0, _stream())
    return out


_ws_cache = {}


_ws_retired = []     # outgrown workspaces stay allocated: a captured HIP graph may have their address baked in


def workspace(nbytes, device, tag="default"):
    key = (tag, str(device))
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        if w is not None:
            _ws_retired.append(w)
            nbytes = max(int(nbytes), w.numel() * 3 // 2)      # geometric growth bounds what is retired
        w = torch.empty(max(int(nbytes), 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = w
    return w


def tn_splits(M, N, K):
    """Slices of M for the weight-gradient GEMM (the C side owns the rule: it depends on which kernel the shape selects)."""
    return lib().call("pvrl_gemm_tn_plan_splits", M, N, K)


def gemm_tn(P, Q, dW, dbias=None, beta=0.0, splits=None, ws_tag="tn", gscale=None, nonfinite=None):
    """dW[N,K] = beta*dW + gscale * P[M,N]^T @ Q[M,K]; dbias = beta*dbias + gscale * colsum(P).  gscale (device scalar tensor or
    None = 1) / nonfinite (device flag tensor or None): the common tail of the gradient-writing entry points, include/pvrl.h."""
    L = lib()
    _chk2d(P, OP16); _chk2d(Q, OP16)
    M, N = P.shape
    K = Q.shape[1]
    assert Q.shape[0] == M and dW.shape == (N, K) and dW.is_contiguous() and dW.dtype == F32
    if splits is None:
        splits = tn_splits(M, N, K)
    nbytes = L.call("pvrl_gemm_tn_workspace_bytes", N, K, splits)
    ws = workspace(nbytes, P.device, ws_tag)
    _timed("gemm_tn_kernel+reduce", 2.0 * M * N * K, lambda: L.call(
        "pvrl_gemm_tn_bf16", _ptr(P), _ld(P), _ptr(Q), _ld(Q), M, N, K, splits, float(beta), _ptr(dW), _ptr(dbias),
        _ptr(ws), ws.numel(), _ptr(gscale), _ptr(nonfinite), _stream()))
    return dW


def gemm_tn_into(P, Q, dW, n_valid, k_valid, dbias=None, beta=0.0, beta_bias=0.0, ws_tag="tn", gscale=None, nonfinite=None):
    """dW[:n_valid, :k_valid] (fp32, row stride dW.stride(0)) = beta*dW + (P^T Q)[:n_valid, :k_valid] for zero-padded
    operands P [M, Np], Q [M, Kp]; dbias[:n_valid] likewise with beta_bias."""
    L = lib()
    _chk2d(P, OP16); _chk2d(Q, OP16)
    M, N = P.shape
    K = Q.shape[1]
    assert Q.shape[0] == M and dW.dtype == F32 and dW.dim() == 2 and dW.stride(1) == 1
    assert dW.shape[0] >= n_valid and dW.shape[1] >= k_valid and n_valid <= N and k_valid <= K
    assert dbias is None or (dbias.dtype == F32 and dbias.is_contiguous() and dbias.numel() >= n_valid)
    splits = tn_splits(M, N, K)
    ws = workspace(L.call("pvrl_gemm_tn_workspace_bytes", N, K, splits), P.device, ws_tag)
    _timed("gemm_tn_kernel+reduce", 2.0 * M * N * K, lambda: L.call(
        "pvrl_gemm_tn_into_bf16", _ptr(P), _ld(P), _ptr(Q), _ld(Q), M, N, K, splits, float(beta), _ptr(dW), dW.stride(0),
        n_valid, k_valid, _ptr(dbias), float(beta_bias), _ptr(ws), ws.numel(), _ptr(gscale), _ptr(nonfinite), _stream()))
    return dW


TN_GROUP_MAX = 8


def gemm_tn_grouped(problems, ws_tag="tn_group"):
    """problems: list of (P, Q, dW, dbias | None, beta[, gscale | None, nonfinite | None]), each as gemm_tn -- all issued as ONE
    grouped launch when every N and K is a multiple of 256 (pvrl_gemm_tn_grouped_bf16), otherwise one gemm_tn per problem."""
    from ._lib import TnProblem
    L = lib()
    if not problems:
        return
    problems = [tuple(pr) + (None,) * (7 - len(pr)) for pr in problems]
    ok = 1 < len(problems) <= TN_GROUP_MAX and all(
        pr[0].shape[1] % 256 == 0 and pr[1].shape[1] % 256 == 0 and pr[0].shape[0] >= 1 for pr in problems)
    if not ok:
        for P, Q, dW, dbias, beta, gsc, nf in problems:
            gemm_tn(P, Q, dW, dbias, beta=beta, ws_tag=ws_tag, gscale=gsc, nonfinite=nf)
        return
    arr = (TnProblem * len(problems))()
    flops = 0.0
    for a, (P, Q, dW, dbias, beta, gsc, nf) in zip(arr, problems):
        _chk2d(P, OP16); _chk2d(Q, OP16)
        M, N = P.shape
        K = Q.shape[1]
        assert Q.shape[0] == M and dW.shape == (N, K) and dW.is_contiguous() and dW.dtype == F32
        assert dbias is None or (dbias.dtype == F32 and dbias.is_contiguous() and dbias.numel() == N)
        a.P, a.ldp, a.Q, a.ldq, a.M, a.N, a.K = P.data_ptr(), _ld(P), Q.data_ptr(), _ld(Q), M, N, K
        a.beta, a.dW, a.dbias = float(beta), dW.data_ptr(), (None if dbias is None else dbias.data_ptr())
        a.gscale, a.nonfinite = (None if gsc is None else gsc.data_ptr()), (None if nf is None else nf.data_ptr())
        flops += 2.0 * M * N * K
    ap = ctypes.addressof(arr)
    splits = L.call("pvrl_gemm_tn_grouped_plan_splits", len(problems), ap)
    nbytes = L.call("pvrl_gemm_tn_grouped_workspace_bytes", len(problems), ap, splits)
    ws = workspace(nbytes, problems[0][0].device, ws_tag)
    _timed("gemm_tn_grouped+reduce", flops, lambda: L.call(
        "pvrl_gemm_tn_grouped_bf16", len(problems), ap, splits, _ptr(ws), ws.numel(), _stream()))


# ----------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------
class SplitRows:
    """A matrix whose first rows live in a 16-bit tensor `lo` [rows16, C] and whose remaining rows in an fp32 tensor `hi` [M - rows16, C]
    (`pvrl_rows`, include/pvrl.h): a stage of the encoder's split residual stream -- patch rows 16-bit, cls rows fp32.  Either part may
    be None where an entry point allows it (a dx_in part that is known to be zero)."""
    __slots__ = ("lo", "hi", "n_lo", "n_hi")

    def __init__(self, lo, hi, n_lo=None, n_hi=None):
        self.lo, self.hi = lo, hi
        self.n_lo = lo.shape[0] if lo is not None else int(n_lo or 0)
        self.n_hi = hi.shape[0] if hi is not None else int(n_hi or 0)
        if lo is not None:
            _chk2d(lo, OP16)
        if hi is not None:
            _chk2d(hi, F32)

    @property
    def shape(self):
        t = self.lo if self.lo is not None else self.hi
        return (self.n_lo + self.n_hi, t.shape[1])

    @property
    def device(self):
        return (self.lo if self.lo is not None else self.hi).device

    def c_rows(self):
        from ._lib import Rows
        r = Rows()
        r.lo, r.ldlo = (self.lo.data_ptr(), _ld(self.lo)) if self.lo is not None else (None, 0)
        r.hi, r.ldhi = (self.hi.data_ptr(), _ld(self.hi)) if self.hi is not None else (None, 0)
        r.rows16 = self.n_lo
        return r


def layernorm_fwd(x, gamma, beta, eps, out_dtype=OP16, out=None, save_stats=True, stats=None):
    """`stats` (optional): (mean, rstd) fp32 [M] tensors to write the row statistics into.  `x`: fp32 [M, C], or a SplitRows."""
    L = lib()
    if isinstance(x, SplitRows):
        M, C = x.shape
        if out is None:
            out = torch.empty((M, C), device=x.device, dtype=out_dtype)
        mean = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
        rstd = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
        L.call("pvrl_layernorm_fwd_split", _ptr(x.lo), _ld(x.lo) if x.lo is not None else 0, x.n_lo, _ptr(x.hi),
               _ld(x.hi) if x.hi is not None else 0, _ptr(gamma), _ptr(beta), float(eps), _ptr(out), _ld(out),
               1 if out.dtype == F32 else 0, _ptr(mean), _ptr(rstd), M, C, _stream())
        return out, mean, rstd
    _chk2d(x, F32)
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=out_dtype)
    if stats is not None:
        mean, rstd = stats
        assert mean.dtype == F32 and rstd.dtype == F32 and mean.is_contiguous() and rstd.is_contiguous() and mean.numel() == M == rstd.numel()
    else:
        mean = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
        rstd = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
    L.call("pvrl_layernorm_fwd", _ptr(x), _ld(x), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), _ld(out),
           1 if out.dtype == F32 else 0, _ptr(mean), _ptr(rstd), M, C, _stream())
    return out, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dgamma, dbeta, dx_in=None, dx_out=None, beta_acc=0.0, dxs=None, dxs_scale=None,
                  dxsum=None, dxsum_beta=None, gscale=None, nonfinite=None, defer=None):
    """`dxs` (optional bf16 [rows <= M, C]) additionally receives bf16(dxs_scale[m] * dx_out[m]); `dxsum` (optional fp32
    [C]) the unscaled column sums of those rows of dx_out (dxsum = dxsum_beta * dxsum + sums).
    `defer` (a list): the per-workgroup partial sums of dgamma / dbeta / dxsum stay in a private workspace and an entry is appended
    for `layernorm_bwd_reduce_batched`, which reduces many LayerNorms' partials in one launch."""
    L = lib()
    if isinstance(x, SplitRows):
        return _layernorm_bwd_split(dy, x, mean, rstd, gamma, dgamma, dbeta, dx_in, dx_out, beta_acc, dxs, dxs_scale, dxsum,
                                    dxsum_beta, gscale, nonfinite, defer)
    _chk2d(dy); _chk2d(x, F32)
    M, C = x.shape
    if dx_out is None:
        dx_out = torch.empty((M, C), device=x.device, dtype=F32)
    nbytes = L.call("pvrl_layernorm_bwd_workspace_bytes", M, C)
    if defer is not None:
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        assert dxsum is None or dxs is not None
        L.call("pvrl_layernorm_bwd", _ptr(dy), _ld(dy), 1 if dy.dtype == F32 else 0, _ptr(x), _ld(x), _ptr(mean),
               _ptr(rstd), _ptr(gamma), _ptr(dx_in), _ld(dx_in) if dx_in is not None else 0, _ptr(dx_out), _ld(dx_out),
               float(beta_acc), None, None, _ptr(ws), ws.numel(), M, C, _ptr(dxs), _ld(dxs) if dxs is not None else 0,
               _ptr(dxs_scale), dxs.shape[0] if dxs is not None else 0, _ptr(dxsum), None, None, _stream())
        defer.append(dict(part=ws, M=M, C=C, beta=float(beta_acc), beta_sum=float(beta_acc if dxsum_beta is None else dxsum_beta),
                          dgamma=dgamma, dbeta=dbeta, dxsum=dxsum))
        return dx_out
    ws = workspace(nbytes, x.device, "ln")
    tgt = dxsum
    if dxsum is not None:
        assert dxs is not None and dxsum.dtype == F32 and dxsum.is_contiguous() and dxsum.numel() == C
        if (beta_acc if dxsum_beta is None else dxsum_beta) != beta_acc:      # the kernel has one beta for all three sums
            tgt = torch.empty_like(dxsum)
    L.call("pvrl_layernorm_bwd", _ptr(dy), _ld(dy), 1 if dy.dtype == F32 else 0, _ptr(x), _ld(x), _ptr(mean),
           _ptr(rstd), _ptr(gamma), _ptr(dx_in), _ld(dx_in) if dx_in is not None else 0, _ptr(dx_out), _ld(dx_out),
           float(beta_acc), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), M, C, _ptr(dxs),
           _ld(dxs) if dxs is not None else 0, _ptr(dxs_scale), dxs.shape[0] if dxs is not None else 0, _ptr(tgt),
           _ptr(gscale), _ptr(nonfinite), _stream())
    if tgt is not dxsum:
        if beta_acc != 0.0:
            raise PvrlError("layernorm_bwd: dxsum_beta = 0 with beta_acc != 0 is not supported")
        dxsum.mul_(float(dxsum_beta)).add_(tgt)
    return dx_out


def _layernorm_bwd_split(dy, x, mean, rstd, gamma, dgamma, dbeta, dx_in, dx_out, beta_acc, dxs, dxs_scale, dxsum, dxsum_beta,
                         gscale, nonfinite, defer):
    """layernorm_bwd over a split matrix (pvrl_layernorm_bwd_split): x, dx_in (optional; a None part reads as zeros) and dx_out are
    SplitRows with the same split"""
    L = lib()
    _chk2d(dy)
    M, C = x.shape
    assert isinstance(dx_out, SplitRows) and dx_out.n_lo == x.n_lo and (dx_in is None or (isinstance(dx_in, SplitRows) and dx_in.n_lo == x.n_lo))
    xr, dor = x.c_rows(), dx_out.c_rows()
    dir_ = dx_in.c_rows() if dx_in is not None else None
    nbytes = L.call("pvrl_layernorm_bwd_workspace_bytes", M, C)
    deferred = defer is not None
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if deferred else workspace(nbytes, x.device, "ln")
    tgt = dxsum
    if dxsum is not None:
        assert dxs is not None and dxsum.dtype == F32 and dxsum.is_contiguous() and dxsum.numel() == C
        if not deferred and (beta_acc if dxsum_beta is None else dxsum_beta) != beta_acc:      # the kernel has one beta for all three sums
            if beta_acc != 0.0:
                raise PvrlError("layernorm_bwd: dxsum_beta = 0 with beta_acc != 0 is not supported")
            tgt = torch.empty_like(dxsum)
    L.call("pvrl_layernorm_bwd_split", _ptr(dy), _ld(dy), 1 if dy.dtype == F32 else 0, ctypes.addressof(xr), _ptr(mean), _ptr(rstd),
           _ptr(gamma), ctypes.addressof(dir_) if dir_ is not None else None, ctypes.addressof(dor), float(beta_acc),
           None if deferred else _ptr(dgamma), None if deferred else _ptr(dbeta), _ptr(ws), ws.numel(), M, C, _ptr(dxs),
           _ld(dxs) if dxs is not None else 0, _ptr(dxs_scale), dxs.shape[0] if dxs is not None else 0, _ptr(tgt),
           None if deferred else _ptr(gscale), None if deferred else _ptr(nonfinite), _stream())
    if tgt is not dxsum:
        dxsum.mul_(float(dxsum_beta)).add_(tgt)
    if deferred:
        defer.append(dict(part=ws, M=M, C=C, beta=float(beta_acc), beta_sum=float(beta_acc if dxsum_beta is None else dxsum_beta),
                          dgamma=dgamma, dbeta=dbeta, dxsum=dxsum))
    return dx_out


def layernorm_bwd_reduce_batched(items, gscale=None, nonfinite=None):
    """items: the entries `layernorm_bwd(..., defer=items)` appended -> dgamma / dbeta / dxsum of all of them in one launch"""
    from ._lib import LnReduce
    if not items:
        return
    arr = (LnReduce * len(items))()
    for a, it in zip(arr, items):
        a.part, a.M, a.C = it["part"].data_ptr(), it["M"], it["C"]
        a.want_sum = 0 if it["dxsum"] is None else 1
        a.beta, a.beta_sum = it["beta"], it["beta_sum"]
        a.dgamma, a.dbeta = it["dgamma"].data_ptr(), it["dbeta"].data_ptr()
        a.dxsum = None if it["dxsum"] is None else it["dxsum"].data_ptr()
    lib().call("pvrl_layernorm_bwd_reduce_batched", len(items), ctypes.addressof(arr), _ptr(gscale), _ptr(nonfinite), _stream())


# ----------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------
def attn_t8_fwd(qkv, nseq, H, scale, out=None):
    L = lib()
    _chk2d(qkv, OP16)
    if out is None:
        out = torch.empty((nseq * 8, H * 64), device=qkv.device, dtype=OP16)
    L.call("pvrl_attn_t8_fwd", _ptr(qkv), _ld(qkv), nseq, H, float(scale), _ptr(out), _ld(out), _stream())
    return out


def attn_t8_bwd(qkv, d_o, nseq, H, scale, dqkv=None):
    L = lib()
    _chk2d(qkv, OP16); _chk2d(d_o, OP16)
    if dqkv is None:
        dqkv = torch.empty((nseq * 8, 3 * H * 64), device=qkv.device, dtype=OP16)
    L.call("pvrl_attn_t8_bwd", _ptr(qkv), _ld(qkv), nseq, H, float(scale), _ptr(d_o), _ld(d_o), _ptr(dqkv), _ld(dqkv),
           _stream())
    return dqkv


def attn_fwd(qkv, nseq, S, H, scale, mode=0, T=1, cls_base=0, causal=False, kpm=None, o=None, o_cls=None, lse=None):
    """o / o_cls must share a leading dimension (o_cls may be a row-slice of the same buffer)."""
    L = lib()
    _chk2d(qkv, OP16)
    if o is None:
        o = torch.empty((nseq * S if mode == 0 else cls_base, H * 64), device=qkv.device, dtype=OP16)
    if mode == 1 and o_cls is None:
        o_cls = torch.empty((nseq, H * 64), device=qkv.device, dtype=OP16)
    if lse is None:
        lse = torch.empty((nseq, H, S), device=qkv.device, dtype=F32)
    if o_cls is not None:
        assert _ld(o_cls) == _ld(o)
    L.call("pvrl_attn_fwd", _ptr(qkv), _ld(qkv), nseq, S, H, mode, T, cls_base, float(scale), 1 if causal else 0,
           _ptr(kpm), _ptr(o), _ptr(o_cls), _ld(o), _ptr(lse), _stream())
    return o, o_cls, lse


def attn_cls_fwd(qkv, nseq, S, H, scale, T, cls_base, o_cls=None, lse=None):
    """spatial attention (mode 1 addressing) for the cls query of every sequence only -> (o_cls [nseq, H*64], lse [nseq, H, S] with entry 0
    of every (sequence, head) valid): the encoder's last block (pvrl_attn_cls_fwd)"""
    _chk2d(qkv, OP16)
    if o_cls is None:
        o_cls = torch.empty((nseq, H * 64), device=qkv.device, dtype=OP16)
    if lse is None:
        lse = torch.empty((nseq, H, S), device=qkv.device, dtype=F32)
    lib().call("pvrl_attn_cls_fwd", _ptr(qkv), _ld(qkv), nseq, S, H, T, cls_base, float(scale), _ptr(o_cls), _ld(o_cls), _ptr(lse),
               _stream())
    return o_cls, lse


def attn_cls_bwd(qkv, o_cls, d_o_cls, lse, nseq, S, H, scale, T, cls_base, dqkv, dqkv_cls, zero_patch_dq=True):
    """backward of attn_cls_fwd: dK / dV of every token, the cls token's partial rows in dqkv_cls; the patch tokens' dQ (zero) is written
    only with `zero_patch_dq`"""
    _chk2d(qkv, OP16)
    assert _ld(o_cls) == _ld(d_o_cls) and _ld(dqkv) == _ld(dqkv_cls)
    lib().call("pvrl_attn_cls_bwd", _ptr(qkv), _ld(qkv), nseq, S, H, T, cls_base, float(scale), _ptr(o_cls), _ptr(d_o_cls), _ld(o_cls),
               _ptr(lse), _ptr(dqkv), _ptr(dqkv_cls), _ld(dqkv), 1 if zero_patch_dq else 0, _stream())
    return dqkv, dqkv_cls


def attn_bwd(qkv, o, o_cls, d_o, d_o_cls, lse, nseq, S, H, scale, mode=0, T=1, cls_base=0, causal=False, kpm=None,
             dqkv=None, dqkv_cls=None):
    L = lib()
    _chk2d(qkv, OP16)
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if mode == 1 and dqkv_cls is None:
        dqkv_cls = torch.empty((nseq, qkv.shape[1]), device=qkv.device, dtype=OP16)
    if dqkv_cls is not None:
        assert _ld(dqkv_cls) == _ld(dqkv)
    if o_cls is not None:
        assert _ld(o_cls) == _ld(o) == _ld(d_o) == _ld(d_o_cls)
    else:
        assert _ld(o) == _ld(d_o)
    dvec = torch.empty_like(lse)
    L.call("pvrl_attn_bwd", _ptr(qkv), _ld(qkv), nseq, S, H, mode, T, cls_base, float(scale), 1 if causal else 0,
           _ptr(kpm), _ptr(o), _ptr(o_cls), _ptr(d_o), _ptr(d_o_cls), _ld(o), _ptr(lse), _ptr(dvec), _ptr(dqkv),
           _ptr(dqkv_cls), _ld(dqkv), _stream())
    return dqkv, dqkv_cls


# ----------------------------------------------------------------------------------------
# layout / cast helpers
# ----------------------------------------------------------------------------------------
def patchify(frames, out=None):
    """frames fp32 [B,3,T,H,W] -> bf16 [(b,n,t), 768]"""
    L = lib()
    assert frames.dtype == F32 and frames.is_contiguous() and frames.is_cuda and frames.shape[1] == 3
    B, _, T, HI, WI = frames.shape
    rows = B * (HI // 16) * (WI // 16) * T
    if out is None:
        out = torch.empty((rows, 768), device=frames.device, dtype=OP16)
    L.call("pvrl_patchify", _ptr(frames), B, T, HI, WI, _ptr(out), _ld(out), _stream())
    return out


def frames_u8_patchify(clips, out=None):
    """transform.DecodedClips (uint8 [B,T,H0,W0,3] + per-clip draws) -> bf16 [(b,n,t), 768] of the normalised,
    rescaled, cropped, flipped clip (the reference's CPU-worker chain, fused into the im2col)."""
    import ctypes
    L = lib()
    fr = clips.frames
    assert fr.is_cuda and fr.dtype == torch.uint8 and fr.is_contiguous()
    B, T, H0, W0, _ = fr.shape
    crop = clips.crop
    ph = clips.params_host
    if bool(((ph[:, 0] - ph[:, 2]) < crop).any()) or bool(((ph[:, 1] - ph[:, 3]) < crop).any()) or bool((ph[:, 2:4] < 0).any()):
        raise ValueError("crop window leaves the rescaled frame")
    rows = B * (crop // 16) * (crop // 16) * T
    if out is None:
        out = torch.empty((rows, 768), device=fr.device, dtype=OP16)
    mean = (ctypes.c_float * 3)(*clips.mean)
    std = (ctypes.c_float * 3)(*clips.std)
    L.call("pvrl_frames_u8_patchify", _ptr(fr), _ptr(clips.params), B, T, H0, W0, crop,
           ctypes.cast(mean, ctypes.c_void_p), ctypes.cast(std, ctypes.c_void_p), _ptr(out), _ld(out), _stream())
    return out


def frames_u8_to_f32(clips):
    """transform.DecodedClips -> fp32 [B, 3, T, crop, crop]: the tensor the reference's CPU workers would have produced"""
    import ctypes
    L = lib()
    fr = clips.frames
    B, T, H0, W0, _ = fr.shape
    crop = clips.crop
    ph = clips.params_host
    if bool(((ph[:, 0] - ph[:, 2]) < crop).any()) or bool(((ph[:, 1] - ph[:, 3]) < crop).any()) or bool((ph[:, 2:4] < 0).any()):
        raise ValueError("crop window leaves the rescaled frame")
    out = torch.empty((B, 3, T, crop, crop), device=fr.device, dtype=F32)
    mean = (ctypes.c_float * 3)(*clips.mean)
    std = (ctypes.c_float * 3)(*clips.std)
    L.call("pvrl_frames_u8_to_f32", _ptr(fr), _ptr(clips.params), B, T, H0, W0, crop, ctypes.cast(mean, ctypes.c_void_p),
           ctypes.cast(std, ctypes.c_void_p), _ptr(out), _stream())
    return out


def embed_table(pos, time, bias, N, T):
    L = lib()
    C = pos.shape[-1]
    E = torch.empty((N * T, C), device=pos.device, dtype=F32)
    L.call("pvrl_embed_table", _ptr(pos), _ptr(time), _ptr(bias), _ptr(E), N, T, C, _stream())
    return E


def batch_sum(dx, B, rows):
    """G[r] = sum_b dx[b * rows + r] (fp32); dx fp32 or the operand type (the 16-bit patch rows of the split gradient stream)"""
    L = lib()
    _chk2d(dx)
    C = dx.shape[1]
    G = torch.empty((rows, C), device=dx.device, dtype=F32)
    if dx.dtype == OP16:
        L.call("pvrl_batch_sum_bf16", _ptr(dx), _ld(dx), B, rows, C, _ptr(G), _stream())
    else:
        _chk2d(dx, F32)
        L.call("pvrl_batch_sum", _ptr(dx), _ld(dx), B, rows, C, _ptr(G), _stream())
    return G


def cast_scale(x, rowscale=None, out=None):
    L = lib()
    _chk2d(x, F32)
    M, C = x.shape
    if out is None:
        out = torch.empty((M, C), device=x.device, dtype=OP16)
    L.call("pvrl_cast_scale_bf16", _ptr(x), _ld(x), _ptr(rowscale), _ptr(out), _ld(out), M, C, _stream())
    return out


def cast_transpose(w, out=None):
    L = lib()
    assert w.dtype == F32 and w.dim() == 2 and w.is_contiguous()
    R, C = w.shape
    if out is None:
        out = torch.empty((C, R), device=w.device, dtype=OP16)
    L.call("pvrl_cast_transpose_bf16", _ptr(w), _ptr(out), R, C, _stream())
    return out


def cast_weight(w, out=None, out_t=None, need_t=True):
    """fp32 [R, C] -> (bf16 [R, C], bf16 [C, R] or None) in one kernel."""
    L = lib()
    assert w.dtype == F32 and w.dim() == 2 and w.is_contiguous()
    R, C = w.shape
    if out is None:
        out = torch.empty((R, C), device=w.device, dtype=OP16)
    if need_t and out_t is None:
        out_t = torch.empty((C, R), device=w.device, dtype=OP16)
    L.call("pvrl_cast_weight_bf16", _ptr(w), _ptr(out), _ptr(out_t) if need_t else None, R, C, _stream())
    return out, (out_t if need_t else None)


def gemv_rows(W, x, out=None, beta=0.0, gscale=None):
    """out[r] = beta * out[r] + gscale * W[r, :] . x  (W fp32 or bf16 [R, C] row-major, x fp32 [C]) -> fp32 [R]"""
    _chk2d(W)
    R, C = W.shape
    assert x.dtype == F32 and x.is_contiguous() and x.numel() == C
    if out is None:
        out = torch.empty(R, device=W.device, dtype=F32)
    assert out.dtype == F32 and out.is_contiguous() and out.numel() == R
    lib().call("pvrl_gemv_rows_f32", _ptr(W), 1 if W.dtype == OP16 else 0, _ld(W), R, C, _ptr(x), float(beta), _ptr(out),
               _ptr(gscale), _stream())
    return out


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def gemv_rows_batched(Ws, xs, outs, betas, gscale=None):
    """out_i = beta_i * out_i + gscale * W_i x_i for equally-shaped W_i [R, C] (all fp32 or all 16-bit), one launch per 16"""
    W0 = _chk2d(Ws[0])
    R, C = W0.shape
    for W, x, o in zip(Ws, xs, outs):
        assert W.shape == (R, C) and W.dtype == W0.dtype and _ld(W) == _ld(W0) and x.dtype == F32 and x.is_contiguous() and x.numel() == C
        assert o.dtype == F32 and o.is_contiguous() and o.numel() == R
    aw, ax, ao = _ptr_array(Ws), _ptr_array(xs), _ptr_array(outs)
    ab = (ctypes.c_float * len(Ws))(*[float(b) for b in betas])
    lib().call("pvrl_gemv_rows_batched_f32", len(Ws), ctypes.addressof(aw), 1 if W0.dtype == OP16 else 0, _ld(W0), R, C,
               ctypes.addressof(ax), ctypes.addressof(ab), ctypes.addressof(ao), _ptr(gscale), _stream())
    return outs


def rank1_add_batched(outs, As, Bs, gscale=None):
    """out_i[r][c] += gscale * a_i[r] * b_i[c] for equally-shaped fp32 out_i [R, C], one launch per 16"""
    o0 = _chk2d(outs[0], F32)
    R, C = o0.shape
    As = [a.contiguous() for a in As]
    Bs = [b.contiguous() for b in Bs]
    for o, a, b in zip(outs, As, Bs):
        assert o.shape == (R, C) and _ld(o) == _ld(o0) and a.dtype == F32 and b.dtype == F32 and a.numel() == R and b.numel() == C
    ao, aa, ab = _ptr_array(outs), _ptr_array(As), _ptr_array(Bs)
    lib().call("pvrl_rank1_add_batched_f32", len(outs), ctypes.addressof(ao), _ld(o0), ctypes.addressof(aa), ctypes.addressof(ab), R, C,
               _ptr(gscale), _stream())
    return outs


def cast_weights_multi(items):
    """items: list of (w fp32 [R, C] contiguous, out bf16 [R, C], out_t bf16 [C, R] or None) -- one launch for all."""
    from ._lib import CastProblem
    if not items:
        return
    arr = (CastProblem * len(items))()
    for a, (w, out, out_t) in zip(arr, items):
        assert w.dtype == F32 and w.dim() == 2 and w.is_contiguous() and out.is_contiguous() and out.dtype == OP16
        assert out_t is None or (out_t.is_contiguous() and out_t.dtype == OP16)
        a.inp, a.out, a.out_t = w.data_ptr(), out.data_ptr(), (None if out_t is None else out_t.data_ptr())
        a.R, a.C = w.shape
    lib().call("pvrl_cast_weights_multi_bf16", len(items), ctypes.addressof(arr), _stream())


def group_reduce(x, groups, G, scale=None, alpha=1.0, resid=None, out=None, out_dtype=F32):
    L = lib()
    _chk2d(x)
    C = x.shape[1]
    if out is None:
        out = torch.empty((groups, C), device=x.device, dtype=out_dtype)
    L.call("pvrl_group_reduce", _ptr(x), 1 if x.dtype == F32 else 0, _ld(x), groups, G, C, _ptr(scale), float(alpha),
           _ptr(resid), _ld(resid) if resid is not None else 0, _ptr(out), 1 if out.dtype == F32 else 0, _ld(out),
           _stream())
    return out


def group_bcast(x, groups, G, scale=None, alpha=1.0, out=None):
    L = lib()
    _chk2d(x, F32)
    C = x.shape[1]
    if out is None:
        out = torch.empty((groups * G, C), device=x.device, dtype=OP16)
    L.call("pvrl_group_bcast_bf16", _ptr(x), _ld(x), groups, G, C, _ptr(scale), float(alpha), _ptr(out), _ld(out),
           _stream())
    return out


# ----------------------------------------------------------------------------------------
# loss head
# ----------------------------------------------------------------------------------------
def l2norm_fwd(x):
    L = lib()
    _chk2d(x, F32)
    M, D = x.shape
    y = torch.empty_like(x)
    inv = torch.empty(M, device=x.device, dtype=F32)
    L.call("pvrl_l2norm_fwd", _ptr(x), _ld(x), _ptr(y), _ld(y), _ptr(inv), M, D, _stream())
    return y, inv


def l2norm_bwd(dy, y, inv):
    L = lib()
    _chk2d(dy, F32)
    M, D = y.shape
    dx = torch.empty_like(y)
    L.call("pvrl_l2norm_bwd", _ptr(dy), _ld(dy), _ptr(y), _ld(y), _ptr(inv), _ptr(dx), _ld(dx), M, D, _stream())
    return dx


def kl_topk(pred, teacher, topk, grad_scale=None, want_target=False):
    """Returns (row_loss[rows], dpred or None, target or None).  loss1 = row_loss.sum() / rows."""
    L = lib()
    _chk2d(pred, F32); _chk2d(teacher, F32)
    rows, K = pred.shape
    row_loss = torch.empty(rows, device=pred.device, dtype=F32)
    dpred = torch.empty_like(pred) if grad_scale is not None else None
    target = torch.empty_like(pred) if want_target else None
    L.call("pvrl_kl_topk", _ptr(pred), _ld(pred), _ptr(teacher), _ld(teacher), rows, K, topk,
           float(grad_scale if grad_scale is not None else 0.0), _ptr(row_loss), _ptr(dpred),
           _ld(dpred) if dpred is not None else 0, _ptr(target), _ld(target) if target is not None else 0, _stream())
    return row_loss, dpred, target


def mse(a, b, grad_scale=None):
    L = lib()
    assert a.dtype == F32 and b.dtype == F32 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    loss = torch.empty(1, device=a.device, dtype=F32)
    da = torch.empty_like(a) if grad_scale is not None else None
    db = torch.empty_like(b) if grad_scale is not None else None
    L.call("pvrl_mse", _ptr(a), _ptr(b), a.numel(), float(grad_scale if grad_scale is not None else 0.0), _ptr(loss),
           _ptr(da), _ptr(db), _stream())
    return loss, da, db


def milnce(x, n, C, grad_scale=None):
    """x fp32 [n, n*C] -> (nom[n], den[n], dx or None)"""
    L = lib()
    assert x.dtype == F32 and x.is_contiguous() and x.numel() == n * n * C
    nom = torch.empty(n, device=x.device, dtype=F32)
    den = torch.empty(n, device=x.device, dtype=F32)
    dx = torch.empty_like(x) if grad_scale is not None else None
    L.call("pvrl_milnce", _ptr(x), n, C, float(grad_scale if grad_scale is not None else 0.0), _ptr(nom), _ptr(den),
           _ptr(dx), _stream())
    return nom, den, dx


def rank1_add(out, a, b, gscale=None):
    """out[r][c] += gscale * a[r] * b[c]  (fp32 [R, C] row-major, in place)"""
    L = lib()
    _chk2d(out, F32)
    assert a.dtype == F32 and b.dtype == F32 and a.numel() == out.shape[0] and b.numel() == out.shape[1]
    L.call("pvrl_rank1_add_f32", _ptr(out), _ld(out), _ptr(a.contiguous()), _ptr(b.contiguous()), out.shape[0], out.shape[1], _ptr(gscale),
           _stream())
    return out


def softmax_rows(x):
    """fp32 [M, N] -> row softmax (eval-mode probabilities)"""
    L = lib()
    _chk2d(x, F32)
    y = torch.empty_like(x)
    L.call("pvrl_softmax_rows_f32", _ptr(x), _ld(x), _ptr(y), _ld(y), x.shape[0], x.shape[1], _stream())
    return y


def gelu_f32(x, dy=None):
    """exact-erf GELU of a contiguous fp32 tensor, or dy * gelu'(x)"""
    L = lib()
    assert x.dtype == F32 and x.is_contiguous() and x.is_cuda
    out = torch.empty_like(x)
    L.call("pvrl_gelu_f32", _ptr(x), _ptr(dy), _ptr(out), x.numel(), _stream())
    return out
